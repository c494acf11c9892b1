"""dev: time of the fused kernel alone on config D (10 evaluations through calico_rp_normal_eq-free path: lm steps
with speculation, results are garbage under RP_FUSED_VARIANT != 0, only the stage time is read)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from calico_amd import capi, synth
cfg = synth.preset("D")
if len(sys.argv) > 1:
    cfg = synth.preset("D", tn=int(sys.argv[1]))
prob = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
synth.fill_problem(prob, cfg, 0, cfg.tn)
x0 = synth.initial_guess(cfg)
opt = capi.default_options(max_num_iterations=3, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0,
                           profile=1)
for rep in range(2):
    prob.stage_times(reset=True)
    for i in range(3):
        try:
            prob.lm_begin(x0, opt)
            for k in range(3):
                prob.lm_step()
        except Exception as e:
            print("lm:", e)
        try:
            prob.lm_end()
        except Exception as e:
            print("lm_end:", e)
    st = prob.stage_times()
print({k: round(st[k] / max(1, st["n_" + k]), 3) for k in ("jacobian_sweep", "accumulate")}, st["n_jacobian_sweep"])
