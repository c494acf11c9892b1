import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, time
from calico_amd import capi, synth
cfg = synth.preset("D", tn=int(sys.argv[1])) if len(sys.argv) > 1 else synth.preset("D")
P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
t=time.time(); obs = synth.fill_problem(P, cfg); n = obs[0].size; print("fill", time.time()-t, n, flush=True)
x0 = synth.initial_guess(cfg)
if len(sys.argv) > 3 and sys.argv[3] == 'nospec':
    P.set_speculation(False)
if len(sys.argv) > 3 and sys.argv[3] == 'materialised':
    P.set_jacobian_mode('materialised')
x, s = P.solve(x0, capi.default_options(verbose=1, max_num_iterations=int(sys.argv[2]) if len(sys.argv) > 2 else 30))
print(s.num_iterations, s.message, np.sqrt(s.final_cost/n))
