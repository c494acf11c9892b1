#!/usr/bin/env python3
"""bench.py — LM iteration time and residual/Jacobian throughput of the HIP path.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched
with torch.distributed.run, one rank per GPU.  A "step" is ONE complete LM iteration of
a running solve: residuals + Jacobian + per-FR block accumulation over every observation
at the current point, block gathers, time-block elimination, Schur SYRK, dense Cholesky,
back-substitution, the cost at the candidate and the accept/reject test (the work
ceres::Solve does per successful iteration, solving-structure.cpp:982).  The solve is
restarted from the same perturbed x0 every --iters-per-solve steps (default 4: from the
1 degree / 5 mm start four steps reach the noise floor, a fifth is rejected by rounding;
the reference allows ten per solve, calico.cpp:709, and stops on its function tolerance);
lm_begin / lm_end of those solves are inside the timed region.  Only accepted steps do the whole work, so the line
reports how many of the K timed steps were accepted (all of them on the default run) and
the run fails if fewer than 80 % were.

Workload: BASELINE.json configs[4] — synthetic 128 cameras x 8 patterns x 10k times
(~102.4 M corner observations).  It fits one MI355X (about 45 GB), so N = 1 runs the
whole network and N > 1 time-shards the SAME network over the ranks (strong scaling),
with the reduced normal equations all-reduced over RCCL each iteration.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6  # MI355X datasheet, fp64 vector = fp64 matrix; tools/mfma_bench.hip
                         # sustains 47 (v_mfma_f64_16x16x4) / 57 (v_fma_f64) TFLOP/s on this part

# The fused kernel on chain columns (calico_amd/csrc/rp_model.hpp: eval_chain, rp_kernels.hip:
# fused7q_kernel).  Algorithmic work per launch:
#   bytes: 24 B per observation read (u, v, corner index, FR index), per FR 512 B of the per-FR
#          table read + 1024 B of record A written (record B, 720 B per (time, pattern) piece, is
#          left out: 58 MB of 4.5 GB on config D)
#   flops: per observation 259 (composed transform 18 + wc 3, projection + distortion 49,
#          d(u,v)/dXc 59, one cross product per residual row 18, upper triangle of the 7-column Gram
#          2 rows x 28 products x 2 = 112); per part 2 x (7 x 7 x 16 + 16 x 7 x 16) = 5152 for
#          C^T G7 C.  (The 16-column form used until round 2 needed 820 per observation.)
# 7.5 flop per byte on config D, under the machine balance of 78.6 TFLOP/s / 8 TB/s = 9.8: the HBM
# roof is the binding one (0.56 ms per launch on config D; the flop roof would be 0.42 ms).
FUSED_BYTES_PER_OBS = 24
FUSED_BYTES_PER_FR = 512 + 1024
FUSED_FLOP_PER_OBS = 18 + 3 + 49 + 59 + 18 + 112
FUSED_FLOP_PER_PART = 2 * (7 * 7 * 16 + 16 * 7 * 16)


def algorithmic_sweep_bytes(obs, cfg):
    """per observation 24 B read + 16 B r + 8 B x (2 x 7 x blocks)
    written: 376 B with three blocks, 264 B with two, 152 B with one; + 276 B per FR."""
    cam, pat, tim = obs[0], obs[1], obs[2]
    nblk = 1 + (pat != cfg.p_star_var).astype(np.int64) + (tim != cfg.t_star_var).astype(np.int64)
    per_obs = 24 + 16 + 112 * nblk
    key = (tim.astype(np.int64) * (cfg.cn + 1) + cam) * (cfg.cn + cfg.pn + 1) + pat
    n_fr = 1 + int(np.count_nonzero(key[1:] != key[:-1])) if key.size else 0
    return int(per_obs.sum()) + 276 * n_fr, n_fr


def cpu_baseline(cfg, seconds_budget=25.0):
    """The oracle ("port": dual-number restatement of the reference functors + dense
    Schur LM, OpenMP) timed on this box's host cores on a bounded time range of the
    same workload.  Reported beside the GPU numbers; never the thing measured."""
    from calico_amd import capi, synth
    from tests.oracle_lib import load_oracle

    api = load_oracle()
    cores = min(16, os.cpu_count() or 1)
    threads = api.lib.oracle_rp_set_threads(cores)
    # a time range sized so that the whole sample is roughly 10-30 s of CPU work
    t1 = max(2, min(cfg.tn, cfg.tn // 16))
    P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, api=api)
    P.set_intrinsics(synth.intrinsics(cfg))
    P.set_points(synth.points(cfg))
    obs = synth.observations(cfg, 0, t1)
    P.add_observations(*obs)
    n = obs[0].size
    x0 = synth.initial_guess(cfg)
    fn = api.lib.oracle_rp_solve_timed
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(capi.Options), C.POINTER(capi.Summary),
                   C.POINTER(C.c_double), C.POINTER(C.c_double)]
    iters = 6
    opt = capi.default_options(max_num_iterations=iters, function_tolerance=0.0, parameter_tolerance=0.0,
                               gradient_tolerance=0.0)
    x = x0.copy()
    s = capi.Summary()
    jac_ms = C.c_double()
    it_ms = (C.c_double * (iters + 2))()
    t0 = time.time()
    fn(P.h, x.ctypes.data_as(C.POINTER(C.c_double)), C.byref(opt), C.byref(s), C.byref(jac_ms), it_ms)
    wall = time.time() - t0
    per_iter = float(np.median([it_ms[i] for i in range(iters)]))
    return {
        "value": n / per_iter / 1e3,  # M observations per second through one LM iteration
        "unit": "Mobs/s",
        "cores": int(threads),
        "kind": "port",
        "sample": f"times [0,{t1}) of the same network: {n} observations, {iters} LM iterations, "
                  f"{per_iter:.1f} ms/iteration, Jacobian evaluation {jac_ms.value:.1f} ms "
                  f"({n / jac_ms.value / 1e3:.2f} Mobs/s), {wall:.1f} s wall",
        "ms_per_iteration": per_iter,
        "jacobian_mobs_per_s": n / jac_ms.value / 1e3,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="D", choices=["A", "B", "C", "D"])
    ap.add_argument("--tn", type=int, default=0, help="override the number of times (debugging)")
    ap.add_argument("--refine-intrinsics", action="store_true",
                    help="free all 12 camera parameters per camera (BASELINE.json configs[2]: config B); the timed "
                         "kernel is then fused7qk_kernel (19 columns per residual row, the form with the rational "
                         "model's denominator because k4..k6 are free), not the one the roofline names")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ramp-seconds", type=float, default=0.3,
                    help="untimed LM steps for this long before the warm-up steps, so that the timed region starts "
                         "on a card that is at its clocks (0: none)")
    ap.add_argument("--single-rank-comm", action="store_true",
                    help="N = 1 only: attach a one-rank RCCL communicator, so that the iteration runs the sharded code path "
                         "(global block set, pack / all-reduce / unpack of the reduced system, scalar all-gather): what the "
                         "exchange costs a rank besides the wire time")
    ap.add_argument("--iters-per-solve", type=int, default=4,
                    help="LM iterations per solve before the restart from x0 (the reference allows 10, calico.cpp:709; "
                         "from the 1 degree / 5 mm start the fifth step is already at the rounding floor of the cost")
    ap.add_argument("--no-speculation", action="store_true",
                    help="include/calico_rp.h: calico_rp_set_speculation(0): cost sweep at every candidate")
    ap.add_argument("--jacobian-mode", default="fused", choices=["fused", "materialised"],
                    help="include/calico_rp.h: calico_rp_set_jacobian_mode; the timed loop runs in this mode")
    ap.add_argument("--materialised-steps", type=int, default=3,
                    help="fused mode only: extra untimed-for-value iterations in materialised mode that give the "
                         "HBM roofline of the stand-alone Jacobian sweep kernel (0 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import torch

    from calico_amd import capi, synth

    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("CALICO_BENCH_ONE_DEVICE"):
            # test hook (tests/test_gpu_comm.py): every rank on device 0, host-side
            # collectives only; RCCL itself refuses two ranks on one GPU
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            # torch.distributed is the launcher's rendezvous, the barrier and two host scalars here:
            # gloo.  The data path's collective (one all-reduce per linear solve) is the library's
            # own RCCL communicator, created below from an id that rank 0 broadcasts.
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    cfg = synth.preset(args.config)
    if args.tn:
        cfg.tn = args.tn

    # ---- shard by time index
    if world > 1:
        opt_counts = synth.obs_per_time(cfg)
        bounds = (C.c_int32 * (world + 1))()
        capi.load().partition_times(cfg.tn, opt_counts.ctypes.data_as(C.POINTER(C.c_int64)), world, bounds)
        t0, t1 = bounds[rank], bounds[rank + 1]
    else:
        t0, t1 = 0, cfg.tn

    t_gen = time.time()
    prob = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=local_rank)
    prob.set_jacobian_mode(args.jacobian_mode)
    prob.set_speculation(not args.no_speculation)
    prob.set_intrinsics(synth.intrinsics(cfg))
    if args.refine_intrinsics:
        prob.set_refine_intrinsics(0xFFF)
    prob.set_points(synth.points(cfg))
    obs = synth.observations(cfg, t0, t1)  # the synthetic stand-in for Stages 1-4: not the library's time
    t_gen = time.time() - t_gen
    # the library's one-off cost per handle: host buffers in (calico_rp_add_observations: index checks,
    # run detection, upload) and the FR / part / piece structure (built by the first call that needs it)
    t_setup = time.time()
    prob.add_observations(*obs)
    n_frs = int(prob.api.num_frs(prob.h)) if obs[0].size else 0
    prob.synchronize()
    t_setup = time.time() - t_setup
    n_local = obs[0].size
    sweep_bytes, n_fr_local = algorithmic_sweep_bytes(obs, cfg)
    del obs
    x0 = synth.initial_guess(cfg)

    if world > 1:
        uid = [None]
        if rank == 0:
            buf = C.create_string_buffer(capi.UNIQUE_ID_BYTES)
            assert capi.load().comm_unique_id(C.cast(buf, C.c_void_p)) == 0
            uid[0] = bytes(buf.raw)
        dist.broadcast_object_list(uid, src=0)
        prob.comm_init(world, rank, uid[0])
        tot = torch.tensor([n_local], dtype=torch.int64)
        dist.all_reduce(tot)  # gloo
        n_total = int(tot.item())
    else:
        n_total = n_local
        if args.single_rank_comm:
            buf = C.create_string_buffer(capi.UNIQUE_ID_BYTES)
            assert capi.load().comm_unique_id(C.cast(buf, C.c_void_p)) == 0
            prob.comm_init(1, 0, bytes(buf.raw))

    opt = capi.default_options(max_num_iterations=1, function_tolerance=0.0, parameter_tolerance=0.0,
                               gradient_tolerance=0.0, profile=1)

    def one_step():
        prob.lm_begin(x0, opt)
        acc, _ = prob.lm_step()
        return acc

    def run_steps(k, profile=2):
        """k LM iterations as solves of --iters-per-solve iterations from x0 (the last one shorter).
        profile 2: HIP events around the dominant (fused / Jacobian sweep) kernel only; 1: around every
        stage, which costs the iteration about 0.1 ms on config D (an event pair per stage keeps
        consecutive kernels from overlapping), so the timed region runs with 2 and the stage table
        comes from a pass of its own."""
        m = max(1, args.iters_per_solve)
        acc = 0
        for i in range(k):
            if i % m == 0:
                if i:
                    prob.lm_end()
                o = capi.default_options(max_num_iterations=min(m, k - i), function_tolerance=0.0,
                                         parameter_tolerance=0.0, gradient_tolerance=0.0, profile=profile)
                prob.lm_begin(x0, o)
            a, done = prob.lm_step()
            acc += int(a)
        return acc

    def barrier():
        if dist is not None:
            dist.barrier()
        prob.synchronize()
        torch.cuda.synchronize()

    # Clock ramp: the card has idled while the network was generated and handed over, and on some boxes
    # a few steps are not enough to bring it back to its clocks (3.61 ms per step over a timed region of
    # 20 steps behind 3 warm-up steps, 3.28 in the pass right behind it).  So the same steps run untimed
    # for a fixed 0.3 s (every rank the same number: at most 400, at least 8) before the warm-up proper.
    ramp_steps = 0
    if args.ramp_seconds > 0:
        t_ramp = time.perf_counter()
        run_steps(8)
        prob.lm_end()
        prob.synchronize()
        per8 = max(time.perf_counter() - t_ramp, 1e-6)
        ramp_steps = 8 + 8 * int(min(49, max(0, round(args.ramp_seconds / per8) - 1)))
        if dist is not None:  # (the ranks' clocks differ: take rank 0's count)
            c = torch.tensor([ramp_steps], dtype=torch.int64)
            dist.broadcast(c, src=0)
            ramp_steps = int(c.item())
        if ramp_steps > 8:
            run_steps(ramp_steps - 8)
            prob.lm_end()
    if args.warmup > 0:
        run_steps(args.warmup)
        prob.lm_end()
    prob.stage_times(reset=True)
    barrier()
    t_start = time.perf_counter()
    accepted = run_steps(args.steps)
    barrier()
    elapsed = time.perf_counter() - t_start
    stage = prob.stage_times()
    _, last = prob.lm_end()
    # the stage table: the same steps once more, untimed, with every stage between HIP events
    prob.stage_times(reset=True)
    barrier()
    t_prof = time.perf_counter()
    run_steps(args.steps, profile=1)
    barrier()
    t_prof = (time.perf_counter() - t_prof) * 1e3 / args.steps
    stage_all = prob.stage_times()
    prob.lm_end()
    if accepted < 0.8 * args.steps:
        sys.exit(f"bench.py: only {accepted} of {args.steps} timed LM steps were accepted; rejected steps skip the "
                 "evaluation, the number would not be a full iteration's")
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed * 1e3 / args.steps
    sweep_ms = stage["jacobian_sweep"] / max(1, stage["n_jacobian_sweep"])
    traffic_file = os.path.join(ROOT, "profiles", "sweep_traffic.json")
    traffic_tab = {}
    # The PMC figures were collected (rocprofv3 --pmc, separate passes; counters cannot be read from in
    # here) for the default workload only: full config D on one GPU, camera parameters fixed.  Any other
    # shape reports null.
    default_shape = (args.config == "D" and not args.tn and world == 1 and not args.refine_intrinsics
                     and not args.single_rank_comm)
    if default_shape and os.path.exists(traffic_file):
        try:
            traffic_tab = json.load(open(traffic_file))
        except Exception:
            traffic_tab = {}

    def hbm_roofline(ms, kernel):
        achieved = sweep_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return {
            "kernel": kernel,
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic_tab.get("hbm_bytes_per_launch_config_%s_n%d" % (args.config, world)),
            "algorithmic_bytes_per_launch": sweep_bytes,
            "avg_launch_ms": ms,
        }

    roofline_mat = None
    if args.jacobian_mode == "materialised":
        roofline = hbm_roofline(sweep_ms, "jacobian_sweep_kernel (rank 0)")
    else:
        fbytes = float(FUSED_BYTES_PER_OBS) * n_local + float(FUSED_BYTES_PER_FR) * n_frs
        flops = float(FUSED_FLOP_PER_OBS) * n_local + float(FUSED_FLOP_PER_PART) * n_frs
        achieved = fbytes / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
        roofline = {
            "kernel": ("fused7qk_kernel (rank 0): the chain-column kernel with the 12 camera-parameter columns beside "
                       "them (rational-model form: k4..k6 are free); the byte and flop counts below are those of "
                       "fused7q_kernel and do NOT include record K (3 072 B per part) or the K columns' work"
                       if args.refine_intrinsics else
                       "fused7q_kernel (rank 0): residual + chain columns + 7-column Gram on the 4x4x4 fp64 MFMA + "
                       "C^T G7 C per part, Jacobian kept in LDS"),
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic_tab.get("fused7q_hbm_bytes_per_launch_config_%s_n%d" % (args.config, world)),
            "algorithmic_bytes_per_launch": fbytes,
            "algorithmic_flop_per_launch": flops,
            "flop_per_byte": flops / fbytes if fbytes else None,
            "avg_launch_ms": sweep_ms,
            "prepass_ms": stage_all["accumulate"] / max(1, stage_all["n_accumulate"]),
            "tflops": flops / (sweep_ms * 1e-3) / 1e12 if sweep_ms > 0 else 0.0,
            "frac_of_fp64_peak": flops / (sweep_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if sweep_ms > 0 else 0.0,
            "note": "the HBM roof is the binding one of the two (7.5 flop/B); what the kernel runs into first is the SIMD "
                    "datapath, which fp64 VALU and fp64 MFMA share on gfx950 (SQ_VALU_MFMA_COEXEC_CYCLES = 0): with the "
                    "evaluation and the Gram steps switched off the same kernel moves the same bytes in 0.70 ms (6.3 TB/s), "
                    "the evaluation adds 0.39 ms and the Gram steps + part-end products 0.24 ms (phase ablations of round 2, DESIGN.md "
                    "section 3; counters in profiles/r0*_pmc.csv); prepass_ms = prep_vars_kernel + prep_frs_kernel "
                    "(per-FR table: 512 B per FR written), booked under stage 'accumulate'; the HBM roofline of the "
                    "stand-alone Jacobian sweep is in roofline_materialised",
        }
        if args.materialised_steps > 0 and not args.refine_intrinsics:  # (refined camera parameters: fused mode only)
            # the kernel split SURVEY.md 8(d) prices, same problem, same start point
            prob.set_jacobian_mode("materialised")
            one_step()
            prob.stage_times(reset=True)
            barrier()
            tm = time.perf_counter()
            for _ in range(args.materialised_steps):
                one_step()
            barrier()
            mat_ms = (time.perf_counter() - tm) * 1e3 / args.materialised_steps
            st2 = prob.stage_times()
            prob.lm_end()
            roofline_mat = hbm_roofline(st2["jacobian_sweep"] / max(1, st2["n_jacobian_sweep"]),
                                        "jacobian_sweep_kernel (rank 0), materialised mode")
            roofline_mat["ms_per_step"] = mat_ms
            roofline_mat["accumulate_ms"] = st2["accumulate"] / max(1, st2["n_accumulate"])
            prob.set_jacobian_mode("fused")

    out = {
        "metric": "LM iteration throughput (corner observations through one full Levenberg-Marquardt "
                  "iteration per second); ms_per_step is the LM iteration time",
        "value": n_total / (ms_per_step * 1e-3) / 1e6,
        "unit": "Mobs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"config {args.config}: synthetic {cfg.cn} cameras x {cfg.pn} patterns x {cfg.tn} times, "
                        f"{n_total} corner observations (BASELINE.json configs[{synth.PRESETS[args.config]}])",
            "observations": n_total,
            "sharding": "time index, %d rank(s)" % world,
            "accepted_steps": accepted,
            "clock_ramp_steps": ramp_steps,
            "cost_initial": last.initial_cost if last else None,
        },
        "jacobian_sweep_mobs_per_s": n_local / (sweep_ms * 1e-3) / 1e6 if sweep_ms > 0 else None,
        "stage_ms_per_step_rank0": {k: (stage_all[k] / args.steps) for k in (
            "jacobian_sweep", "accumulate", "reduce", "schur", "cholesky", "backsub", "cost_sweep", "exchange")},
        "stage_note": "from a second, untimed pass over the same steps with HIP events around every stage (%.3f ms per "
                      "step with them); the timed region has events around the fused / Jacobian sweep kernel only "
                      "(roofline.avg_launch_ms); 'reduce' (the block gathers) runs on the library's second stream BESIDE 'schur' on "
                      "networks of 32 768 FRs or more, so the stages add up to more than the step" % t_prof,
        "clock_note": "shader clock while the two long kernels run, from GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration in the "
                      "counter pass of tools/profile_round.sh (profiles/r03_z_pmc.csv, SHADER_CLOCK_GHZ rows; a stored "
                      "figure, counters cannot be read from in here): fused7q_kernel 2.11 GHz, syrk_kernel 2.10 GHz, "
                      "against 2.4 GHz peak; the sustained run (profiles/r03_z_bench_sustained.json) holds the same rate "
                      "over 6.5 s",
        "roofline": roofline,
        "setup_s": t_setup,
        "setup_note": "calico_rp_add_observations + structure build + upload for %d observations in %d FRs on rank 0 "
                      "(PCIe-inclusive, pageable host buffers); generating the synthetic network took %.2f s"
                      % (n_local, n_frs, t_gen),
    }
    # Untimed: the same network solved to convergence with Ceres' defaults (budget 40 iterations; the
    # reference's is 10 per batch, calico.cpp:709), for the RMS the north star asks about.
    prob.set_jacobian_mode(args.jacobian_mode)
    xf, sf = prob.solve(x0, capi.default_options(max_num_iterations=40))
    out["config"]["rms_final_px"] = float(np.sqrt(sf.final_cost / n_total)) if n_total else None
    out["config"]["converged_solve"] = {
        "iterations": int(sf.num_iterations), "rejected_steps": int(sf.num_unsuccessful_steps),
        "termination": capi.TERMINATION.get(int(sf.termination_type)), "message": sf.message.decode(errors="replace"),
        "rms_initial_px": float(np.sqrt(sf.initial_cost / n_total)) if n_total else None,
        "total_ms": float(sf.total_ms),
        "note": "untimed, after the timed region: Ceres defaults, max_num_iterations 40; rms = sqrt(final_cost / "
                "observations) = sqrt(sum r^2 / residuals); the generator's pixel noise is 0.1 px"}
    out["config"]["single_rank_comm"] = bool(args.single_rank_comm)
    out["config"]["jacobian_mode"] = args.jacobian_mode
    out["config"]["refine_intrinsics"] = bool(args.refine_intrinsics)
    out["config"]["iters_per_solve"] = args.iters_per_solve
    out["config"]["speculative_evaluation"] = not args.no_speculation
    out["config"]["sweeps_in_timed_steps"] = {"fused_or_jacobian": stage["n_jacobian_sweep"],
                                              "cost_only": stage_all["n_cost_sweep"]}
    if roofline_mat is not None:
        out["roofline_materialised"] = roofline_mat
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg)
    if rank == 0:
        print(json.dumps(out))
    prob.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
