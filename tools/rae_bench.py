#!/usr/bin/env python3
"""Point reconstruction (calico_rae_reconstruct) on the views of a synthetic config.
Prints one JSON line; not the driver's bench.

  python tools/rae_bench.py --config C [--times 1000] [--cpu-times 20]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from calico_amd import capi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C")
    ap.add_argument("--times", type=int, default=0, help="use only the first N times (0: all)")
    ap.add_argument("--cpu-times", type=int, default=0, help="oracle leg on the first N times")
    a = ap.parse_args()
    cfg = synth.preset(a.config)
    t1 = a.times or cfg.tn
    t = time.time()
    args = synth.rae_inputs(cfg, x=synth.initial_guess(cfg), t1=t1)
    gen_s = time.time() - t
    n_obs, n_views = int(args[6].size), int(args[2].size)
    capi.rae_reconstruct(*args, options=capi.default_options(max_num_iterations=0))  # warm-up (library load)
    t = time.time()
    xyz, has, its, fc, s = capi.rae_reconstruct(*args)
    wall = time.time() - t
    # the same through a handle that keeps the views and observations on the device (calico_rae_create / _solve)
    t = time.time()
    H = capi.RaeProblem(args[1], args[2], args[4], args[5], args[6], args[7], args[8], args[9])
    create_s = time.time() - t
    H.solve(args[0], args[3])
    t = time.time()
    xh, hh, _, _, sh = H.solve(args[0], args[3])
    solve_wall = time.time() - t
    assert np.array_equal(xh, xyz)
    H.close()
    pts = synth.points(cfg)
    out = {"metric": "rae_reconstruct", "config": a.config, "times": t1, "observations": n_obs, "views": n_views,
           "points": int(has.size), "solved": int(has.sum()), "sweeps": s.sweeps,
           "max_point_iterations": s.max_point_iterations, "sweep_ms": s.sweep_ms, "update_ms": s.update_ms,
           "total_ms": s.total_ms, "wall_s": wall, "generate_s": gen_s,
           "handle": {"create_s": create_s, "solve_total_ms": sh.total_ms, "solve_wall_s": solve_wall,
                      "sweep_ms": sh.sweep_ms, "note": "calico_rae_create once, then calico_rae_solve per set of poses"},
           "sweep_gbps": 20.0 * n_obs / (s.sweep_ms * 1e-3) / 1e9 if s.sweep_ms else None,
           "max_err_mm_vs_model": float(np.abs(xyz[has] - pts[has]).max())}
    if a.cpu_times:
        from tests.oracle_lib import load_oracle
        cargs = synth.rae_inputs(cfg, x=synth.initial_guess(cfg), t1=a.cpu_times)
        t = time.time()
        capi.rae_reconstruct(*cargs, api=load_oracle())
        dt = time.time() - t
        out["cpu_baseline"] = {"kind": "port", "observations": int(cargs[6].size), "seconds": dt,
                               "cores": os.cpu_count()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
