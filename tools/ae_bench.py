#!/usr/bin/env python3
"""Stage-4 (algebraic error) LM iteration time on the FR set of a synthetic config.
Prints one JSON line; not the driver's bench (bench.py measures BASELINE.json's metric).

  python tools/ae_bench.py --config D --steps 10 --warmup 2 [--cpu-sample 0.05]
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
    ap.add_argument("--config", default="D")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cpu-sample", type=float, default=0.0, help="fraction of the time range for the oracle leg")
    a = ap.parse_args()
    cfg = synth.preset(a.config)
    t = time.time()
    cam, pat, tim, a7, _xt, x0 = synth.ae_relations(cfg, shuffle=False)
    gen_s = time.time() - t
    P = capi.AeProblem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
    P.add_relations(cam, pat, tim, a7)
    opt = capi.default_options(max_num_iterations=10 ** 6, profile=1, function_tolerance=0.0, parameter_tolerance=0.0,
                               gradient_tolerance=0.0)
    P.lm_begin(x0, opt)
    for _ in range(a.warmup):
        P.lm_step()
    P.synchronize()
    P.stage_times(reset=True)
    t = time.time()
    acc = 0
    for _ in range(a.steps):
        ok, done = P.lm_step()
        acc += ok
        if done:
            break
    P.synchronize()
    dt = time.time() - t
    st = P.stage_times()
    _x, s = P.lm_end()
    out = {"metric": "ae_lm_iteration", "config": a.config, "relations": int(cam.size), "steps": a.steps,
           "ms_per_step": 1e3 * dt / a.steps, "relations_per_s": cam.size * a.steps / dt, "accepted": acc,
           "initial_cost": s.initial_cost, "final_cost": s.final_cost, "generate_s": gen_s,
           "stage_ms": {k: (st[k] / max(1, st["n_" + k])) for k in st if not k.startswith("n_")},
           "stage_calls": {k: st[k] for k in st if k.startswith("n_")}}
    if a.cpu_sample > 0:
        from tests.oracle_lib import load_oracle
        api = load_oracle()
        t1 = max(2, int(cfg.tn * a.cpu_sample))
        keep = tim < cfg.cn + cfg.pn + t1
        keep |= tim == cfg.t_star_var
        O = capi.AeProblem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, api=api)
        O.add_relations(cam[keep], pat[keep], tim[keep], a7[keep])
        t = time.time()
        _x, so = O.solve(x0, capi.default_options(max_num_iterations=3))
        cdt = time.time() - t
        out["cpu_baseline"] = {"kind": "port", "relations": int(keep.sum()), "iterations": int(so.num_iterations),
                               "ms_per_iteration": 1e3 * cdt / max(1, so.num_iterations),
                               "relations_per_s": keep.sum() * so.num_iterations / cdt, "cores": os.cpu_count()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
