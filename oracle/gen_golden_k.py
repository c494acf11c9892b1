#!/usr/bin/env python3
"""Generate tests/golden/rp_golden_k.json — golden vectors for the reprojection residual with
the 12 camera parameters as a fourth parameter block (TEST INFRASTRUCTURE; see
oracle/README.md).

BASELINE.json configs[2] asks for "intrinsics + distortion refined".  The reference keeps
[fx cx fy cy k1 k2 p1 p2 k3 k4 k5 k6] constant inside its functors
(/root/reference/src/solving-structure.hpp:738-749), so there is nothing in it to compare
the extra derivatives with: the residual formula (hpp:658-790, restated in gen_golden.py) is
differentiated here with respect to all 33 parameters [camera 7 | pattern 7 | time 7 |
camera parameters 12] by central differences in mpmath at 100 digits (h = 1e-30), with no
hand-derived derivative.  Inputs are exact doubles / float32, outputs 40-digit strings.

Run:  python oracle/gen_golden_k.py   (deterministic; seed fixed)
"""
import json
import os
import random
import sys

import mpmath as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import H, make_case, residual  # noqa: E402

mp.mp.dps = 100


def main():
    rng = random.Random(0xCA11C0 + 2)
    out = []
    for idx in range(16):
        case, cam, pat, tim, K, uv, X = make_case(rng, idx)
        # every vector has tangential terms and k3; every other one the rational terms too
        K[6], K[7], K[8] = rng.uniform(-2e-3, 2e-3), rng.uniform(-2e-3, 2e-3), rng.uniform(-0.05, 0.05)
        if idx % 2 == 1:
            K[9], K[10], K[11] = rng.uniform(-0.1, 0.1), rng.uniform(-0.05, 0.05), rng.uniform(-0.01, 0.01)
        use_p = case in (0, 2)
        use_t = case in (0, 1)

        def fun(params):
            c = params[0:7]
            p = params[7:14] if use_p else None
            t = params[14:21] if use_t else None
            return residual(c, p, t, params[21:33], [mp.mpf(a) for a in uv], [mp.mpf(a) for a in X])

        base = [mp.mpf(v) for v in cam + pat + tim + K]
        r = fun(base)
        J = [[mp.mpf(0)] * 33 for _ in range(2)]
        for k in range(33):
            blk = k // 7
            if k < 21 and ((blk == 1 and not use_p) or (blk == 2 and not use_t)):
                continue
            hi = list(base)
            lo = list(base)
            hi[k] += H
            lo[k] -= H
            rh, rl = fun(hi), fun(lo)
            for rr in range(2):
                J[rr][k] = (rh[rr] - rl[rr]) / (2 * H)
        out.append({
            "case": case,
            "camera": cam, "pattern": pat, "time": tim,
            "intrinsics": K, "uv": uv, "point": X,
            "r": [mp.nstr(v, 40) for v in r],
            "J": [[mp.nstr(v, 40) for v in row] for row in J],
        })
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "rp_golden_k.json")
    with open(dst, "w") as f:
        json.dump({"generator": "oracle/gen_golden_k.py", "dps": 100, "h": "1e-30",
                   "columns": "camera 7 | pattern 7 | time 7 | fx cx fy cy k1 k2 p1 p2 k3 k4 k5 k6", "vectors": out}, f, indent=1)
    print("wrote", os.path.normpath(dst), len(out), "vectors")


if __name__ == "__main__":
    main()
