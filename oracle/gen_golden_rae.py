#!/usr/bin/env python3
"""Generate tests/golden/rae_golden.json — high-precision golden vectors for the point
reconstruction residual (ReconstructXStruct, /root/reference/src/solving-structure.hpp:1313-1409)
and its 2x3 Jacobian with respect to the 3-D point (TEST INFRASTRUCTURE; oracle/README.md).
mpmath at 100 digits, central differences with h = 1e-30; no hand-derived derivative.

Run:  python oracle/gen_golden_rae.py   (deterministic; seed fixed)
"""
import json
import os
import random

import mpmath as mp

from gen_golden import quat_to_rot, rand_quat

mp.mp.dps = 100
H = mp.mpf(10) ** -30


def residual(Ah, K, X, uv):
    xc = [sum(Ah[4 * i + j] * X[j] for j in range(3)) + Ah[4 * i + 3] for i in range(3)]
    fx, cx, fy, cy, k1, k2, p1, p2, k3, k4, k5, k6 = K
    xn, yn = xc[0] / xc[2], xc[1] / xc[2]
    r2 = xn * xn + yn * yn
    numer = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
    denom = 1 + k4 * r2 + k5 * r2 ** 2 + k6 * r2 ** 3
    xpp = xn * numer / denom + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn)
    ypp = yn * numer / denom + 2 * p2 * xn * yn + p1 * (r2 + 2 * yn * yn)
    return [xpp * fx + cx - uv[0], ypp * fy + cy - uv[1]]


def main():
    rng = random.Random(0x4AE)
    out = []
    for idx in range(16):
        q = rand_quat(rng, small=True)
        R = quat_to_rot([mp.mpf(v) for v in q])
        t = [rng.uniform(-200, 200), rng.uniform(-200, 200), rng.uniform(1500, 3000)]
        Ah = [float(R[i][j]) if j < 3 else t[i] for i in range(3) for j in range(4)]
        K = [rng.uniform(900, 1600), rng.uniform(500, 700), rng.uniform(900, 1600), rng.uniform(350, 550),
             rng.uniform(-0.3, 0.3), rng.uniform(-0.2, 0.2), 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
        if idx % 3 == 1:
            K[6], K[7], K[8] = rng.uniform(-2e-3, 2e-3), rng.uniform(-2e-3, 2e-3), rng.uniform(-0.05, 0.05)
        if idx % 5 == 2:
            K[9], K[10], K[11] = rng.uniform(-0.1, 0.1), rng.uniform(-0.05, 0.05), rng.uniform(-0.01, 0.01)
        X = [rng.uniform(0, 400), rng.uniform(0, 300), rng.uniform(-50, 50)]
        uv = [rng.uniform(0, 1200), rng.uniform(0, 900)]
        f = lambda P: residual([mp.mpf(v) for v in Ah], [mp.mpf(v) for v in K], P, [mp.mpf(v) for v in uv])
        base = [mp.mpf(v) for v in X]
        r = f(base)
        J = [[mp.mpf(0)] * 3 for _ in range(2)]
        for k in range(3):
            hi, lo = list(base), list(base)
            hi[k] += H
            lo[k] -= H
            rh, rl = f(hi), f(lo)
            for rr in range(2):
                J[rr][k] = (rh[rr] - rl[rr]) / (2 * H)
        out.append({"a_hat": Ah, "intrinsics": K, "point": X, "uv": uv,
                    "r": [mp.nstr(v, 40) for v in r], "J": [[mp.nstr(v, 40) for v in row] for row in J]})
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "rae_golden.json")
    with open(dst, "w") as f:
        json.dump({"generator": "oracle/gen_golden_rae.py", "dps": 100, "h": "1e-30", "vectors": out}, f, indent=1)
    print("wrote", os.path.normpath(dst), len(out), "vectors")


if __name__ == "__main__":
    main()
