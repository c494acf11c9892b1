#!/usr/bin/env python3
"""Generate tests/golden/ae_golden.json — high-precision golden vectors for the Stage-4
algebraic-error residual and its Jacobian (TEST INFRASTRUCTURE; see oracle/README.md).

The reference holds no golden vectors or tests for this path either, so the formula of
/root/reference/src/solving-structure.hpp:173-462 (residual = rows 0..2 of C - A P T, with
P and/or T dropped for p* / t*; 7-vector -> matrix by hpp:32-58) is evaluated here
independently of the C++ oracle and the HIP kernel: mpmath at 100 digits, Jacobian by
central differences with h = 1e-30 (no hand-derived derivative formula).

Run:  python oracle/gen_golden_ae.py   (deterministic; seed fixed)
"""
import json
import os
import random

import mpmath as mp

from gen_golden import quat_to_rot, rand_quat

mp.mp.dps = 100
H = mp.mpf(10) ** -30


def mat4(p):
    R = quat_to_rot(p[:4])
    return [R[0] + [p[4]], R[1] + [p[5]], R[2] + [p[6]], [mp.mpf(0), mp.mpf(0), mp.mpf(0), mp.mpf(1)]]


def mul(A, B):
    return [[sum(A[i][k] * B[k][j] for k in range(4)) for j in range(4)] for i in range(4)]


def residual(cam, pat, tim, a7):
    M = mat4(a7)
    if pat is not None:
        M = mul(M, mat4(pat))
    if tim is not None:
        M = mul(M, mat4(tim))
    Cm = mat4(cam)
    return [Cm[i][j] - M[i][j] for i in range(3) for j in range(4)]


def main():
    rng = random.Random(0xA1CE)
    out = []
    for idx in range(16):
        case = idx % 4  # 0: c,p,t  1: p==p*  2: t==t*  3: both
        cam = rand_quat(rng) + [rng.uniform(-900, 900) for _ in range(3)]
        pat = rand_quat(rng) + [rng.uniform(-500, 500) for _ in range(3)]
        tim = rand_quat(rng) + [rng.uniform(-500, 500) for _ in range(3)]
        a7 = rand_quat(rng) + [rng.uniform(-900, 900) for _ in range(3)]
        use_p, use_t = case in (0, 2), case in (0, 1)

        def fun(params):
            return residual(params[0:7], params[7:14] if use_p else None, params[14:21] if use_t else None,
                            [mp.mpf(v) for v in a7])

        base = [mp.mpf(v) for v in cam + pat + tim]
        r = fun(base)
        J = [[mp.mpf(0)] * 21 for _ in range(12)]
        for k in range(21):
            blk = k // 7
            if (blk == 1 and not use_p) or (blk == 2 and not use_t):
                continue
            hi, lo = list(base), list(base)
            hi[k] += H
            lo[k] -= H
            rh, rl = fun(hi), fun(lo)
            for rr in range(12):
                J[rr][k] = (rh[rr] - rl[rr]) / (2 * H)
        out.append({"case": case, "camera": cam, "pattern": pat, "time": tim, "a7": a7,
                    "r": [mp.nstr(v, 40) for v in r], "J": [[mp.nstr(v, 40) for v in row] for row in J]})
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ae_golden.json")
    with open(dst, "w") as f:
        json.dump({"generator": "oracle/gen_golden_ae.py", "dps": 100, "h": "1e-30", "vectors": out}, f, indent=1)
    print("wrote", os.path.normpath(dst), len(out), "vectors")


if __name__ == "__main__":
    main()
