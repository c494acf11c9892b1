#!/usr/bin/env python3
"""Generate tests/golden/rp_golden.json — high-precision golden vectors for the
reprojection residual and its Jacobian (TEST INFRASTRUCTURE; see oracle/README.md).

The reference holds no golden vectors or tests for this path and
cannot be built here (no Ceres/Eigen/OpenCV), so the residual formula of
/root/reference/src/solving-structure.hpp:658-790 (cases :820-1310) is evaluated here
independently of both the C++ oracle and the HIP kernel:
  * in mpmath at 100 significant digits,
  * the Jacobian by central differences with h = 1e-30 (truncation ~1e-60), i.e.
    with no hand-derived derivative formula at all.
Inputs are exact doubles (and float32 model points, as the reference's cv::Point3f),
outputs are written as 40-digit decimal strings.

Run:  python oracle/gen_golden.py   (deterministic; seed fixed)
"""
import json
import os
import random
import struct

import mpmath as mp

mp.mp.dps = 100
H = mp.mpf(10) ** -30


def f32(x):
    return struct.unpack("f", struct.pack("f", x))[0]


def quat_to_rot(q):
    # ceres::QuaternionToRotation: scaled rotation / |q|^2   
    a, b, c, d = q
    n = a * a + b * b + c * c + d * d
    R = [
        [a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (a * c + b * d)],
        [2 * (a * d + b * c), a * a - b * b + c * c - d * d, 2 * (c * d - a * b)],
        [2 * (b * d - a * c), 2 * (a * b + c * d), a * a - b * b - c * c + d * d],
    ]
    return [[R[i][j] / n for j in range(3)] for i in range(3)]


def apply7(p, X):
    R = quat_to_rot(p[:4])
    return [sum(R[i][j] * X[j] for j in range(3)) + p[4 + i] for i in range(3)]


def residual(cam, pat, tim, K, uv, X):
    """A_hat = C * T * P (solving-structure.hpp:733-734): x_cam = C(T(P(X)))."""
    z = apply7(pat, X) if pat is not None else list(X)
    y = apply7(tim, z) if tim is not None else z
    xc = apply7(cam, y)
    fx, cx, fy, cy, k1, k2, p1, p2, k3, k4, k5, k6 = K
    xn, yn = xc[0] / xc[2], xc[1] / xc[2]
    r2 = xn * xn + yn * yn
    numer = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
    denom = 1 + k4 * r2 + k5 * r2 ** 2 + k6 * r2 ** 3
    xpp = xn * numer / denom + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn)
    ypp = yn * numer / denom + 2 * p2 * xn * yn + p1 * (r2 + 2 * yn * yn)
    return [xpp * fx + cx - uv[0], ypp * fy + cy - uv[1]]


def rand_quat(rng, small=False):
    # random rotation of up to ~40 deg (or arbitrary), non-unit norm in [0.5, 2]
    ax = [rng.gauss(0, 1) for _ in range(3)]
    n = sum(a * a for a in ax) ** 0.5
    ang = rng.uniform(-0.7, 0.7) if small else rng.uniform(-3.0, 3.0)
    import math

    q = [math.cos(ang / 2)] + [math.sin(ang / 2) * a / n for a in ax]
    s = rng.uniform(0.5, 2.0)
    return [s * v for v in q]


def make_case(rng, idx):
    case = idx % 4  # 0: c,p,t  1: p==p*  2: t==t*  3: both
    cam = rand_quat(rng, small=True) + [rng.uniform(-200, 200), rng.uniform(-200, 200), rng.uniform(1500, 3000)]
    pat = rand_quat(rng) + [rng.uniform(-300, 300) for _ in range(3)]
    tim = rand_quat(rng, small=True) + [rng.uniform(-150, 150) for _ in range(3)]
    K = [rng.uniform(900, 1600), rng.uniform(500, 700), rng.uniform(900, 1600), rng.uniform(350, 550),
         rng.uniform(-0.3, 0.3), rng.uniform(-0.2, 0.2), 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
    if idx % 3 == 1:  # tangential + k3
        K[6], K[7], K[8] = rng.uniform(-2e-3, 2e-3), rng.uniform(-2e-3, 2e-3), rng.uniform(-0.05, 0.05)
    if idx % 5 == 2:  # rational model terms k4..k6
        K[9], K[10], K[11] = rng.uniform(-0.1, 0.1), rng.uniform(-0.05, 0.05), rng.uniform(-0.01, 0.01)
    X = [f32(rng.uniform(0, 400)), f32(rng.uniform(0, 300)), f32(0.0 if idx % 2 == 0 else rng.uniform(-50, 50))]
    uv = [f32(rng.uniform(0, 1200)), f32(rng.uniform(0, 900))]
    return case, cam, pat, tim, K, uv, X


def main():
    rng = random.Random(0xCA11C0)
    out = []
    for idx in range(32):
        case, cam, pat, tim, K, uv, X = make_case(rng, idx)
        use_p = case in (0, 2)
        use_t = case in (0, 1)

        def fun(params):
            c = params[0:7]
            p = params[7:14] if use_p else None
            t = params[14:21] if use_t else None
            return residual(c, p, t, [mp.mpf(k) for k in K], [mp.mpf(a) for a in uv], [mp.mpf(a) for a in X])

        base = [mp.mpf(v) for v in cam + pat + tim]
        r = fun(base)
        J = [[mp.mpf(0)] * 21 for _ in range(2)]
        for k in range(21):
            blk = k // 7
            if (blk == 1 and not use_p) or (blk == 2 and not use_t):
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
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "rp_golden.json")
    with open(dst, "w") as f:
        json.dump({"generator": "oracle/gen_golden.py", "dps": 100, "h": "1e-30", "vectors": out}, f, indent=1)
    print("wrote", os.path.normpath(dst), len(out), "vectors")


if __name__ == "__main__":
    main()
