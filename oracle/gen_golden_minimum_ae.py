#!/usr/bin/env python3
"""Generate tests/golden/ae_minimum_*.npz — converged states of the Stage-4 algebraic-error
problem found WITHOUT the oracle's or the library's Levenberg-Marquardt code (TEST
INFRASTRUCTURE; see oracle/README.md and oracle/gen_golden_minimum.py, whose reasoning applies
unchanged: the minimum of 1/2 sum r^2 does not depend on the algorithm that finds it).

Residual: rows 0..2 of C - A P T per relation (/root/reference/src/solving-structure.hpp:173-462,
P / T dropped for p* / t*), the formula oracle/gen_golden_ae.py evaluates in mpmath, here in
complex128 over all relations at once so that the Jacobian is a complex-step derivative.  The
float64 evaluation is first checked against the mpmath vectors of tests/golden/ae_golden.json.
scipy.optimize.least_squares (trust-region reflective, exact sub-problems) minimises it; a few
minimum-norm Gauss-Newton steps polish the result (the quaternion scales are null directions).

Run:  python oracle/gen_golden_minimum_ae.py     (build container only: needs scipy)
"""
import json
import os
import sys

import numpy as np
from scipy.optimize import least_squares

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))

from calico_amd import synth  # noqa: E402  (inputs only)


def mats(p):
    """[N, 7] (any arithmetic type) -> [N, 4, 4]: ceres::QuaternionToRotation (scaled rotation
    over |q|^2, hpp:32-58) and the translation."""
    a, b, c, d = p[:, 0], p[:, 1], p[:, 2], p[:, 3]
    n = a * a + b * b + c * c + d * d
    M = np.zeros((p.shape[0], 4, 4), dtype=p.dtype)
    M[:, 0, 0] = (a * a + b * b - c * c - d * d) / n
    M[:, 0, 1] = 2 * (b * c - a * d) / n
    M[:, 0, 2] = 2 * (a * c + b * d) / n
    M[:, 1, 0] = 2 * (a * d + b * c) / n
    M[:, 1, 1] = (a * a - b * b + c * c - d * d) / n
    M[:, 1, 2] = 2 * (c * d - a * b) / n
    M[:, 2, 0] = 2 * (b * d - a * c) / n
    M[:, 2, 1] = 2 * (a * b + c * d) / n
    M[:, 2, 2] = (a * a - b * b - c * c + d * d) / n
    M[:, :3, 3] = p[:, 4:7]
    M[:, 3, 3] = 1
    return M


def residuals(x, cam, pat, tim, a7):
    """x: [vn, 7] with the identity block at p* / t* (so that dropping the factor and multiplying
    by it are the same thing); returns [N, 12]."""
    C, P, T, A = mats(x[cam]), mats(x[pat]), mats(x[tim]), mats(a7.astype(x.dtype))
    M = np.einsum("nij,njk,nkl->nil", A, P, T)
    return (C - M)[:, :3, :].reshape(-1, 12)


def check_against_golden():
    G = json.load(open(os.path.join(HERE, "..", "tests", "golden", "ae_golden.json")))["vectors"]
    ident = [1.0, 0, 0, 0, 0, 0, 0]
    for g in G:
        case = g["case"]
        x = np.array([g["camera"], g["pattern"] if case in (0, 2) else ident, g["time"] if case in (0, 1) else ident])
        r = residuals(x, np.array([0]), np.array([1]), np.array([2]), np.array([g["a7"]]))[0]
        want = np.array([float(s) for s in g["r"]])
        assert np.abs(r - want).max() <= 1e-12 * np.abs(want).max()


def solve_one(name, cfg, **kw):
    cam, pat, tim, a7, _xt, x0 = synth.ae_relations(cfg, **kw)
    free = [b for b in range(cfg.vn) if b not in (cfg.p_star_var, cfg.t_star_var)]

    def unpack(p):
        x = np.zeros((cfg.vn, 7), dtype=p.dtype)
        x[:, 0] = 1.0
        x[free] = p.reshape(-1, 7)
        return x

    def fun(p):
        return residuals(unpack(p), cam, pat, tim, a7).reshape(-1)

    p0 = x0.reshape(-1, 7)[free].reshape(-1)
    res = least_squares(fun, p0, jac="cs", method="trf", tr_solver="exact", x_scale="jac", ftol=1e-15, xtol=1e-15,
                        gtol=1e-15, max_nfev=300)
    p = res.x
    for _ in range(5):
        r = fun(p)
        J = np.empty((r.size, p.size))
        for k in range(p.size):
            ph = p.astype(np.complex128)
            ph[k] += 1e-30j
            J[:, k] = fun(ph).imag / 1e-30
        step = np.linalg.lstsq(J, -r, rcond=1e-12)[0]
        if 0.5 * (fun(p + step) ** 2).sum() <= 0.5 * (r ** 2).sum():
            p = p + step
        if np.abs(step).max() < 1e-11:
            break
    r = fun(p)
    gmax = float(np.abs(J.T @ r).max())
    dst = os.path.join(HERE, "..", "tests", "golden", f"ae_minimum_{name}.npz")
    np.savez_compressed(dst, cn=cfg.cn, pn=cfg.pn, tn=cfg.tn, p_star_var=cfg.p_star_var, t_star_var=cfg.t_star_var,
                        cam=cam, pat=pat, tim=tim, a7=a7, x0=x0, x_star=unpack(p).reshape(-1),
                        half_sum_r2=0.5 * float((r ** 2).sum()), gradient_max=gmax)
    print(f"{name}: {cam.size} relations, {p.size} parameters, nfev {res.nfev}, 1/2 sum r^2 {0.5 * (r ** 2).sum():.12e}, "
          f"|g|max {gmax:.2e}, {os.path.getsize(dst) / 1024:.0f} KB")


def main():
    check_against_golden()
    solve_one("tiny", synth.custom(3, 2, 4))
    solve_one("small", synth.custom(6, 3, 12, fr_drop_prob=0.2), noise_deg=0.5, noise_mm=3.0)


if __name__ == "__main__":
    main()
