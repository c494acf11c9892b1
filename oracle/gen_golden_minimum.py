#!/usr/bin/env python3
"""Generate tests/golden/rp_minimum_*.npz — converged states found WITHOUT the oracle's or the
library's Levenberg-Marquardt code (TEST INFRASTRUCTURE; see oracle/README.md).

ceres::Solve is an un-vendored dependency of the reference and its iterate sequence stays
"parity unpinned" (DESIGN.md 5): oracle/rp_oracle.cpp and calico_amd/csrc/rp_api.hip both
restate it.  The MINIMUM of 1/2 sum r^2 does not depend on the algorithm, so it can be pinned
independently: scipy.optimize.least_squares (trust-region reflective, exact SVD sub-problems)
minimises the residual of /root/reference/src/solving-structure.hpp:658-790 as restated in
oracle/gen_golden.py (the function the mpmath golden vectors come from, evaluated here in
complex128 so that the Jacobian is a complex-step derivative: no hand-derived formula and no
code shared with the oracle's Jets or the kernels).  The exemplar pattern and time are not
parameters (solving-structure.cpp:528-544); quaternions are free 4-vectors, as in the reference
(no manifold): their scale is a gauge, so the tests compare R(q) and t.

Each file holds the inputs (so that nothing depends on the generator staying bit-stable) and
the minimiser's x*, refined intrinsics (where refined) and RMS.

Run:  python oracle/gen_golden_minimum.py     (build container only: needs scipy; ~1 min)
"""
import os
import sys

import numpy as np
from scipy.optimize import least_squares

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
from gen_golden import residual  # noqa: E402  (generic in the arithmetic type)

from calico_amd import synth  # noqa: E402  (inputs only)


def make_fun(cfg, obs, pts, K0, mask):
    cam, pat, tim, cor, u, v = obs
    free = [b for b in range(cfg.vn) if b not in (cfg.p_star_var, cfg.t_star_var)]
    kcols = [k for k in range(12) if (mask >> k) & 1]
    X = [pts[cor, 0].astype(np.float64), pts[cor, 1].astype(np.float64), pts[cor, 2].astype(np.float64)]
    nb = 7 * len(free)

    def unpack(p):
        x = np.zeros((cfg.vn, 7), dtype=p.dtype)
        x[:, 0] = 1.0
        x[free] = p[:nb].reshape(-1, 7)
        K = K0.astype(p.dtype).copy()
        if kcols:
            K[:, kcols] = p[nb:].reshape(cfg.cn, len(kcols))
        return x, K

    def fun(p):
        x, K = unpack(p)
        c, pp, tt, kk = x[cam], x[pat], x[tim], K[cam]
        # p* / t* rows are the identity block: R = I, t = 0 (hpp:861-874, 1028-1038, 1197-1204)
        r = residual([c[:, k] for k in range(7)], [pp[:, k] for k in range(7)], [tt[:, k] for k in range(7)],
                     [kk[:, k] for k in range(12)], [u, v], X)
        return np.concatenate([r[0], r[1]])

    def pack(x, K):
        return np.concatenate([x.reshape(-1, 7)[free].reshape(-1), K[:, kcols].reshape(-1)])

    return fun, pack, unpack


def check_against_golden():
    """The float64 evaluation of gen_golden.residual against its own mpmath vectors."""
    import json
    G = json.load(open(os.path.join(HERE, "..", "tests", "golden", "rp_golden.json")))["vectors"]
    for g in G:
        case = g["case"]
        r = residual(g["camera"], g["pattern"] if case in (0, 2) else None, g["time"] if case in (0, 1) else None,
                     g["intrinsics"], g["uv"], g["point"])
        want = [float(s) for s in g["r"]]
        assert max(abs(a - b) for a, b in zip(r, want)) <= 1e-12 * max(abs(w) for w in want)


def solve_one(name, cfg, mask, K0):
    obs = synth.observations(cfg)
    pts = synth.points(cfg)
    x0 = synth.initial_guess(cfg)
    fun, pack, unpack = make_fun(cfg, obs, pts, K0, mask)
    p0 = pack(x0, K0)
    res = least_squares(fun, p0, jac="cs", method="trf", tr_solver="exact", x_scale="jac", ftol=1e-15, xtol=1e-15,
                        gtol=1e-15, max_nfev=300)
    # polish: plain Gauss-Newton steps on the minimum-norm solution (the quaternion scales are
    # null directions of J) until the step is at rounding level
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
    x, K = unpack(p)
    r = fun(p)
    n = obs[0].size
    rms = float(np.sqrt(0.5 * (r ** 2).sum() / n))
    gmax = float(np.abs(J.T @ r).max())
    dst = os.path.join(HERE, "..", "tests", "golden", f"rp_minimum_{name}.npz")
    np.savez_compressed(dst, cn=cfg.cn, pn=cfg.pn, tn=cfg.tn, p_star_var=cfg.p_star_var, t_star_var=cfg.t_star_var,
                        cam=obs[0], pat=obs[1], tim=obs[2], corner=obs[3], u=obs[4], v=obs[5], points=pts,
                        intrinsics0=K0, mask=mask, x0=x0, x_star=x.reshape(-1), intrinsics_star=K, rms=rms,
                        gradient_max=gmax, half_sum_r2=0.5 * float((r ** 2).sum()))
    print(f"{name}: {n} observations, {p.size} parameters, nfev {res.nfev}, rms {rms:.12f} px, |g|max {gmax:.2e}, "
          f"{os.path.getsize(dst) / 1024:.0f} KB")


def main():
    check_against_golden()
    cfg = synth.custom(3, 2, 4)
    solve_one("tiny", cfg, 0, synth.intrinsics(cfg))
    cfg = synth.custom(5, 3, 10, fr_drop_prob=0.2, tangential=1)
    solve_one("small", cfg, 0, synth.intrinsics(cfg))
    # camera parameters refined (the extension): fx cx fy cy k1 k2 p1 p2 k3 from a perturbed start
    cfg = synth.custom(4, 2, 12, tangential=1)
    rng = np.random.default_rng(21)
    K0 = synth.intrinsics(cfg).copy()
    K0[:, [0, 2]] *= 1.0 + rng.uniform(-0.01, 0.01, size=(cfg.cn, 2))
    K0[:, [1, 3]] += rng.uniform(-3.0, 3.0, size=(cfg.cn, 2))
    K0[:, 4:6] += rng.uniform(-0.01, 0.01, size=(cfg.cn, 2))
    solve_one("refined", cfg, 0x1FF, K0)


if __name__ == "__main__":
    main()
