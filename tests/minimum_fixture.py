"""Shared by the CPU (oracle) and GPU tests: the converged states that
oracle/gen_golden_minimum.py found with scipy.optimize.least_squares on the mpmath-checked
residual, i.e. without any of the Levenberg-Marquardt code under test."""
import os

import numpy as np

from calico_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["tiny", "small", "refined"]
TIGHT = dict(max_num_iterations=200, function_tolerance=1e-16, parameter_tolerance=1e-14, gradient_tolerance=1e-14)


def load(name):
    return np.load(os.path.join(ROOT, "tests", "golden", f"rp_minimum_{name}.npz"))


def solve_and_compare(name, api=None, device=0):
    """Solve the fixture's network from its x0 with `api` (None: the HIP library) at tight
    tolerances; the result must be the independent minimum: RMS within 1e-6 px, every
    R(q) within 1e-6 rad and t within 1e-6 mm (BASELINE.json north_star), refined focal
    lengths / principal points within 1e-5 px."""
    F = load(name)
    cn, pn, tn = int(F["cn"]), int(F["pn"]), int(F["tn"])
    P = capi.Problem(cn, pn, tn, int(F["p_star_var"]), int(F["t_star_var"]), device=device, api=api)
    P.set_intrinsics(F["intrinsics0"])
    P.set_points(F["points"])
    P.add_observations(F["cam"], F["pat"], F["tim"], F["corner"], F["u"], F["v"])
    mask = int(F["mask"])
    if mask:
        P.set_refine_intrinsics(mask)
    x, s = P.solve(F["x0"], capi.default_options(**TIGHT))
    n = F["cam"].size
    rms = np.sqrt(s.final_cost / n)
    assert abs(rms - float(F["rms"])) < 1e-6, (rms, float(F["rms"]))
    # the same minimum VALUE to rounding (1e-14 seen): neither a worse nor a different minimum
    assert abs(s.final_cost - float(F["half_sum_r2"])) <= 1e-12 * float(F["half_sum_r2"])
    xs = F["x_star"].reshape(-1, 7)
    xr = x.reshape(-1, 7)
    rot = tr = 0.0
    for v in range(cn + pn + tn):
        qa, qb = xr[v, :4] / np.linalg.norm(xr[v, :4]), xs[v, :4] / np.linalg.norm(xs[v, :4])
        qb = qb if float(qa @ qb) >= 0 else -qb
        rot = max(rot, 2.0 * np.arcsin(min(1.0, 0.5 * np.linalg.norm(qa - qb))))
        tr = max(tr, np.abs(xr[v, 4:] - xs[v, 4:]).max())
    dk = None
    if mask:
        K, Ks = P.get_intrinsics(), F["intrinsics_star"]
        dk = np.abs(K - Ks).max(0)
    P.close()
    return rot, tr, dk


AE_NAMES = ["tiny", "small"]


def ae_solve_and_compare(name, api=None, device=0):
    """Stage 4: the fixture's relations solved from its x0 with `api` (None: the HIP library) at
    tight tolerances against the minimum oracle/gen_golden_minimum_ae.py found with scipy: the
    same minimum value, every R(q) and t of the solution."""
    F = np.load(os.path.join(ROOT, "tests", "golden", f"ae_minimum_{name}.npz"))
    cn, pn, tn = int(F["cn"]), int(F["pn"]), int(F["tn"])
    P = capi.AeProblem(cn, pn, tn, int(F["p_star_var"]), int(F["t_star_var"]), device=device, api=api)
    P.add_relations(F["cam"], F["pat"], F["tim"], F["a7"])
    x, s = P.solve(F["x0"], capi.default_options(**TIGHT))
    assert abs(s.final_cost - float(F["half_sum_r2"])) <= 1e-11 * float(F["half_sum_r2"]), (s.final_cost, float(F["half_sum_r2"]))
    xs, xr = F["x_star"].reshape(-1, 7), x.reshape(-1, 7)
    rot = tr = 0.0
    for v in range(cn + pn + tn):
        qa, qb = xr[v, :4] / np.linalg.norm(xr[v, :4]), xs[v, :4] / np.linalg.norm(xs[v, :4])
        qb = qb if float(qa @ qb) >= 0 else -qb
        rot = max(rot, 2.0 * np.arcsin(min(1.0, 0.5 * np.linalg.norm(qa - qb))))
        tr = max(tr, np.abs(xr[v, 4:] - xs[v, 4:]).max())
    P.close()
    return rot, tr


def rae_solve_and_compare(name, api=None, device=0):
    """Point reconstruction: the fixture's views through calico_rae_reconstruct (api None: the HIP
    library) at tight tolerances against the per-corner minima oracle/gen_golden_minimum_rae.py
    found with scipy.  Returns (largest and median corner distance in mm, largest relative
    difference of the per-corner minimum values)."""
    F = np.load(os.path.join(ROOT, "tests", "golden", f"rae_minimum_{name}.npz"))
    opt = capi.default_options(max_num_iterations=100, function_tolerance=1e-15, parameter_tolerance=1e-13,
                               gradient_tolerance=1e-14)
    xyz, has, _its, cost, _s = capi.rae_reconstruct(F["cam_params"], F["image_cols_rows"], F["fr_cam"], F["fr_a_hat"],
                                                    int(F["n_points"]), int(F["corners_per_pattern"]), F["obs_fr"],
                                                    F["obs_point"], F["u"], F["v"], options=opt, device=device, api=api)
    want = F["has_values"].astype(bool)
    assert np.array_equal(has, want)
    d = np.abs(xyz[want] - F["x_star"][want]).max(axis=1)
    dc = np.abs(cost[want] - F["cost"][want]) / np.maximum(F["cost"][want], 1e-300)
    return d.max(), np.median(d), dc.max()
