"""CPU tests of the Stage-4 algebraic-error widening:
 * the oracle's dual-number AE functor and the kernels' closed-form header
   (calico_amd/csrc/ae_model.hpp, compiled for the host) against the mpmath golden
   vectors tests/golden/ae_golden.json (oracle/gen_golden_ae.py) — the reference holds no
   fixtures for this path, so these pin the r/J arithmetic;
 * the oracle's LM loop on AE relations through properties (noise-free recovery, exemplars
   fixed): PARITY UNPINNED for the minimiser, as for the reprojection path."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from calico_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "ae_golden.json")))["vectors"]


def golden_problem(api, g):
    """One relation on a 1-camera / 2-pattern / 2-time layout; p* = var 1, t* = var 3."""
    case = g["case"]
    pv = 1 if case in (1, 3) else 2
    tv = 3 if case in (2, 3) else 4
    P = capi.AeProblem(1, 2, 2, 1, 3, api=api)
    P.add_relations([0], [pv], [tv], [g["a7"]])
    x = np.zeros(35)
    x[0::7] = 1
    x[0:7] = g["camera"]
    if case in (0, 2):
        x[7 * pv:7 * pv + 7] = g["pattern"]
    if case in (0, 1):
        x[7 * tv:7 * tv + 7] = g["time"]
    return P, x


def golden_rj(g):
    return (np.array([float(v) for v in g["r"]]), np.array([[float(v) for v in row] for row in g["J"]]))


@pytest.mark.parametrize("idx", range(len(GOLDEN)))
def test_oracle_matches_ae_golden(oracle_api, idx):
    g = GOLDEN[idx]
    P, x = golden_problem(oracle_api, g)
    cost, r, J = P.eval(x)
    rg, Jg = golden_rj(g)
    assert np.abs(r[0] - rg).max() <= 1e-12 * np.abs(rg).max()
    assert np.abs(J[0] - Jg).max() <= 1e-12 * np.abs(Jg).max()
    assert abs(cost - 0.5 * (rg ** 2).sum()) <= 1e-12 * cost
    if g["case"] in (1, 3):
        assert not J[0][:, 7:14].any()
    if g["case"] in (2, 3):
        assert not J[0][:, 14:21].any()


@pytest.fixture(scope="module")
def model_host():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "helpers")])
    L = C.CDLL(os.path.join(ROOT, "tests", "_build", "libmodel_host.so"))
    dp = C.POINTER(C.c_double)
    L.ae_model_host_eval.argtypes = [dp, dp, dp, C.c_int, C.c_int, dp, dp, dp]
    return L


def host_eval(L, cam, pat, tim, has_p, has_t, a7):
    dp = C.POINTER(C.c_double)
    a = [np.ascontiguousarray(t, dtype=np.float64) for t in (cam, pat, tim, a7)]
    r, J = np.zeros(12), np.zeros(252)
    L.ae_model_host_eval(a[0].ctypes.data_as(dp), a[1].ctypes.data_as(dp), a[2].ctypes.data_as(dp), int(has_p),
                         int(has_t), a[3].ctypes.data_as(dp), r.ctypes.data_as(dp), J.ctypes.data_as(dp))
    return r, J.reshape(12, 21)


@pytest.mark.parametrize("idx", range(len(GOLDEN)))
def test_kernel_arithmetic_matches_ae_golden_on_host(model_host, idx):
    """calico_amd/csrc/ae_model.hpp (the code inside the AE kernels) vs mpmath, 1e-12."""
    g = GOLDEN[idx]
    r, J = host_eval(model_host, g["camera"], g["pattern"], g["time"], g["case"] in (0, 2), g["case"] in (0, 1), g["a7"])
    rg, Jg = golden_rj(g)
    assert np.abs(r - rg).max() <= 1e-12 * np.abs(rg).max()
    assert np.abs(J - Jg).max() <= 1e-12 * np.abs(Jg).max()


@pytest.mark.parametrize("idx", range(len(GOLDEN)))
def test_column_wise_evaluation_is_the_same_arithmetic(model_host, idx):
    """ae_model.hpp: ae_eval_column (what four lanes per relation run in ae_records_kernel), the
    four matrix columns put together, against ae_eval_prepared: bit for bit."""
    g = GOLDEN[idx]
    has_p, has_t = g["case"] in (0, 2), g["case"] in (0, 1)
    r, J = host_eval(model_host, g["camera"], g["pattern"], g["time"], has_p, has_t, g["a7"])
    dp = C.POINTER(C.c_double)
    a = [np.ascontiguousarray(t, dtype=np.float64) for t in (g["camera"], g["pattern"], g["time"], g["a7"])]
    M = np.full(12 * 22, np.nan)
    model_host.ae_model_host_eval_columns(a[0].ctypes.data_as(dp), a[1].ctypes.data_as(dp), a[2].ctypes.data_as(dp),
                                          int(has_p), int(has_t), a[3].ctypes.data_as(dp), M.ctypes.data_as(dp))
    M = M.reshape(12, 22)
    assert np.array_equal(M[:, :21], J)
    assert np.array_equal(M[:, 21], r)


def test_kernel_arithmetic_matches_oracle_on_synthetic_relations(model_host, oracle_api):
    cfg = synth.custom(3, 2, 4)
    cam, pat, tim, a7, _xt, x0 = synth.ae_relations(cfg)
    P = capi.AeProblem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, api=oracle_api)
    P.add_relations(cam, pat, tim, a7)
    _, ro, Jo = P.eval(x0)
    xb = x0.reshape(-1, 7)
    seen = set()
    for i in range(cam.size):
        hp, ht = pat[i] != cfg.p_star_var, tim[i] != cfg.t_star_var
        seen.add((bool(hp), bool(ht)))
        r, J = host_eval(model_host, xb[cam[i]], xb[pat[i]], xb[tim[i]], hp, ht, a7[i])
        assert np.abs(r - ro[i]).max() <= 1e-12 * max(1.0, np.abs(ro[i]).max())
        assert np.abs(J - Jo[i]).max() <= 1e-12 * np.abs(Jo[i]).max()
    assert len(seen) == 4  # all four functor cases occur


def test_ae_relations_are_consistent_at_the_truth():
    """Generator check: without noise, C - A P T vanishes at x_true."""
    cfg = synth.custom(3, 2, 4)
    cam, pat, tim, a7, xt, _ = synth.ae_relations(cfg, noise_deg=0.0, noise_mm=0.0)
    xb = xt.reshape(-1, 7)
    for i in range(cam.size):
        A, Cm = synth.quat_to_matrix(a7[i]), synth.quat_to_matrix(xb[cam[i]])
        M = A @ synth.quat_to_matrix(xb[pat[i]]) @ synth.quat_to_matrix(xb[tim[i]])
        assert np.abs(Cm - M).max() < 1e-9


def test_oracle_ae_noise_free_recovery(oracle_api):
    cfg = synth.custom(3, 2, 5, fr_drop_prob=0.0)
    cam, pat, tim, a7, xt, x0 = synth.ae_relations(cfg, noise_deg=0.0, noise_mm=0.0)
    P = capi.AeProblem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, api=oracle_api)
    P.add_relations(cam, pat, tim, a7)
    opt = capi.default_options(max_num_iterations=50, function_tolerance=1e-16, parameter_tolerance=1e-14)
    x, s = P.solve(x0, opt)
    assert s.final_cost < 1e-12
    xr, gr = x.reshape(-1, 7), xt.reshape(-1, 7)
    for v in range(cfg.vn):
        assert np.abs(synth.quat_to_matrix(xr[v]) - synth.quat_to_matrix(gr[v])).max() < 1e-5
    for v in (cfg.p_star_var, cfg.t_star_var):
        assert np.array_equal(xr[v], x0.reshape(-1, 7)[v])


def test_ae_entry_points_refuse_the_wrong_handle_kind(oracle_api):
    P = capi.Problem(1, 2, 2, 1, 3, api=oracle_api)
    with pytest.raises(capi.RpError):
        capi.AeProblem.add_relations(P, [0], [2], [4], [[1, 0, 0, 0, 0, 0, 0]])
