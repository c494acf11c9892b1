"""CPU tests of the point-reconstruction widening:
 * the oracle's Jet functor and the kernels' closed-form header (csrc/rae_model.hpp on the
   host) against mpmath golden vectors (tests/golden/rae_golden.json,
   oracle/gen_golden_rae.py) — the reference holds no fixtures for this path;
 * the DLT rows of the header against a numpy restatement of solving-structure.cpp:324-346;
 * the oracle end to end through properties (noise-free recovery of the model corners,
   corners with fewer than two views left at zero).  PARITY UNPINNED for the minimiser."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from calico_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "rae_golden.json")))["vectors"]
dp = C.POINTER(C.c_double)


def _arr(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _golden(g):
    return (np.array([float(v) for v in g["r"]]), np.array([[float(v) for v in row] for row in g["J"]]))


@pytest.mark.parametrize("idx", range(len(GOLDEN)))
def test_oracle_functor_matches_rae_golden(oracle_api, idx):
    g = GOLDEN[idx]
    f = oracle_api.lib.oracle_rae_eval
    f.argtypes = [dp, dp, dp, C.c_double, C.c_double, dp, dp]
    f.restype = None
    K, A, X = _arr(g["intrinsics"]), _arr(g["a_hat"]), _arr(g["point"])
    r, J = np.zeros(2), np.zeros(6)
    f(K.ctypes.data_as(dp), A.ctypes.data_as(dp), X.ctypes.data_as(dp), g["uv"][0], g["uv"][1], r.ctypes.data_as(dp),
      J.ctypes.data_as(dp))
    rg, Jg = _golden(g)
    assert np.abs(r - rg).max() <= 1e-12 * np.abs(rg).max()
    assert np.abs(J.reshape(2, 3) - Jg).max() <= 1e-12 * np.abs(Jg).max()


@pytest.fixture(scope="module")
def model_host():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "helpers")])
    L = C.CDLL(os.path.join(ROOT, "tests", "_build", "libmodel_host.so"))
    L.rae_model_host_eval.argtypes = [dp, dp, dp, C.c_double, C.c_double, dp, dp]
    L.rae_model_host_dlt_rows.argtypes = [dp, dp, C.c_double, C.c_double, C.c_double, C.c_double, dp, dp]
    return L


@pytest.mark.parametrize("idx", range(len(GOLDEN)))
def test_kernel_arithmetic_matches_rae_golden_on_host(model_host, idx):
    g = GOLDEN[idx]
    K, A, X = _arr(g["intrinsics"]), _arr(g["a_hat"]), _arr(g["point"])
    r, J = np.zeros(2), np.zeros(6)
    model_host.rae_model_host_eval(A.ctypes.data_as(dp), K.ctypes.data_as(dp), X.ctypes.data_as(dp), g["uv"][0],
                                   g["uv"][1], r.ctypes.data_as(dp), J.ctypes.data_as(dp))
    rg, Jg = _golden(g)
    assert np.abs(r - rg).max() <= 1e-12 * np.abs(rg).max()
    assert np.abs(J.reshape(2, 3) - Jg).max() <= 1e-12 * np.abs(Jg).max()


def test_dlt_rows_match_the_reference_construction(model_host):
    """solving-structure.cpp:324-346 with numpy matrices: T, P = T K A_hat, x = T (u, v, 1)."""
    rng = np.random.default_rng(3)
    for g in GOLDEN:
        K12, A = _arr(g["intrinsics"]), _arr(g["a_hat"]).reshape(3, 4)
        cols, rows = 1200.0, 900.0
        u, v = rng.uniform(0, cols), rng.uniform(0, rows)
        T = np.array([[2.0 / cols, 0, -1], [0, 2.0 / rows, -1], [0, 0, 1]])
        K = np.array([[K12[0], 0, K12[1]], [0, K12[2], K12[3]], [0, 0, 1]])
        P = T @ (K @ A)
        xc = T @ np.array([u, v, 1.0])
        want1, want2 = xc[0] * P[2] - P[0], xc[1] * P[2] - P[1]
        r1, r2 = np.zeros(4), np.zeros(4)
        model_host.rae_model_host_dlt_rows(_arr(A).ctypes.data_as(dp), K12.ctypes.data_as(dp), cols, rows, u, v,
                                           r1.ctypes.data_as(dp), r2.ctypes.data_as(dp))
        scale = max(np.abs(want1).max(), np.abs(want2).max())
        assert np.abs(r1 - want1).max() <= 1e-13 * scale and np.abs(r2 - want2).max() <= 1e-13 * scale


def test_oracle_recovers_the_model_corners_without_noise(oracle_api):
    cfg = synth.custom(4, 2, 8, noise_px=0.0)
    xyz, has, its, fc, s = capi.rae_reconstruct(*synth.rae_inputs(cfg), api=oracle_api)
    pts = synth.points(cfg)
    assert has.all() and s.n_solved == len(pts)
    # pixels are rounded to float32 (camera-calibration.cpp:257-258): ~1e-5 px, ~1e-4 mm
    assert np.abs(xyz - pts).max() < 1e-3
    assert its.max() <= 10


def test_oracle_dlt_only_and_unseen_corners(oracle_api):
    cfg = synth.custom(3, 2, 6)
    args = list(synth.rae_inputs(cfg))
    obs_fr, obs_point, u, v = args[6:10]
    # corner 5 keeps a single view, corner 7 none
    first5 = np.flatnonzero(obs_point == 5)[0]
    keep = ~(((obs_point == 5) & (np.arange(obs_point.size) != first5)) | (obs_point == 7))
    args[6:10] = obs_fr[keep], obs_point[keep], u[keep], v[keep]
    xyz, has, its, _fc, _s = capi.rae_reconstruct(*args, api=oracle_api)
    assert not has[5] and not has[7] and has.sum() == has.size - 2
    assert not xyz[5].any() and not xyz[7].any() and its[5] == 0
    dlt, has0, its0, _, _ = capi.rae_reconstruct(*args, options=capi.default_options(max_num_iterations=0), api=oracle_api)
    assert np.array_equal(has0, has)
    pts = synth.points(cfg)
    assert np.abs(dlt[has] - pts[has]).max() < 5.0      # DLT ignores distortion: millimetres off
    assert np.abs(xyz[has] - pts[has]).max() < 1.0      # refined: pixel noise 0.1 px
    assert np.abs(xyz[has] - pts[has]).max() < np.abs(dlt[has] - pts[has]).max()
