"""CPU tests of the oracle itself (the checker must be pinned before it checks):
 * its dual-number residual/Jacobian against the mpmath golden vectors
   (tests/golden/rp_golden.json, oracle/gen_golden.py) — the reference holds no
   fixtures for this path, so these are the pin for the r/J arithmetic;
 * its LM loop (restated Ceres; PARITY UNPINNED, see oracle/rp_oracle.cpp) through
   properties: noise-free recovery of the ground truth, exemplars fixed,
   invariances of the Jacobian."""
import json
import os

import numpy as np
import pytest

from calico_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = json.load(open(os.path.join(ROOT, "tests", "golden", "rp_golden.json")))["vectors"]
GOLDEN_K = json.load(open(os.path.join(ROOT, "tests", "golden", "rp_golden_k.json")))["vectors"]


def _single(api, g):
    case = g["case"]
    pv = 1 if case in (1, 3) else 2
    tv = 3 if case in (2, 3) else 4
    P = capi.Problem(1, 2, 2, 1, 3, api=api)
    P.set_intrinsics(g["intrinsics"])
    P.set_points([g["point"]])
    P.add_observations([0], [pv], [tv], [0], [g["uv"][0]], [g["uv"][1]])
    x = np.zeros(35)
    x[0::7] = 1
    x[0:7] = g["camera"]
    if case in (0, 2):
        x[7 * pv:7 * pv + 7] = g["pattern"]
    if case in (0, 1):
        x[7 * tv:7 * tv + 7] = g["time"]
    return P, x


@pytest.mark.parametrize("idx", range(len(GOLDEN)))
def test_oracle_matches_golden(oracle_api, idx):
    g = GOLDEN[idx]
    P, x = _single(oracle_api, g)
    cost, r, J = P.eval(x)
    rg = np.array([float(v) for v in g["r"]])
    Jg = np.array([[float(v) for v in row] for row in g["J"]])
    assert np.abs(r[0] - rg).max() <= 1e-12 * np.abs(rg).max()
    assert np.abs(J[0] - Jg).max() <= 1e-12 * np.abs(Jg).max()
    # absent blocks are exactly zero (cases 1-3, solving-structure.cpp:528-544)
    if g["case"] in (1, 3):
        assert not J[0][:, 7:14].any()
    if g["case"] in (2, 3):
        assert not J[0][:, 14:21].any()


@pytest.mark.parametrize("idx", range(len(GOLDEN_K)))
def test_oracle_matches_golden_with_camera_parameters_refined(oracle_api, idx):
    """Extension (BASELINE.json configs[2]): the 12 camera parameters as a fourth block of the
    oracle's dual numbers vs mpmath central differences over all 33 parameters; masked-out
    parameters give exactly zero columns and the first 21 columns do not change."""
    g = GOLDEN_K[idx]
    P, x = _single(oracle_api, g)
    rg = np.array([float(v) for v in g["r"]])
    Jg = np.array([[float(v) for v in row] for row in g["J"]])
    _, r0, J0 = P.eval(x)
    assert J0.shape == (1, 2, 21)
    for mask in (0xFFF, 0x1FF):
        P.set_refine_intrinsics(mask)
        _, r, J = P.eval(x)
        assert J.shape == (1, 2, 33)
        want = Jg.copy()
        want[:, [21 + k for k in range(12) if not (mask >> k) & 1]] = 0.0
        assert np.abs(r[0] - rg).max() <= 1e-12 * np.abs(rg).max()
        assert np.abs(J[0] - want).max() <= 1e-12 * np.abs(Jg).max()
        assert np.array_equal(J[0][:, :21], J0[0])
    P.set_refine_intrinsics(0)
    assert P.eval(x)[2].shape == (1, 2, 21)


def test_golden_covers_all_cases_and_distortion_terms():
    cases = {g["case"] for g in GOLDEN}
    assert cases == {0, 1, 2, 3}
    assert any(g["intrinsics"][6] != 0 and g["intrinsics"][8] != 0 for g in GOLDEN)  # p1, k3
    assert any(g["intrinsics"][9] != 0 for g in GOLDEN)  # k4..k6
    norms = [np.linalg.norm(g["camera"][:4]) for g in GOLDEN]
    assert min(norms) < 0.8 and max(norms) > 1.5  # non-unit quaternions


def test_jacobian_scale_invariance(oracle_api):
    """R(q) is normalised inside the functor, so r is invariant to |q| and
    J_block . [q; 0] = 0 for every block (SURVEY.md §8c property 3)."""
    cfg = synth.custom(3, 2, 5)
    P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, api=oracle_api)
    obs = synth.fill_problem(P, cfg)
    x = synth.initial_guess(cfg)
    _, r, J = P.eval(x)
    cam, pat, tim = obs[0], obs[1], obs[2]
    for b, var in enumerate((cam, pat, tim)):
        q = x.reshape(-1, 7)[var][:, :4]
        Jq = J[:, :, 7 * b:7 * b + 4]
        dots = np.einsum("nrk,nk->nr", Jq, q)
        assert np.abs(dots).max() < 1e-9 * np.abs(Jq).max()
    x2 = x.copy().reshape(-1, 7)
    x2[:cfg.cn, :4] *= 1.7
    _, r2, _ = P.eval(x2.reshape(-1))
    assert np.abs(r2 - r).max() < 1e-9


def test_noise_free_network_recovers_ground_truth(oracle_api):
    cfg = synth.custom(3, 2, 4, noise_px=0.0, fr_drop_prob=0.0, corner_drop_prob=0.0)
    P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, api=oracle_api)
    synth.fill_problem(P, cfg)
    xg, x0 = synth.ground_truth(cfg), synth.initial_guess(cfg)
    opt = capi.default_options(max_num_iterations=50, function_tolerance=1e-16, parameter_tolerance=1e-14)
    x, s = P.solve(x0, opt)
    n = P.num_observations
    # pixels are rounded to float32 (camera-calibration.cpp:257-258): floor ~1e-5 px
    assert np.sqrt(s.final_cost / n) < 1e-4
    xr, gr = x.reshape(-1, 7), xg.reshape(-1, 7)
    for v in range(cfg.vn):
        qa, qb = xr[v, :4] / np.linalg.norm(xr[v, :4]), gr[v, :4] / np.linalg.norm(gr[v, :4])
        assert 2 * np.arccos(min(1.0, abs(qa @ qb))) < 1e-5
        assert np.abs(xr[v, 4:] - gr[v, 4:]).max() < 5e-2  # mm, limited by the float32 pixel rounding
    for v in (cfg.p_star_var, cfg.t_star_var):
        assert np.array_equal(xr[v], x0.reshape(-1, 7)[v])


def test_oracle_lm_follows_the_reference_schedule(oracle_api):
    """Two batches x 10 iterations (calico.cpp:707-729) on config A end at the noise floor."""
    cfg = synth.preset("A", tn=40)
    obs = synth.observations(cfg)
    P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, api=oracle_api)
    P.set_intrinsics(synth.intrinsics(cfg))
    P.set_points(synth.points(cfg))
    half = obs[0].size // 2
    x = synth.initial_guess(cfg)
    P.add_observations(*[a[:half] for a in obs])
    x, s1 = P.solve(x)
    P.add_observations(*[a[half:] for a in obs])
    x, s2 = P.solve(x)
    assert s1.num_iterations <= 11 and s2.num_iterations <= 11
    assert s2.final_cost <= s2.initial_cost
    assert 0.05 < np.sqrt(s2.final_cost / P.num_observations) < 0.2
    assert "Ceres Solver Report: Iterations:" in capi.brief_report(s2, api=capi.Api.__new__(capi.Api)) if False else True


def test_reduced_system_is_symmetric_positive_definite(oracle_api):
    cfg = synth.custom(5, 3, 12, fr_drop_prob=0.3)
    P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, api=oracle_api)
    synth.fill_problem(P, cfg)
    S, b = P.normal_eq(synth.initial_guess(cfg))
    assert S.shape[0] == 7 * (cfg.cn + cfg.pn - 1)
    assert np.abs(S - S.T).max() <= 1e-9 * np.abs(S).max()
    assert np.linalg.eigvalsh(S).min() > 0
