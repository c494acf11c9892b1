"""GPU parity tests of the Stage-4 algebraic-error widening: the HIP
path through calico_ae_* / calico_rp_* against the CPU oracle on the same seeded relations
and against the mpmath golden vectors tests/golden/ae_golden.json.  Same bars as the
reprojection path: r and J 1e-12 relative, normal equations 1e-10, converged parameters
1e-6."""
import json
import os

import numpy as np
import pytest

from calico_amd import capi, synth
from tests.test_ae_cpu import golden_problem, golden_rj

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _pair(cfg, oracle_api, rel):
    g = capi.AeProblem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
    o = capi.AeProblem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, api=oracle_api)
    g.add_relations(*rel[:4])
    o.add_relations(*rel[:4])
    return g, o


def test_ae_golden_vectors_on_gpu():
    G = json.load(open(os.path.join(ROOT, "tests", "golden", "ae_golden.json")))["vectors"]
    for g in G:
        P, x = golden_problem(capi.load(), g)
        cost, r, J = P.eval(x)
        rg, Jg = golden_rj(g)
        assert _relerr(r[0], rg) < 1e-12
        assert _relerr(J[0], Jg) < 1e-12
        assert abs(cost - 0.5 * (rg ** 2).sum()) <= 1e-12 * cost
        P.close()


@pytest.mark.parametrize("shape", [(3, 2, 4), (6, 3, 25)])
def test_ae_residual_jacobian_vs_oracle(oracle_api, shape):
    cfg = synth.custom(*shape)
    rel = synth.ae_relations(cfg)  # shuffled relation order: results come back in the caller's order
    g, o = _pair(cfg, oracle_api, rel)
    cg, rg, Jg = g.eval(rel[5])
    co, ro, Jo = o.eval(rel[5])
    assert _relerr(rg, ro) < 1e-12
    scale = np.abs(Jo).max(axis=2, keepdims=True)
    assert (np.abs(Jg - Jo) / scale).max() < 1e-12
    assert abs(cg - co) < 1e-12 * co


@pytest.mark.parametrize("jacobi", [True, False])
def test_ae_reduced_normal_equations_vs_oracle(oracle_api, jacobi):
    cfg = synth.custom(6, 3, 25, fr_drop_prob=0.3)
    rel = synth.ae_relations(cfg)
    g, o = _pair(cfg, oracle_api, rel)
    Sg, bg = g.normal_eq(rel[5], radius=1e4, jacobi_scaling=jacobi)
    So, bo = o.normal_eq(rel[5], radius=1e4, jacobi_scaling=jacobi)
    assert Sg.shape == So.shape
    assert _relerr(Sg, So) < 1e-10
    assert _relerr(bg, bo) < 1e-10


def test_ae_repeated_relations_across_wave_ranges(oracle_api):
    """Several relations per FR (the interface allows it) and one FR with more relations than
    a wave's observation range (256 at this size), so that it is split into parts."""
    cfg = synth.custom(4, 2, 6)
    cam, pat, tim, a7, _xt, x0 = synth.ae_relations(cfg)
    rng = np.random.default_rng(5)
    rep = rng.integers(1, 4, size=cam.size)
    rep[3] = 300
    idx = np.repeat(np.arange(cam.size), rep)
    rng.shuffle(idx)
    a7r = a7[idx] + 1e-3 * rng.normal(size=(idx.size, 7))
    rel = (cam[idx], pat[idx], tim[idx], a7r)
    g, o = _pair(cfg, oracle_api, rel)
    cg, rg, _ = g.eval(x0, want_J=False)
    co, ro, _ = o.eval(x0, want_J=False)
    assert _relerr(rg, ro) < 1e-12 and abs(cg - co) < 1e-12 * co
    Sg, bg = g.normal_eq(x0)
    So, bo = o.normal_eq(x0)
    assert _relerr(Sg, So) < 1e-10
    assert _relerr(bg, bo) < 1e-10


@pytest.mark.parametrize("shape", [(3, 2, 4), (6, 3, 40), (20, 4, 30)])
def test_ae_converged_solution_vs_oracle(oracle_api, shape):
    cfg = synth.custom(*shape)
    rel = synth.ae_relations(cfg)
    g, o = _pair(cfg, oracle_api, rel)
    opt = capi.default_options(max_num_iterations=100, function_tolerance=1e-15, parameter_tolerance=1e-13,
                               gradient_tolerance=1e-14)
    x0 = rel[5]
    xg, sg = g.solve(x0, opt)
    xo, so = o.solve(x0, opt)
    assert abs(sg.initial_cost - so.initial_cost) <= 1e-12 * so.initial_cost
    assert abs(sg.final_cost - so.final_cost) <= 1e-9 * so.final_cost
    assert sg.final_cost < 0.05 * sg.initial_cost
    for v in range(cfg.vn):
        Mg, Mo = synth.quat_to_matrix(xg[7 * v:7 * v + 7]), synth.quat_to_matrix(xo[7 * v:7 * v + 7])
        assert np.abs(Mg[:3, :3] - Mo[:3, :3]).max() < 1e-6 and np.abs(Mg[:3, 3] - Mo[:3, 3]).max() < 1e-6
    for v in (cfg.p_star_var, cfg.t_star_var):
        assert np.array_equal(xg[7 * v:7 * v + 7], x0[7 * v:7 * v + 7])


def test_ae_default_schedule_matches_oracle_step_for_step(oracle_api):
    """Ceres defaults with the iteration cap the reference passes for this stage
    (calico.cpp:684/697 -> solving-structure.cpp:930): same accepted/rejected sequence."""
    cfg = synth.custom(8, 3, 30, fr_drop_prob=0.2)
    rel = synth.ae_relations(cfg, perturb_deg=10.0, perturb_mm=100.0)
    g, o = _pair(cfg, oracle_api, rel)
    opt = capi.default_options(max_num_iterations=15)
    xg, sg = g.solve(rel[5], opt)
    xo, so = o.solve(rel[5], opt)
    assert (sg.num_successful_steps, sg.num_unsuccessful_steps, sg.termination_type) == \
           (so.num_successful_steps, so.num_unsuccessful_steps, so.termination_type)
    assert abs(sg.final_cost - so.final_cost) <= 1e-9 * so.final_cost


def test_ae_handle_kinds_are_checked():
    a = capi.AeProblem(2, 2, 2, 2, 4, device=0)
    with pytest.raises(capi.RpError):
        a.set_points(np.zeros((4, 3), np.float32))
    with pytest.raises(capi.RpError):
        a.set_jacobian_mode("materialised")
    with pytest.raises(capi.RpError):
        a.add_relations([0], [9], [5], [[1, 0, 0, 0, 0, 0, 0]])  # pattern index out of range
    with pytest.raises(capi.RpError):
        a.add_relations([0], [3], [5], [[0, 0, 0, 0, 1, 2, 3]])  # zero quaternion
    with pytest.raises(capi.RpError):
        a.solve(np.zeros(42))  # no relations
    r = capi.Problem(2, 2, 2, 2, 4, device=0)
    with pytest.raises(capi.RpError):
        capi.AeProblem.add_relations(r, [0], [3], [5], [[1, 0, 0, 0, 0, 0, 0]])


_SEEDS = range(int(os.environ["CALICO_TEST_SEEDS"])) if "CALICO_TEST_SEEDS" in os.environ else range(12)


@pytest.mark.parametrize("seed", _SEEDS)
def test_ae_random_networks_vs_oracle(oracle_api, seed):
    """Stage 4 over shapes nobody picked by hand (the sweep of test_gpu_parity.py's
    test_random_networks_vs_oracle): random block counts, p* / t*, share of relations dropped,
    some repeated, one block or a few times removed; residuals 1e-12, normal equations 1e-10 and
    three LM iterations against the oracle.  CALICO_TEST_SEEDS=2000 is a fuzzing run."""
    rng = np.random.default_rng(5000 + seed)
    cn, pn, tn = int(rng.integers(1, 10)), int(rng.integers(1, 5)), int(rng.integers(1, 40))
    cfg = synth.custom(cn, pn, tn, fr_drop_prob=float(rng.choice([0.0, 0.3, 0.7])),
                       one_board_per_cam_time=int(rng.integers(0, 2)), p_star_idx=int(rng.integers(0, pn)),
                       t_star_idx=int(rng.integers(0, tn)), seed=123 + seed)
    cam, pat, tim, a7, _xt, x0 = synth.ae_relations(cfg)
    if cam.size == 0:
        pytest.skip("the draw left no relation")
    idx = np.repeat(np.arange(cam.size), rng.integers(1, 3, size=cam.size))
    rng.shuffle(idx)
    if rng.random() < 0.5 and idx.size > 10:
        col = (cam, pat, tim)[int(rng.integers(0, 3))][idx]
        uniq = np.unique(col)
        if uniq.size > 1:
            idx = idx[col != rng.choice(uniq)]
    rel = (cam[idx], pat[idx], tim[idx], a7[idx] + 1e-3 * rng.normal(size=(idx.size, 7)))
    g, o = _pair(cfg, oracle_api, rel)
    cg, rg, _ = g.eval(x0, want_J=False)
    co, ro, _ = o.eval(x0, want_J=False)
    assert _relerr(rg, ro) < 1e-12 and abs(cg - co) <= 1e-12 * co
    Sg, bg = g.normal_eq(x0)
    So, bo = o.normal_eq(x0)
    # twelve residual rows per relation against seven unknowns per block
    determined = 12 * idx.size > 6 * 7 * cfg.vn
    tol = 1e-10 if determined else 1e-6
    assert Sg.shape == So.shape
    assert _relerr(Sg, So) < tol and _relerr(bg, bo) < tol
    opt = capi.default_options(max_num_iterations=3)
    xg, sg = g.solve(x0, opt)
    xo, so = o.solve(x0, opt)
    assert np.all(np.isfinite(xg)) and np.isfinite(sg.final_cost)
    if determined:
        assert (sg.num_iterations, sg.num_successful_steps, sg.termination_type) == (
            so.num_iterations, so.num_successful_steps, so.termination_type)
        assert abs(sg.final_cost - so.final_cost) <= 1e-7 * so.final_cost + 1e-13 * so.initial_cost
