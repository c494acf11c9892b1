"""GPU parity tests of the camera-parameter refinement (calico_rp_set_refine_intrinsics,
BASELINE.json configs[2] "intrinsics + distortion refined").  An extension: the reference keeps
the 12 numbers constant (solving-structure.hpp:738-749), so the pins are the mpmath golden
vectors over all 33 parameters (tests/golden/rp_golden_k.json, oracle/gen_golden_k.py) and the
oracle with the same fourth parameter block (oracle/rp_oracle.cpp).  Bars as in
test_gpu_parity.py: r, J 1e-12; normal equations 1e-10; converged RMS 1e-6 px, poses 1e-6."""
import json
import os

import numpy as np
import pytest

from calico_amd import capi, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MASK_OPENCV = 0x1FF  # fx cx fy cy k1 k2 p1 p2 k3: what cv::calibrateCamera hands over (solving-structure.cpp:134-159)
MASK_ALL = 0xFFF


def _relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _pair(cfg, oracle_api, mask, intr=None, obs=None):
    g = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
    o = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, api=oracle_api)
    obs = synth.observations(cfg) if obs is None else obs
    intr = synth.intrinsics(cfg) if intr is None else intr
    for P in (g, o):
        P.set_intrinsics(intr)
        P.set_points(synth.points(cfg))
        P.add_observations(*obs)
        P.set_refine_intrinsics(mask)
    return g, o, obs


def _perturbed_intrinsics(cfg, seed=3, rational=False):
    """What a per-camera calibration that is a little off would hand over."""
    rng = np.random.default_rng(seed)
    k = synth.intrinsics(cfg).copy()
    k[:, [0, 2]] *= 1.0 + rng.uniform(-0.01, 0.01, size=(cfg.cn, 2))
    k[:, [1, 3]] += rng.uniform(-3.0, 3.0, size=(cfg.cn, 2))
    k[:, 4:6] += rng.uniform(-0.01, 0.01, size=(cfg.cn, 2))
    k[:, 6:8] += rng.uniform(-5e-4, 5e-4, size=(cfg.cn, 2))
    k[:, 8] += rng.uniform(-0.005, 0.005, size=cfg.cn)
    if rational:
        k[:, 9:12] += rng.uniform(-0.01, 0.01, size=(cfg.cn, 3))
    return k


def _assert_intrinsics_close(kg, ko):
    assert np.abs(kg[:, :4] - ko[:, :4]).max() < 2e-5  # fx cx fy cy, pixels
    assert np.abs(kg[:, [4, 6, 7]] - ko[:, [4, 6, 7]]).max() < 1e-7  # k1 p1 p2
    # k2 and k3 multiply r^4 <= 1e-3 and r^6 <= 3e-5 in these images: 1e-5 / 1e-4 of them is < 1e-8 px
    assert np.abs(kg[:, 5] - ko[:, 5]).max() < 1e-5 and np.abs(kg[:, 8] - ko[:, 8]).max() < 1e-4
    assert np.array_equal(kg[:, 9:], ko[:, 9:])


def _pose_deltas(xa, xb, vn):
    rot, tr = 0.0, 0.0
    for v in range(vn):
        qa, qb = xa[7 * v:7 * v + 4], xb[7 * v:7 * v + 4]
        qa, qb = qa / np.linalg.norm(qa), qb / np.linalg.norm(qb)
        rot = max(rot, 2.0 * np.arccos(min(1.0, abs(float(qa @ qb)))))
        tr = max(tr, np.abs(xa[7 * v + 4:7 * v + 7] - xb[7 * v + 4:7 * v + 7]).max())
    return rot, tr


def _oracle_gradient_max(o, cfg, obs, x, intr):
    """max |J^T r| of the oracle's objective at (x, intr), over the pose blocks that are
    optimised and the refined camera parameters."""
    o.set_intrinsics(intr)
    _, r, J = o.eval(x)
    g = np.zeros((cfg.vn, 7))
    for b, var in enumerate(obs[:3]):
        np.add.at(g, var, np.einsum("nrk,nr->nk", J[:, :, 7 * b:7 * b + 7], r))
    g[[cfg.p_star_var, cfg.t_star_var]] = 0.0
    gk = np.zeros((cfg.cn, 12))
    np.add.at(gk, obs[0], np.einsum("nrk,nr->nk", J[:, :, 21:], r))
    return max(np.abs(g).max(), np.abs(gk).max())


def test_golden_vectors_with_camera_parameter_columns_on_gpu():
    G = json.load(open(os.path.join(ROOT, "tests", "golden", "rp_golden_k.json")))["vectors"]
    for g in G:
        case = g["case"]
        pv = 1 if case in (1, 3) else 2
        tv = 3 if case in (2, 3) else 4
        P = capi.Problem(1, 2, 2, 1, 3, device=0)
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
        rg = np.array([float(v) for v in g["r"]])
        Jg = np.array([[float(v) for v in row] for row in g["J"]])
        for mask in (MASK_ALL, MASK_OPENCV):
            P.set_refine_intrinsics(mask)
            cost, r, J = P.eval(x)
            want = Jg.copy()
            want[:, [21 + k for k in range(12) if not (mask >> k) & 1]] = 0.0
            assert J.shape == (1, 2, 33)
            assert _relerr(r[0], rg) < 1e-12
            assert np.abs(J[0] - want).max() < 1e-12 * np.abs(Jg).max()
        P.close()


@pytest.mark.parametrize("mask", [MASK_OPENCV, MASK_ALL])
def test_residual_jacobian_vs_oracle_refined(oracle_api, mask):
    cfg = synth.custom(5, 3, 17, tangential=1)
    g, o, _ = _pair(cfg, oracle_api, mask, intr=_perturbed_intrinsics(cfg, rational=True))
    x = synth.initial_guess(cfg)
    cg, rg, Jg = g.eval(x)
    co, ro, Jo = o.eval(x)
    assert Jg.shape == Jo.shape and Jg.shape[2] == 33
    assert _relerr(rg, ro) < 1e-12
    for cols in (slice(0, 21), slice(21, 33)):
        scale = np.abs(Jo[:, :, cols]).max(axis=2, keepdims=True)
        assert (np.abs(Jg[:, :, cols] - Jo[:, :, cols]) / scale).max() < 1e-12
    assert abs(cg - co) < 1e-12 * co


@pytest.mark.parametrize("jacobi", [True, False])
@pytest.mark.parametrize("mask", [MASK_OPENCV, MASK_ALL])
def test_reduced_normal_equations_vs_oracle_refined(oracle_api, jacobi, mask):
    """n = 7 (cameras + patterns - 1) + 12 cameras; the 12-parameter blocks follow the others."""
    cfg = synth.custom(6, 3, 25, fr_drop_prob=0.3, tangential=1)
    g, o, _ = _pair(cfg, oracle_api, mask, intr=_perturbed_intrinsics(cfg, rational=mask == MASK_ALL))
    x = synth.initial_guess(cfg)
    Sg, bg = g.normal_eq(x, radius=1e4, jacobi_scaling=jacobi)
    So, bo = o.normal_eq(x, radius=1e4, jacobi_scaling=jacobi)
    assert Sg.shape == So.shape == (7 * (cfg.cn + cfg.pn - 1) + 12 * cfg.cn,) * 2
    assert _relerr(Sg, So) < 1e-10
    assert _relerr(bg, bo) < 1e-10
    # block by block: the small blocks must not hide behind the large ones
    n7 = 7 * (cfg.cn + cfg.pn - 1)
    for a, b in ((slice(n7, None), slice(n7, None)), (slice(n7, None), slice(0, n7))):
        assert _relerr(Sg[a, b], So[a, b]) < 1e-10
    assert _relerr(bg[n7:], bo[n7:]) < 1e-10
    assert np.abs(Sg - Sg.T).max() == 0.0


@pytest.mark.parametrize("drop,tn,mask", [(0.7, 60, MASK_OPENCV), (0.95, 300, MASK_ALL)])
def test_ragged_unsorted_refined(oracle_api, drop, tn, mask):
    """Ragged FRs (1..48 corners), several wave ranges, observations in random order; with 0.95
    FRs of one to a handful of corners: a slab then spans more FRs than the fused kernel's table
    ring holds and a 16-observation group several parts (and with all 12 parameters free and
    k4..k6 nonzero the rational-model form of the kernel runs)."""
    cfg = synth.custom(5, 3, tn, fr_drop_prob=0.2, corner_drop_prob=drop, tangential=1)
    obs = synth.observations(cfg)
    assert obs[0].size > 2 * 4096
    perm = np.random.default_rng(4).permutation(obs[0].size)
    g, o, _ = _pair(cfg, oracle_api, mask, obs=[a[perm] for a in obs],
                    intr=_perturbed_intrinsics(cfg, rational=mask == MASK_ALL))
    x = synth.initial_guess(cfg)
    Sg, bg = g.normal_eq(x)
    So, bo = o.normal_eq(x)
    assert _relerr(Sg, So) < 1e-10 and _relerr(bg, bo) < 1e-10
    _, rg, Jg = g.eval(x)
    _, ro, Jo = o.eval(x)
    assert _relerr(rg, ro) < 1e-12 and _relerr(Jg, Jo) < 1e-11


@pytest.mark.parametrize("name,shape", [("small", (6, 3, 40)), ("medium", (20, 4, 30))])
def test_converged_solution_vs_oracle_refined(oracle_api, name, shape):
    cfg = synth.custom(*shape, tangential=1)
    k0 = _perturbed_intrinsics(cfg)
    g, o, _ = _pair(cfg, oracle_api, MASK_OPENCV, intr=k0)
    x0 = synth.initial_guess(cfg)
    opt = capi.default_options(max_num_iterations=100, function_tolerance=1e-15, parameter_tolerance=1e-13,
                               gradient_tolerance=1e-14)
    xg, sg = g.solve(x0, opt)
    xo, so = o.solve(x0, opt)
    n = g.num_observations
    rms_g, rms_o = np.sqrt(sg.final_cost / n), np.sqrt(so.final_cost / n)
    assert abs(rms_g - rms_o) < 1e-6
    assert rms_g < 0.11  # the 0.1 px noise floor: the perturbed intrinsics were recovered
    rot, tr = _pose_deltas(xg, xo, cfg.vn)
    kg, ko, kt = g.get_intrinsics(), o.get_intrinsics(), synth.intrinsics(cfg)
    print(f"{name}: rms {rms_g:.9f} d_rms {abs(rms_g - rms_o):.2e} rot {rot:.2e} tr {tr:.2e} "
          f"dK {np.abs(kg - ko).max(0)} iters {sg.num_iterations}/{so.num_iterations}")
    # With the focal lengths free, depth along the optical axis and focal length are nearly
    # interchangeable (3 m stand-off, boards 0.4 m wide): both minimisers stop where rounding of
    # the cost (1e-13 relative) decides, which leaves that combination undetermined to ~1e-5 mm
    # / 1e-5 px.  Rotations, RMS and everything else meet the north_star bars.
    assert rot < 1e-6 and tr < 2e-5
    _assert_intrinsics_close(kg, ko)
    assert np.array_equal(kg[:, 9:], k0[:, 9:])  # masked-out parameters never move
    assert np.abs(kg[:, :4] - kt[:, :4]).max() < np.abs(k0[:, :4] - kt[:, :4]).max()
    for v in (cfg.p_star_var, cfg.t_star_var):
        assert np.array_equal(xg[7 * v:7 * v + 7], x0[7 * v:7 * v + 7])
    # the refined values are what the handle reports with from now on
    rr_g = g.fr_errors(xg)[5]
    assert rr_g < 0.2


def test_all_twelve_free_from_an_opencv_hand_over(oracle_api):
    """The natural use of MASK_ALL: cv::calibrateCamera hands over k4 = k5 = k6 = 0
    (solving-structure.cpp:134-159 fills zeros) and the mask frees them.  The handle then starts
    with "every denominator is 1", and the kernels without the rational model's division must
    NOT be chosen, or k4..k6 drift in x while residual, cost and Jacobian ignore them.
    Normal equations, r / J, the ten-iteration schedule and the converged state against the oracle;
    the per-FR report at the returned intrinsics reproduces the solve's final cost."""
    cfg = synth.custom(6, 3, 40, tangential=1)
    k0 = _perturbed_intrinsics(cfg, seed=21)  # rational=False: k4..k6 exactly zero
    assert np.all(k0[:, 9:] == 0.0)
    g, o, obs = _pair(cfg, oracle_api, MASK_ALL, intr=k0)
    x0 = synth.initial_guess(cfg)
    Sg, bg = g.normal_eq(x0)
    So, bo = o.normal_eq(x0)
    assert _relerr(Sg, So) < 1e-10 and _relerr(bg, bo) < 1e-10
    n7 = 7 * (cfg.cn + cfg.pn - 1)
    assert _relerr(Sg[n7:, n7:], So[n7:, n7:]) < 1e-10
    _, rg, Jg = g.eval(x0)
    _, ro, Jo = o.eval(x0)
    assert _relerr(rg, ro) < 1e-12
    scale = np.abs(Jo[:, :, 21:]).max(axis=2, keepdims=True)
    assert (np.abs(Jg[:, :, 21:] - Jo[:, :, 21:]) / scale).max() < 1e-12
    assert np.abs(Jo[:, :, 30:]).max() > 0.0  # the k4..k6 columns are there
    # the reference's budget: same step sequence
    xg, sg = g.solve(x0)
    xo, so = o.solve(x0)
    assert (sg.num_iterations, sg.num_successful_steps, sg.termination_type) == (
        so.num_iterations, so.num_successful_steps, so.termination_type)
    assert abs(sg.final_cost - so.final_cost) <= 1e-8 * so.final_cost
    kg = g.get_intrinsics()
    assert np.abs(kg[:, 9:]).max() > 0.0  # they moved ...
    n = g.num_observations
    cam, pat, tim, cnt, err, rrmse = g.fr_errors(xg)
    # ... and the model the report evaluates at the returned intrinsics is the one the solve minimised
    assert abs(0.5 * float((err * cnt).sum()) - sg.final_cost) <= 1e-9 * sg.final_cost
    # converged, from the refined state
    opt = capi.default_options(max_num_iterations=100, function_tolerance=1e-15, parameter_tolerance=1e-13,
                               gradient_tolerance=1e-14)
    xg, sg = g.solve(xg, opt)
    xo, so = o.solve(xo, opt)
    assert abs(np.sqrt(sg.final_cost / n) - np.sqrt(so.final_cost / n)) < 1e-6
    assert np.sqrt(sg.final_cost / n) < 0.11
    # With numerator AND denominator of the rational model free (k1..k3 against k4..k6 at r^2 <= 0.1)
    # the minimum is the floor of a nearly flat valley: within their 100 iterations the two
    # minimisers reach the same RMS to 1e-6 px (costs within 1e-4 relative, the HIP one lower) and
    # stand 5e-4 rad apart along it.  That the HIP result is as stationary for the ORACLE's objective as
    # the oracle's own is shown by the oracle's gradient at both (at the start: ~1e7).
    assert abs(sg.final_cost - so.final_cost) <= 1e-4 * so.final_cost
    rot, tr = _pose_deltas(xg, xo, cfg.vn)
    assert rot < 5e-3 and tr < 5.0
    kg_final, ko_final = g.get_intrinsics(), o.get_intrinsics()  # (the gradient helper resets the oracle's)
    gmax_0 = _oracle_gradient_max(o, cfg, obs, x0, k0)
    gmax_o = _oracle_gradient_max(o, cfg, obs, xo, ko_final)
    gmax_g = _oracle_gradient_max(o, cfg, obs, xg, kg_final)
    print(f"all twelve free: rot {rot:.2e} tr {tr:.2e} gmax start {gmax_0:.2e} hip {gmax_g:.2e} oracle {gmax_o:.2e} "
          f"iterations {sg.num_iterations}/{so.num_iterations}")
    assert gmax_g < 1e-6 * gmax_0 or gmax_g < 10.0 * gmax_o, (gmax_g, gmax_o, gmax_0)
    cam, pat, tim, cnt, err, rrmse = g.fr_errors(xg)
    assert abs(0.5 * float((err * cnt).sum()) - sg.final_cost) <= 1e-9 * sg.final_cost


def test_ten_iterations_follow_the_oracle_refined(oracle_api):
    """The reference's own budget (10 iterations, Ceres defaults): same step sequence."""
    cfg = synth.custom(8, 3, 30, tangential=1)
    g, o, _ = _pair(cfg, oracle_api, MASK_OPENCV, intr=_perturbed_intrinsics(cfg, seed=9))
    x0 = synth.initial_guess(cfg)
    xg, sg = g.solve(x0)
    xo, so = o.solve(x0)
    assert (sg.num_iterations, sg.num_successful_steps, sg.termination_type) == (
        so.num_iterations, so.num_successful_steps, so.termination_type)
    assert abs(sg.initial_cost - so.initial_cost) <= 1e-10 * so.initial_cost
    assert abs(sg.final_cost - so.final_cost) <= 1e-8 * so.final_cost
    _assert_intrinsics_close(g.get_intrinsics(), o.get_intrinsics())


def test_mask_zero_is_the_reference_behaviour():
    """Flag off reproduces the library without the extension bit for bit, also after it was on."""
    cfg = synth.custom(6, 3, 20, tangential=1)
    x0 = synth.initial_guess(cfg)
    plain = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
    synth.fill_problem(plain, cfg)
    x_ref, s_ref = plain.solve(x0)
    S_ref, b_ref = plain.normal_eq(x0)
    P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
    synth.fill_problem(P, cfg)
    P.set_refine_intrinsics(MASK_OPENCV)
    Sk, _ = P.normal_eq(x0)
    assert Sk.shape[0] == S_ref.shape[0] + 12 * cfg.cn
    with pytest.raises(capi.RpError):
        P.set_jacobian_mode("materialised")
    with pytest.raises(capi.RpError):
        P.set_refine_intrinsics(1 << 12)
    P.set_refine_intrinsics(0)
    x1, s1 = P.solve(x0)
    S1, b1 = P.normal_eq(x0)
    assert np.array_equal(x1, x_ref) and s1.final_cost == s_ref.final_cost
    assert np.array_equal(S1, S_ref) and np.array_equal(b1, b_ref)
    assert np.array_equal(P.get_intrinsics(), synth.intrinsics(cfg))


def test_speculation_changes_nothing_refined():
    cfg = synth.custom(6, 3, 40, tangential=1)
    x0 = synth.initial_guess(cfg)
    out = []
    for spec in (True, False):
        P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
        obs = synth.observations(cfg)
        P.set_intrinsics(_perturbed_intrinsics(cfg))
        P.set_points(synth.points(cfg))
        P.add_observations(*obs)
        P.set_refine_intrinsics(MASK_OPENCV)
        P.set_speculation(spec)
        x, s = P.solve(x0, capi.default_options(max_num_iterations=8))
        out.append((x, s, P.get_intrinsics()))
        P.close()
    (x1, s1, k1), (x0_, s0, k0) = out
    assert (s1.num_iterations, s1.termination_type) == (s0.num_iterations, s0.termination_type)
    assert abs(s1.final_cost - s0.final_cost) <= 1e-10 * s0.final_cost
    assert np.abs(x1 - x0_).max() <= 1e-9 * np.abs(x0_).max()
    assert np.abs(k1 - k0).max() <= 1e-9 * np.abs(k0).max()
