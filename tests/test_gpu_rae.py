"""GPU parity tests of the point reconstruction: calico_rae_reconstruct
against the CPU oracle on the same seeded views.  Bars: DLT estimates 1e-6 mm (normal
equations here, pivoted QR in the oracle); refined corners at tight tolerances: 1e-5 mm for
every corner and 1e-9 mm for the median one (a corner seen by two or three views has a flat
direction along the rays, where the minimiser stops wherever the cost change reaches
rounding level); the same per-corner iteration counts with Ceres' default tolerances."""
import numpy as np
import pytest

from calico_amd import capi, synth

pytestmark = pytest.mark.gpu

TIGHT = dict(max_num_iterations=100, function_tolerance=1e-15, parameter_tolerance=1e-13, gradient_tolerance=1e-14)


@pytest.mark.parametrize("shape", [(3, 2, 6), (8, 3, 40)])
def test_rae_dlt_estimates_vs_oracle(oracle_api, shape):
    cfg = synth.custom(*shape)
    args = synth.rae_inputs(cfg, x=synth.initial_guess(cfg))
    opt = capi.default_options(max_num_iterations=0)
    xg, hg, _, _, sg = capi.rae_reconstruct(*args, options=opt)
    xo, ho, _, _, _ = capi.rae_reconstruct(*args, options=opt, api=oracle_api)
    assert np.array_equal(hg, ho) and hg.all()
    assert np.abs(xg - xo).max() < 1e-6
    assert sg.sweeps == 1


@pytest.mark.parametrize("shape", [(3, 2, 6), (8, 3, 40), (20, 4, 30)])
def test_rae_refined_corners_vs_oracle(oracle_api, shape):
    cfg = synth.custom(*shape, fr_drop_prob=0.2, corner_drop_prob=0.3)
    args = synth.rae_inputs(cfg, x=synth.initial_guess(cfg))
    opt = capi.default_options(**TIGHT)
    xg, hg, ig, cg, sg = capi.rae_reconstruct(*args, options=opt)
    xo, ho, io, co, so = capi.rae_reconstruct(*args, options=opt, api=oracle_api)
    assert np.array_equal(hg, ho)
    d = np.abs(xg - xo).max(axis=1)
    assert d.max() < 1e-5 and np.median(d) < 1e-9
    assert np.abs(cg - co).max() <= 1e-9 * co.max()
    assert sg.n_solved == so.n_solved == hg.sum()


def test_rae_default_options_follow_the_oracle_step_for_step(oracle_api):
    """opt = NULL: Ceres defaults, 50 iterations (solving-structure.cpp:358-362)."""
    cfg = synth.custom(6, 3, 25)
    args = synth.rae_inputs(cfg, x=synth.initial_guess(cfg))
    xg, hg, ig, cg, sg = capi.rae_reconstruct(*args)
    xo, ho, io, co, so = capi.rae_reconstruct(*args, api=oracle_api)
    assert np.array_equal(ig, io)
    assert sg.max_point_iterations == so.max_point_iterations
    assert np.abs(xg - xo).max() < 1e-6
    assert sg.sweeps == 1 + ig.max()  # DLT + iteration zero + one sweep per candidate


def test_rae_unseen_and_single_view_corners(oracle_api):
    cfg = synth.custom(3, 2, 6)
    args = list(synth.rae_inputs(cfg))
    obs_fr, obs_point, u, v = args[6:10]
    first5 = np.flatnonzero(obs_point == 5)[0]
    keep = ~(((obs_point == 5) & (np.arange(obs_point.size) != first5)) | (obs_point == 7))
    args[6:10] = obs_fr[keep], obs_point[keep], u[keep], v[keep]
    xg, hg, ig, _, _ = capi.rae_reconstruct(*args)
    xo, ho, io, _, _ = capi.rae_reconstruct(*args, api=oracle_api)
    assert np.array_equal(hg, ho) and not hg[5] and not hg[7]
    assert not xg[5].any() and not xg[7].any()
    assert np.abs(xg - xo).max() < 1e-6


def test_rae_noise_free_recovers_model_corners():
    cfg = synth.custom(4, 2, 8, noise_px=0.0)
    xyz, has, its, fc, s = capi.rae_reconstruct(*synth.rae_inputs(cfg))
    assert has.all()
    assert np.abs(xyz - synth.points(cfg)).max() < 1e-3


def test_rae_argument_checks():
    cfg = synth.custom(3, 2, 4)
    args = list(synth.rae_inputs(cfg))
    bad = list(args)
    bad[7] = args[7].copy()
    bad[7][[0, 1]] = bad[7][[1, 0]]  # corners of a view not increasing
    with pytest.raises(capi.RpError):
        capi.rae_reconstruct(*bad)
    bad = list(args)
    bad[6] = args[6][::-1].copy()  # views not grouped
    with pytest.raises(capi.RpError):
        capi.rae_reconstruct(*bad)
    bad = list(args)
    bad[2] = args[2] + 100  # camera index out of range
    with pytest.raises(capi.RpError):
        capi.rae_reconstruct(*bad)
    with pytest.raises(capi.RpError):
        capi.rae_reconstruct(*args, device=99)


def test_rae_handle_keeps_the_observations_between_calls(oracle_api):
    """calico_rae_create / _solve: the same corners as the stateless call, bit for bit, for two
    different states of the poses (the reference reconstructs once per solution type from the
    same detections, multicamera.cpp:812-833) and changed camera parameters, the second solve
    against the oracle."""
    cfg = synth.custom(8, 3, 40, fr_drop_prob=0.2, corner_drop_prob=0.3)
    a0 = synth.rae_inputs(cfg, x=synth.initial_guess(cfg))
    a1 = synth.rae_inputs(cfg, x=synth.ground_truth(cfg))
    cam, cr, fr_cam, ahat0, n_points, ncp, obs_fr, obs_point, u, v = a0
    assert np.array_equal(fr_cam, a1[2]) and np.array_equal(obs_point, a1[7])
    H = capi.RaeProblem(cr, fr_cam, n_points, ncp, obs_fr, obs_point, u, v)
    for ahat, args in ((ahat0, a0), (a1[3], a1), (ahat0, a0)):
        xh, hh, ih, ch, sh = H.solve(cam, ahat)
        xs, hs, is_, cs, ss = capi.rae_reconstruct(*args)
        assert np.array_equal(xh, xs) and np.array_equal(hh, hs) and np.array_equal(ih, is_) and np.array_equal(ch, cs)
        assert sh.sweeps == ss.sweeps
    cam2 = cam.copy()
    cam2[:, 0] *= 1.002
    cam2[:, 4] += 0.003
    xh, hh, _, _, _ = H.solve(cam2, a1[3])
    args2 = list(a1)
    args2[0] = cam2
    xo, ho, _, _, _ = capi.rae_reconstruct(*args2, api=oracle_api)
    assert np.array_equal(hh, ho) and np.abs(xh - xo).max() < 1e-5
    assert np.abs(xh - H.solve(cam, a1[3])[0]).max() > 1e-3  # the camera parameters of the call are the ones used
    with pytest.raises(capi.RpError):
        H.solve(cam, np.full_like(ahat0, np.nan))
    H.close()


import os

_SEEDS = range(int(os.environ["CALICO_TEST_SEEDS"])) if "CALICO_TEST_SEEDS" in os.environ else range(12)


@pytest.mark.parametrize("seed", _SEEDS)
def test_rae_random_view_sets_vs_oracle(oracle_api, seed):
    """Point reconstruction over shapes nobody picked by hand: random block counts and board
    sizes, views and corners dropped at random (corners seen in one view only or not at all, views
    of a single corner), a random subset of the observations removed on top.  Which corners get a
    value, the DLT estimates and the refined corners against the oracle, the handle against the
    stateless call bit for bit.  CALICO_TEST_SEEDS=1000 is a fuzzing run."""
    rng = np.random.default_rng(9000 + seed)
    cn, pn, tn = int(rng.integers(1, 9)), int(rng.integers(1, 4)), int(rng.integers(1, 30))
    cfg = synth.custom(cn, pn, tn, sx=int(rng.integers(3, 9)), sy=int(rng.integers(3, 10)),
                       fr_drop_prob=float(rng.choice([0.0, 0.3, 0.7])), corner_drop_prob=float(rng.choice([0.0, 0.5, 0.9])),
                       one_board_per_cam_time=int(rng.integers(0, 2)), tangential=int(rng.integers(0, 2)), seed=31 + seed)
    if synth.count(cfg) == 0:
        pytest.skip("the draw left no observation")
    args = list(synth.rae_inputs(cfg, x=synth.initial_guess(cfg)))
    obs_fr, obs_point, u, v = args[6:10]
    keep = rng.random(obs_fr.size) < float(rng.choice([1.0, 0.8, 0.3]))
    args[6:10] = obs_fr[keep], obs_point[keep], u[keep], v[keep]
    dlt = capi.default_options(max_num_iterations=0)
    xg, hg, _, _, _ = capi.rae_reconstruct(*args, options=dlt)
    xo, ho, _, _, _ = capi.rae_reconstruct(*args, options=dlt, api=oracle_api)
    assert np.array_equal(hg, ho)
    # a corner seen in two views whose rays are nearly parallel has an ill-conditioned DLT: compare
    # where the two estimates' own normal equations are well posed, i.e. relative to the board's size
    scale = max(1.0, np.abs(xo).max())
    views = np.bincount(args[7], minlength=args[4])
    well = hg & (views >= 3)
    if well.any():
        assert np.abs(xg[well] - xo[well]).max() < 1e-5 * scale
    xg, hg, ig, cg, sg = capi.rae_reconstruct(*args)
    xo, ho, io, co, so = capi.rae_reconstruct(*args, api=oracle_api)
    assert np.array_equal(hg, ho)
    assert np.all(np.isfinite(xg)) and not xg[~hg].any()
    if well.any():
        assert np.abs(xg[well] - xo[well]).max() < 1e-4 * scale
        assert np.abs(cg[well] - co[well]).max() <= 1e-6 * max(co.max(), 1e-30)
    H = capi.RaeProblem(args[1], args[2], args[4], args[5], *args[6:10])
    xh, hh, ih, ch, sh = H.solve(args[0], args[3])
    assert np.array_equal(xh, xg) and np.array_equal(hh, hg) and np.array_equal(ih, ig) and np.array_equal(ch, cg)
    H.close()
