"""The C++ mirror of CeresProblemClass (calico_amd/host/ceres_problem_class.hpp) driven
by a re-typed copy of the reference's Stage-5 loop (calico.cpp:703-731)."""
import os
import subprocess

import numpy as np
import pytest

from calico_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "_build", "host_shim_test")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "helpers")])


def test_pose_conversions_selfcheck():
    _build()
    out = subprocess.check_output([EXE, "--selfcheck"], text=True)
    assert "selfcheck ok" in out


@pytest.mark.gpu
def test_stage5_loop_through_the_shim_matches_the_c_abi():
    assert os.path.exists(EXE), "build tests/helpers first (CPU side)"
    tn = 30
    out = subprocess.check_output([EXE, "--run", str(tn)], text=True, timeout=120)
    solves = [l for l in out.splitlines() if l.startswith("SOLVE")]
    assert len(solves) == 2
    final_shim = float(solves[-1].split("final=")[1])
    x_shim = np.array([float(t) for t in [l for l in out.splitlines() if l.startswith("X ")][0].split()[1:]])

    cfg = synth.preset("A", tn=tn)
    obs = synth.observations(cfg)
    key = np.stack([obs[0], obs[1], obs[2]], 1)
    starts = np.concatenate([[0], np.flatnonzero(np.any(key[1:] != key[:-1], axis=1)) + 1])
    cut = starts[int(round(0.5 * starts.size))]
    P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
    P.set_intrinsics(synth.intrinsics(cfg))
    P.set_points(synth.points(cfg))
    x = synth.initial_guess(cfg)
    P.add_observations(*[a[:cut] for a in obs])
    x, _ = P.solve(x)
    P.add_observations(*[a[cut:] for a in obs])
    x, s2 = P.solve(x)
    # the shim goes 7-vector -> 4x4 -> (inverse) -> 7-vector at both ends, so agreement
    # is to rounding of those conversions, not bitwise
    assert abs(final_shim - s2.final_cost) <= 1e-6 * s2.final_cost
    xs, xp = x_shim.reshape(-1, 7), x.reshape(-1, 7)
    for v in range(cfg.vn):
        qa, qb = xs[v, :4] / np.linalg.norm(xs[v, :4]), xp[v, :4] / np.linalg.norm(xp[v, :4])
        assert 2 * np.arccos(min(1.0, abs(qa @ qb))) < 1e-6
        assert np.abs(xs[v, 4:] - xp[v, 4:]).max() < 1e-4


def _replay(cfg, intr, mask=0):
    obs = synth.observations(cfg)
    key = np.stack([obs[0], obs[1], obs[2]], 1)
    starts = np.concatenate([[0], np.flatnonzero(np.any(key[1:] != key[:-1], axis=1)) + 1])
    n_eq = starts.size
    batch = int(np.float32(0.5) * np.float32(n_eq) + 0.5)  # round(perc_rp * number_equations), calico.cpp:708
    P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
    P.set_intrinsics(intr)
    P.set_points(synth.points(cfg))
    if mask:
        P.set_refine_intrinsics(mask)
    x = synth.initial_guess(cfg)
    bounds = np.concatenate([starts, [obs[0].size]])
    sums = []
    for lo in range(0, n_eq, batch):
        hi = min(lo + batch, n_eq)
        P.add_observations(*[a[bounds[lo]:bounds[hi]] for a in obs])
        x, s = P.solve(x)
        sums.append(s)
    return x, sums, P.get_intrinsics()


@pytest.mark.gpu
@pytest.mark.parametrize("config_id,name,tn,mask", [(2, "B", 16, 0), (3, "C", 20, 0), (2, "B", 16, 0x1FF)])
def test_stage5_loop_on_configs_B_C_through_the_shim(config_id, name, tn, mask):
    """The reference's Stage-5 loop (calico.cpp:703-731) through the mirror class on the
    AprilTag rig B and the non-overlapping network C (reduced time range), and on B with the
    camera parameters refined (SetRefineIntrinsics, the extension: camera_params updated in
    place); MCcali::TestSolutionHasIdentityPstarTstar runs before and after every solve
    (solving-structure.cpp:971,1021)."""
    assert os.path.exists(EXE), "build tests/helpers first (CPU side)"
    args = [EXE, "--run", str(tn), str(config_id)] + ([hex(mask)] if mask else [])
    out = subprocess.check_output(args, text=True, timeout=120)
    solves = [l for l in out.splitlines() if l.startswith("SOLVE")]
    assert len(solves) == 2
    assert int([l for l in out.splitlines() if l.startswith("IDENTITY_CHECKS")][0].split()[1]) == 4
    x_shim = np.array([float(t) for t in [l for l in out.splitlines() if l.startswith("X ")][0].split()[1:]])
    k_shim = np.array([float(t) for t in [l for l in out.splitlines() if l.startswith("K ")][0].split()[1:]]).reshape(-1, 12)
    cfg = synth.preset(name, tn=tn)
    intr = synth.intrinsics(cfg).copy()
    if mask:
        c = np.arange(cfg.cn)
        intr[:, 0] *= 1.0 + 0.004 * ((c % 3) - 1)
        intr[:, 2] *= 1.0 - 0.003 * np.where(c % 2, 1, -1)
        intr[:, 1] += 1.5 * ((c % 4) - 1.5)
        intr[:, 3] -= 1.0 * ((c % 5) - 2)
        intr[:, 4] += 0.004 * ((c % 3) - 1)
    x, sums, k = _replay(cfg, intr, mask)
    for line, s in zip(solves, sums):
        assert int(line.split("iterations=")[1].split()[0]) == s.num_iterations
        assert abs(float(line.split("final=")[1]) - s.final_cost) <= 1e-6 * s.final_cost
    xs, xp = x_shim.reshape(-1, 7), x.reshape(-1, 7)
    for v in range(cfg.vn):
        qa, qb = xs[v, :4] / np.linalg.norm(xs[v, :4]), xp[v, :4] / np.linalg.norm(xp[v, :4])
        assert 2 * np.arccos(min(1.0, abs(qa @ qb))) < 1e-6
        assert np.abs(xs[v, 4:] - xp[v, 4:]).max() < 1e-3
    if mask:
        assert np.abs(k_shim[:, :4] - k[:, :4]).max() < 1e-3 and np.abs(k_shim[:, 4:9] - k[:, 4:9]).max() < 1e-4
        assert np.abs(k_shim - intr).max() > 0.1  # they did move
    else:
        assert np.array_equal(k_shim, intr)
    # The refined values reach the calibration objects (CopyToCalibration, the inverse of
    # solving-structure.cpp:134-159) and through them the result file, which prints
    # CCV[i]->internal_parameters / distortion (multicamera.cpp:1519-1560): K as 3x3, R, t, then
    # the five distortion coefficients cv::calibrateCamera handed over.
    ccv = np.array([[float(t) for t in l.split()[2:]] for l in out.splitlines() if l.startswith("CCV ")])
    assert ccv.shape == (cfg.cn, 9)  # the distortion vectors keep their five entries: k4..k6 are not freed
    assert np.array_equal(ccv, k_shim[:, :9])
    assert [l for l in out.splitlines() if l.startswith("FILE_HEAD")][0].split()[1] == str(cfg.cn)
    rows = [l.split()[1:] for l in out.splitlines() if l.startswith("FILE ")]
    assert len(rows) == cfg.cn and all(len(r) == 1 + 9 + 9 + 3 + 5 for r in rows)
    for c, r in enumerate(rows):
        assert r[0] == f"cam{c}"
        K = np.array([float(t) for t in r[1:10]]).reshape(3, 3)
        d = np.array([float(t) for t in r[22:27]])
        want = np.array([[k_shim[c, 0], 0, k_shim[c, 1]], [0, k_shim[c, 2], k_shim[c, 3]], [0, 0, 1]])
        assert np.abs(K - want).max() <= 1e-5 * 1500  # `out << double`: six significant digits
        assert np.abs(d - k_shim[c, 4:9]).max() <= 1e-5 * max(1e-6, np.abs(k_shim[c, 4:9]).max())
        if mask:
            assert abs(K[0, 0] - intr[c, 0]) > 0.01 or abs(K[0, 2] - intr[c, 1]) > 0.01  # not the hand-over any more


@pytest.mark.gpu
def test_points_txt_ingest_feeds_the_gpu_handle():
    """corners written the way CameraCali writes points.txt
    (camera-calibration.cpp:283-294: 9 significant digits of float32 detections), read back with
    calico_amd/host/points_txt.hpp and added to a handle camera by camera: the reduced normal
    equations equal those of the direct feed."""
    assert os.path.exists(EXE), "build tests/helpers first (CPU side)"
    out = subprocess.check_output([EXE, "--run-points", "25"], text=True, timeout=120)
    line = [l for l in out.splitlines() if l.startswith("POINTS")][0]
    assert int(line.split("n_obs=")[1].split()[0]) > 10000
    assert float(line.split("S_rel=")[1].split()[0]) <= 1e-12
    assert float(line.split("rhs_rel=")[1].split()[0]) <= 1e-12


def _lcg(seed):
    s = seed
    while True:
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        yield (s >> 8) / float(1 << 24) - 0.5


@pytest.mark.gpu
def test_stage4_loop_through_the_shim_matches_the_c_abi():
    """AddToProblemAlgebraicError / SolveWriteBackToMCAlgebraicError of the mirror class, driven
    as calico.cpp:637-697 drives them (variables become available in three waves), against
    the same schedule replayed on calico_ae_* from Python."""
    assert os.path.exists(EXE), "build tests/helpers first (CPU side)"
    tn = 24
    out = subprocess.check_output([EXE, "--run-ae", str(tn)], text=True, timeout=120)
    solves = [l for l in out.splitlines() if l.startswith("AE_SOLVE")]
    assert len(solves) == 3, out
    shim = [(int(l.split("relations=")[1].split()[0]), int(l.split("iterations=")[1].split()[0]),
             float(l.split("initial=")[1].split()[0]), float(l.split("final=")[1])) for l in solves]
    truth = [l for l in out.splitlines() if l.startswith("AE_TRUTH")][0]
    assert float(truth.split("rot=")[1].split()[0]) < 1e-3 and float(truth.split("trans=")[1]) < 2.0

    cfg = synth.preset("A", tn=tn)
    k = synth.frs(cfg)  # (cam, pat, tim) in first-occurrence order = the helper's equation order
    cam, pat, tim = k[:, 0], k[:, 1], k[:, 2]

    def uninvert(x):  # generator 7-vectors hold patterns / times inverted
        M = synth.quats_to_matrices(x.reshape(-1, 7))
        M[cfg.cn:] = synth.rigid_inverses(M[cfg.cn:])
        return M

    Vg, V0 = uninvert(synth.ground_truth(cfg)), uninvert(synth.initial_guess(cfg))
    A = Vg[cam] @ np.linalg.inv(Vg[pat] @ Vg[tim])
    g = _lcg(99)
    for i in range(len(k)):
        for r in range(3):
            A[i, r, 3] += 0.5 * next(g)
    a7 = synth.matrices_to_quats(A)
    x = synth.matrices_to_quats(V0).reshape(-1)
    P = capi.AeProblem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
    ready = np.zeros(cfg.vn, bool)
    ready[[cfg.p_star_var, cfg.t_star_var]] = True
    added = np.zeros(len(k), bool)
    bounds = [(0, cfg.cn + cfg.pn), (cfg.cn + cfg.pn, cfg.cn + cfg.pn + tn // 2), (cfg.cn + cfg.pn + tn // 2, cfg.vn)]
    for (lo, hi), (rel_s, it_s, init_s, fin_s) in zip(bounds, shim):
        ready[lo:hi] = True
        new = ready[cam] & ready[pat] & ready[tim] & ~added
        added |= new
        P.add_relations(cam[new], pat[new], tim[new], a7[new])
        x, s = P.solve(x)
        assert P.num_observations == rel_s
        assert s.num_iterations == it_s
        assert abs(s.initial_cost - init_s) <= 1e-6 * init_s
        assert abs(s.final_cost - fin_s) <= 1e-6 * fin_s
    assert added.all()


@pytest.mark.gpu
def test_point_reconstruction_through_the_shim_matches_the_c_abi():
    """ReconstructXFunctionIDsMCwithDLTs of calico_amd/host/reconstruct_x.hpp (gathers views
    and detected corners from the reference's structures) against calico_rae_reconstruct
    fed from Python with the same views."""
    assert os.path.exists(EXE), "build tests/helpers first (CPU side)"
    tn = 12
    out = subprocess.check_output([EXE, "--run-rae", str(tn)], text=True, timeout=120)
    rows = [l.split() for l in out.splitlines() if l.startswith("RAE ")]
    cfg = synth.preset("A", tn=tn)
    xyz, has, _its, _fc, _s = capi.rae_reconstruct(*synth.rae_inputs(cfg))
    assert len(rows) == has.size
    shim_has = np.array([int(r[2]) for r in rows], bool)
    shim_xyz = np.array([[float(t) for t in r[3:6]] for r in rows])
    assert np.array_equal(shim_has, has) and has.all()
    # A_hat goes 7-vector -> 4x4 -> inverse -> product on the shim side: rounding-level differences
    assert np.abs(shim_xyz - xyz).max() < 1e-6
    assert np.abs(xyz - synth.points(cfg)).max() < 1.0


@pytest.mark.gpu
def test_report_composition_through_the_shim():
    """WriteSolutionAssessErrorII / ReconstructionAccuracyErrorAndWriteII / OutputRunResults of
    calico_amd/host/assess.hpp (multicamera.cpp:1625-1897, 792-1046, 693-789) after the Stage-5
    loop, as calico.cpp:731-798 calls them: `total_results.txt` and the files beside it, every
    number recomputed here without the mirror - the algebraic error per FR ||C - A P T||_F^2 in
    numpy, the reprojection error per FR over ALL detected corners (the minimisation used every
    other one) through the C ABI, the RAE statistics from calico_rae_reconstruct."""
    assert os.path.exists(EXE), "build tests/helpers first (CPU side)"
    tn = 10
    out = subprocess.check_output([EXE, "--run-report", str(tn)], text=True, timeout=180)
    lines = out.splitlines()
    x_shim = np.array([float(t) for t in [l for l in lines if l.startswith("X ")][0].split()[1:]])
    k_shim = np.array([float(t) for t in [l for l in lines if l.startswith("K ")][0].split()[1:]]).reshape(-1, 12)
    tr = [l[len("FILE[total_results.txt] "):] for l in lines if l.startswith("FILE[total_results.txt]")]
    cfg = synth.preset("A", tn=tn)
    assert np.array_equal(k_shim, synth.intrinsics(cfg))
    obs = synth.observations(cfg)
    k = synth.frs(cfg)  # equation order of the helper
    cam, pat, tim = k[:, 0], k[:, 1], k[:, 2]
    n_fr = len(k)

    def uninvert(x):
        M = synth.quats_to_matrices(x.reshape(-1, 7))
        M[cfg.cn:] = synth.rigid_inverses(M[cfg.cn:])
        return M

    # ---- algebraic error
    Vg, V = uninvert(synth.ground_truth(cfg)), uninvert(x_shim)
    A = Vg[cam] @ np.linalg.inv(Vg[pat] @ Vg[tim])
    g = _lcg(99)
    for i in range(n_fr):
        for r in range(3):
            A[i, r, 3] += 0.5 * next(g)
    c1 = ((V[cam] - A @ V[pat] @ V[tim]) ** 2).sum(axis=(1, 2))
    assert tr[0].startswith("Algebraic error cost function error, summed over all FR")
    name, val = tr[1].split(":")
    assert name == "incremental".ljust(20) and abs(float(val) - c1.sum()) <= 2e-5 * c1.sum()
    avg_line = tr[tr.index("Algebraic error cost function error, averaged by number of FRs (equation 16) ") + 1]
    assert abs(float(avg_line.split(":")[1]) - c1.sum() / n_fr) <= 2e-5 * c1.sum() / n_fr

    # ---- reprojection error over all detected corners, at the refined poses
    P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
    P.set_intrinsics(k_shim)
    P.set_points(synth.points(cfg))
    P.add_observations(*obs)
    fc, fp, ft, cnt, mse, rrmse = P.fr_errors(x_shim)
    where = {(int(a), int(b), int(c)): i for i, (a, b, c) in enumerate(zip(cam, pat, tim))}
    mse_eq = np.zeros(n_fr)
    for a, b, c, m in zip(fc, fp, ft, mse):
        mse_eq[where[(int(a), int(b), int(c))]] = m
    rr_line = [l for l in tr if "Reprojection error, rrmse:" in l][0]
    assert rr_line.startswith("incremental: ")
    assert abs(float(rr_line.split("rrmse:")[1]) - np.sqrt(mse_eq.sum() / n_fr)) <= 2e-5 * rrmse
    assert 0.1 < rrmse < 0.2  # sqrt(2) x the generator's 0.1 px
    # the minimisation saw every other corner only: the report's number is over all of them
    half = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
    half.set_intrinsics(k_shim)
    half.set_points(synth.points(cfg))
    sel = obs[3] % 2 == 0
    half.add_observations(*[a[sel] for a in obs])
    assert abs(half.fr_errors(x_shim)[5] - rrmse) > 1e-4

    # ---- per-FR blocks
    i0 = tr.index("Algebraic error, per foundational relationship. ")
    per_c1 = np.array([float(t) for t in tr[i0 + 1].split(":")[1].split()])
    assert per_c1.shape == (n_fr,) and np.abs(per_c1 - c1).max() <= 2e-5 * c1.max()
    i1 = tr.index("Reprojection error, per foundational relationship. ")
    body, total = tr[i1 + 1].split(":")[1].split(" Total ")
    per_rp = np.array([float(t) for t in body.split()])
    assert per_rp.shape == (n_fr,) and np.abs(per_rp - mse_eq).max() <= 2e-5 * mse_eq.max()
    assert abs(float(total) - mse_eq.sum()) <= 2e-5 * mse_eq.sum()

    # ---- RAE
    xyz, has, _its, _fc, _s = capi.rae_reconstruct(*synth.rae_inputs(cfg, x=x_shim))
    assert has.all()
    norms = ((xyz - synth.points(cfg)) ** 2).sum(axis=1)
    mu, sd, med = norms.mean(), np.sqrt(((norms - norms.mean()) ** 2).mean()), np.sort(norms)[norms.size // 2]
    assert [l for l in tr if l.startswith("Number valid image points")][0].split()[-1] == str(norms.size)
    rae = [l for l in tr if l.startswith("incremental: RAE w/ BA, average, stddev, median :")][0]
    got = [float(t) for t in rae.split(":")[2].split(",")]
    assert np.allclose(got, [mu, sd, med], rtol=2e-5, atol=0)
    sq = [l for l in tr if l.startswith("SQRT -- RAE w/ BA, average, stddev, median :")][0]
    assert np.allclose([float(t) for t in sq.split(":")[1].split(",")], np.sqrt([mu, sd, med]), rtol=2e-5, atol=0)

    # ---- the files beside it
    est = [l for l in lines if l.startswith("FILE[EstimatedVsGroundTruth2.txt]")]
    assert est[0].endswith(" 0") and est[3].split("squared norm ")[1] and abs(float(est[3].split("squared norm ")[1]) - norms[0]) <= 2e-5 * norms[0]
    a_est = [l[len("FILE[cameras-incremental/A_est.txt] "):] for l in lines if l.startswith("FILE[cameras-incremental/A_est.txt]")]
    assert a_est[0] == "Equation 0"
    Ap = V[cam[0]] @ np.linalg.inv(V[tim[0]]) @ np.linalg.inv(V[pat[0]])
    got = np.array([[float(t) for t in a_est[1 + r].split()] for r in range(4)])
    assert np.abs(got - Ap).max() <= 1e-5 * np.abs(Ap).max()
    var = [l[len("FILE[cameras-incremental/variables.txt] "):] for l in lines if l.startswith("FILE[cameras-incremental/variables.txt]")]
    assert var[0] == "C 0" and var[5] == "C 1"
    assert [l for l in lines if l.startswith("FILE[camera_cali_incremental.txt]")][0].split()[-1] == str(cfg.cn)
