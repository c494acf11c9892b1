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
