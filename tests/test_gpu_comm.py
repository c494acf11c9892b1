"""Exercises the RCCL code path (communicator, packed all-reduce, scalar all-gather,
time-block gather at lm_end) on the one GPU available to tests: a 1-rank communicator
must leave every result unchanged.  Multi-rank algebra is covered on CPU by
tests/test_dist_gloo.py; real N > 1 runs are the driver's (bench.py --gpus N)."""
import ctypes as C

import numpy as np
import pytest

from calico_amd import capi, synth

pytestmark = pytest.mark.gpu


def test_single_rank_communicator_changes_nothing(oracle_api):
    cfg = synth.custom(6, 3, 20, fr_drop_prob=0.2)
    x0 = synth.initial_guess(cfg)
    plain = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
    synth.fill_problem(plain, cfg)
    xa, sa = plain.solve(x0)

    comm = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
    synth.fill_problem(comm, cfg)
    buf = C.create_string_buffer(capi.UNIQUE_ID_BYTES)
    assert capi.load().comm_unique_id(C.cast(buf, C.c_void_p)) == 0
    comm.comm_init(1, 0, bytes(buf.raw))
    xb, sb = comm.solve(x0)
    assert sa.num_iterations == sb.num_iterations
    assert abs(sa.final_cost - sb.final_cost) <= 1e-12 * sa.final_cost
    assert np.abs(xa - xb).max() <= 1e-9 * np.abs(xa).max()


def test_sharded_handle_uses_the_global_block_set():
    """A rank that sees only some cameras still solves for the full reduced system."""
    cfg = synth.custom(6, 3, 20)
    obs = synth.observations(cfg)
    keep = obs[0] < 4  # this "rank" never observes cameras 4 and 5
    P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
    P.set_intrinsics(synth.intrinsics(cfg))
    P.set_points(synth.points(cfg))
    P.add_observations(*[a[keep] for a in obs])
    buf = C.create_string_buffer(capi.UNIQUE_ID_BYTES)
    assert capi.load().comm_unique_id(C.cast(buf, C.c_void_p)) == 0
    P.comm_init(1, 0, bytes(buf.raw))
    x0 = synth.initial_guess(cfg)
    S, b = P.normal_eq(x0)
    assert S.shape[0] == 7 * (cfg.cn + cfg.pn - 1)
    x, s = P.solve(x0)
    # unobserved cameras do not move, observed ones do
    xr, x0r = x.reshape(-1, 7), x0.reshape(-1, 7)
    assert np.array_equal(xr[4], x0r[4]) and np.array_equal(xr[5], x0r[5])
    assert not np.array_equal(xr[0], x0r[0])
    assert s.final_cost < s.initial_cost
