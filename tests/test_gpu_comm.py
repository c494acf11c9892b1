"""Exercises the RCCL code path (communicator, packed all-reduce, scalar all-gather,
time-block gather at lm_end) on the one GPU available to tests: a 1-rank communicator
must leave every result unchanged.  Multi-rank algebra is covered on CPU by
tests/test_dist_gloo.py; real N > 1 runs are the driver's (bench.py --gpus N)."""
import ctypes as C
import os

import numpy as np
import pytest

from calico_amd import capi, synth


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


HELPERS = os.path.join(ROOT, "tests", "helpers")
SHM_LIB = os.path.join(ROOT, "tests", "_build", "libshm_collectives.so")
# the library built from the same sources with -DCALICO_RP_TEST_HOOKS (calico_amd/csrc/Makefile):
# the product library has neither the collectives stand-in nor the injected failure
HOOKS_LIB = os.path.join(ROOT, "tests", "_build", "libcalico_rp_testhooks.so")


@pytest.mark.parametrize("mode,nranks,shape", [("fused", 2, (6, 3, 24)), ("materialised", 2, (6, 3, 24)), ("ae", 2, (6, 3, 24)),
                                               ("fused", 4, (5, 3, 30)), ("fused", 3, (4, 2, 2)), ("refined", 2, (6, 3, 24)),
                                               ("refined", 3, (5, 2, 30))])
def test_two_ranks_on_one_gpu_match_the_single_process_solve(tmp_path, mode, nranks, shape):
    """RCCL refuses two ranks on one device, so the N > 1 control flow (time sharding,
    global block set, packed all-reduce, scalar all-gather, time-block merge at the end)
    runs here with two processes on device 0 and a shared-memory stand-in for the six
    collective entry points (CALICO_RP_COLLECTIVES_LIB): the sharded solve must take the
    same steps and reach the same point as one process holding everything."""
    import subprocess
    import sys

    env = dict(os.environ, CALICO_RP_LIBRARY=HOOKS_LIB, CALICO_RP_COLLECTIVES_LIB=SHM_LIB, CALICO_RP_COMM_TIMEOUT_S="60")
    procs = [subprocess.Popen([sys.executable, os.path.join(HELPERS, "two_rank_solve.py"), str(r), str(nranks), str(tmp_path),
                               *map(str, shape), mode], env=env) for r in range(nranks)]
    for p in procs:
        assert p.wait(timeout=240) == 0
    out = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(nranks)]
    r0 = out[0]
    for a, b in zip(out, out[1:]):
        assert a["t1"] == b["t0"]
        assert np.array_equal(a["x"], b["x"])  # every rank returns the full x
    assert int(out[0]["t0"]) == 0 and int(out[-1]["t1"]) == shape[2]
    if nranks <= shape[2]:
        assert all(int(o["t1"]) > int(o["t0"]) for o in out)
    else:  # more ranks than times (4 cameras x 2 boards x 2 times over 3 ranks): a rank without observations takes part
        assert any(int(o["t1"]) == int(o["t0"]) for o in out)
    cfg = synth.custom(*shape, fr_drop_prob=0.2)
    if mode == "ae":  # Stage-4 relations on the same sharding and exchange
        cam, pat, tim, a7, _xt, x0 = synth.ae_relations(cfg)
        P = capi.AeProblem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
        P.add_relations(cam, pat, tim, a7)
    else:
        P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
        P.set_jacobian_mode("fused" if mode == "refined" else mode)
        synth.fill_problem(P, cfg)
        if mode == "refined":  # the camera-parameter blocks join the replicated (reduced) set
            from tests.test_gpu_refine import MASK_OPENCV, _perturbed_intrinsics
            P.set_intrinsics(_perturbed_intrinsics(cfg, seed=11))
            P.set_refine_intrinsics(MASK_OPENCV)
        x0 = synth.initial_guess(cfg)
    # function tolerance above the rounding floor of the cost (1e-14 relative: there the sign of the
    # last cost changes, and with it the step count, depends on the summation order of the shards)
    opt = capi.default_options(max_num_iterations=25, function_tolerance=1e-12, parameter_tolerance=1e-12,
                               gradient_tolerance=1e-14)
    x, s = P.solve(x0, opt)
    assert (int(r0["it"]), int(r0["ok"]), int(r0["term"])) == (s.num_iterations, s.num_successful_steps,
                                                               s.termination_type)
    assert abs(float(r0["c0"]) - s.initial_cost) <= 1e-12 * s.initial_cost
    assert abs(float(r0["c1"]) - s.final_cost) <= 1e-9 * s.final_cost
    assert np.abs(r0["x"] - x).max() < 1e-8
    if mode == "refined":
        for o in out:  # every rank holds the refined values
            assert np.array_equal(o["k"], out[0]["k"])
        assert np.abs(out[0]["k"] - P.get_intrinsics()).max() < 1e-7
        assert np.abs(P.get_intrinsics() - _perturbed_intrinsics(cfg, seed=11)).max() > 1e-3
    P.close()


def test_a_failing_rank_does_not_leave_its_peer_blocked(tmp_path):
    """Rank 1 fails at iteration 2 of a two-rank solve, between the all-gather of one iteration
    and the all-reduce of the next (CALICO_RP_TEST_FAIL_AT, a test hook that throws where a HIP
    error would).  It must abort the communicator before returning the error, its peer's
    pending collective must come back with an error instead of blocking, and both handles must
    refuse further work."""
    import subprocess
    import sys

    env = dict(os.environ, CALICO_RP_LIBRARY=HOOKS_LIB, CALICO_RP_COLLECTIVES_LIB=SHM_LIB, CALICO_RP_TEST_FAIL_AT="1:2", CALICO_RP_COMM_TIMEOUT_S="20")
    procs = [subprocess.Popen([sys.executable, os.path.join(HELPERS, "two_rank_solve.py"), str(r), "2", str(tmp_path),
                               "6", "3", "24", "fused"], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=120) == 0
    out = [open(tmp_path / ("rank%d.txt" % r)).read().splitlines() for r in range(2)]
    assert "injected failure" in out[1][0] and "communicator aborted" in out[1][0]
    assert out[0][0] != "ok" and "communicator aborted" in out[0][0]  # the peer: error, not a hang
    for r in range(2):
        assert "aborted" in out[r][1]  # second call on the same handle: CALICO_RP_ESTATE


def test_bench_two_ranks_on_one_gpu(tmp_path):
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one rank per
    process), with both ranks on device 0 and the shared-memory collectives."""
    import json
    import subprocess
    import sys

    env = dict(os.environ, CALICO_RP_LIBRARY=HOOKS_LIB, CALICO_RP_COLLECTIVES_LIB=SHM_LIB, CALICO_BENCH_ONE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "A", "--steps",
           "3", "--warmup", "1", "--no-cpu-baseline", "--materialised-steps", "1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["accepted_steps"] == 3
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "A", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", "--materialised-steps", "0"], capture_output=True, text=True, timeout=400)
    d1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["observations"] == d1["config"]["observations"]
    assert abs(d["config"]["cost_initial"] - d1["config"]["cost_initial"]) <= 1e-12 * d1["config"]["cost_initial"]


def test_a_peer_that_never_turns_up_ends_in_an_error(tmp_path):
    """Rank 0 of a two-rank job whose peer never starts: the communicator's set-up must come back
    with an error after CALICO_RP_COMM_TIMEOUT_S instead of waiting for ever (with RCCL the
    bootstrap's own time-out does that; the stand-in's waits carry the same limit, so that a test
    that loses a rank cannot leave processes spinning on the box)."""
    import subprocess
    import sys
    import time

    env = dict(os.environ, CALICO_RP_LIBRARY=HOOKS_LIB, CALICO_RP_COLLECTIVES_LIB=SHM_LIB, CALICO_RP_COMM_TIMEOUT_S="5")
    t = time.time()
    r = subprocess.run([sys.executable, os.path.join(HELPERS, "two_rank_solve.py"), "0", "2", str(tmp_path), "4", "2", "6", "fused"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and time.time() - t < 60
    assert "comm_init" in r.stderr, r.stderr[-1500:]


_SEEDS = range(int(os.environ["CALICO_TEST_SEEDS"])) if "CALICO_TEST_SEEDS" in os.environ else range(3)


@pytest.mark.parametrize("seed", _SEEDS)
def test_random_networks_over_two_to_four_ranks(tmp_path, seed):
    """The sharded solve over shapes nobody picked by hand: random block counts, p* / t*, shares of
    FRs and corners dropped, two to four ranks (more ranks than times included: ranks without
    observations) on one GPU through the collectives stand-in, against one process holding
    everything.  CALICO_TEST_SEEDS=100 is a fuzzing run."""
    import json
    import subprocess
    import sys

    rng = np.random.default_rng(4000 + seed)
    shape = (int(rng.integers(1, 8)), int(rng.integers(1, 4)), int(rng.integers(1, 30)))
    nranks = int(rng.integers(2, 5))
    fields = dict(fr_drop_prob=float(rng.choice([0.0, 0.3])), corner_drop_prob=float(rng.choice([0.0, 0.5])),
                  one_board_per_cam_time=int(rng.integers(0, 2)), p_star_idx=int(rng.integers(0, shape[1])),
                  t_star_idx=int(rng.integers(0, shape[2])), seed=500 + seed)
    cfg = synth.custom(*shape, **fields)
    if synth.count(cfg) == 0:
        pytest.skip("the draw left no observation")
    env = dict(os.environ, CALICO_RP_LIBRARY=HOOKS_LIB, CALICO_RP_COLLECTIVES_LIB=SHM_LIB, CALICO_RP_COMM_TIMEOUT_S="60")
    procs = [subprocess.Popen([sys.executable, os.path.join(HELPERS, "two_rank_solve.py"), str(r), str(nranks), str(tmp_path),
                               *map(str, shape), "fused", json.dumps(fields)], env=env) for r in range(nranks)]
    for p in procs:
        assert p.wait(timeout=240) == 0
    out = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(nranks)]
    for a, b in zip(out, out[1:]):
        assert a["t1"] == b["t0"] and np.array_equal(a["x"], b["x"])
    P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
    n_obs = synth.fill_problem(P, cfg)[0].size
    x0 = synth.initial_guess(cfg)
    opt = capi.default_options(max_num_iterations=25, function_tolerance=1e-12, parameter_tolerance=1e-12,
                               gradient_tolerance=1e-14)
    x, s = P.solve(x0, opt)
    P.close()
    r0 = out[0]
    assert abs(float(r0["c0"]) - s.initial_cost) <= 1e-12 * s.initial_cost
    # a sharded handle keeps every camera and pattern as a reduced block, seen or not: its reduced system is
    # the single process's plus damped identity blocks, the steps are the same where the network determines them
    if 2 * n_obs > 6 * 7 * cfg.vn:
        # (one draw in 1 000 stops one iteration apart: the last cost change sits at the function tolerance)
        assert int(r0["term"]) == s.termination_type and abs(int(r0["it"]) - s.num_iterations) <= 1
        assert abs(float(r0["c1"]) - s.final_cost) <= 1e-8 * s.final_cost + 1e-13 * s.initial_cost
        assert np.abs(r0["x"] - x).max() < 1e-7 * max(1.0, np.abs(x).max())
    else:
        assert np.isfinite(float(r0["c1"])) and float(r0["c1"]) <= float(r0["c0"])
