"""world_size-2 gloo test (CPU) of the multi-GPU exchange design:
observations are sharded by time index, every rank eliminates its own time blocks
with purely local scaling/damping, and ONE all-reduce of [P | q | diag(H_rr)] followed
by the replicated reduced-block scaling/damping reproduces the single-process reduced
system.  This is the algebra of assemble_reduced() in calico_amd/csrc/rp_api.hip,
restated in numpy on top of the oracle's per-observation r/J; what travels is the product's
own exchange buffer (calico_rp_exchange_pack_host / _unpack_host, the layout functions the
device pack / unpack kernels use), through a real torch.distributed collective between two
processes."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _local_partial(cfg, oracle_api, t0, t1, x, radius):
    """[P | q | hdiag] of one shard, reduced blocks = all cameras and patterns but p*."""
    from calico_amd import capi, synth

    P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, api=oracle_api)
    P.set_intrinsics(synth.intrinsics(cfg))
    P.set_points(synth.points(cfg))
    obs = synth.observations(cfg, t0, t1)
    P.add_observations(*obs)
    _, r, J = P.eval(x)
    cam, pat, tim = obs[0], obs[1], obs[2]
    rvars = [v for v in range(cfg.cn + cfg.pn) if v != cfg.p_star_var]
    ridx = {v: i for i, v in enumerate(rvars)}
    n = 7 * len(rvars)
    Hrr, gr = np.zeros((n, n)), np.zeros(n)
    Pm = None
    per_time = {}
    for i in range(cam.size):
        blocks = [(cam[i], 0)]
        if pat[i] != cfg.p_star_var:
            blocks.append((pat[i], 1))
        Ji, ri = J[i], r[i]
        for (va, ba) in blocks:
            ia = 7 * ridx[va]
            gr[ia:ia + 7] += Ji[:, 7 * ba:7 * ba + 7].T @ ri
            for (vb, bb) in blocks:
                ib = 7 * ridx[vb]
                Hrr[ia:ia + 7, ib:ib + 7] += Ji[:, 7 * ba:7 * ba + 7].T @ Ji[:, 7 * bb:7 * bb + 7]
        if tim[i] != cfg.t_star_var:
            E, g, H = per_time.setdefault(tim[i], (np.zeros((7, 7)), np.zeros(7), np.zeros((n, 7))))
            Jt = Ji[:, 14:21]
            E += Jt.T @ Jt
            g += Jt.T @ ri
            for (va, ba) in blocks:
                ia = 7 * ridx[va]
                H[ia:ia + 7] += Ji[:, 7 * ba:7 * ba + 7].T @ Jt
    Pm, q = Hrr.copy(), gr.copy()
    for t, (E, g, H) in per_time.items():
        d = np.diag(E)
        s = 1.0 / (1.0 + np.sqrt(d))
        D2 = np.clip(s * s * d, 1e-6, 1e32) / radius
        Es = (s[:, None] * E * s[None, :]) + np.diag(D2)
        L = np.linalg.cholesky(Es)
        Y = np.linalg.solve(L, (H * s[None, :]).T).T  # H s L^-T
        z = np.linalg.solve(L, s * g)
        Pm -= Y @ Y.T
        q -= Y @ z
    return Pm, q, gr, np.diag(Hrr).copy(), n


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import ctypes as C

    import torch
    import torch.distributed as dist

    from calico_amd import capi, synth
    from tests.oracle_lib import load_oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = synth.custom(5, 3, 14, fr_drop_prob=0.2)
    counts = synth.obs_per_time(cfg)
    b = (C.c_int32 * (world + 1))()
    assert capi.load().partition_times(cfg.tn, counts.ctypes.data_as(C.POINTER(C.c_int64)), world, b) == 0
    x = synth.initial_guess(cfg)
    radius = 1e4
    Pm, q, gr, hd, n = _local_partial(cfg, load_oracle(), b[rank], b[rank + 1], x, radius)
    # the rank's working buffer [P ld x ld | q | g_r | diag] -> the product's exchange layout
    # (calico_rp_exchange_pack_host: the offset functions of the device kernels) -> all-reduce -> back
    api = capi.load()
    dp = C.POINTER(C.c_double)
    ld = api.exchange_ld(n)
    work = np.zeros(ld * ld + 3 * ld)
    work[:ld * ld].reshape(ld, ld)[:n, :n] = np.tril(Pm) + np.triu(np.full((n, n), np.nan), 1)  # only the lower triangle travels
    for k, vec in enumerate((q, gr, hd)):
        work[ld * ld + k * ld:ld * ld + k * ld + n] = vec
    buf = np.empty(api.exchange_count(n))
    assert buf.size == n * (n + 1) // 2 + 3 * n
    assert api.exchange_pack_host(n, work.ctypes.data_as(dp), buf.ctypes.data_as(dp)) == 0
    assert np.isfinite(buf).all()
    t = torch.from_numpy(buf)
    dist.all_reduce(t)  # the one exchange per linear solve
    work[:] = 0.0
    assert api.exchange_unpack_host(n, buf.ctypes.data_as(dp), work.ctypes.data_as(dp)) == 0
    Pm = work[:ld * ld].reshape(ld, ld)[:n, :n].copy()
    assert np.array_equal(Pm, Pm.T)
    q, hd = work[ld * ld:ld * ld + n].copy(), work[ld * ld + 2 * ld:ld * ld + 2 * ld + n].copy()
    s = 1.0 / (1.0 + np.sqrt(hd))
    D2 = np.clip(s * s * hd, 1e-6, 1e32) / radius
    S = s[:, None] * Pm * s[None, :] + np.diag(D2)
    rhs = s * q
    if rank == 0:
        np.savez(out_path, S=S, rhs=rhs, bounds=np.array(b[:]))
    dist.barrier()
    dist.destroy_process_group()


def test_time_sharded_exchange_reproduces_single_process_system(oracle_api, tmp_path):
    import torch.multiprocessing as mp

    from calico_amd import capi, synth

    out = str(tmp_path / "sharded.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    cfg = synth.custom(5, 3, 14, fr_drop_prob=0.2)
    P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, api=oracle_api)
    synth.fill_problem(P, cfg)
    S, rhs = P.normal_eq(synth.initial_guess(cfg), radius=1e4, jacobi_scaling=True)
    assert got["S"].shape == S.shape
    assert 0 < got["bounds"][1] < cfg.tn  # both ranks own times
    assert np.abs(got["S"] - S).max() <= 1e-10 * np.abs(S).max()
    assert np.abs(got["rhs"] - rhs).max() <= 1e-10 * np.abs(rhs).max()
