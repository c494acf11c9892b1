"""Test helper: one rank of a time-sharded solve, all ranks on HIP device 0, collectives
through CALICO_RP_COLLECTIVES_LIB (tests/helpers/shm_collectives.cpp).
usage: two_rank_solve.py <rank> <nranks> <workdir> <cn> <pn> <tn> <mode> [<json: generator fields>]
mode: fused | materialised (Stage 5), refined (fused, camera parameters refined from perturbed
values) or ae (Stage 4 relations, sharded by time as well)"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from calico_amd import capi, synth  # noqa: E402

rank, nranks, work = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
cn, pn, tn, mode = int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
import json  # noqa: E402

cfg = synth.custom(cn, pn, tn, **(json.loads(sys.argv[8]) if len(sys.argv) > 8 else dict(fr_drop_prob=0.2)))
counts = synth.obs_per_time(cfg)
bounds = (C.c_int32 * (nranks + 1))()
assert capi.load().partition_times(cfg.tn, counts.ctypes.data_as(C.POINTER(C.c_int64)), nranks, bounds) == 0
t0, t1 = bounds[rank], bounds[rank + 1]
if mode == "ae":
    cam, pat, tim, a7, _xt, x0 = synth.ae_relations(cfg)
    mine = (tim - cfg.cn - cfg.pn >= t0) & (tim - cfg.cn - cfg.pn < t1)
    P = capi.AeProblem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
    P.add_relations(cam[mine], pat[mine], tim[mine], a7[mine])
else:
    P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
    P.set_jacobian_mode("fused" if mode == "refined" else mode)
    synth.fill_problem(P, cfg, t0, t1)
    if mode == "refined":
        from tests.test_gpu_refine import MASK_OPENCV, _perturbed_intrinsics  # noqa: E402
        P.set_intrinsics(_perturbed_intrinsics(cfg, seed=11))
        P.set_refine_intrinsics(MASK_OPENCV)
    x0 = synth.initial_guess(cfg)
uid_file = os.path.join(work, "uid.bin")
if rank == 0:
    buf = C.create_string_buffer(capi.UNIQUE_ID_BYTES)
    assert capi.load().comm_unique_id(C.cast(buf, C.c_void_p)) == 0
    with open(uid_file + ".tmp", "wb") as f:
        f.write(buf.raw)
    os.rename(uid_file + ".tmp", uid_file)
else:
    for _ in range(20000):
        if os.path.exists(uid_file):
            break
        time.sleep(0.001)
uid = open(uid_file, "rb").read()
P.comm_init(nranks, rank, uid)
opt = capi.default_options(max_num_iterations=25, function_tolerance=1e-12, parameter_tolerance=1e-12,
                           gradient_tolerance=1e-14)
if os.environ.get("CALICO_RP_TEST_FAIL_AT"):
    # abort-path test: a rank fails between two collectives; every rank must come back with an
    # error (no hang) and the handle must refuse further work
    try:
        P.solve(x0, opt)
        outcome = "ok"
    except RuntimeError as e:
        outcome = str(e)
    try:
        P.solve(x0, opt)
        again = "ok"
    except RuntimeError as e:
        again = str(e)
    with open(os.path.join(work, "rank%d.txt" % rank), "w") as f:
        f.write(outcome + "\n" + again + "\n")
    P.close()
    sys.exit(0)
x, s = P.solve(x0, opt)
np.savez(os.path.join(work, "rank%d.npz" % rank), x=x, it=s.num_iterations, ok=s.num_successful_steps,
         c0=s.initial_cost, c1=s.final_cost, term=s.termination_type, t0=t0, t1=t1,
         k=P.get_intrinsics() if mode == "refined" else np.zeros(1))
P.close()
