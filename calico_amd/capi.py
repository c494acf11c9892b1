"""ctypes view of the C ABI declared in include/calico_rp.h.

Python is only plumbing here (tests, bench.py, torch.distributed bootstrap): the
product is libcalico_rp.so.  `Api` binds a shared library that exports the
calico_rp_* entry points under a given symbol prefix; the same binder is used by
tests/ to drive the CPU checker, which exports the same shapes under its own prefix.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "_build", "libcalico_rp.so")

UNIQUE_ID_BYTES = 128

TERMINATION = {0: "CONVERGENCE", 1: "NO_CONVERGENCE", 2: "FAILURE"}


JACOBIAN_FUSED, JACOBIAN_MATERIALISED = 0, 1


class Options(C.Structure):
    _fields_ = [
        ("max_num_iterations", C.c_int32),
        ("jacobi_scaling", C.c_int32),
        ("verbose", C.c_int32),
        ("profile", C.c_int32),
        ("initial_trust_region_radius", C.c_double),
        ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double),
        ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double),
        ("function_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
    ]


def default_options(**kw):
    """Ceres defaults + max_num_iterations=10 (calico.cpp:709); same as
    calico_rp_options_default()."""
    o = Options(10, 1, 0, 0, 1e4, 1e16, 1e-32, 1e-3, 1e-6, 1e32, 1e-6, 1e-10, 1e-8)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class StageMs(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "jacobian_sweep", "accumulate", "reduce", "schur", "cholesky", "backsub", "cost_sweep", "exchange")] + [
        (n, C.c_int32) for n in (
            "n_jacobian_sweep", "n_accumulate", "n_reduce", "n_schur", "n_cholesky", "n_backsub",
            "n_cost_sweep", "n_exchange")]


class Summary(C.Structure):
    _fields_ = [
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
        ("num_iterations", C.c_int32),
        ("num_successful_steps", C.c_int32),
        ("num_unsuccessful_steps", C.c_int32),
        ("termination_type", C.c_int32),
        ("num_residual_blocks", C.c_int64),
        ("total_ms", C.c_double),
        ("stage", StageMs),
        ("message", C.c_char * 128),
    ]


class RaeSummary(C.Structure):
    _fields_ = [("n_points", C.c_int32), ("n_solved", C.c_int32), ("sweeps", C.c_int32),
                ("max_point_iterations", C.c_int32), ("total_ms", C.c_double), ("sweep_ms", C.c_double),
                ("update_ms", C.c_double)]


class RpError(RuntimeError):
    pass


def _p(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


class Api:
    """Binds <prefix>create/... of one shared library."""

    def __init__(self, cdll, prefix="calico_rp_"):
        self.lib = cdll
        self.prefix = prefix
        f = self._f
        vp = C.c_void_p
        dp, ip, fp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_float)
        f("create", C.c_int, [C.c_int32] * 6 + [C.POINTER(vp)])
        f("destroy", C.c_int, [vp])
        f("set_intrinsics", C.c_int, [vp, dp])
        f("set_points", C.c_int, [vp, fp, C.c_int32])
        f("add_observations", C.c_int, [vp, C.c_int64, ip, ip, ip, ip, dp, dp])
        f("num_observations", C.c_int64, [vp])
        f("eval", C.c_int, [vp, dp, dp, dp, dp])
        f("normal_eq", C.c_int, [vp, dp, C.c_double, C.c_int32, ip, dp, dp])
        f("solve", C.c_int, [vp, dp, C.POINTER(Options), C.POINTER(Summary)])
        for name, res, args in [
            ("last_error", C.c_char_p, [vp]),
            ("lm_begin", C.c_int, [vp, dp, C.POINTER(Options)]),
            ("lm_step", C.c_int, [vp, ip, ip]),
            ("lm_end", C.c_int, [vp, dp, C.POINTER(Summary)]),
            ("synchronize", C.c_int, [vp]),
            ("set_jacobian_mode", C.c_int, [vp, C.c_int32]),
            ("set_speculation", C.c_int, [vp, C.c_int32]),
            ("set_refine_intrinsics", C.c_int, [vp, C.c_uint32]),
            ("get_intrinsics", C.c_int, [vp, dp]),
            ("num_frs", C.c_int32, [vp]),
            ("fr_errors", C.c_int, [vp, dp, ip, ip, ip, ip, dp, dp]),
            ("fr_errors_noalloc", C.c_int, [vp, dp, ip, ip, ip, ip, dp, dp]),
            ("stage_times", C.c_int, [vp, C.POINTER(StageMs), C.c_int32]),
            ("partition_times", C.c_int, [C.c_int32, C.POINTER(C.c_int64), C.c_int32, ip]),
            ("exchange_count", C.c_int64, [C.c_int32]),
            ("exchange_ld", C.c_int32, [C.c_int32]),
            ("exchange_pack_host", C.c_int, [C.c_int32, dp, dp]),
            ("exchange_unpack_host", C.c_int, [C.c_int32, dp, dp]),
            ("dense_solve", C.c_int, [vp, C.c_int32, dp, dp, dp]),
            ("comm_unique_id", C.c_int, [vp]),
            ("comm_init", C.c_int, [vp, C.c_int32, C.c_int32, vp]),
            ("brief_report", C.c_int, [C.POINTER(Summary), C.c_char_p, C.c_size_t]),
            ("options_default", None, [C.POINTER(Options)]),
        ]:
            f(name, res, args, optional=True)
        # Stage 4 (algebraic error): <prefix with ae_ for rp_>create / add_relations / eval
        for name, res, args in [
            ("ae_create", C.c_int, [C.c_int32] * 6 + [C.POINTER(vp)]),
            ("ae_add_relations", C.c_int, [vp, C.c_int64, ip, ip, ip, dp]),
            ("ae_eval", C.c_int, [vp, dp, dp, dp, dp]),
        ]:
            f(name, res, args, optional=True)
        # point reconstruction: <prefix with rae_ for rp_>reconstruct
        up = C.POINTER(C.c_uint8)
        f("rae_reconstruct", C.c_int,
          [C.c_int32, C.c_int32, dp, ip, C.c_int32, ip, dp, C.c_int32, C.c_int32, C.c_int64, ip, ip, dp, dp,
           C.POINTER(Options), dp, up, ip, dp, C.POINTER(RaeSummary), C.c_char_p, C.c_size_t], optional=True)

        f("rae_create", C.c_int, [C.c_int32, C.c_int32, ip, C.c_int32, ip, C.c_int32, C.c_int32, C.c_int64, ip, ip, dp, dp,
                                  C.POINTER(vp), C.c_char_p, C.c_size_t], optional=True)
        f("rae_solve", C.c_int, [vp, dp, dp, C.POINTER(Options), dp, up, ip, dp, C.POINTER(RaeSummary)], optional=True)
        f("rae_last_error", C.c_char_p, [vp], optional=True)
        f("rae_destroy", C.c_int, [vp], optional=True)

    def _f(self, name, restype, argtypes, optional=False):
        symbol = self.prefix + name
        if name.startswith("ae_"):
            symbol = self.prefix.replace("rp_", "ae_") + name[3:]
        if name.startswith("rae_"):
            symbol = self.prefix.replace("rp_", "rae_") + name[4:]
        try:
            fn = getattr(self.lib, symbol)
        except AttributeError:
            if optional:
                return
            raise
        fn.restype = restype
        fn.argtypes = argtypes
        setattr(self, name, fn)


_api = None


def load():
    """Load libcalico_rp.so (built in-tree by __graft_entry__.build()).  There is no
    fallback: a missing or unloadable HIP library is an error."""
    global _api
    if _api is None:
        # tests/test_gpu_comm.py points its child processes at the test build of the same
        # sources (tests/_build/libcalico_rp_testhooks.so: collectives stand-in, injected failure)
        path = os.environ.get("CALICO_RP_LIBRARY", LIB_PATH)
        if path != LIB_PATH:
            _api = Api(C.CDLL(path, mode=C.RTLD_GLOBAL))
            return _api
        if not os.path.exists(LIB_PATH):
            raise RpError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _api = Api(C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL))
    return _api


class Problem:
    """Thin owner of one calico_rp_problem handle (or oracle handle)."""

    _create = "create"

    def __init__(self, cn, pn, tn, p_star_var, t_star_var, device=0, api=None):
        self.api = api or load()
        self.cn, self.pn, self.tn = cn, pn, tn
        self.vn = cn + pn + tn
        self.p_star_var, self.t_star_var = p_star_var, t_star_var
        self.refine_mask = 0
        self.h = C.c_void_p()
        rc = getattr(self.api, self._create)(cn, pn, tn, p_star_var, t_star_var, device, C.byref(self.h))
        if rc != 0:
            raise RpError(f"create failed: {rc}")

    def _check(self, rc, what):
        if rc != 0:
            msg = b""
            if hasattr(self.api, "last_error"):
                msg = self.api.last_error(self.h) or b""
            raise RpError(f"{what} failed ({rc}): {msg.decode(errors='replace')}")

    def close(self):
        if self.h:
            self.api.destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_intrinsics(self, k):
        k = np.ascontiguousarray(k, dtype=np.float64).reshape(-1)
        assert k.size == 12 * self.cn
        self._check(self.api.set_intrinsics(self.h, _p(k, C.c_double)), "set_intrinsics")

    def set_points(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        self._check(self.api.set_points(self.h, _p(xyz, C.c_float), xyz.shape[0]), "set_points")

    def add_observations(self, cam, pat_var, time_var, corner, u, v):
        cam = np.ascontiguousarray(cam, dtype=np.int32)
        pat_var = np.ascontiguousarray(pat_var, dtype=np.int32)
        time_var = np.ascontiguousarray(time_var, dtype=np.int32)
        corner = np.ascontiguousarray(corner, dtype=np.int32)
        u = np.ascontiguousarray(u, dtype=np.float64)
        v = np.ascontiguousarray(v, dtype=np.float64)
        n = cam.size
        assert pat_var.size == n and time_var.size == n and corner.size == n and u.size == n and v.size == n
        self._check(self.api.add_observations(
            self.h, n, _p(cam, C.c_int32), _p(pat_var, C.c_int32), _p(time_var, C.c_int32),
            _p(corner, C.c_int32), _p(u, C.c_double), _p(v, C.c_double)), "add_observations")

    @property
    def num_observations(self):
        return int(self.api.num_observations(self.h))

    def eval(self, x, want_r=True, want_J=True):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        assert x.size == 7 * self.vn
        n = self.num_observations
        cost = C.c_double()
        r = np.empty(2 * n) if want_r else None
        J = np.empty((n, 2, 33 if self.refine_mask else 21)) if want_J else None
        self._check(self.api.eval(self.h, _p(x, C.c_double), C.byref(cost),
                                  _p(r, C.c_double) if want_r else None,
                                  _p(J, C.c_double) if want_J else None), "eval")
        return cost.value, (r.reshape(n, 2) if want_r else None), J

    def normal_eq(self, x, radius=1e4, jacobi_scaling=True):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        n = C.c_int32()
        self._check(self.api.normal_eq(self.h, _p(x, C.c_double), radius, int(jacobi_scaling),
                                       C.byref(n), None, None), "normal_eq(query)")
        S = np.empty((n.value, n.value))
        rhs = np.empty(n.value)
        self._check(self.api.normal_eq(self.h, _p(x, C.c_double), radius, int(jacobi_scaling),
                                       C.byref(n), _p(S, C.c_double), _p(rhs, C.c_double)), "normal_eq")
        return S, rhs

    def solve(self, x, options=None):
        x = np.array(x, dtype=np.float64).reshape(-1).copy()
        assert x.size == 7 * self.vn
        options = options or default_options()
        s = Summary()
        self._check(self.api.solve(self.h, _p(x, C.c_double), C.byref(options), C.byref(s)), "solve")
        return x, s

    # --- stepping interface (bench.py)
    def lm_begin(self, x, options=None):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        self._opt = options or default_options()
        self._check(self.api.lm_begin(self.h, _p(x, C.c_double), C.byref(self._opt)), "lm_begin")

    def lm_step(self):
        acc, done = C.c_int32(), C.c_int32()
        self._check(self.api.lm_step(self.h, C.byref(acc), C.byref(done)), "lm_step")
        return bool(acc.value), bool(done.value)

    def lm_end(self):
        x = np.empty(7 * self.vn)
        s = Summary()
        self._check(self.api.lm_end(self.h, _p(x, C.c_double), C.byref(s)), "lm_end")
        return x, s

    def stage_times(self, reset=False):
        st = StageMs()
        self._check(self.api.stage_times(self.h, C.byref(st), int(reset)), "stage_times")
        return {name: getattr(st, name) for name, _t in StageMs._fields_}

    def synchronize(self):
        self._check(self.api.synchronize(self.h), "synchronize")

    def fr_errors(self, x, noalloc=False):
        """Per-FR mean squared reprojection error and rrmse (multicamera.cpp:717-731,
        1860-1891): returns (cam, pat_var, time_var, n_corners, mean_sq_err, rrmse).
        noalloc: through calico_rp_fr_errors_noalloc (the library's file-scope scratch array)."""
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        n = int(self.api.num_frs(self.h))
        cam, pat, tim, cnt = (np.empty(n, np.int32) for _ in range(4))
        err = np.empty(n)
        rr = C.c_double()
        fn = self.api.fr_errors_noalloc if noalloc else self.api.fr_errors
        self._check(fn(self.h, _p(x, C.c_double), _p(cam, C.c_int32), _p(pat, C.c_int32),
                       _p(tim, C.c_int32), _p(cnt, C.c_int32), _p(err, C.c_double), C.byref(rr)),
                    "fr_errors_noalloc" if noalloc else "fr_errors")
        return cam, pat, tim, cnt, err, rr.value

    def dense_solve(self, S, b):
        """x = S^-1 b through the reduced system's solver (include/calico_rp.h)."""
        S = np.ascontiguousarray(S, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1)
        n = b.size
        assert S.shape == (n, n)
        x = np.empty(n)
        self._check(self.api.dense_solve(self.h, n, _p(S, C.c_double), _p(b, C.c_double), _p(x, C.c_double)),
                    "dense_solve")
        return x

    def set_jacobian_mode(self, mode):
        """"fused" (default) or "materialised" (include/calico_rp.h)."""
        code = {"fused": JACOBIAN_FUSED, "materialised": JACOBIAN_MATERIALISED}.get(mode, mode)
        self._check(self.api.set_jacobian_mode(self.h, int(code)), "set_jacobian_mode")

    def set_refine_intrinsics(self, mask):
        """Extension (BASELINE.json configs[2]): bit k of mask frees parameter k of every
        camera's [fx cx fy cy k1 k2 p1 p2 k3 k4 k5 k6]; 0 = constants, as in the reference.
        eval() then returns 2 x 33 Jacobians, normal_eq() the 12-parameter blocks after the
        7-parameter ones, and solve() refines the handle's intrinsics (get_intrinsics())."""
        self._check(self.api.set_refine_intrinsics(self.h, int(mask)), "set_refine_intrinsics")
        self.refine_mask = int(mask)

    def get_intrinsics(self):
        k = np.empty((self.cn, 12))
        self._check(self.api.get_intrinsics(self.h, _p(k, C.c_double)), "get_intrinsics")
        return k

    def set_speculation(self, on):
        """Speculative evaluation of the candidate point (include/calico_rp.h), default on."""
        self._check(self.api.set_speculation(self.h, 1 if on else 0), "set_speculation")

    def comm_init(self, nranks, rank, uid_bytes):
        buf = C.create_string_buffer(bytes(uid_bytes), UNIQUE_ID_BYTES)
        self._check(self.api.comm_init(self.h, nranks, rank, C.cast(buf, C.c_void_p)), "comm_init")


class AeProblem(Problem):
    """Stage-4 algebraic-error handle (calico_ae_create): relations instead of corner
    observations; normal_eq / solve / lm_* / comm_init as on Problem."""

    _create = "ae_create"

    def add_relations(self, cam, pat_var, time_var, a7):
        cam = np.ascontiguousarray(cam, dtype=np.int32)
        pat_var = np.ascontiguousarray(pat_var, dtype=np.int32)
        time_var = np.ascontiguousarray(time_var, dtype=np.int32)
        a7 = np.ascontiguousarray(a7, dtype=np.float64).reshape(-1, 7)
        n = cam.size
        assert pat_var.size == n and time_var.size == n and a7.shape[0] == n
        self._check(self.api.ae_add_relations(
            self.h, n, _p(cam, C.c_int32), _p(pat_var, C.c_int32), _p(time_var, C.c_int32),
            _p(a7, C.c_double)), "ae_add_relations")

    def eval(self, x, want_r=True, want_J=True):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        assert x.size == 7 * self.vn
        n = self.num_observations
        cost = C.c_double()
        r = np.empty(12 * n) if want_r else None
        J = np.empty((n, 12, 21)) if want_J else None
        self._check(self.api.ae_eval(self.h, _p(x, C.c_double), C.byref(cost),
                                     _p(r, C.c_double) if want_r else None,
                                     _p(J, C.c_double) if want_J else None), "ae_eval")
        return cost.value, (r.reshape(n, 12) if want_r else None), J


def rae_reconstruct(cam_params, image_cols_rows, fr_cam, fr_a_hat, n_points, corners_per_pattern, obs_fr, obs_point,
                    u, v, options=None, device=0, api=None):
    """calico_rae_reconstruct (include/calico_rp.h): returns (xyz [n_points,3], has_values,
    iterations, final_cost, RaeSummary)."""
    api = api or load()
    cam_params = np.ascontiguousarray(cam_params, dtype=np.float64).reshape(-1, 12)
    image_cols_rows = np.ascontiguousarray(image_cols_rows, dtype=np.int32).reshape(-1, 2)
    fr_cam = np.ascontiguousarray(fr_cam, dtype=np.int32)
    fr_a_hat = np.ascontiguousarray(fr_a_hat, dtype=np.float64).reshape(-1, 12)
    obs_fr = np.ascontiguousarray(obs_fr, dtype=np.int32)
    obs_point = np.ascontiguousarray(obs_point, dtype=np.int32)
    u = np.ascontiguousarray(u, dtype=np.float64)
    v = np.ascontiguousarray(v, dtype=np.float64)
    assert image_cols_rows.shape[0] == cam_params.shape[0] and fr_a_hat.shape[0] == fr_cam.size
    assert obs_point.size == obs_fr.size == u.size == v.size
    xyz = np.zeros((n_points, 3))
    has = np.zeros(n_points, np.uint8)
    its = np.zeros(n_points, np.int32)
    fc = np.zeros(n_points)
    summ = RaeSummary()
    err = C.create_string_buffer(256)
    rc = api.rae_reconstruct(device, cam_params.shape[0], _p(cam_params, C.c_double), _p(image_cols_rows, C.c_int32),
                             fr_cam.size, _p(fr_cam, C.c_int32), _p(fr_a_hat, C.c_double), n_points,
                             corners_per_pattern, obs_fr.size, _p(obs_fr, C.c_int32), _p(obs_point, C.c_int32),
                             _p(u, C.c_double), _p(v, C.c_double), C.byref(options) if options is not None else None,
                             _p(xyz, C.c_double), _p(has, C.c_uint8), _p(its, C.c_int32), _p(fc, C.c_double),
                             C.byref(summ), err, 256)
    if rc != 0:
        raise RpError(f"rae_reconstruct failed ({rc}): {err.value.decode(errors='replace')}")
    return xyz, has.astype(bool), its, fc, summ


class RaeProblem:
    """calico_rae_create / _solve / _destroy (include/calico_rp.h): the views and their observations stay
    on the device; solve() takes the camera parameters and one A_hat per view."""

    def __init__(self, image_cols_rows, fr_cam, n_points, corners_per_pattern, obs_fr, obs_point, u, v, device=0, api=None):
        self.api = api or load()
        image_cols_rows = np.ascontiguousarray(image_cols_rows, dtype=np.int32).reshape(-1, 2)
        fr_cam = np.ascontiguousarray(fr_cam, dtype=np.int32)
        obs_fr = np.ascontiguousarray(obs_fr, dtype=np.int32)
        obs_point = np.ascontiguousarray(obs_point, dtype=np.int32)
        u = np.ascontiguousarray(u, dtype=np.float64)
        v = np.ascontiguousarray(v, dtype=np.float64)
        assert obs_point.size == obs_fr.size == u.size == v.size
        self.cn, self.n_frs, self.n_points = image_cols_rows.shape[0], fr_cam.size, n_points
        self.h = C.c_void_p()
        err = C.create_string_buffer(256)
        rc = self.api.rae_create(device, self.cn, _p(image_cols_rows, C.c_int32), fr_cam.size, _p(fr_cam, C.c_int32), n_points,
                                 corners_per_pattern, obs_fr.size, _p(obs_fr, C.c_int32), _p(obs_point, C.c_int32),
                                 _p(u, C.c_double), _p(v, C.c_double), C.byref(self.h), err, 256)
        if rc != 0:
            raise RpError(f"rae_create failed ({rc}): {err.value.decode(errors='replace')}")

    def solve(self, cam_params, fr_a_hat, options=None):
        cam_params = np.ascontiguousarray(cam_params, dtype=np.float64).reshape(-1, 12)
        fr_a_hat = np.ascontiguousarray(fr_a_hat, dtype=np.float64).reshape(-1, 12)
        assert cam_params.shape[0] == self.cn and fr_a_hat.shape[0] == self.n_frs
        xyz = np.zeros((self.n_points, 3))
        has = np.zeros(self.n_points, np.uint8)
        its = np.zeros(self.n_points, np.int32)
        fc = np.zeros(self.n_points)
        summ = RaeSummary()
        rc = self.api.rae_solve(self.h, _p(cam_params, C.c_double), _p(fr_a_hat, C.c_double),
                                C.byref(options) if options is not None else None, _p(xyz, C.c_double),
                                _p(has, C.c_uint8), _p(its, C.c_int32), _p(fc, C.c_double), C.byref(summ))
        if rc != 0:
            raise RpError(f"rae_solve failed ({rc}): {(self.api.rae_last_error(self.h) or b'').decode(errors='replace')}")
        return xyz, has.astype(bool), its, fc, summ

    def close(self):
        if self.h:
            self.api.rae_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def brief_report(summary, api=None):
    api = api or load()
    buf = C.create_string_buffer(256)
    api.brief_report(C.byref(summary), buf, 256)
    return buf.value.decode()
