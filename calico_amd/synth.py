"""ctypes view of the synthetic camera-network generator (csrc/synth.cpp)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "_build", "libcalico_synth.so")


class Config(C.Structure):
    _fields_ = [
        ("cn", C.c_int32), ("pn", C.c_int32), ("tn", C.c_int32),
        ("board_type", C.c_int32), ("sx", C.c_int32), ("sy", C.c_int32),
        ("square_mm", C.c_double),
        ("one_board_per_cam_time", C.c_int32),
        ("fr_drop_prob", C.c_double), ("corner_drop_prob", C.c_double),
        ("k_select", C.c_int32),
        ("noise_px", C.c_double),
        ("perturb_rot_deg", C.c_double), ("perturb_trans_mm", C.c_double),
        ("tangential", C.c_int32),
        ("p_star_idx", C.c_int32), ("t_star_idx", C.c_int32),
        ("seed", C.c_uint64),
    ]

    @property
    def vn(self):
        return self.cn + self.pn + self.tn

    @property
    def p_star_var(self):
        return self.cn + self.p_star_idx

    @property
    def t_star_var(self):
        return self.cn + self.pn + self.t_star_idx


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run __graft_entry__.build()")
        L = C.CDLL(LIB_PATH)
        cp = C.POINTER(Config)
        ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
        L.calico_synth_preset.argtypes = [C.c_int32, cp]
        L.calico_synth_corners_per_pattern.argtypes = [cp]
        L.calico_synth_points.argtypes = [cp, C.POINTER(C.c_float)]
        L.calico_synth_intrinsics.argtypes = [cp, dp]
        L.calico_synth_ground_truth.argtypes = [cp, dp]
        L.calico_synth_initial_guess.argtypes = [cp, dp]
        L.calico_synth_observations.argtypes = [cp, C.c_int32, C.c_int32, ip, ip, ip, ip, dp, dp, C.c_int64]
        L.calico_synth_observations.restype = C.c_int64
        L.calico_synth_obs_per_time.argtypes = [cp, C.POINTER(C.c_int64)]
        L.calico_synth_obs_per_time.restype = None
        L.calico_synth_points.restype = None
        L.calico_synth_intrinsics.restype = None
        L.calico_synth_ground_truth.restype = None
        L.calico_synth_initial_guess.restype = None
        _lib = L
    return _lib


PRESETS = {"A": 1, "B": 2, "C": 3, "D": 4}


def preset(name, **overrides):
    """Configs A–D of SURVEY.md §8(d) (BASELINE.json configs[1..4])."""
    cfg = Config()
    if lib().calico_synth_preset(PRESETS[name], C.byref(cfg)) != 0:
        raise ValueError(name)
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg


def custom(cn, pn, tn, sx=7, sy=9, board_type=0, **kw):
    cfg = preset("A")
    cfg.cn, cfg.pn, cfg.tn, cfg.sx, cfg.sy, cfg.board_type = cn, pn, tn, sx, sy, board_type
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def points(cfg):
    n = cfg.pn * lib().calico_synth_corners_per_pattern(C.byref(cfg))
    xyz = np.empty((n, 3), dtype=np.float32)
    lib().calico_synth_points(C.byref(cfg), xyz.ctypes.data_as(C.POINTER(C.c_float)))
    return xyz


def intrinsics(cfg):
    k = np.empty((cfg.cn, 12))
    lib().calico_synth_intrinsics(C.byref(cfg), _dp(k))
    return k


def ground_truth(cfg):
    x = np.empty(7 * cfg.vn)
    lib().calico_synth_ground_truth(C.byref(cfg), _dp(x))
    return x


def initial_guess(cfg):
    x = np.empty(7 * cfg.vn)
    lib().calico_synth_initial_guess(C.byref(cfg), _dp(x))
    return x


def count(cfg, t0=0, t1=None):
    t1 = cfg.tn if t1 is None else t1
    return int(lib().calico_synth_observations(C.byref(cfg), t0, t1, None, None, None, None, None, None, 0))


def observations(cfg, t0=0, t1=None):
    """(cam, pat_var, time_var, corner, u, v) for time superscripts [t0, t1)."""
    t1 = cfg.tn if t1 is None else t1
    n = count(cfg, t0, t1)
    cam = np.empty(n, np.int32)
    pat = np.empty(n, np.int32)
    tim = np.empty(n, np.int32)
    cor = np.empty(n, np.int32)
    u = np.empty(n)
    v = np.empty(n)
    got = lib().calico_synth_observations(C.byref(cfg), t0, t1, _ip(cam), _ip(pat), _ip(tim), _ip(cor),
                                          _dp(u), _dp(v), n)
    assert got == n
    return cam, pat, tim, cor, u, v


def obs_per_time(cfg):
    """Observation count per time superscript (for calico_rp_partition_times)."""
    out = np.empty(cfg.tn, np.int64)
    lib().calico_synth_obs_per_time(C.byref(cfg), out.ctypes.data_as(C.POINTER(C.c_int64)))
    return out


def fill_problem(prob, cfg, t0=0, t1=None):
    """Load intrinsics, model points and the observations of [t0,t1) into a Problem."""
    prob.set_intrinsics(intrinsics(cfg))
    prob.set_points(points(cfg))
    obs = observations(cfg, t0, t1)
    prob.add_observations(*obs)
    return obs
