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


# ---- Stage 4 (algebraic error) relations, numpy: small configurations only ----------------
def quat_to_matrix(b7):
    """4x4 of a 7-vector [qw qx qy qz tx ty tz]: R(q)/|q|^2 and t
    (Convert7ParameterQuaternionRepresentationIntoMatrix, solving-structure.hpp:32-58)."""
    w, x, y, z = b7[:4]
    n2 = w * w + x * x + y * y + z * z
    R = np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]]) / n2
    M = np.eye(4)
    M[:3, :3] = R
    M[:3, 3] = b7[4:7]
    return M


def matrix_to_quat(M):
    """Unit quaternion (w >= 0) + translation of a rigid 4x4."""
    R = M[:3, :3]
    K = np.array([[R[0, 0] + R[1, 1] + R[2, 2], R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]],
                  [R[2, 1] - R[1, 2], R[0, 0] - R[1, 1] - R[2, 2], R[0, 1] + R[1, 0], R[0, 2] + R[2, 0]],
                  [R[0, 2] - R[2, 0], R[0, 1] + R[1, 0], R[1, 1] - R[0, 0] - R[2, 2], R[1, 2] + R[2, 1]],
                  [R[1, 0] - R[0, 1], R[0, 2] + R[2, 0], R[1, 2] + R[2, 1], R[2, 2] - R[0, 0] - R[1, 1]]]) / 3.0
    w, v = np.linalg.eigh(K)
    q = v[:, -1]
    if q[0] < 0:
        q = -q
    return np.concatenate([q, M[:3, 3]])


def quats_to_matrices(b):
    """Batched quat_to_matrix: [n,7] -> [n,4,4]."""
    b = np.asarray(b, dtype=np.float64)
    w, x, y, z = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    n2 = w * w + x * x + y * y + z * z
    M = np.zeros((b.shape[0], 4, 4))
    M[:, 0, 0], M[:, 0, 1], M[:, 0, 2] = w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)
    M[:, 1, 0], M[:, 1, 1], M[:, 1, 2] = 2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)
    M[:, 2, 0], M[:, 2, 1], M[:, 2, 2] = 2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z
    M[:, :3, :3] /= n2[:, None, None]
    M[:, :3, 3] = b[:, 4:7]
    M[:, 3, 3] = 1.0
    return M


def matrices_to_quats(M):
    """Batched rigid 4x4 -> unit quaternion (w >= 0) + translation; largest-component branch
    of the usual trace method, chosen per matrix."""
    R = M[:, :3, :3]
    r00, r11, r22 = R[:, 0, 0], R[:, 1, 1], R[:, 2, 2]
    cand = np.stack([1 + r00 + r11 + r22, 1 + r00 - r11 - r22, 1 - r00 + r11 - r22, 1 - r00 - r11 + r22], 1)
    k = cand.argmax(1)
    t = 2.0 * np.sqrt(cand[np.arange(len(k)), k])  # 4 * |largest component|
    q = np.empty((len(k), 4))
    a, b_, c = R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]
    s01, s02, s12 = R[:, 0, 1] + R[:, 1, 0], R[:, 0, 2] + R[:, 2, 0], R[:, 1, 2] + R[:, 2, 1]
    rows = [np.stack([t / 4, a / t, b_ / t, c / t], 1), np.stack([a / t, t / 4, s01 / t, s02 / t], 1),
            np.stack([b_ / t, s01 / t, t / 4, s12 / t], 1), np.stack([c / t, s02 / t, s12 / t, t / 4], 1)]
    for j in range(4):
        q[k == j] = rows[j][k == j]
    q[q[:, 0] < 0] *= -1.0
    return np.concatenate([q, M[:, :3, 3]], axis=1)


def rigid_inverses(M):
    Rt = np.transpose(M[:, :3, :3], (0, 2, 1))
    out = np.zeros_like(M)
    out[:, :3, :3] = Rt
    out[:, :3, 3] = -np.einsum("nij,nj->ni", Rt, M[:, :3, 3])
    out[:, 3, 3] = 1.0
    return out


def _random_rigids(rng, n, max_deg, max_mm):
    axis = rng.normal(size=(n, 3))
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    ang = np.deg2rad(max_deg) * rng.uniform(-1, 1, size=n)
    return np.concatenate([np.cos(ang / 2)[:, None], np.sin(ang / 2)[:, None] * axis,
                           rng.uniform(-max_mm, max_mm, size=(n, 3))], axis=1)


def frs(cfg, chunk=250):
    """The (camera, pattern var, time var) triples the corner generator observes, in its
    order, found chunk by chunk from the runs of equal triples."""
    out = []
    for t0 in range(0, cfg.tn, chunk):
        cam, pat, tim, _c, _u, _v = observations(cfg, t0, min(cfg.tn, t0 + chunk))
        if cam.size == 0:
            continue
        key = np.stack([cam, pat, tim], 1)
        start = np.concatenate([[0], np.flatnonzero(np.any(key[1:] != key[:-1], axis=1)) + 1])
        out.append(key[start])
    return np.concatenate(out)  # an FR has one time, so no FR straddles two chunks


def ae_relations(cfg, noise_deg=0.2, noise_mm=1.0, perturb_deg=3.0, perturb_mm=20.0, seed=7, shuffle=True):
    """Relations C = A P T for the FRs of `cfg` (one per (camera, pattern, time) that the
    corner generator observes): returns (cam, pat_var, time_var, a7, x_true, x0) with
    x in the Stage-4 convention (nothing inverted; p*, t* identity)."""
    rng = np.random.default_rng(seed)
    k = frs(cfg)
    if shuffle:  # the library sorts; the reference adds equations in detection order
        rng.shuffle(k)
    cam, pat, tim = k[:, 0], k[:, 1], k[:, 2]
    ident = np.array([1.0, 0, 0, 0, 0, 0, 0])
    xt = _random_rigids(rng, cfg.vn, 60.0, 800.0)
    xt[:, :4] *= rng.uniform(0.8, 1.25, size=(cfg.vn, 1))  # the reference never normalises q
    xt[cfg.p_star_var] = ident
    xt[cfg.t_star_var] = ident
    Ms = quats_to_matrices(xt)
    A = Ms[cam] @ rigid_inverses(Ms[pat] @ Ms[tim])
    A = A @ quats_to_matrices(_random_rigids(rng, len(k), noise_deg, noise_mm))
    a7 = matrices_to_quats(A)
    x0 = matrices_to_quats(quats_to_matrices(_random_rigids(rng, cfg.vn, perturb_deg, perturb_mm)) @ Ms)
    x0[cfg.p_star_var] = ident
    x0[cfg.t_star_var] = ident
    return (cam.astype(np.int32), pat.astype(np.int32), tim.astype(np.int32), a7, xt.reshape(-1), x0.reshape(-1))


# ---- point reconstruction inputs (ReconstructXFunctionIDsMCwithDLTs) ----------------------
IMAGE_COLS_ROWS = (1200, 900)  # the generator's image size


def rae_inputs(cfg, x=None, t0=0, t1=None):
    """Arguments of calico_rae_reconstruct for the views of `cfg` at state x (Stage-5
    convention, default: the ground truth): (cam_params, image_cols_rows, fr_cam, fr_a_hat,
    n_points, corners_per_pattern, obs_fr, obs_point, u, v).  A_hat = C T~ P~ with the
    stored (already inverted) pattern / time blocks, i.e. the reference's C T^-1 P^-1."""
    x = ground_truth(cfg) if x is None else x
    cam, pat, tim, cor, u, v = observations(cfg, t0, t1)
    key = np.stack([cam, pat, tim], 1)
    start = np.concatenate([[0], np.flatnonzero(np.any(key[1:] != key[:-1], axis=1)) + 1])
    obs_fr = np.zeros(cam.size, np.int32)
    obs_fr[start[1:]] = 1
    obs_fr = np.cumsum(obs_fr, dtype=np.int32)
    M = quats_to_matrices(x.reshape(-1, 7))
    fc, fp, ft = cam[start], pat[start], tim[start]
    A = M[fc] @ M[ft] @ M[fp]
    ncp = lib().calico_synth_corners_per_pattern(C.byref(cfg))
    cr = np.tile(np.array(IMAGE_COLS_ROWS, np.int32), (cfg.cn, 1))
    return (intrinsics(cfg), cr, fc.astype(np.int32), A[:, :3, :].reshape(-1, 12).copy(), cfg.pn * ncp, ncp, obs_fr,
            cor, u, v)
