"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on
the same seeded synthetic inputs, and against the committed mpmath golden vectors.
Bars (BASELINE.json north_star): r and J at fixed parameters 1e-12 relative;
converged RMS within 1e-6 px, rotations 1e-6 rad, translations 1e-6 units."""
import json
import os

import numpy as np
import pytest

from calico_amd import capi, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


MODES = ["fused", "materialised"]  # include/calico_rp.h: calico_rp_set_jacobian_mode


def _pair(cfg, oracle_api, t0=0, t1=None, mode="fused"):
    g = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
    g.set_jacobian_mode(mode)
    o = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, api=oracle_api)
    obs = synth.fill_problem(g, cfg, t0, t1)
    o.set_intrinsics(synth.intrinsics(cfg))
    o.set_points(synth.points(cfg))
    o.add_observations(*obs)
    return g, o, obs


def _relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_golden_vectors_on_gpu():
    G = json.load(open(os.path.join(ROOT, "tests", "golden", "rp_golden.json")))["vectors"]
    for g in G:
        case = g["case"]
        pv = 1 if case in (1, 3) else 2
        tv = 3 if case in (2, 3) else 4
        P = capi.Problem(1, 2, 2, 1, 3, device=0)
        P.set_intrinsics(g["intrinsics"])
        P.set_points([g["point"]])
        P.add_observations([0], [pv], [tv], [0], [g["uv"][0]], [g["uv"][1]])
        x = np.zeros(35)
        x[0::7] = 1
        x[0:7] = g["camera"]
        if case in (0, 2):
            x[7 * pv:7 * pv + 7] = g["pattern"]
        if case in (0, 1):
            x[7 * tv:7 * tv + 7] = g["time"]
        cost, r, J = P.eval(x)
        rg = np.array([float(v) for v in g["r"]])
        Jg = np.array([[float(v) for v in row] for row in g["J"]])
        assert _relerr(r[0], rg) < 1e-12
        assert _relerr(J[0], Jg) < 1e-12
        assert abs(cost - 0.5 * (rg ** 2).sum()) <= 1e-12 * 0.5 * (rg ** 2).sum()
        P.close()


@pytest.mark.parametrize("shape", [(3, 2, 4), (5, 3, 17)])
def test_residual_jacobian_vs_oracle(oracle_api, shape):
    cfg = synth.custom(*shape)
    g, o, _ = _pair(cfg, oracle_api)
    x = synth.initial_guess(cfg)
    # non-unit quaternions, as LM produces them (no manifold in the reference)
    rng = np.random.default_rng(1)
    for v in range(cfg.vn):
        if v not in (cfg.p_star_var, cfg.t_star_var):
            x[7 * v:7 * v + 4] *= rng.uniform(0.6, 1.7)
    cg, rg, Jg = g.eval(x)
    co, ro, Jo = o.eval(x)
    assert _relerr(rg, ro) < 1e-12
    # per-row relative to the row's largest entry
    scale = np.abs(Jo).max(axis=2, keepdims=True)
    assert (np.abs(Jg - Jo) / scale).max() < 1e-12
    assert abs(cg - co) < 1e-12 * co


@pytest.mark.parametrize("mode", MODES)
def test_unsorted_and_batched_observations(oracle_api, mode):
    """Observations may arrive in any order and in several calls (two batches in the
    reference, calico.cpp:718-729); results come back in the caller's order."""
    cfg = synth.custom(4, 2, 9)
    obs = synth.observations(cfg)
    n = obs[0].size
    rng = np.random.default_rng(2)
    perm = rng.permutation(n)
    g = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
    g.set_jacobian_mode(mode)
    g.set_intrinsics(synth.intrinsics(cfg))
    g.set_points(synth.points(cfg))
    half = n // 2
    g.add_observations(*[a[perm[:half]] for a in obs])
    g.add_observations(*[a[perm[half:]] for a in obs])
    o = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, api=oracle_api)
    o.set_intrinsics(synth.intrinsics(cfg))
    o.set_points(synth.points(cfg))
    o.add_observations(*[a[perm] for a in obs])
    x = synth.initial_guess(cfg)
    cg, rg, Jg = g.eval(x)
    co, ro, Jo = o.eval(x)
    assert _relerr(rg, ro) < 1e-12
    assert _relerr(Jg, Jo) < 1e-11
    Sg, bg = g.normal_eq(x)
    So, bo = o.normal_eq(x)
    assert _relerr(Sg, So) < 1e-10
    assert _relerr(bg, bo) < 1e-10


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("jacobi", [True, False])
def test_reduced_normal_equations_vs_oracle(oracle_api, jacobi, mode):
    cfg = synth.custom(6, 3, 25, fr_drop_prob=0.3)
    g, o, _ = _pair(cfg, oracle_api, mode=mode)
    x = synth.initial_guess(cfg)
    Sg, bg = g.normal_eq(x, radius=1e4, jacobi_scaling=jacobi)
    So, bo = o.normal_eq(x, radius=1e4, jacobi_scaling=jacobi)
    assert Sg.shape == So.shape
    assert _relerr(Sg, So) < 1e-10
    assert _relerr(bg, bo) < 1e-10
    assert np.abs(Sg - Sg.T).max() == 0.0


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("drop,tn", [(0.7, 60), (0.95, 300)])
def test_normal_equations_ragged_frs(oracle_api, mode, drop, tn):
    """FRs of 1..48 corners (corner_drop_prob 0.7) so that wave slabs and 16-observation
    groups hold several parts, and enough observations for several wave ranges; with 0.95
    FRs of one to a handful of corners: a slab then spans more FRs than the fused kernel's
    table ring holds, a 16-observation group several parts."""
    cfg = synth.custom(5, 3, tn, fr_drop_prob=0.2, corner_drop_prob=drop)
    g, o, obs = _pair(cfg, oracle_api, mode=mode)
    assert obs[0].size > 2 * 4096
    x = synth.initial_guess(cfg)
    Sg, bg = g.normal_eq(x)
    So, bo = o.normal_eq(x)
    assert _relerr(Sg, So) < 1e-10
    assert _relerr(bg, bo) < 1e-10


def _pose_deltas(xa, xb, vn):
    rot, tr = 0.0, 0.0
    for v in range(vn):
        qa, qb = xa[7 * v:7 * v + 4], xb[7 * v:7 * v + 4]
        qa, qb = qa / np.linalg.norm(qa), qb / np.linalg.norm(qb)
        d = min(1.0, abs(float(qa @ qb)))
        rot = max(rot, 2.0 * np.arccos(d))
        tr = max(tr, np.abs(xa[7 * v + 4:7 * v + 7] - xb[7 * v + 4:7 * v + 7]).max())
    return rot, tr


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name,shape", [("tiny", (3, 2, 4)), ("small", (6, 3, 40)), ("medium", (20, 4, 30))])
def test_converged_solution_vs_oracle(oracle_api, name, shape, mode):
    cfg = synth.custom(*shape)
    g, o, _ = _pair(cfg, oracle_api, mode=mode)
    x0 = synth.initial_guess(cfg)
    opt = capi.default_options(max_num_iterations=100, function_tolerance=1e-15, parameter_tolerance=1e-13,
                               gradient_tolerance=1e-14)
    xg, sg = g.solve(x0, opt)
    xo, so = o.solve(x0, opt)
    n = g.num_observations
    rms_g = np.sqrt(2 * sg.final_cost / (2 * n))
    rms_o = np.sqrt(2 * so.final_cost / (2 * n))
    assert abs(rms_g - rms_o) < 1e-6
    rot, tr = _pose_deltas(xg, xo, cfg.vn)
    assert rot < 1e-6 and tr < 1e-6
    # exemplars never move (MCcali::TestSolutionHasIdentityPstarTstar, multicamera.cpp:1493-1514)
    for v in (cfg.p_star_var, cfg.t_star_var):
        assert np.array_equal(xg[7 * v:7 * v + 7], x0[7 * v:7 * v + 7])


def test_reference_schedule_config_A(oracle_api):
    """BASELINE.json configs[1] with the reference's own schedule: two batches of FRs,
    ten LM iterations each, Ceres defaults (calico.cpp:707-729)."""
    cfg = synth.preset("A")
    obs = synth.observations(cfg)
    tim = obs[2]
    # FR order = generator order; split at half of the FRs (perc_rp = 0.5)
    key = np.stack([obs[0], obs[1], tim], 1)
    change = np.flatnonzero(np.any(key[1:] != key[:-1], axis=1)) + 1
    starts = np.concatenate([[0], change])
    cut = starts[int(round(0.5 * starts.size))]
    results = []
    for api, dev in ((None, 0), (oracle_api, 0)):
        P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=dev, api=api)
        P.set_intrinsics(synth.intrinsics(cfg))
        P.set_points(synth.points(cfg))
        x = synth.initial_guess(cfg)
        P.add_observations(*[a[:cut] for a in obs])
        x, s1 = P.solve(x)
        P.add_observations(*[a[cut:] for a in obs])
        x, s2 = P.solve(x)
        results.append((x, s1, s2, P.num_observations))
    (xg, g1, g2, n), (xo, o1, o2, _) = results
    assert abs(g1.initial_cost - o1.initial_cost) < 1e-9 * o1.initial_cost
    rms_g = np.sqrt(g2.final_cost / n)
    rms_o = np.sqrt(o2.final_cost / n)
    assert abs(rms_g - rms_o) < 1e-6
    assert 0.05 < rms_g < 0.2  # pixel noise is 0.1
    rot, tr = _pose_deltas(xg, xo, cfg.vn)
    assert rot < 1e-6 and tr < 1e-4  # 10+10 iterations is not converged to 1e-6 mm; see DESIGN.md
    assert g2.num_iterations == o2.num_iterations


def test_fr_errors_report_vs_oracle(oracle_api):
    """per-FR reprojection error and rrmse as the reference reports them
    (multicamera.cpp:717-731), against the oracle's residuals grouped on the host."""
    cfg = synth.custom(5, 3, 12, fr_drop_prob=0.2, corner_drop_prob=0.3)
    g, o, obs = _pair(cfg, oracle_api)
    x = synth.initial_guess(cfg)
    cam, pat, tim, cnt, err, rrmse = g.fr_errors(x)
    _, ro, _ = o.eval(x)
    sq = (ro ** 2).sum(axis=1)
    ref = {}
    for c, p, t, e in zip(obs[0], obs[1], obs[2], sq):
        ref.setdefault((int(c), int(p), int(t)), []).append(e)
    assert len(ref) == cam.size
    for c, p, t, n, e in zip(cam, pat, tim, cnt, err):
        v = ref[(int(c), int(p), int(t))]
        assert n == len(v)
        assert abs(e - np.mean(v)) <= 1e-12 * np.mean(v)
    want = np.sqrt(np.mean([np.mean(v) for v in ref.values()]))
    assert abs(rrmse - want) <= 1e-12 * want


def test_full_size_config_D_properties():
    """BASELINE.json's full size (102.4 M observations) is out of the oracle's reach, so the
    two independent device paths check each other: the fused kernel (reduced-column Gram in
    LDS) against the materialised Jacobian + accumulation kernels on the reduced normal
    equations, the cost sweep against the r^T r entries of the accumulation records, and
    one LM step from the perturbed start in both modes."""
    cfg = synth.preset("D")
    P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
    obs = synth.fill_problem(P, cfg)
    n = obs[0].size
    assert n == 102_400_000
    del obs
    x0 = synth.initial_guess(cfg)
    cost_sweep = P.eval(x0, want_r=False, want_J=False)[0]
    out = {}
    for mode in MODES:
        P.set_jacobian_mode(mode)
        S, b = P.normal_eq(x0)
        opt = capi.default_options(max_num_iterations=1)
        x1, s = P.solve(x0, opt)
        out[mode] = (S, b, x1, s.initial_cost, s.final_cost)
    Sf, bf, xf, cf0, cf1 = out["fused"]
    Sm, bm, xm, cm0, cm1 = out["materialised"]
    assert _relerr(Sf, Sm) < 1e-10 and _relerr(bf, bm) < 1e-10
    assert np.abs(Sf - Sf.T).max() == 0.0
    for c0 in (cf0, cm0):
        assert abs(c0 - cost_sweep) <= 1e-12 * cost_sweep
    assert cf1 < 0.2 * cf0  # the first step from a 1 degree / 5 mm perturbation removes most of the cost
    assert abs(cf1 - cm1) <= 1e-8 * cm1  # one LM step amplifies the 1e-10 of the systems; 1.5e-9 seen with other wave ranges
    assert np.abs(xf - xm).max() < 1e-9
    P.close()


@pytest.mark.parametrize("mode", MODES)
def test_rejected_steps_follow_the_oracle(oracle_api, mode):
    """A start far from the solution (60 degrees, 800 mm): the run contains rejected steps,
    so the accept/reject sequence, the radius update (decrease factor) and the retry on the
    kept Jacobian must follow the oracle's Ceres restatement step for step.  (With an
    initial radius of 1e12 the damped system is singular to rounding along the quaternion
    scale directions and whether LLT survives is a matter of summation order: not tested.)"""
    cfg = synth.custom(5, 3, 20, perturb_rot_deg=60.0, perturb_trans_mm=800.0)
    g, o, _ = _pair(cfg, oracle_api, mode=mode)
    x0 = synth.initial_guess(cfg)
    opt = capi.default_options(max_num_iterations=30)
    xg, sg = g.solve(x0, opt)
    xo, so = o.solve(x0, opt)
    assert so.num_unsuccessful_steps > 1, "the case is meant to exercise rejections (the terminating step counts as one)"
    assert (sg.num_iterations, sg.num_successful_steps, sg.num_unsuccessful_steps, sg.termination_type) == (
        so.num_iterations, so.num_successful_steps, so.num_unsuccessful_steps, so.termination_type)
    assert abs(sg.final_cost - so.final_cost) <= 1e-8 * so.final_cost


@pytest.mark.parametrize("far", [False, True])
def test_speculative_evaluation_changes_nothing_but_the_sweeps(far):
    """include/calico_rp.h: calico_rp_set_speculation.  The fused kernel run at the candidate
    replaces the cost sweep and, when the step is accepted, the next iteration's evaluation:
    the same accept/reject sequence and the same iterates (the candidate's cost comes from the
    Gram records instead of the cost sweep: rounding only), with fewer sweeps.  `far` starts
    60 degrees away so that the run contains rejected steps (cost sweep until the next accept)."""
    cfg = synth.custom(5, 3, 20, perturb_rot_deg=60.0, perturb_trans_mm=800.0) if far else synth.custom(6, 3, 40)
    x0 = synth.initial_guess(cfg)
    runs = {}
    for spec in (True, False):
        P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
        synth.fill_problem(P, cfg)
        P.set_speculation(spec)
        opt = capi.default_options(max_num_iterations=30 if far else 8, profile=1)
        P.lm_begin(x0, opt)
        with pytest.raises(capi.RpError):  # not between lm_begin and lm_end
            P.set_speculation(not spec)
        trace = []
        while True:
            acc, done = P.lm_step()
            if done:
                break
            trace.append(int(acc))
        x, s = P.lm_end()
        runs[spec] = (trace, x, s)
        P.close()
    (t1, x1, s1), (t0, x0_, s0) = runs[True], runs[False]
    assert t1 == t0
    if far:
        assert 0 in t1, "the case is meant to exercise rejections"
    assert (s1.num_iterations, s1.termination_type) == (s0.num_iterations, s0.termination_type)
    assert abs(s1.final_cost - s0.final_cost) <= 1e-10 * s0.final_cost
    assert np.abs(x1 - x0_).max() <= 1e-9 * np.abs(x0_).max()
    # without speculation: one cost sweep per step and one evaluation per accepted point
    assert s0.stage.n_cost_sweep >= len(t0)
    assert s1.stage.n_cost_sweep < s0.stage.n_cost_sweep
    assert s1.stage.n_jacobian_sweep + s1.stage.n_cost_sweep <= s0.stage.n_jacobian_sweep + s0.stage.n_cost_sweep


def _subset_pair(cfg, oracle_api, keep, mode):
    obs = [a[keep] for a in synth.observations(cfg)]
    g = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
    g.set_jacobian_mode(mode)
    o = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, api=oracle_api)
    for P in (g, o):
        P.set_intrinsics(synth.intrinsics(cfg))
        P.set_points(synth.points(cfg))
        P.add_observations(*obs)
    return g, o, obs


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case", ["only_t_star", "only_p_star_and_t_star", "unobserved_blocks", "one_observation_per_fr"])
def test_degenerate_networks(oracle_api, mode, case):
    """Edge cases of the block structure (solving-structure.cpp:528-544 cases 1-3 on their own):
    no eliminated time segment at all (every observation at t*), camera blocks only (p* and t*),
    cameras / patterns / times that no observation mentions (their blocks must keep their values),
    and foundational relationships of a single corner."""
    cfg = synth.custom(5, 3, 12)
    obs = synth.observations(cfg)
    cam, pat, tim = obs[0], obs[1], obs[2]
    if case == "only_t_star":
        keep = tim == cfg.t_star_var
    elif case == "only_p_star_and_t_star":
        keep = (tim == cfg.t_star_var) & (pat == cfg.p_star_var)
    elif case == "unobserved_blocks":
        tvars = np.unique(tim)
        dropped_t = [t for t in tvars if t != cfg.t_star_var][:2]
        dropped_p = [p for p in np.unique(pat) if p != cfg.p_star_var][:1]
        keep = (cam != 0) & ~np.isin(tim, dropped_t) & ~np.isin(pat, dropped_p)
    else:
        key = np.stack([cam, pat, tim], 1)
        first = np.concatenate([[True], np.any(key[1:] != key[:-1], axis=1)])
        keep = first
    assert keep.sum() > 0
    g, o, sub = _subset_pair(cfg, oracle_api, keep, mode)
    x0 = synth.initial_guess(cfg)
    if case != "one_observation_per_fr":  # (one corner per FR leaves the poses undetermined: normal equations only)
        opt = capi.default_options(max_num_iterations=6)
        xg, sg = g.solve(x0, opt)
        xo, so = o.solve(x0, opt)
        assert (sg.num_iterations, sg.termination_type) == (so.num_iterations, so.termination_type)
        assert abs(sg.final_cost - so.final_cost) <= 1e-9 * max(so.final_cost, 1e-30)
        assert np.abs(xg - xo).max() <= 1e-8 * np.abs(xo).max()
        mentioned = np.zeros(cfg.vn, bool)
        for a in sub[:3]:
            mentioned[np.unique(a)] = True
        for v in np.flatnonzero(~mentioned):
            assert np.array_equal(xg[7 * v:7 * v + 7], x0[7 * v:7 * v + 7])
    Sg, bg = g.normal_eq(x0)
    So, bo = o.normal_eq(x0)
    assert Sg.shape == So.shape
    assert _relerr(Sg, So) < 1e-10 and _relerr(bg, bo) < 1e-10


def test_million_observations_in_random_order():
    """More than 2^20 observations, every one a run of its own: the multi-threaded host paths of
    add_observations / build_structure (chunked checks and copies, run detection, the parallel
    merge sort of the runs, the permuted gathers) against the same network added FR by FR."""
    cfg = synth.custom(16, 4, 1400)
    obs = synth.observations(cfg)
    n = obs[0].size
    assert n > (1 << 20)
    perm = np.random.default_rng(5).permutation(n)
    x = synth.initial_guess(cfg)
    out = []
    for order in (None, perm):
        P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
        P.set_intrinsics(synth.intrinsics(cfg))
        P.set_points(synth.points(cfg))
        P.add_observations(*([a[order] for a in obs] if order is not None else obs))
        S, b = P.normal_eq(x)
        cost = P.eval(x, want_r=False, want_J=False)[0]
        out.append((S, b, cost))
        P.close()
    (S0, b0, c0), (S1, b1, c1) = out
    assert _relerr(S1, S0) < 1e-11 and _relerr(b1, b0) < 1e-11
    assert abs(c1 - c0) <= 1e-12 * c0
    with pytest.raises(capi.RpError):  # the chunked index check still names an offender
        P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
        P.set_points(synth.points(cfg))
        bad = obs[3].copy()
        bad[n // 2 + 17] = 10 ** 6
        P.add_observations(obs[0], obs[1], obs[2], bad, obs[4], obs[5])


def test_error_codes_and_states():
    """Error convention of INTEGRATION.md: negative codes + last_error text, no exit()."""
    cfg = synth.custom(3, 2, 4)
    P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
    x0 = synth.initial_guess(cfg)
    with pytest.raises(capi.RpError):  # no observations yet
        P.solve(x0)
    obs = synth.fill_problem(P, cfg)
    with pytest.raises(capi.RpError):  # camera index out of range
        P.add_observations(np.array([cfg.cn], np.int32), obs[1][:1], obs[2][:1], obs[3][:1], obs[4][:1], obs[5][:1])
    with pytest.raises(capi.RpError):
        P.set_jacobian_mode(7)
    P.lm_begin(x0)
    with pytest.raises(capi.RpError):  # not between lm_begin and lm_end
        P.set_jacobian_mode("materialised")
    P.lm_step()
    P.lm_end()
    P.set_jacobian_mode("materialised")
    x1, s = P.solve(x0)
    assert s.final_cost < s.initial_cost
    P.close()


def test_zero_rational_terms_fast_path_is_bit_identical():
    """rp_model.hpp NODEN: when every camera's k4 = k5 = k6 is zero (what CopyFromCalibration
    hands over) the fused kernel drops the denominator's division.  A camera that no observation
    mentions gets k4 = 0.5, which switches the whole launch to the general path: every number
    must come out bit for bit the same."""
    cfg = synth.custom(5, 3, 12)
    obs = synth.observations(cfg)
    keep = obs[0] != 4
    x0 = synth.initial_guess(cfg)
    out = []
    for k4 in (0.0, 0.5):
        K = synth.intrinsics(cfg).copy()
        assert not K[:, 9:].any()
        K[4, 9] = k4
        P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
        P.set_intrinsics(K)
        P.set_points(synth.points(cfg))
        P.add_observations(*[a[keep] for a in obs])
        S, b = P.normal_eq(x0)
        x, s = P.solve(x0)
        out.append((S, b, x, s.final_cost))
        P.close()
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)


# 16 draws by default; CALICO_TEST_SEEDS=3000 is a fuzzing run (30 s).  Draws 24, 284 and 486 are the ones that found
# the square root of a Gram diagonal rounded to -4e-15 (time block seen only in the corner at the board's origin:
# fused mode reported "not positive definite", or returned NaN without an error when no time block is eliminated)
_SEEDS = (range(int(os.environ["CALICO_TEST_SEEDS"])) if "CALICO_TEST_SEEDS" in os.environ
          else list(range(16)) + [24, 284, 486])


@pytest.mark.parametrize("seed", _SEEDS)
def test_random_networks_vs_oracle(oracle_api, seed):
    """A sweep over network shapes nobody picked by hand: random numbers of cameras, patterns and
    times, random p* / t*, random board size, a random share of FRs and of corners dropped, the
    observations handed over in random order and in two batches, one camera or pattern or a few
    times possibly never observed.  Normal equations at the start point in both modes (1e-10) and
    three LM iterations (same decisions, same cost) against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    big = 3 if os.environ.get("CALICO_TEST_BIG") else 1  # (larger draws: several wave ranges, SYRK tiles and Cholesky panels)
    cn, pn, tn = int(rng.integers(1, 10 * big)), int(rng.integers(1, 5 * big)), int(rng.integers(1, 40 * big))
    cfg = synth.custom(cn, pn, tn, sx=int(rng.integers(3, 9)), sy=int(rng.integers(3, 10)),
                       fr_drop_prob=float(rng.choice([0.0, 0.3, 0.7])), corner_drop_prob=float(rng.choice([0.0, 0.5, 0.9])),
                       one_board_per_cam_time=int(rng.integers(0, 2)), p_star_idx=int(rng.integers(0, pn)),
                       t_star_idx=int(rng.integers(0, tn)), tangential=int(rng.integers(0, 2)), seed=77 + seed)
    obs = synth.observations(cfg)
    if obs[0].size == 0:
        pytest.skip("the draw left no observation")
    perm = rng.permutation(obs[0].size)
    if rng.random() < 0.5 and obs[0].size > 50:  # drop a whole camera / pattern / a few times
        which = int(rng.integers(0, 3))
        col = obs[which][perm]
        uniq = np.unique(col)
        if uniq.size > 1:
            victims = rng.choice(uniq, size=1 if which < 2 else min(3, uniq.size - 1), replace=False)
            perm = perm[~np.isin(col, victims)]
    obs = [a[perm] for a in obs]
    cut = int(rng.integers(0, perm.size + 1))
    # the fewest corners any camera, pattern or time is seen in
    min_per_block = int(min(np.unique(obs[k], return_counts=True)[1].min() for k in range(3)))
    # one draw in three with camera parameters refined (the extension; fused mode only)
    mask = int(rng.choice([0, 0, 0, 0, 0xF, 0x1FF, 0xFFF]))
    x0 = synth.initial_guess(cfg)
    ref = None
    for mode, api in (("oracle", oracle_api), ("fused", None), ("materialised", None)):
        if mask and mode == "materialised":
            continue
        P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0, api=api)
        if api is None:
            P.set_jacobian_mode(mode)
        P.set_intrinsics(synth.intrinsics(cfg))
        P.set_points(synth.points(cfg))
        P.add_observations(*[a[:cut] for a in obs])
        P.add_observations(*[a[cut:] for a in obs])
        if mask:
            P.set_refine_intrinsics(mask)
        S, b = P.normal_eq(x0)
        if mode == "oracle":
            sq = (P.eval(x0)[1] ** 2).sum(axis=1)
        elif mode == "fused":
            # the per-FR report (multicamera.cpp:717-731) over the same draw, both entry points
            cam, pat, tim, cnt, err, rrmse = P.fr_errors(x0)
            frs = {}
            for c, p_, t, e in zip(obs[0], obs[1], obs[2], sq):
                frs.setdefault((int(c), int(p_), int(t)), []).append(e)
            assert len(frs) == cam.size
            for c, p_, t, n, e in zip(cam, pat, tim, cnt, err):
                v = frs[(int(c), int(p_), int(t))]
                # (residuals are differences of pixel coordinates of a few hundred to thousand: 1e-13 px
                # absolute, which is more than 1e-12 relative in the square of a residual of 0.02 px)
                assert n == len(v) and abs(e - np.mean(v)) <= 1e-12 * max(np.mean(v), 1.0)
            want = np.sqrt(np.mean([np.mean(v) for v in frs.values()]))
            assert abs(rrmse - want) <= 1e-12 * want
            again = P.fr_errors(x0, noalloc=True)
            assert all(np.array_equal(a, b_) for a, b_ in zip(again[:5], (cam, pat, tim, cnt, err)))
            assert abs(again[5] - rrmse) <= 1e-14 * rrmse  # (the two entry points add the FRs' errors in different orders)
        x, s = P.solve(x0, capi.default_options(max_num_iterations=3))
        got = (S, b, x, s.final_cost, s.num_iterations, s.num_successful_steps, s.termination_type)
        initial_cost = s.initial_cost
        P.close()
        if ref is None:
            ref = got
            continue
        # fewer residuals than about three per unknown: the poses are barely determined, the time blocks'
        # E_tt are singular up to the damping, and both the elimination and three LM steps amplify the
        # rounding of the sums accordingly (5e-8 seen on a network of 14 observations)
        determined = 2 * perm.size > 6 * (7 * cfg.vn + (12 * cfg.cn if mask else 0))
        # (a fuzzing run: 3 of 30 000 draws exceed 1e-10 on an ill-conditioned elimination, 4.1e-10 at most)
        tol = (1e-9 if "CALICO_TEST_SEEDS" in os.environ else 1e-10) if determined and min_per_block >= 4 else 1e-6
        assert got[0].shape == ref[0].shape, mode
        assert _relerr(got[0], ref[0]) < tol and _relerr(got[1], ref[1]) < tol, mode
        assert np.isfinite(got[3]) and np.all(np.isfinite(got[2])), mode
        # The LM sequence is compared where every block is seen in at least four corners (the
        # reference counts a pattern from ten detected corners on and minimises over k = 8 of them,
        # calico.cpp:291-307).  A block seen in a single corner, or in the corner at the board's
        # origin, has columns of J that are exactly zero; the fused kernel's chain form gives their
        # Gram entries as differences of large terms, 1e-13 instead of 0, which is of the size of the
        # damping min_lm_diagonal / radius that alone keeps such a block positive definite: a step can
        # then count as invalid where exact zeros let it pass (3 of 3 500 draws).
        if min_per_block >= 4:
            assert got[4:] == ref[4:], (mode, got[4:], ref[4:])
            # (relative to the final cost, with a floor where the draw has an exact solution and the final
            # cost is rounding)
            assert abs(got[3] - ref[3]) <= (1e-7 if determined else 1e-4) * ref[3] + 1e-13 * initial_cost, mode


def test_two_handles_solve_in_two_threads():
    """The reference is one instance per run; the C ABI keeps all state in the handle
    (two streams, events, buffers), so two handles may run at the same time from two threads of
    one process: same results as one after the other, bit for bit."""
    import threading

    cfgs = [synth.custom(6, 3, 40), synth.custom(5, 2, 25, fr_drop_prob=0.3, seed=9)]

    def run(cfg, out, k):
        P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
        synth.fill_problem(P, cfg)
        res = []
        for _ in range(3):
            x, s = P.solve(synth.initial_guess(cfg), capi.default_options(max_num_iterations=8))
            res.append((x.tobytes(), s.final_cost, s.num_iterations))
        P.close()
        out[k] = res

    serial, threaded = {}, {}
    for k, cfg in enumerate(cfgs):
        run(cfg, serial, k)
    th = [threading.Thread(target=run, args=(cfg, threaded, k)) for k, cfg in enumerate(cfgs)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
        assert not t.is_alive()
    assert threaded == serial
    assert len({r[0] for r in serial[0]}) == 1  # and repeatable from one solve to the next


@pytest.mark.parametrize("seed", range(int(os.environ.get("CALICO_TEST_SEEDS", "8"))))
def test_random_cumulative_batches_vs_oracle(oracle_api, seed):
    """The Stage-5 loop's shape (calico.cpp:703-731): observations arrive in batches and the
    cumulative problem is solved after each, starting from the previous result.  Random networks,
    two to four random batches in random order, both sides doing the same sequence: the structure
    is rebuilt between solves (new FRs, new blocks becoming active, record buffers growing)."""
    rng = np.random.default_rng(7000 + seed)
    cn, pn, tn = int(rng.integers(2, 9)), int(rng.integers(1, 4)), int(rng.integers(4, 40))
    cfg = synth.custom(cn, pn, tn, fr_drop_prob=float(rng.choice([0.0, 0.3])), corner_drop_prob=float(rng.choice([0.0, 0.5])),
                       one_board_per_cam_time=int(rng.integers(0, 2)), seed=900 + seed)
    obs = synth.observations(cfg)
    if obs[0].size < 200:
        pytest.skip("too small a draw")
    # batches are sets of whole FRs, as in the reference (equation_ordering is a permutation of FRs)
    key = (obs[0].astype(np.int64) * cfg.vn + obs[1]) * cfg.vn + obs[2]
    frs = rng.permutation(np.unique(key))
    cuts = np.sort(rng.choice(np.arange(1, frs.size), size=min(int(rng.integers(1, 4)), frs.size - 1), replace=False))
    batches = [np.flatnonzero(np.isin(key, part)) for part in np.split(frs, cuts)]
    results = []
    for api in (oracle_api, None):
        P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0, api=api)
        P.set_intrinsics(synth.intrinsics(cfg))
        P.set_points(synth.points(cfg))
        x = synth.initial_guess(cfg)
        trace = []
        for idx in batches:
            P.add_observations(*[a[idx] for a in obs])
            x, s = P.solve(x, capi.default_options(max_num_iterations=4))
            trace.append((s.num_iterations, s.num_successful_steps, s.termination_type, s.initial_cost, s.final_cost))
        results.append((x, trace))
        P.close()
    (xo, to), (xg, tg) = results
    n_seen, determined_so_far = 0, True
    for k, ((io, so, termo, c0o, c1o), (ig, sg, termg, c0g, c1g), idx) in enumerate(zip(to, tg, batches)):
        n_seen += idx.size
        # (the first batches of a draw may leave most blocks barely determined; what follows such a batch
        # starts from a point that differs at the level that solve left)
        determined_so_far = determined_so_far and 2 * n_seen > 6 * 7 * cfg.vn
        if k == 0:
            assert abs(c0g - c0o) <= 1e-12 * c0o
        if determined_so_far:
            assert abs(c0g - c0o) <= 1e-6 * c0o
            assert (ig, sg, termg) == (io, so, termo)
            assert abs(c1g - c1o) <= 1e-6 * c1o + 1e-13 * c0o
    assert np.all(np.isfinite(xg))
    if determined_so_far:
        assert np.abs(xg - xo).max() <= 1e-5 * max(1.0, np.abs(xo).max())


@pytest.mark.parametrize("mode", MODES)
def test_one_fr_longer_than_a_wave_range(oracle_api, mode):
    """An FR with far more observations than a wave's range (the interface does not forbid repeated
    corners): its part is cut at every wave boundary, the pieces' sums must still add up."""
    cfg = synth.custom(4, 2, 6)
    obs = list(synth.observations(cfg))
    rng = np.random.default_rng(3)
    key = (obs[0].astype(np.int64) * cfg.vn + obs[1]) * cfg.vn + obs[2]
    victim = np.flatnonzero(key == key[len(key) // 2])
    rep = np.tile(victim, 400)  # ~19 000 observations in one FR
    idx = rng.permutation(np.concatenate([np.arange(key.size), rep]))
    obs = [a[idx] for a in obs]
    obs[4] = obs[4] + 0.05 * rng.normal(size=idx.size)
    obs[5] = obs[5] + 0.05 * rng.normal(size=idx.size)
    out = []
    for api in (oracle_api, None):
        P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0, api=api)
        if api is None:
            P.set_jacobian_mode(mode)
        P.set_intrinsics(synth.intrinsics(cfg))
        P.set_points(synth.points(cfg))
        P.add_observations(*obs)
        x0 = synth.initial_guess(cfg)
        S, b = P.normal_eq(x0)
        x, s = P.solve(x0, capi.default_options(max_num_iterations=5))
        out.append((S, b, x, s))
        P.close()
    (So, bo, xo, so), (Sg, bg, xg, sg) = out
    assert _relerr(Sg, So) < 1e-10 and _relerr(bg, bo) < 1e-10
    assert (sg.num_iterations, sg.num_successful_steps, sg.termination_type) == (so.num_iterations, so.num_successful_steps,
                                                                                 so.termination_type)
    assert abs(sg.final_cost - so.final_cost) <= 1e-8 * so.final_cost


def test_handles_give_their_device_memory_back():
    """create / add / solve / report / destroy a few hundred times, all three kinds of handle: neither the device's free memory
    nor the process's resident size may drift (every buffer, stream and event of a handle goes with it)."""
    import torch

    cfg = synth.custom(6, 3, 30)
    obs = synth.observations(cfg)
    x0 = synth.initial_guess(cfg)
    rel = synth.ae_relations(cfg)
    rae = synth.rae_inputs(cfg, x=x0)

    def cycle(n):
        for _ in range(n):
            P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
            P.set_intrinsics(synth.intrinsics(cfg))
            P.set_points(synth.points(cfg))
            P.add_observations(*obs)
            P.solve(x0, capi.default_options(max_num_iterations=3))
            P.fr_errors(x0)
            P.normal_eq(x0)
            P.close()
            A = capi.AeProblem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
            A.add_relations(*rel[:4])
            A.solve(rel[5], capi.default_options(max_num_iterations=2))
            A.close()
            R = capi.RaeProblem(rae[1], rae[2], rae[4], rae[5], *rae[6:10])
            R.solve(rae[0], rae[3])
            R.close()

    import psutil

    cycle(20)  # (first use: code objects, the runtime's and the allocator's own pools)
    torch.cuda.synchronize()
    free0, _total = torch.cuda.mem_get_info(0)
    rss0 = psutil.Process().memory_info().rss
    cycle(200)
    torch.cuda.synchronize()
    free1, _total = torch.cuda.mem_get_info(0)
    rss1 = psutil.Process().memory_info().rss
    assert free0 - free1 < 64 << 20, (free0, free1)
    assert rss1 - rss0 < 64 << 20, (rss0, rss1)  # ... and the host's: pinned buffers, vectors, threads
