"""The one entry point that allocates nothing (include/calico_rp.h:
calico_rp_fr_errors_noalloc) against the existing calico_rp_fr_errors, through the C ABI:
two calls in a row on one GPU, the second with a smaller input (both share the library's
file-scope scratch array, which keeps the first call's contents), then the limits."""
import numpy as np
import pytest

from calico_amd import capi, synth

pytestmark = pytest.mark.gpu


def _ready_problem(cfg):
    P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
    synth.fill_problem(P, cfg)
    return P


def _compare(P, x):
    want = P.fr_errors(x)            # existing entry point (allocates per call; also readies the handle)
    got = P.fr_errors(x, noalloc=True)
    for a, b in zip(want[:4], got[:4]):  # cam, pattern, time, corner counts: bit-exact
        assert np.array_equal(a, b)
    assert np.array_equal(want[4], got[4])  # the same kernel: identical per-FR means
    assert want[5] > 0 and abs(got[5] - want[5]) <= 1e-13 * want[5]  # rrmse: device tree sum vs host sum
    return want


def test_noalloc_matches_fr_errors_twice_second_smaller():
    big = synth.custom(6, 3, 40, fr_drop_prob=0.2, corner_drop_prob=0.3)
    small = synth.custom(3, 2, 5)
    Pb, Ps = _ready_problem(big), _ready_problem(small)
    wb = _compare(Pb, synth.initial_guess(big))
    ws = _compare(Ps, synth.initial_guess(small))
    assert ws[0].size < wb[0].size  # fewer FRs: stale means of the first call sit behind them in the scratch
    # and once more on the first handle, at another point
    xg = synth.ground_truth(big)
    wg = _compare(Pb, xg)
    assert wg[5] < wb[5]
    Pb.close()
    Ps.close()


def test_noalloc_states_and_limits():
    cfg = synth.custom(3, 2, 5)
    P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
    x = synth.initial_guess(cfg)
    with pytest.raises(capi.RpError):  # nothing added yet
        P._check(P.api.fr_errors_noalloc(P.h, capi._p(x, capi.C.c_double), None, None, None, None, None, None), "noalloc")
    synth.fill_problem(P, cfg)
    with pytest.raises(capi.RpError, match="not on the device yet"):  # would have to allocate: refused
        P._check(P.api.fr_errors_noalloc(P.h, capi._p(x, capi.C.c_double), None, None, None, None, None, None), "noalloc")
    _compare(P, x)
    P.close()
    # more variables than the scratch array holds: a clear error, no launch
    tn = 32768 + 8
    Q = capi.Problem(2, 2, tn, 2, 4, device=0)
    Q.set_intrinsics(synth.intrinsics(synth.custom(2, 2, 4)))
    Q.set_points(synth.points(synth.custom(2, 2, 4)))
    Q.add_observations([0, 1], [3, 3], [5, 5], [0, 1], [10.0, 20.0], [30.0, 40.0])
    xq = np.zeros(7 * Q.vn)
    xq[0::7] = 1.0
    xq[6::7] = 1000.0
    Q.fr_errors(xq)
    with pytest.raises(capi.RpError, match="exceed the limit"):
        Q.fr_errors(xq, noalloc=True)
    Q.close()
