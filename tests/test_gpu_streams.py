"""The two orders of an LM iteration give the same bits (rp_api.hip: gathers_beside).

Networks with many parts run the segment diagonals of a speculative evaluation and the pair /
block gathers of an accepted step on a second stream, beside the cost sum and the SYRK; small ones
run them inline on the main stream.  Same kernels, same sums, same order inside each sum: the
iterates must agree bit for bit, and a missing event wait between the streams would show up as a
difference (or as a different accept / reject sequence).  The size threshold keeps every network the
oracle can check on the inline order, so the test build of the library
(tests/_build/libcalico_rp_testhooks.so, -DCALICO_RP_TEST_HOOKS) reads CALICO_RP_TEST_BESIDE = 0 / 1
to force either order; the product build has no such switch (tests/test_host_logic.py).
Child processes: the library is chosen when calico_amd.capi loads it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOOKS_LIB = os.path.join(ROOT, "tests", "_build", "libcalico_rp_testhooks.so")

CHILD = r"""
import hashlib, json, os, sys
import numpy as np
from calico_amd import capi, synth

case = sys.argv[1]
if case == "far":        # rejected steps: cost sweeps in between, the candidate's records discarded
    cfg = synth.custom(5, 3, 20, perturb_rot_deg=60.0, perturb_trans_mm=800.0)
    iters, mask = 30, 0
elif case == "D":        # the headline shape, 945 unknowns
    cfg = synth.preset("D", tn=40)
    iters, mask = 12, 0
else:                    # camera parameters refined: the fourth block's gathers
    cfg = synth.preset("B", tn=40)
    iters, mask = 12, 0xFF
x0 = synth.initial_guess(cfg)
out = {}
for beside in ("0", "1"):
    os.environ["CALICO_RP_TEST_BESIDE"] = beside
    P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
    synth.fill_problem(P, cfg)
    if mask:
        P.set_refine_intrinsics(mask)
    P.lm_begin(x0, capi.default_options(max_num_iterations=iters))
    trace = []
    while True:
        acc, done = P.lm_step()
        if done:
            break
        trace.append(int(acc))
    x, s = P.lm_end()
    k = P.get_intrinsics() if mask else np.zeros(1)
    out[beside] = dict(trace=trace, x=hashlib.sha256(x.tobytes()).hexdigest(), k=hashlib.sha256(k.tobytes()).hexdigest(),
                       cost=float(s.final_cost).hex(), iterations=int(s.num_iterations),
                       termination=int(s.termination_type))
    P.close()
print("RESULT " + json.dumps(out))
"""


@pytest.mark.parametrize("case", ["far", "D", "B_refined"])
def test_gathers_beside_and_inline_agree_bit_for_bit(case):
    assert os.path.exists(HOOKS_LIB), "tests/_build/libcalico_rp_testhooks.so is built by __graft_entry__.build()"
    env = dict(os.environ, CALICO_RP_LIBRARY=HOOKS_LIB, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", CHILD, case], cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    inline, beside = out["0"], out["1"]
    assert len(inline["trace"]) >= 3
    if case == "far":
        assert 0 in inline["trace"] and 1 in inline["trace"], "the case is meant to mix accepted and rejected steps"
    assert beside == inline
