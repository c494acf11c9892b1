#!/bin/bash
# dev: same-box A/B of the bench line: the library in the tree against tools/_build/libcalico_rp_base.so
# (a build of an earlier commit), alternating, ms per step.  usage: tools/dev_ab.sh [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do
  for L in base new; do
    if [ $L = base ]; then export CALICO_RP_LIBRARY=$R/tools/_build/libcalico_rp_base.so; else unset CALICO_RP_LIBRARY; fi
    timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --materialised-steps 0 --steps 40 --warmup 8 "$@" > /tmp/ab_$L.json || exit 1
    python3 - $L <<'PY'
import json, sys
d = json.loads([l for l in open('/tmp/ab_%s.json' % sys.argv[1]) if l.startswith('{')][-1])
print("%-5s %.4f ms/step  fused %.3f  stages %s" % (sys.argv[1], d["ms_per_step"], d["roofline"]["avg_launch_ms"],
      {k: round(v, 3) for k, v in d["stage_ms_per_step_rank0"].items() if v}))
PY
  done
done
