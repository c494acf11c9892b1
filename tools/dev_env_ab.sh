#!/bin/bash
# dev: same-box A/B of one environment switch of a development build: tools/dev_env_ab.sh VAR [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=$1; shift
for rep in 1 2 3; do
  for L in off on; do
    if [ $L = on ]; then export $V=1; else unset $V; fi
    timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --materialised-steps 0 --steps 40 --warmup 8 "$@" > /tmp/ab_$L.json || exit 1
    python3 - $L <<'PY'
import json, sys
d = json.loads([l for l in open('/tmp/ab_%s.json' % sys.argv[1]) if l.startswith('{')][-1])
print("%-5s %.4f ms/step  fused %.3f  stages %s" % (sys.argv[1], d["ms_per_step"], d["roofline"]["avg_launch_ms"],
      {k: round(v, 3) for k, v in d["stage_ms_per_step_rank0"].items() if v}))
PY
  done
done
