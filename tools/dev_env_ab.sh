#!/bin/bash
# dev: same-box A/B of one environment variable (a switch of a development build, or a hook of the test build
# with CALICO_RP_LIBRARY pointing at it): tools/dev_env_ab.sh VAR VALUE_A VALUE_B [bench args]; "-" = unset
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=$1; A=$2; B=$3; shift 3
for rep in 1 2 3; do
  for L in "$A" "$B"; do
    if [ "$L" = "-" ]; then unset $V; else export $V=$L; fi
    timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --materialised-steps 0 --steps 40 --warmup 8 "$@" > /tmp/ab_env.json || exit 1
    python3 - "$L" <<'PY'
import json, sys
d = json.loads([l for l in open('/tmp/ab_env.json') if l.startswith('{')][-1])
print("%-5s %.4f ms/step  fused %.3f  stages %s" % (sys.argv[1], d["ms_per_step"], d["roofline"]["avg_launch_ms"],
      {k: round(v, 3) for k, v in d["stage_ms_per_step_rank0"].items() if v}))
PY
  done
done
