#!/bin/bash
# dev: idle time on the GPU between consecutive kernels of the default bench loop (kernel trace)
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp
rm -rf /tmp/pcg
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pcg -- python3 $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --materialised-steps 0 ${@} > /dev/null 2>&1
cp $(find /tmp/pcg -name '*kernel_trace.csv' | head -1) $R/gpurun_out/kernel_trace.csv
python3 - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# iterations start at the first list_sum after a flag_to_scalar
ev = [(r['Kernel_Name'].split('(')[0].replace('void ', ''), int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
gaps = {}
for (n0, s0, e0), (n1, s1, e1) in zip(ev, ev[1:]):
    g = s1 - e0
    gaps.setdefault((n0, n1), []).append(g)
tot = sorted(((sum(v) / len(v), len(v), k) for k, v in gaps.items() if len(v) >= 8), reverse=True)
for avg, n, k in tot[:14]:
    print("%8.1f us avg x %4d  %s -> %s" % (avg / 1e3, n, k[0][-40:], k[1][-40:]))
PY
