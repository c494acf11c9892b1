#!/bin/bash
# dev: the kernels of one steady-state LM iteration of the bench loop in launch order: start offset, duration and
# the idle gap in front of each (rocprofv3 kernel trace).  usage: tools/dev_timeline.sh <out-name> [bench args]
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp
OUT=$1; shift
rm -rf /tmp/pct
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pct -- python3 $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --materialised-steps 0 ${@} > /dev/null 2>&1 || exit 1
cp $(find /tmp/pct -name '*kernel_trace.csv' | head -1) /tmp/pct/trace.csv
python3 - "$R/gpurun_out/$OUT" <<'PY'
import csv, sys
rows = list(csv.DictReader(open('/tmp/pct/trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ev = [(r['Kernel_Name'].split('(')[0].replace('void ', '').replace('rp::', ''), int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
# an iteration ends with flag_to_scalar_kernel; take the 8th of the timed region (index from the end: the profiled pass follows)
def tail(i):  # the read-back of an iteration's scalars follows the cost sum / flag kernel (a gather may sit between)
    return any(ev[j][0].startswith(('sum_partials', 'flag_to_scalar')) for j in (i - 1, i - 2) if j >= 0)
ends = [i for i, e in enumerate(ev) if e[0].startswith('__amd_rocclr_copyBuffer') and tail(i)
        and not ev[i - 1][0].startswith('__amd_rocclr_copyBuffer')]
# iterations in launch order: 4 warm-up, 8 timed (events around the fused kernel only), 8 of the stage pass (events
# around every stage: about 10 us of gap at each stage boundary), then the converged solve.  A steady-state
# iteration of the TIMED region (accepted speculative step -> the next one): the shortest such span with a fused
# kernel and no cost sweep in it
allspans = list(zip(ends, ends[1:]))
spans = [(a + 1, b) for a, b in allspans[4:11] if any(e[0].startswith('fused7q') for e in ev[a + 1:b + 1])
         and not any(e[0].startswith('cost_sweep') for e in ev[a + 1:b + 1])]
k0, k = min(spans, key=lambda ab: ab[1] - ab[0])
t0 = ev[k0][1]
with open(sys.argv[1], 'w') as f:
    prev = ev[k0 - 1][2]
    tot = 0
    for n, s, e in ev[k0:k + 1]:
        f.write("%9.1f us  +%7.1f us  gap %6.1f  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, n))
        prev = max(prev, e)
        tot += e - s
    f.write("iteration: %d kernels, %.1f us from first start to last end, %.1f us of kernel time\n" % (k + 1 - k0, (ev[k][2] - t0) / 1e3, tot / 1e3))
print(open(sys.argv[1]).read())
PY
