#!/bin/bash
# dev: kernel statistics of a short default bench run, lines matching $1 (grep -i pattern)
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp
rm -rf /tmp/pck
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pck -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --materialised-steps 0 ${@:2} > $R/gpurun_out/kstat_bench.json 2>/dev/null
grep -i -h "$1" $(find /tmp/pck -name '*kernel_stats.csv' | head -1) | cut -c1-170
