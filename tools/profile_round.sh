#!/bin/bash
# Kernel stats + PMC passes of the default bench loop (config D, fused mode) on the GPU box.
# usage: tools/profile_round.sh <tag>   -> gpurun_out/<tag>_kernel_stats.csv, <tag>_pmc.csv, <tag>_bench.json
set -e
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp
B="python3 $ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --materialised-steps 3"
rm -rf /tmp/prof_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/stats -- $B > $OUT/${TAG}_bench.json
cp $(find /tmp/prof_$TAG/stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
S="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --materialised-steps 0"
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
         "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_LEVEL_WAVES" \
         "GRBM_GUI_ACTIVE GRBM_COUNT" \
         "WRITE_SIZE" "FETCH_SIZE"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$TAG/pmc$i -- $S > /dev/null
done
python3 $ROOT/tools/pmc_summary.py /tmp/prof_$TAG/pmc* > $OUT/${TAG}_pmc.csv
# the default line, one eighth of the time range (what one of eight ranks holds), and a sustained run: 2000 steps,
# about seven seconds of back-to-back LM iterations (does the rate hold when the part has warmed up?)
timeout -k 10 300 python3 $ROOT/bench.py > $OUT/${TAG}_bench_default.json
timeout -k 10 300 python3 $ROOT/bench.py --no-cpu-baseline --materialised-steps 0 --tn 1250 > $OUT/${TAG}_bench_eighth.json
timeout -k 10 300 python3 $ROOT/bench.py --no-cpu-baseline --materialised-steps 0 --steps 2000 --warmup 8 > $OUT/${TAG}_bench_sustained.json
echo done
