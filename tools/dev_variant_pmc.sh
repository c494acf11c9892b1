#!/bin/bash
# dev: instruction mix of fused7_kernel phase ablations (RP_FUSED_VARIANT), one --pmc pass each
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for V in 0 6 4 2; do
  export RP_FUSED_VARIANT=$V
  rm -rf /tmp/vp_$V
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d /tmp/vp_$V -- python3 $ROOT/tools/dev_fused_time.py > /dev/null 2>&1 || exit 1
  echo "variant $V"
  python3 $ROOT/tools/pmc_summary.py /tmp/vp_$V | grep fused7 | sed -e 's/(rp::SweepArgs.*)"/"/'
done
