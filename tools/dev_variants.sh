#!/bin/bash
# dev: fused7_kernel phase ablations (RP_FUSED_VARIANT); prints the stage time of the kernel per variant
for V in 0 2 4 6 8 14; do
  echo "variant $V"
  RP_FUSED_VARIANT=$V timeout -k 10 100 python tools/dev_fused_time.py || exit 1
done
