#!/bin/bash
# dev: fused7q_kernel phase ablations (RP_FUSED_VARIANT: 2 no Gram, 4 no evaluation, 8 no part-end products, 16 no record stores);
# prints the stage time of the kernel per variant (RP_FUSED_GRAM16=1: the same on fused7_kernel, variants 2..14)
for V in 0 2 4 6 8 14 16; do
  echo "variant $V"
  RP_FUSED_VARIANT=$V timeout -k 10 100 python tools/dev_fused_time.py || exit 1
done
