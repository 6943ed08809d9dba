#!/bin/bash
# Isolated decode timings over the tuning knobs of dh_yolo_decode (unroll factor, non-temporal policy:
# bit 0 = loads, bit 1 = stores).
for u in 1 2 4 8; do for n in 0 1 2 3; do
  echo "unroll=$u nt=$n: $(DH_DECODE_UNROLL=$u DH_DECODE_NT=$n timeout -k 10 100 python tools/decode_variants.py 2>&1 | grep 'obj + logit')"
done; done
