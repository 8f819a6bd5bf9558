#!/bin/bash
# decode at 8 rows (ONE group): grid split-K counts of the q/k/v + o and the down projections (VR2_SPLITS="q,d")
for s in 4,4 5,5 6,6 7,7 7,9 8,8 9,9 12,12 16,16; do
  echo "splits $s"; VR2_SPLITS=$s python profiles/src/decode_only.py 1 256 2>&1 | grep "ms/token" | tail -1
done
