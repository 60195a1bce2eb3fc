#!/bin/bash
# round 3 experiment: contexts on disjoint CU sets (HCONV_CU_SPLIT)
O=gpurun_out/r3cu; mkdir -p $O
for cfg in "0 4 4" "2 2 8" "2 4 4" "4 4 4" "-2 2 8" "-2 4 4" "-4 4 4" "8 8 2" "0 4 4"; do
  set -- $cfg
  HCONV_CU_SPLIT=$1 timeout 300 python bench.py --steps 30 --warmup 5 --streams $2 --batch $3 --no-cpu-baseline > $O/b.json 2>> $O/err.txt
  python -c "
import json; d=json.load(open('$O/b.json')); print('CU_SPLIT=$1 streams=$2 batch=$3 =>', round(d['value'],1), 'conv/s  single', round(d['roofline']['single_conv_ms'],3))"
done
