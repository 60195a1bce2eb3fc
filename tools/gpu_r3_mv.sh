#!/bin/bash
# round 3 experiment: mv_split / lanes with more hardware queues per process (GPU_MAX_HW_QUEUES)
O=gpurun_out/r3mv2; mkdir -p $O
run() { env "$1" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $2 > $O/b.json 2>> $O/err.txt; python -c "
import json; d=json.load(open('$O/b.json')); print('$1 $2 =>', round(d['value'],1), 'conv/s  single', round(d['roofline']['single_conv_ms'],3))"; }
run GPU_MAX_HW_QUEUES=4 ""
run GPU_MAX_HW_QUEUES=8 ""
run GPU_MAX_HW_QUEUES=16 ""
run GPU_MAX_HW_QUEUES=16 "--opt mv_split=16"
run GPU_MAX_HW_QUEUES=16 "--opt mv_split=12"
run GPU_MAX_HW_QUEUES=16 "--opt mv_split=20"
run GPU_MAX_HW_QUEUES=16 "--streams 2 --batch 8 --opt mv_split=14"
run GPU_MAX_HW_QUEUES=16 "--lanes 2"
run GPU_MAX_HW_QUEUES=16 "--lanes 4"
run GPU_MAX_HW_QUEUES=16 "--streams 8 --batch 2"
run GPU_MAX_HW_QUEUES=4 ""
