#!/bin/bash
# interleaved A/B of two builds in one box: A = product .so, B = blueberry_amd/libabl_B.so
out=gpurun_out/ab.txt; : > $out
run() { echo -n "$1 bins $3: " >> $out; env BB_LIB=$2 timeout -k 10 120 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --converge-steps 0 --bins $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step_ms', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'red', round(d['roofline']['reduce_update_ms'],4))" >> $out 2>&1; }
A=$PWD/blueberry_amd/libblueberry_hip.so; B=$PWD/blueberry_amd/libabl_B.so
for rep in 1 2 3; do for n in 50000 17700; do run A $A $n; run B $B $n; done; done
cat $out
