#!/bin/bash
# interleaved comparison of several builds in one box: product .so + tools/variants/libabl_*.so
out=gpurun_out/abn.txt; : > $out
run() { echo -n "$1 bins $3: " >> $out; env BB_LIB=$2 timeout -k 10 120 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --converge-steps 0 --bins $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step_ms', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4))" >> $out 2>&1; }
for rep in $(seq 1 ${ABN_REPS:-3}); do for n in ${ABN_BINS:-50000 17700}; do run base $PWD/blueberry_amd/libblueberry_hip.so $n; for f in tools/variants/libabl_*.so; do run $(basename $f .so) $PWD/$f $n; done; done; done
sort -k3,3n -k1,1 $out
