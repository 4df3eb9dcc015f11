#!/bin/bash
out=gpurun_out/f64ab.txt; : > $out
run() { echo -n "$1 $3 bins $4: " >> $out; env BB_LIB=$2 timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --converge-steps 0 --dtype $3 --bins $4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('Gpair/s', round(d['value'],1), 'step_ms', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3))" >> $out 2>&1; }
for rep in 1 2; do for cfg in "float64 20000" "float64 963" "float64 40000" "float32 50000"; do set -- $cfg; run new $PWD/blueberry_amd/libblueberry_hip.so $1 $2; run OLD $PWD/tools/variants/libabl_OLD.so $1 $2; done; done
sort -k2,2 -k4,4n -k1,1 $out
