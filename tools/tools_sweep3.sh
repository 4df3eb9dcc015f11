#!/bin/bash
# waves per CU x problem size (BB_WAVES_PER_CU overrides the built-in policy)
out=gpurun_out/sweep_wpc.txt; : > $out
run() { n=$1; shift; echo -n "bins $n $*: " >> $out; env "$@" timeout -k 10 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --converge-steps 0 --bins $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'step_ms', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'red', round(d['roofline']['reduce_update_ms'],4))" >> $out 2>&1; }
for rep in 1 2; do for n in ${SWEEP_BINS:-17700 24926 35000}; do for w in 4 6 8; do run $n BB_WAVES_PER_CU=$w; done; done; done
sort -k2,2n -k3,3 $out
