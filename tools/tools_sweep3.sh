#!/bin/bash
out=gpurun_out/sweep4.txt; : > $out
run() { n=$1; shift; echo "== bins $n $*" >> $out; env "$@" timeout -k 10 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --converge-steps 0 --bins $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'step_ms', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'red', round(d['roofline']['reduce_update_ms'],4))" >> $out 2>&1; }
run 17700 A=1; run 25000 A=1; run 50000 A=1
cat $out
