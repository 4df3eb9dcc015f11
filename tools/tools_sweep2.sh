#!/bin/bash
out=gpurun_out/sweep2.txt; : > $out
run() { echo "== $*" >> $out; env "$@" timeout -k 10 120 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --bins 17700 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'step_ms', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'red', round(d['roofline']['reduce_update_ms'],4))" >> $out 2>&1; }
for wpc in 4 6 8 10 12 16; do run BB_WAVES_PER_CU=$wpc; done
cat $out
