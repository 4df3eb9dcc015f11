#!/bin/bash
out=gpurun_out/sweep3.txt; : > $out
run() { n=$1; shift; echo "== bins $n $*" >> $out; env "$@" timeout -k 10 120 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --bins $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'step_ms', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'red', round(d['roofline']['reduce_update_ms'],4), 'read_ms', round(d['roofline']['stream_read_ms'],4))" >> $out 2>&1; }
for wpc in 4 8 4 8 16; do run 50000 BB_WAVES_PER_CU=$wpc; done
for wpc in 2 4; do run 17700 BB_WAVES_PER_CU=$wpc; done
cat $out
