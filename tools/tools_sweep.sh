#!/bin/bash
# tuning sweep (scratch tool): prints kernel_ms / value for env-var variants
out=gpurun_out/sweep.txt; : > $out
run() { echo "== $*" >> $out; env "$@" timeout -k 10 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'red', round(d['roofline']['reduce_update_ms'],4), 'frac', round(d['roofline']['frac'],4), 'read_ms', round(d['roofline']['stream_read_ms'],4), 'readGBs', round(d['roofline']['stream_read_GBs'],0))" >> $out 2>&1; }
for il in 1 16; do for nt in 0 1; do run BB_INTERLEAVE=$il BB_NT=$nt; done; done
for wpc in 4 8 12; do run BB_WAVES_PER_CU=$wpc BB_INTERLEAVE=1 BB_NT=0; done
cat $out
