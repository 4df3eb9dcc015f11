#!/bin/bash
# fp64 2 x 512 units: the MFMA row reduction + parked row sums (product) against the generic
# unit (libabl_F64GEN.so, BB_DEFER_ROWS=0), interleaved in one box.
out=gpurun_out/f64_ab.txt; : > $out
run() { echo -n "$1 bins $2: " >> $out; shift; n=$1; shift
  env "$@" timeout -k 10 300 python bench.py --dtype float64 --steps 30 --warmup 5 --no-cpu-baseline --converge-steps 0 --settle-ms 150 --reps 3 --bins $n 2>/dev/null \
   | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('step_ms', round(d['ms_per_step'],4), 'kernel_ms', round(r['kernel_ms'],4), 'frac', round(r['frac'],3), 'Gpair/s', round(d['value'],1))" >> $out 2>&1; }
for rep in 1 2 3; do for n in 20000 40000; do
  run mfma $n BB_X=1
  run generic $n BB_LIB=$PWD/tools/variants/libabl_F64GEN.so BB_DEFER_ROWS=0
done; done
sort -k3,3n -k1,1 $out
