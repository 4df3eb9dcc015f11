#!/bin/bash
# Throughput and roofline fraction over problem sizes, fp32 and fp64, one line each
# (bench.py, 60 timed steps, settle phase on): the table in profiles/archive/r02_size_sweep.txt.
out=gpurun_out/size_sweep.txt; : > $out
for dt in float32 float64; do
  for n in ${SWEEP_BINS:-963 2000 4096 5000 8000 12000 17700 24926 35000 50000 61914 90000}; do
    [ $dt = float64 ] && [ $n -gt 62000 ] && continue
    timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --converge-steps 0 --reps 3 --bins $n --dtype $dt 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$dt N=%6d  %8.1f Gpair-updates/s  step %.4f ms  kernel %.4f ms  frac %.3f  path %s' % ($n, d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r['kernel']))" >> $out 2>&1
  done
done
cat $out
