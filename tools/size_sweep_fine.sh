#!/bin/bash
# fp32 sweep kernel fraction over a fine grid of sizes (is a dip at one size the size or the run?)
out=gpurun_out/size_sweep_fine.txt; : > $out
for rep in 1 2; do
  for n in ${SWEEP_BINS:-24926 28000 30000 32000 33000 34000 35000 36000 37000 38000 40000 42000 45000 50000}; do
    timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --converge-steps 0 --reps 3 --bins $n 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('N=%6d  %8.1f Gpair-updates/s  step %.4f ms  kernel %.4f ms  frac %.3f  of read sweep %.3f' % ($n, d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r['frac_of_stream_read']))" >> $out 2>&1
  done
done
sort -k1,1 -s $out
