#!/bin/bash
# Interleaved A/B of environment settings with ONE binary, in one box: for every size in
# $AB_BINS and every setting given ("NAME=VALUE[,NAME=VALUE...]"), $AB_REPS times.
#   usage: AB_BINS="50000 24926" AB_REPS=3 tools/ab_env.sh "BB_PAIR=0" "BB_PAIR=1"
out=gpurun_out/${AB_OUT:-ab_env}.txt; : > $out
for rep in $(seq 1 ${AB_REPS:-3}); do for n in ${AB_BINS:-50000 24926}; do for setting in "$@"; do
  echo -n "bins $n $setting: " >> $out
  env ${setting//,/ } timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --converge-steps 0 --settle-ms 150 --reps 3 --bins $n ${AB_ARGS:-} 2>/dev/null \
   | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('step_ms', round(d['ms_per_step'],4), 'reps_med', round(d['ms_per_step_reps']['median'],4), 'kernel_ms', round(r['kernel_ms'],4), 'reduce_ms', round(r['reduce_update_ms'],4), 'frac', round(r['frac'],3))" >> $out 2>&1
done; done; done
sort -k2,2n -k3,3 $out
