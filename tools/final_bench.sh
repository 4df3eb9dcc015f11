#!/bin/bash
# End-of-round evidence in one call: default bench (N=50,000) plain and under rocprofv3
# kernel-trace, the two other BASELINE sizes, PMC passes at the three sizes.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_final_n50k.json 2> $O/bench_final_n50k.err
python3 $R/bench.py --bins 24926 --no-cpu-baseline > $O/bench_final_n24926.json 2>/dev/null
python3 $R/bench.py --bins 61914 --no-cpu-baseline > $O/bench_final_n61914.json 2>/dev/null
python3 $R/bench.py --bins 963 --dtype float64 --steps 2000 --warmup 100 --no-cpu-baseline --converge-steps 0 > $O/bench_final_n963_f64.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final_default -- python3 $R/bench.py --no-cpu-baseline > $O/bench_final_under_rocprof.json 2> $O/prof_final_default.log
cp $O/prof_final_default/*/*kernel_stats.csv $O/final_kernel_stats_default_bench.csv 2>/dev/null
for n in 50000 24926 61914; do $R/tools/tools_pmc.sh final $n || exit 1; done
echo done
