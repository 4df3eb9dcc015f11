#!/bin/bash
# End-of-round evidence in one call: default bench (N=50,000) plain and under rocprofv3
# kernel-trace, the other BASELINE sizes, the size sweep, the exchange fixed cost, the
# kernel-trace timeline, PMC passes (fp32 at the three sizes, fp64 at two).
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_final_n50k.json 2> $O/bench_final_n50k.err
python3 $R/bench.py --bins 24926 --no-cpu-baseline > $O/bench_final_n24926.json 2>/dev/null
python3 $R/bench.py --bins 61914 --no-cpu-baseline > $O/bench_final_n61914.json 2>/dev/null
python3 $R/bench.py --bins 17700 --no-cpu-baseline > $O/bench_final_n17700.json 2>/dev/null
python3 $R/bench.py --bins 963 --dtype float64 --steps 2000 --warmup 100 --no-cpu-baseline --converge-steps 0 > $O/bench_final_n963_f64.json 2>/dev/null
python3 $R/bench.py --bins 24926 --dtype float64 --no-cpu-baseline > $O/bench_final_n24926_f64.json 2>/dev/null
python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_final_steps20.json 2>/dev/null
# BASELINE config 5: the whole genome at 10 kb as blocked-sparse tiles, plain and under the kernel trace
python3 $R/bench.py --workload genome10kb > $O/bench_final_genome10kb.json 2> $O/bench_final_genome10kb.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final_genome10kb -- python3 $R/bench.py --workload genome10kb --no-cpu-baseline --converge-steps 0 --reps 0 > $O/bench_final_genome10kb_under_rocprof.json 2> $O/prof_final_genome10kb.log
cp $O/prof_final_genome10kb/*/*kernel_stats.csv $O/final_kernel_stats_genome10kb.csv 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final_default -- python3 $R/bench.py --no-cpu-baseline > $O/bench_final_under_rocprof.json 2> $O/prof_final_default.log
cp $O/prof_final_default/*/*kernel_stats.csv $O/final_kernel_stats_default_bench.csv 2>/dev/null
$R/tools/size_sweep.sh > /dev/null 2>&1
$R/tools/timeline.sh final 5000 8000 12000 17700 24926 50000 > /dev/null 2>&1
python3 $R/tools/exchange_timing.py 17700 > $O/exch_final.txt 2>&1
for n in 50000 24926 61914; do $R/tools/tools_pmc.sh final $n || exit 1; done
for n in 24926 50000; do $R/tools/tools_pmc.sh final $n float64 || exit 1; done
$R/tools/tools_pmc.sh final 309568 float32 genome10kb || exit 1
python3 $R/tools/spectral_timing.py > $O/spectral_final.txt 2>&1
python3 $R/tools/batch_timing.py 50000 100000 25000 > $O/batch_final.txt 2>&1
python3 $R/tools/pipeline_timing.py > $O/pipeline_final.txt 2>&1
echo done
