#!/bin/bash
# round 4, first GPU call: the new parity tests, then config 5's first figures
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; export TMPDIR=/tmp; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -q -x -s -m gpu -k "at_depth or headline or genome10kb or spill" > $O/r04_tests1.log 2>&1; echo "tests rc=$?" | tee -a $O/r04_tests1.log
tail -5 $O/r04_tests1.log
timeout -k 10 600 python3 bench.py --workload genome10kb > $O/bench_genome10kb.json 2> $O/bench_genome10kb.err; echo "genome bench rc=$?"
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $O/bench_dense.json 2> $O/bench_dense.err; echo "dense bench rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_genome10kb -- python3 bench.py --workload genome10kb --no-cpu-baseline --converge-steps 0 --reps 0 > $O/bench_genome10kb_under_rocprof.json 2> $O/prof_genome10kb.log
cp $O/prof_genome10kb/*/*kernel_stats.csv $O/genome10kb_kernel_stats.csv 2>/dev/null
head -c 1500 $O/bench_genome10kb.json; echo
