#!/bin/bash
# round 4, sixth GPU call: whole suite, ContactMap stage with counters, batch timing, gram A/B,
# config 5 on two ranks (gloo rehearsal on one GPU), dense bench
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; export TMPDIR=/tmp; mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests -q -x -m gpu > $O/r04_tests6.log 2>&1; echo "tests rc=$?"
tail -4 $O/r04_tests6.log
timeout -k 10 300 python3 tools/batch_timing.py 50000 100000 25000 > $O/r04_batch.txt 2>&1; echo "batch rc=$?"; cat $O/r04_batch.txt
timeout -k 10 400 python3 tools/bench_contactmap.py > $O/r04_contactmap.txt 2>&1; echo "cm rc=$?"; grep -E "normalize|correlation|marginals" $O/r04_contactmap.txt
BB_LIB=$R/tools/variants/libabl_GRAMIL.so timeout -k 10 400 python3 tools/bench_contactmap.py 2>&1 | grep -E "correlation" > $O/r04_gram_il.txt; cat $O/r04_gram_il.txt
timeout -k 10 900 bash tools/cm_pmc.sh; echo "cm_pmc rc=$?"; cat $O/cm_pmc.txt
timeout -k 10 600 python3 bench.py --workload genome10kb --gpus 2 --backend gloo --no-cpu-baseline --steps 20 --reps 1 > $O/bench_genome10kb_gloo2.json 2> $O/bench_genome10kb_gloo2.err; echo "gloo2 rc=$?"; head -c 600 $O/bench_genome10kb_gloo2.json; echo
timeout -k 10 300 python3 bench.py > $O/bench_dense_full.json 2> $O/bench_dense_full.err; echo "dense rc=$?"; head -c 400 $O/bench_dense_full.json; echo
