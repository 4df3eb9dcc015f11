#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; export TMPDIR=/tmp; mkdir -p $O
cd $R
timeout -k 10 1000 python3 -m pytest tests -q -x -m gpu > $O/r04_tests8.log 2>&1; echo "tests rc=$?"; tail -3 $O/r04_tests8.log
BB_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 20 --reps 1 > $O/bench_forcedist_world1.json 2> $O/bench_forcedist_world1.err; echo "forcedist rc=$?"
python3 -c "
import json; d=json.load(open('$O/bench_forcedist_world1.json')); print(d['value'], d['ms_per_step'], d['config']['exchange'], d['config']['exchange_trial'])"
BB_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 20 --reps 1 --bins 17700 > $O/bench_forcedist_world1_n17700.json 2>/dev/null; echo "forcedist 17700 rc=$?"
python3 -c "
import json; d=json.load(open('$O/bench_forcedist_world1_n17700.json')); print(d['value'], d['ms_per_step'], d['config']['exchange'], d['config']['exchange_trial'])"
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --no-cpu-baseline --steps 20 --reps 1 > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo2 rc=$?"
python3 -c "
import json; d=json.load(open('$O/bench_gloo2.json')); print(d['value'], d['ms_per_step'], d['config']['exchange'])"
