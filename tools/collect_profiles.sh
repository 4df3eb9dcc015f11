#!/bin/bash
# Copy the end-of-round evidence (tools/final_bench.sh, run on the GPU box) from gpurun_out/
# into profiles/ under this round's names and merge the PMC entries into pmc_latest.json.
r=${1:-r04}; O=gpurun_out; P=profiles
cp $O/bench_final_n50k.json $P/bench_${r}_n50k.json
cp $O/bench_final_steps20.json $P/bench_${r}_n50k_steps20.json
cp $O/bench_final_n24926.json $P/bench_${r}_n24926.json
cp $O/bench_final_n61914.json $P/bench_${r}_n61914.json
cp $O/bench_final_n17700.json $P/bench_${r}_n17700.json
cp $O/bench_final_n963_f64.json $P/bench_${r}_n963_f64.json
cp $O/bench_final_n24926_f64.json $P/bench_${r}_n24926_f64.json
cp $O/bench_final_under_rocprof.json $P/${r}_bench_under_rocprof.json
cp $O/final_kernel_stats_default_bench.csv $P/${r}_kernel_stats_default_bench.csv
cp $O/size_sweep.txt $P/${r}_size_sweep.txt
cp $O/timeline_final.txt $P/${r}_timeline.txt
cp $O/exch_final.txt $P/${r}_share_step.txt
for f in $O/pmc_final_n*.txt; do cp $f $P/${r}_$(basename $f); done
cp $O/bench_final_genome10kb.json $P/bench_${r}_genome10kb.json
cp $O/bench_final_genome10kb_under_rocprof.json $P/${r}_bench_genome10kb_under_rocprof.json
cp $O/final_kernel_stats_genome10kb.csv $P/${r}_kernel_stats_genome10kb.csv
cp $O/spectral_final.txt $P/${r}_spectral_timing.txt
cp $O/batch_final.txt $P/${r}_batch.txt
cp $O/pipeline_final.txt $P/${r}_pipeline_timing.txt
python3 tools/pmc_merge.py final
[ -f $O/utrace_f64.txt ] && cp $O/utrace_f64.txt $P/${r}_unit_trace_f64_n24926.txt
ls $P | grep -c $r
