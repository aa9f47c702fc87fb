#!/bin/bash
# Copy the summaries of a tools/profile_final.sh TAG run from gpurun_out/ (scratch) into profiles/ (tracked):  tools/collect_profiles.sh TAG
set -e
T=$1
cp gpurun_out/prof_$T/kernel_stats.csv profiles/${T}_kernel_stats.csv
cp gpurun_out/prof_$T/pmc_summary.json profiles/${T}_pmc_summary.json
grep '^{"metric"' gpurun_out/prof_$T/bench_stats.log | tail -1 > profiles/${T}_bench_under_rocprof.json
cp gpurun_out/prof_${T}_C5/kernel_stats.csv profiles/${T}_C5_kernel_stats.csv
cp gpurun_out/prof_${T}_C5/pmc_summary.json profiles/${T}_C5_pmc_summary.json
grep '^{"metric"' gpurun_out/prof_${T}_C5/bench_stats.log | tail -1 > profiles/${T}_C5_bench_under_rocprof.json
cp gpurun_out/prof_${T}_loss/kernel_stats.csv profiles/${T}_loss_kernel_stats.csv
cp gpurun_out/prof_${T}_loss/pmc_summary.json profiles/${T}_loss_pmc_summary.json
grep us_per_call gpurun_out/prof_${T}_loss/loss_stats.log | tail -1 > profiles/${T}_loss_bench.json
cp gpurun_out/${T}_tile_tail_C3.json profiles/
for c in "" _C1 _C2 _C4 _C5; do tail -1 gpurun_out/${T}_bench$c.json > profiles/${T}_bench$c.json; done
ls -la profiles | grep $T
