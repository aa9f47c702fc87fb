#!/bin/bash
# Everything the round's profiles/ entries come from, in one GPU-box call:  tools/profile_final.sh TAG
set -e -o pipefail
TAG=$1
export TMPDIR=/tmp
bash tools/profile_round.sh ${TAG} > gpurun_out/${TAG}_profile.log 2>&1
python3 tools/tile_tail.py C3 gpurun_out/${TAG}_tile_tail_C3.json > gpurun_out/${TAG}_tail.log 2>&1
bash tools/profile_round.sh ${TAG}_C5 --config C5 > gpurun_out/${TAG}_C5_profile.log 2>&1
O=$PWD/gpurun_out/prof_${TAG}_loss; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/loss_prof.py > $O/loss_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/loss_prof.py > $O/loss_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/loss_prof.py > $O/loss_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $O/pmc_valu -- python3 tools/loss_prof.py > $O/loss_valu.log 2>&1
python3 tools/pmc_summary.py $O/pmc_summary.json $O/pmc_valu $O/pmc_fetch $O/pmc_write --config loss_1920x1080 > $O/pmc_summary.txt
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
grep us_per_call $O/loss_stats.log | tail -1
python3 bench.py --steps 100 --warmup 10 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
for c in C1 C2 C5; do python3 bench.py --config $c --steps 40 --warmup 6 --no-cpu-baseline > gpurun_out/${TAG}_bench_$c.json 2> gpurun_out/${TAG}_bench_$c.err; done
python3 bench.py --config C4 --steps 20 --warmup 4 --no-cpu-baseline > gpurun_out/${TAG}_bench_C4.json 2> gpurun_out/${TAG}_bench_C4.err
echo done
