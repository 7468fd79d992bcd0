#!/bin/bash
# Produce the profile artefacts of a round on the GPU box (run through gpurun from the repo root):
#   SRF_COMMIT=<short hash> tools/profile_round.sh r04_j
# writes gpurun_out/<tag>_*:
#   _driver_kernel_stats.md   rocprofv3 --kernel-trace of the DRIVER's command (python bench.py --gpus 1 --steps 20 --warmup 5), per
#                             (kernel, launch size): the averages the line's roofline fractions can be recomputed from
#   _rocprofv3_kernel_stats.* rocprofv3 --kernel-trace --stats of the timed region alone (--headline-only)
#   _step_trace.md            one replayed step of that run, kernel by kernel
#   _pmc_hbm.json / .txt      FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes), with the digest of the kernel sources they were
#                             taken on (bench.py: roofline.traffic_fresh); copied to profiles/ for the bench run below
#   _bench.json               the driver's command once more, outside the profiler; _inlib_events_kernels.json its in-library event table
tag=${1:-rXX}
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
DRV="python $R/bench.py --gpus 1 --steps 20 --warmup 5"
HEAD="$DRV --headline-only"
rm -rf $O/${tag}_kt $O/${tag}_kd
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/${tag}_kd -o p -- $DRV --no-cpu-baseline > $O/${tag}_kd.log 2>&1
{ echo "# rocprofv3 --kernel-trace -- $DRV --no-cpu-baseline   (per kernel and launch size; the headline's launches: mlp_wide_kernel at grid 307200 = 1,200 blocks)"; echo; python $R/tools/condense_trace.py $O/${tag}_kd 48; } > $O/${tag}_driver_kernel_stats.md
rm -rf $O/${tag}_kd
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_kt -o p -- $HEAD > $O/${tag}_kt.log 2>&1
f=$(ls $O/${tag}_kt/*kernel_stats.csv $O/${tag}_kt/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then
  cp $f $O/${tag}_rocprofv3_kernel_stats.csv
  { echo "# rocprofv3 --kernel-trace --stats -- $HEAD"; echo; python $R/tools/condense_rocprof.py $f 40; } > $O/${tag}_rocprofv3_kernel_stats.md
fi
python $R/tools/step_trace.py $O/${tag}_kt 5 > $O/${tag}_step_trace.md 2>&1
rm -rf $O/${tag}_kt
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_pmc_fetch -o p -- $HEAD > $O/${tag}_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_pmc_write -o p -- $HEAD > $O/${tag}_pmc_write.log 2>&1
python $R/tools/pmc_hbm.py $O/${tag}_pmc_fetch $O/${tag}_pmc_write $O/${tag}_pmc_hbm.json > $O/${tag}_pmc_hbm.txt 2>&1
rm -rf $O/${tag}_pmc_fetch $O/${tag}_pmc_write
cd $R
mkdir -p profiles && cp $O/${tag}_pmc_hbm.json profiles/${tag}_pmc_hbm.json   # so that bench.py's roofline.traffic cites this round's counters
timeout 900 $DRV --kernels-json $O/${tag}_inlib_events_kernels.json > $O/${tag}_bench.json 2> $O/${tag}_bench.err
tail -1 $O/${tag}_bench.json | cut -c1-400
