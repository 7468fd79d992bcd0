#!/bin/bash
# Produce the profile artefacts of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01_d
# writes gpurun_out/<tag>_*: rocprofv3 kernel stats (csv + md), PMC HBM bytes per kernel, bench JSON line, in-library event table.
tag=${1:-rXX}
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-eager-baseline --no-fp32-mode --no-roofline --no-extra-legs"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_kt -o p -- $CMD > $O/${tag}_kt.log 2>&1
f=$(ls $O/${tag}_kt/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then
  cp $f $O/${tag}_rocprofv3_kernel_stats.csv
  { echo "# rocprofv3 --kernel-trace --stats -- $CMD"; echo; python $R/tools/condense_rocprof.py $f 40; } > $O/${tag}_rocprofv3_kernel_stats.md
fi
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_pmc_fetch -o p -- $CMD > $O/${tag}_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_pmc_write -o p -- $CMD > $O/${tag}_pmc_write.log 2>&1
python $R/tools/pmc_hbm.py $O/${tag}_pmc_fetch $O/${tag}_pmc_write $O/${tag}_pmc_hbm.json > $O/${tag}_pmc_hbm.txt 2>&1
cd $R
mkdir -p profiles && cp $O/${tag}_pmc_hbm.json profiles/${tag}_pmc_hbm.json   # so that bench.py's roofline.traffic cites this round's counters
timeout 900 python bench.py --kernels-json $O/${tag}_inlib_events_kernels.json > $O/${tag}_bench.json 2> $O/${tag}_bench.err
tail -1 $O/${tag}_bench.json | cut -c1-600
