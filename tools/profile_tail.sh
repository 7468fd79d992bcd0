#!/bin/bash
# The per-ray tail the product launches against the HBM roofline, with counters:
#   tools/profile_tail.sh r05_e  ->  gpurun_out/<tag>_tail_{probe.txt,probe.json,rocprofv3_kernel_stats.md,pmc_hbm.json,pmc_hbm.txt}
tag=${1:-rXX}
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
TAIL_PROBE_JSON=$O/${tag}_tail_probe.json python tools/tail_probe.py > $O/${tag}_tail_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/tail_probe.py"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_tail_kt -o p -- $CMD > $O/${tag}_tail_kt.log 2>&1
f=$(ls $O/${tag}_tail_kt/*kernel_stats.csv $O/${tag}_tail_kt/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && { echo "# rocprofv3 --kernel-trace --stats -- $CMD"; echo; python $R/tools/condense_rocprof.py $f 12; } > $O/${tag}_tail_rocprofv3_kernel_stats.md
rm -rf $O/${tag}_tail_kt
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_tail_fetch -o p -- $CMD > $O/${tag}_tail_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_tail_write -o p -- $CMD > $O/${tag}_tail_write.log 2>&1
python $R/tools/pmc_hbm.py $O/${tag}_tail_fetch $O/${tag}_tail_write $O/${tag}_tail_pmc_hbm.json > $O/${tag}_tail_pmc_hbm.txt 2>&1
rm -rf $O/${tag}_tail_fetch $O/${tag}_tail_write
cd $R
cat $O/${tag}_tail_probe.txt; grep -i "ray_tail" $O/${tag}_tail_pmc_hbm.txt | head -8
