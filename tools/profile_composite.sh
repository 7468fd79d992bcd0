#!/bin/bash
# Compositing pass against the HBM roofline at inference-sized chunks, with tracked evidence:
#   tools/profile_composite.sh r02_b   ->  gpurun_out/<tag>_composite_{probe.txt,rocprofv3_kernel_stats.md,pmc_hbm.json}
# (HIP-event timing by the probe itself, rocprofv3 --kernel-trace --stats of the same command, PMC FETCH_SIZE / WRITE_SIZE passes)
tag=${1:-rXX}
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
python tools/composite_probe.py > $O/${tag}_composite_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/composite_probe.py"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_comp_kt -o p -- $CMD > $O/${tag}_comp_kt.log 2>&1
f=$(ls $O/${tag}_comp_kt/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && { echo "# rocprofv3 --kernel-trace --stats -- $CMD"; echo; python $R/tools/condense_rocprof.py $f 12; } > $O/${tag}_composite_rocprofv3_kernel_stats.md
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_comp_fetch -o p -- $CMD > $O/${tag}_comp_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_comp_write -o p -- $CMD > $O/${tag}_comp_write.log 2>&1
python $R/tools/pmc_hbm.py $O/${tag}_comp_fetch $O/${tag}_comp_write $O/${tag}_composite_pmc_hbm.json > $O/${tag}_composite_pmc_hbm.txt 2>&1
cd $R
cat $O/${tag}_composite_probe.txt; cat $O/${tag}_composite_pmc_hbm.txt | head -8
