mkdir -p gpurun_out
R=$(pwd); O=$R/gpurun_out
timeout 600 python -m pytest tests/test_gpu_stages.py -m gpu -x -q -k "feature or dfeat or mlp_backward" 2>&1 | tail -2
H="python bench.py --gpus 1 --steps 300 --warmup 20 --headline-only"
one() { python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1  %.3f ms/step  %.0f rays/s' % (b['ms_per_step'], b['value']))"; }
for i in 1 2 3; do
  $H 2>/dev/null | one "coarser levels first "
  SRF_LIB_TAG=old $H 2>/dev/null | one "finest level first   "
done
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o p -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --headline-only > $O/kt.log 2>&1
python $R/tools/step_trace.py $O/kt 5 > $O/cur_step_trace.md 2>&1
rm -rf $O/kt
