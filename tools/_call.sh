( timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_graph.py tests/test_gpu_render.py -q -x 2>&1 | tail -4 )
for i in 1 2 3; do timeout 300 python bench.py --steps 40 --warmup 5 --headline-only 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['value'], b['ms_per_step'])"; done
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r04_i_kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/r04_i_kt -o p -- python $R/bench.py --steps 20 --warmup 5 --device-warm-steps 30 --headline-only > $O/r04_i_kt.log 2>&1
cd $R
python tools/step_trace.py $O/r04_i_kt 5 > $O/r04_i_step_trace.md 2>&1
head -26 $O/r04_i_step_trace.md | cut -c1-120; tail -3 $O/r04_i_step_trace.md | cut -c1-200
rm -rf $O/r04_i_kt
