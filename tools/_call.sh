mkdir -p gpurun_out
R=$(pwd); O=$R/gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
H="python bench.py --gpus 1 --steps 300 --warmup 20 --headline-only"
one() { python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1  %.3f ms/step  %.0f rays/s' % (b['ms_per_step'], b['value']))"; }
for i in 1 2; do
  $H 2>/dev/null | one "current tree  "
  python .base_tree/bench.py --gpus 1 --steps 300 --warmup 20 --no-cpu-baseline --no-eager-baseline --no-fp32-mode --no-roofline --no-extra-legs 2>/dev/null | one "round-3 tree  "
done
