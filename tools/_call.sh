mkdir -p gpurun_out
R=$(pwd); O=$R/gpurun_out
for i in 1 2; do
python bench.py --gpus 1 --steps 300 --warmup 20 --headline-only --graph off 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eager  %.3f ms/step  %.0f rays/s  host %.3f ms' % (b['ms_per_step'], b['value'], b['host_issue_ms_per_step']))"
done
python -m cProfile -o $O/eager.prof bench.py --gpus 1 --steps 300 --warmup 20 --headline-only --graph off > $O/eager_prof.json 2>$O/eager_prof.err
python - <<'PY'
import pstats
p = pstats.Stats('gpurun_out/eager.prof')
p.sort_stats('cumtime').print_stats(60)
PY
