for i in 1 2 3; do for v in 1 4 0; do timeout 300 python bench.py --steps 400 --warmup 20 --headline-only --set renderer.PREFILL_AT=$v 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PREFILL_AT=$v', b['value'], b['ms_per_step'])"; done; done
