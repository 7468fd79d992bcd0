#!/bin/bash
# Same-box A/B of two source trees (box-to-box spread is +-4 %, larger than most single changes): the committed baseline unpacked under
# .base_tree/ (git archive <commit> | tar -x -C .base_tree; python -m scenerf_amd.build there) against the working tree, bench.py run
# alternately.   usage: tools/ab_trees.sh [rounds] [extra bench args]
cd "$(dirname "$0")/.." || exit 1
rounds=${1:-3}; shift
for i in $(seq 1 $rounds); do
  for t in .base_tree .; do
    python $t/bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-eager-baseline --no-fp32-mode --no-roofline --no-extra-legs "$@" 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-12s %.3f ms/step  %.0f rays/s  host %.3f' % ('$t', b['ms_per_step'], b['value'], b.get('host_issue_ms_per_step', float('nan'))))"
  done
done
