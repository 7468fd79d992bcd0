#!/bin/bash
# Same-box A/B of replayed training steps in ALTERNATING PROCESSES (a process holds one GraphedStep per parameter set, and box-to-box
# spread is larger than most single changes): each argument is "label|bench args", e.g.
#   tools/ab_set.sh 3 "fill behind pack|" "fill behind sampler|--set renderer.PREFILL_AT=1" "other lib|SRF_LIB_TAG=old"
# (a leading integer = rounds, default 3; an entry of the form NAME=VALUE before the bar is exported to that run's environment, everything
# after it is appended to `bench.py --gpus 1 --steps 300 --warmup 20 --headline-only`).  Add TRACE=1 to leave one step's kernel trace of
# the FIRST entry in gpurun_out/ab_step_trace.md.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
rounds=3
if [[ "$1" =~ ^[0-9]+$ ]]; then rounds=$1; shift; fi
run() {   # $1 = "label|args"
    local label="${1%%|*}" rest="${1#*|}" envs=()
    while [[ "$rest" =~ ^([A-Za-z_][A-Za-z0-9_]*=[^ ]*)\ ?(.*)$ && "${BASH_REMATCH[1]}" != --* ]]; do envs+=("${BASH_REMATCH[1]}"); rest="${BASH_REMATCH[2]}"; done
    env "${envs[@]}" python bench.py --gpus 1 --steps 300 --warmup 20 --headline-only $rest 2>/dev/null | python -c "
import json, sys
lines = sys.stdin.read().strip().splitlines()
if not lines: print('%-40s FAILED' % sys.argv[1]); sys.exit(0)
b = json.loads(lines[-1]); print('%-40s %.3f ms/step  %.0f rays/s' % (sys.argv[1], b['ms_per_step'], b['value']))" "$label"
}
for i in $(seq 1 $rounds); do for e in "$@"; do run "$e"; done; done
if [ -n "$TRACE" ]; then
    R=$(pwd); first="$1"; rest="${first#*|}"
    cd /tmp && export TMPDIR=/tmp
    rm -rf $R/gpurun_out/ab_kt
    timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ab_kt -o p -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --headline-only $rest > $R/gpurun_out/ab_kt.log 2>&1
    python $R/tools/step_trace.py $R/gpurun_out/ab_kt 5 > $R/gpurun_out/ab_step_trace.md 2>&1
    rm -rf $R/gpurun_out/ab_kt
fi
