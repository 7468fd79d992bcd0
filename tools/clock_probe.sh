#!/bin/bash
# what clock / power does the chip hold while the 128-row forward (PROBE_LOOP_WHAT=fwd|bwd|both) runs back to back?  Samples rocm-smi once a
# second for as long as the probe process lives (idle samples before and after the busy stretch show the contrast).
cd "$(dirname "$0")/.." || exit 1
( PROBE_LOOP_S=${PROBE_LOOP_S:-8} python tools/wide_time.py "$@" > /tmp/clock_probe_run.txt 2>&1 ) &
pid=$!
rocm-smi --showclocks --showpower 2>&1 | head -30
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -iE "sclk|power" | sed 's/^.*GPU\[[0-9]*\][ \t]*: //; s/  */ /g' | tr '\n' '|'; echo
  sleep 1
done
tail -2 /tmp/clock_probe_run.txt
