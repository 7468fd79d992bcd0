#!/bin/bash
# A/B of kernel-variant libraries on one GPU box (same clocks, same thermal state): for each tag a short bench.py run, printing the
# step time and the fused / weight-gradient kernel averages from the in-library HIP-event table.
#   build:  SRF_LIB_TAG=<tag> SRF_EXTRA_FLAGS="-D..." python -m scenerf_amd.build      (tag "" = the default library)
#   run:    tools/variants.sh "" old "" old pc pp        (repeat a tag to see the run-to-run spread)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for tag in "$@"; do
    for rep in 1; do
        SRF_LIB_TAG=$tag python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eager-baseline --no-extra-legs --no-fp32-mode --kernels-json gpurun_out/var_k.json \
            > gpurun_out/var_b.json 2> gpurun_out/var_err.log || { echo "variant '$tag' FAILED"; tail -5 gpurun_out/var_err.log; continue; }
        python - "$tag" <<'PY'
import json, sys
b = json.loads(open("gpurun_out/var_b.json").read().strip().splitlines()[-1])
k = {r["name"]: r for r in json.load(open("gpurun_out/var_k.json"))}
def us(n):
    r = k.get(n)
    return r["total_ms"] * 1e3 / r["launches"] if r else float("nan")
print("variant %-6r  %.3f ms/step  %7.0f rays/s | fwd_fused %.1f us  bwd_fused %.1f us  fwd_fused/g %.1f  bwd_fused/g %.1f  wgrad batch %.1f us"
      % (sys.argv[1], b["ms_per_step"], b["value"], us("mlp_fwd_fused"), us("mlp_bwd_fused"), us("mlp_fwd_fused/g"), us("mlp_bwd_fused/g"),
         us("gemm_wgrad_fc")))
print("               head (side stream) kernels per step: %.0f us  [wgrad batch/g %.1f us]" % (
    sum(r["total_ms"] * 1e3 for n, r in k.items() if n.endswith("/g")) / 20.0, us("gemm_wgrad_fc/g")))
PY
    done
done
