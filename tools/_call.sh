for t in A B C D; do
echo "variant $t"
SRF_LIB_TAG=$t PROBE_MASK=0 PROBE_LEAN=1 timeout 100 python tools/fused_probe.py 153600 20 fused 2>&1 | tail -2
done
