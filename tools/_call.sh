PROBE_MASKS=1 SRF_LIB_TAG=cyc timeout 300 python tools/dfeat_probe.py 2>&1 | grep -E "mask|dfeat  M|pc=" | tail -8
timeout 300 python tools/dfeat_probe.py 2>&1 | grep -E "mask|dfeat  M|pc=|level" | tail -8
