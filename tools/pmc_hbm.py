"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM bytes per launch.
usage: pmc_hbm.py <dir_fetch> <dir_write> <out.json>
Correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE/WRITE_SIZE count KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of
wide (16 B/lane) coalesced reads -> doubled.  WRITE_SIZE is taken as is (uncalibrated per the guide)."""
import collections, csv, glob, json, re, sys

def agg(d, counter):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            for key in (k, k + " @grid=" + r["Grid_Size"]):   # per kernel function, and per (function, launch size)
                tot[key] += float(r["Counter_Value"])
                cnt[key] += 1
    return tot, cnt

ft, fc = agg(sys.argv[1], "FETCH_SIZE")
wt, wc = agg(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(ft) | set(wt)):
    if "at::" in k or "rocsolver" in k or "rocclr" in k:
        continue
    n = max(fc.get(k, 0), wc.get(k, 0), 1)
    fetch = ft.get(k, 0.0) * 1024 * 2 / max(fc.get(k, 1), 1)
    write = wt.get(k, 0.0) * 1024 / max(wc.get(k, 1), 1)
    out[k] = {"launches": n, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
              "hbm_bytes_per_launch": fetch + write}
# which kernel sources these counters were taken on: bench.py compares the digest with the tree it runs on (roofline.traffic_fresh)
import os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
meta = {}
try:
    from scenerf_amd import build as _b
    meta["src_digest"] = _b._digest()
except Exception as e:
    meta["src_digest_error"] = repr(e)
try:
    meta["commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=os.path.dirname(os.path.abspath(__file__)), text=True,
                                             stderr=subprocess.DEVNULL).strip()
except Exception:
    meta["commit"] = os.environ.get("SRF_COMMIT", "")       # (the GPU box has no .git: tools/profile_round.sh passes the commit in)
json.dump(dict(out, _meta=meta), open(sys.argv[3], "w"), indent=1)
for k, v in sorted(((k, v) for k, v in out.items() if "@grid" not in k), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
    print("%-60s n=%4d fetch %8.1f MB  write %8.1f MB" % (k[:60], v["launches"], v["fetch_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6))
