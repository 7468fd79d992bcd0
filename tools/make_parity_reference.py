"""tests/golden/parity_full_measured.json from the parity_full_*.json files a GPU run of tests/test_gpu_parity_full.py leaves in
gpurun_out/ (or profiles/r*_parity_full_*): per case the measured worst errors at matched discrete choices -- per output (max abs for the
absolute-gated ones, max rel otherwise), per gradient group (worst tensor's relative L2), the proxy loss, the gaussian head's offsets.
The test refuses results more than 1.25 x worse than these.
usage: make_parity_reference.py [glob, default gpurun_out/parity_full_*.json]"""
import glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ABS_KEYS = ("color", "alphas", "weights")
GATED = ("depth", "color", "gaussian_means", "gaussian_stds", "depth_volumes", "alphas", "weights", "densities")
pat = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_full_*.json")
out = {}
for f in sorted(glob.glob(pat)):
    d = json.load(open(f))
    key = "%s_%s%s" % (d["case"], d["precision"], "" if d.get("maps", "chw") == "chw" else "_" + d["maps"])
    m = d["matched"]
    groups = {}
    for nm, v in m["grad"].items():
        if "rel_l2" in v:
            g = nm.split(".")[0] + "."
            groups[g] = max(groups.get(g, 0.0), v["rel_l2"])
    out[key] = {"out": {k: (m["out"][k]["max_abs"] if k in ABS_KEYS else m["out"][k]["max_rel"]) for k in GATED if k in m["out"]},
                "grad": groups, "loss": m["loss"]["rel"], "head": d["head_offsets"]["rel_l2"], "source": os.path.basename(f)}
dst = os.path.join(ROOT, "tests", "golden", "parity_full_measured.json")
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print("wrote %s: %d cases" % (dst, len(out)))
