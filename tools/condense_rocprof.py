"""Condense a rocprofv3 *_kernel_stats.csv into a short table (kernel names truncated)."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("| kernel | calls | total ms | avg us | % |")
print("|---|---|---|---|---|")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
    name = re.sub(r"at::native::(\(anonymous namespace\)::)?", "torch:", name)[:70]
    print("| %s | %s | %.3f | %.1f | %.2f |" % (name, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                              100 * float(r["TotalDurationNs"]) / tot))
