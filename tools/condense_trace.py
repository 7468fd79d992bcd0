"""Per (kernel, grid size) launch statistics out of a `rocprofv3 --kernel-trace --output-format csv` run: the averages bench.py's
`roofline` figures can be recomputed from (a kernel NAME alone mixes launch sizes: the 128-row forward runs at 153,600 rows in the
headline, at 76,800 in the N = 64 leg and at 2 M rows in the inference leg).
usage: condense_trace.py <dir with *kernel_trace.csv> [rows to print, default 40]"""
import collections, csv, glob, re, sys

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    name = re.sub(r"at::native::(\(anonymous namespace\)::)?", "torch:", name)[:64]
    agg[(name, r.get("Grid_Size", r.get("Grid_Size_X", "?")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
print("| kernel | grid (work-items) | launches | total ms | avg us | min us | max us | % of kernel time |")
print("|---|---|---|---|---|---|---|---|")
for (name, grid), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print("| %s | %s | %d | %.3f | %.1f | %.1f | %.1f | %.2f |" % (name, grid, len(v), sum(v) / 1e3, sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))
