"""One training step out of a `rocprofv3 --kernel-trace --output-format csv` run of bench.py: the kernels between two consecutive
adamw_kernel launches (the optimizer ends a step), with start offsets, durations and HSA queue -- profiles/*_step_trace.md.
usage: step_trace.py <dir with *kernel_trace.csv> [which step from the end, default 2]"""
import csv, glob, sys

d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adamw_kernel")]
a, b = ends[-back - 1] + 1, ends[-back] + 1
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
queues = {}
for r in step:
    queues.setdefault(r["Queue_Id"], len(queues) + 1)
print("# one training step (kernels between two adamw_kernel launches); times in us from the step's first kernel; q = HSA queue in order of appearance\n")
busy = {}
lib = 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = queues[r["Queue_Id"]]
    busy[q] = busy.get(q, 0.0) + (e - s) / 1e3
    name = r["Kernel_Name"]
    lib += 0 if name.startswith(("at::", "void at::", "__amd")) else 1
    print("%8.1f %7.1f q%d grid %-9s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, r.get("Grid_Size", r.get("Grid_Size_X", "?")), name[:90]))
span = (max(int(r["End_Timestamp"]) for r in step) - t0) / 1e3
print("\n%d launches in the step, %d of them library kernels; kernel time per queue (us): %s; step span %.1f us under the profiler" % (
    len(step), lib, {k: round(v, 1) for k, v in busy.items()}, span))
