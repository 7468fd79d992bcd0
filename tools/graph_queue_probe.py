"""Which HSA queue does a replayed hipGraph put each captured kernel on, and what does a dependency that crosses queues cost?
Small capture topologies (a main stream M, side streams S / T, fork and join by events) whose nodes are elementwise kernels of distinct
sizes; run under `rocprofv3 --kernel-trace --output-format csv`, then `graph_queue_probe.py --parse <dir>` prints, per topology, each
node's queue and the idle time between it and the latest of its predecessors.  Written to find the capture ORDER that keeps a step's
critical chain on one queue (scenerf_amd.graph: in the replayed training step every cross-queue edge of the chain shows 10-19 us idle,
consecutive kernels of one queue none).
usage: rocprofv3 --kernel-trace --output-format csv -d OUT -o p -- python tools/graph_queue_probe.py ; python tools/graph_queue_probe.py --parse OUT"""
import csv, glob, json, sys

BASE, STEP = 1 << 22, 1 << 14          # node k works on BASE + k * STEP floats: the grid size identifies it in the trace
MARK = 1 << 18                         # an eager kernel of MARK + t * 256 floats separates topology t's replays

# a topology = list of (stream, node name, [streams waited for right before the launch])
TOPOS = {
    "fork_side_first": [("M", "A", []), ("S", "X", ["M"]), ("M", "B", []), ("M", "C", ["S"])],
    "fork_main_first": [("M", "A", []), ("M", "B", []), ("S", "X", ["M@A"]), ("M", "C", ["S"])],
    "two_forks_side_first": [("M", "A", []), ("S", "X1", ["M"]), ("M", "B", []), ("M", "C", ["S"]), ("S", "X2", ["M"]), ("M", "D", []), ("M", "E", ["S"])],
    "two_forks_main_first": [("M", "A", []), ("M", "B", []), ("S", "X1", ["M@A"]), ("M", "C", ["S"]), ("M", "D", []), ("S", "X2", ["M@C"]), ("M", "E", ["S"])],
    "chain_only": [("M", "A", []), ("M", "B", []), ("M", "C", [])],
    "side_root": [("S", "X", ["M"]), ("M", "A", []), ("M", "B", ["S"])],            # the side stream's node captured before any main node
    "main_root_then_side": [("M", "A", []), ("S", "X", ["M@root"]), ("M", "B", ["S"])],  # X depends on nothing but the capture origin
    "three_streams": [("M", "A", []), ("S", "X", ["M"]), ("T", "Y", ["M"]), ("M", "B", []), ("M", "C", ["S", "T"])],
    # joins whose side branch finished long before (what a satisfied cross-queue dependency costs)
    "early_join": [("M", "A", []), ("M", "B1", []), ("S", "X", ["M@A"]), ("M", "B2", []), ("M", "B3", []), ("M", "B4", []), ("M", "C", ["S"]), ("M", "D", [])],
    "early_join_side_root": [("M", "A", []), ("S", "X", ["M@root"]), ("M", "B1", []), ("M", "B2", []), ("M", "B3", []), ("M", "C", ["S"]), ("M", "D", [])],
    # the step's front as planned: chain root first, packs as a second root, fill forked behind the head's forward AFTER the sampler's launch
    "planned_front": [("M", "RS", []), ("M", "EH", []), ("M", "GH", []), ("S", "P1", ["M@root"]), ("S", "P2", []), ("S", "P3", []), ("S", "P4", []),
                      ("M", "F0", ["S@P2"]), ("M", "SS", []), ("S", "FILL", ["M@F0"]), ("M", "EN", []), ("M", "GA", []), ("M", "W0", ["S@P4"]), ("M", "TAIL", []),
                      ("M", "L4", []), ("S", "L2", ["M@TAIL"]), ("S", "H1", []), ("M", "W1", []), ("M", "DF", ["S@FILL"]), ("T", "WG", ["M@W1"]), ("M", "OPT", ["S", "T"])],
    # two side branches that land on the same queue: in which order does it run them?  H (two short kernels) hangs off B, W off D
    "order_h_then_w": [("M", "A", []), ("M", "B", []), ("M", "C", []), ("S", "H1", ["M@B"]), ("S", "H2", []), ("M", "D", []), ("M", "D2", []), ("M", "D3", []), ("M", "E", []),
                       ("T", "W", ["M@D3"]), ("M", "F", ["S", "T"])],
    "order_w_then_h": [("M", "A", []), ("M", "B", []), ("M", "C", []), ("M", "D", []), ("M", "D2", []), ("M", "D3", []), ("M", "E", []), ("T", "W", ["M@D3"]),
                       ("S", "H1", ["M@B"]), ("S", "H2", []), ("M", "F", ["S", "T"])],
    # ... and a third successor of one node: does it get a third queue?
    "third_child": [("M", "A", []), ("M", "B", []), ("M", "C", []), ("S", "X", ["M@B"]), ("T", "H1", ["M@B"]), ("T", "H2", []), ("M", "D", []), ("M", "D2", []), ("M", "D3", []),
                    ("M", "E", []), ("U", "W", ["M@D3"]), ("M", "F", ["S", "T", "U"])],
    "three_streams_main_first": [("M", "A", []), ("M", "B", []), ("S", "X", ["M@A"]), ("T", "Y", ["M@A"]), ("M", "C", ["S", "T"])],
}


def run():
    import torch
    dev = torch.device("cuda:0")
    streams = {"M": torch.cuda.Stream(dev), "S": torch.cuda.Stream(dev), "T": torch.cuda.Stream(dev), "U": torch.cuda.Stream(dev)}
    graphs = {}
    bufs = {}
    for ti, (tname, topo) in enumerate(TOPOS.items()):
        for k, (_, node, _) in enumerate(topo):
            bufs[(tname, node)] = torch.zeros(BASE + (ti * 32 + k) * STEP, device=dev)
    marks = [torch.zeros(MARK + t * 256, device=dev) for t in range(len(TOPOS))]
    torch.cuda.synchronize()
    for tname, topo in TOPOS.items():
        g = torch.cuda.CUDAGraph()
        M = streams["M"]
        with torch.cuda.stream(M):
            g.capture_begin()
            events = {"root": M.record_event()}
            for st, node, waits in topo:
                s = streams[st]
                for w in waits:
                    if "@" in w:
                        s.wait_event(events[w.split("@")[1]])      # "S@P2" / "M@A": the event recorded behind that node ("root": capture origin)
                    else:
                        s.wait_stream(streams[w])
                with torch.cuda.stream(s):
                    bufs[(tname, node)].add_(1.0)
                    events[node] = s.record_event()
            for st in ("S", "T", "U"):      # every side stream joins the origin before the capture ends
                M.wait_stream(streams[st])
            g.capture_end()
        graphs[tname] = g
    torch.cuda.synchronize()
    for ti, (tname, g) in enumerate(graphs.items()):
        for _ in range(4):
            marks[ti].add_(1.0)
            g.replay()
        torch.cuda.synchronize()
    print("done")


def parse(d):
    f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))

    def numel(r):   # elementwise kernels: 4 elements per thread in torch's vectorised form; identify by the closest match instead
        return int(r.get("Grid_Size", r.get("Grid_Size_X", 0)))
    grids = {}
    for ti, (tname, topo) in enumerate(TOPOS.items()):
        for k, (_, node, _) in enumerate(topo):
            grids[(tname, node)] = BASE + (ti * 32 + k) * STEP
    # the grid size of torch's vectorised add_ is numel / 4 rounded up to a block: find the factor from the marks
    by_grid = {}
    for r in rows:
        by_grid.setdefault(numel(r), []).append(r)
    # one vt (elements per thread) for every node: the one under which the most nodes are found
    def found(vt):
        return sum(1 for n in grids.values() if any(0 <= g * vt - n < 256 * vt for g in by_grid))
    VT = max((1, 2, 4, 8, 16), key=found)
    for ti, (tname, topo) in enumerate(TOPOS.items()):
        print("\n## %s   %s" % (tname, "  ".join("%s:%s%s" % (st, n, "<-" + ",".join(w) if w else "") for st, n, w in topo)))
        inst = {}
        for st, node, _ in topo:
            n = grids[(tname, node)]
            cand = [g for g in by_grid if 0 <= g * VT - n < 256 * VT]   # grid = ceil(n / (256 vt)) * 256 threads
            if not cand:
                print("   %s: not found in the trace" % node)
                continue
            inst[node] = by_grid[cand[0]][-4:]     # the replays (the eager warm-up, if any, comes first)
        qn = {}
        for rep in range(4):
            line = []
            for st, node, waits in topo:
                if node not in inst or len(inst[node]) <= rep:
                    continue
                r = inst[node][rep]
                q = qn.setdefault(r["Queue_Id"], "q%d" % (len(qn) + 1))
                s = int(r["Start_Timestamp"])
                # predecessors: the previous node of the same stream and the last node of every waited stream / named event
                preds = []
                idx = [n for _, n, _ in topo].index(node)
                for j in range(idx - 1, -1, -1):
                    if topo[j][0] == st:
                        preds.append(topo[j][1]); break
                for w in waits:
                    if "@" in w:
                        if w.split("@")[1] != "root":
                            preds.append(w.split("@")[1])
                    else:
                        for j in range(idx - 1, -1, -1):
                            if topo[j][0] == w:
                                preds.append(topo[j][1]); break
                ends = [int(inst[p][rep]["End_Timestamp"]) for p in preds if p in inst and len(inst[p]) > rep]
                gap = (s - max(ends)) / 1e3 if ends else float("nan")
                line.append("%s %s dur %.1f gap %.1f" % (node, q, (int(r["End_Timestamp"]) - s) / 1e3, gap))
            print("   replay %d: %s" % (rep, " | ".join(line)))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--parse":
        parse(sys.argv[2])
    else:
        run()
