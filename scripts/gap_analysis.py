"""per-level kernel time and idle gaps from a rocprofv3 kernel trace csv (dev aid)"""
import csv, sys, glob, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", 0))) for r in rows]
# last full cycle: split on fold_metric_k
idx = [i for i, e in enumerate(ev) if "fold_metric" in e[2]]
a, b = idx[-3] + 1, idx[-2] + 1
cyc = ev[a:b]
wall = cyc[-1][1] - ev[a - 1][1]
busy = sum(e[1] - e[0] for e in cyc)
print(f"cycle: {len(cyc)} kernels, wall {wall/1e3:.1f} us, busy {busy/1e3:.1f} us, gaps {(wall-busy)/1e3:.1f} us")
small = [e for e in cyc if e[1] - e[0] < 60000]
print(f"kernels < 60 us: {len(small)}, busy {sum(e[1]-e[0] for e in small)/1e3:.1f} us")
prev = ev[a - 1][1]
g_small = 0
for e in cyc:
    if e[1] - e[0] < 60000: g_small += e[0] - prev
    prev = e[1]
print(f"gaps before small kernels: {g_small/1e3:.1f} us")
from collections import defaultdict
d = defaultdict(lambda: [0, 0])
for e in cyc:
    k = re.sub(r"\(anonymous namespace\)::", "", e[2])[:70]
    d[k][0] += 1; d[k][1] += e[1] - e[0]
for k, v in sorted(d.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[0]:4d} {v[1]/1e3:9.1f} us  {k}")
