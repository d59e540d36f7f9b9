"""Aggregate a rocprofv3 kernel-trace CSV by (kernel, grid): avg us x calls, share."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"]
    key = (name[:100], r.get("Grid_Size_X", ""), r.get("Workgroup_Size_X", ""))
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(key, [0, 0, 1 << 62, 0]); a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
for k, (n, t, lo, hi) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{t/n/1e3:9.1f} us x {n:5d} = {t/1e6:8.3f} ms ({100*t/tot:5.1f}%)  [{lo/1e3:7.1f} .. {hi/1e3:7.1f}]  grid {k[1]:>8} wg {k[2]:>4}  {k[0]}")
