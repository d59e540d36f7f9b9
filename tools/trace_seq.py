"""Print the launches of kernels whose name contains argv[2], in time order, from a rocprofv3 kernel-trace CSV: start offset, duration."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
for r in rows[-int(sys.argv[3]) if len(sys.argv) > 3 else 0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:12.1f} us  +{(e - s) / 1e3:8.1f} us  grid {r['Grid_Size_X']:>8}  {r['Kernel_Name'][:70]}")
