"""The launch sequence of the LAST whole predict() step in a rocprofv3 kernel-trace CSV (+ memory-copy rows if present): start offset,
duration, gap to the previous launch's end, name.  A step starts at the pack_scan launch."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nm = lambda r: re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0][:60]
first = sys.argv[2] if len(sys.argv) > 2 else "pack_scan_kernel"
st = [i for i, r in enumerate(rows) if nm(r).startswith(first)]
a, b = st[-3], st[-2]
for i in range(len(st) - 2, 0, -1):          # the last segment that is a whole step (ends in the post kernel), not an encoder-only forward
    if any(nm(r).startswith("predict_post_kernel") for r in rows[st[i - 1]:st[i]]):
        a, b = st[i - 1], st[i]
        break
t0 = int(rows[a]["Start_Timestamp"]); prev = None
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%10.1f us  +%7.1f us  gap %6.1f  grid %8s wg %4s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, 0 if prev is None else (s - prev) / 1e3,
                                                                  r["Grid_Size_X"], r["Workgroup_Size_X"], nm(r)))
    prev = e
