"""From a rocprofv3 kernel trace (csv): for every launch of kernel A (substring), the launches of other kernels that
overlap it in time, with both durations. python tools/lab/overlap_report.py <trace.csv> <substring of A> [n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
a_idx = [i for i, x in enumerate(iv) if key in x[2]]
for i in a_idx[-n:]:
    b, e, name = iv[i]
    print(f"{name[:50]:50s} {(e - b) / 1e3:8.1f} us")
    for j in range(max(0, i - 40), min(len(iv), i + 40)):
        if j == i:
            continue
        b2, e2, n2 = iv[j]
        ov = min(e, e2) - max(b, b2)
        if ov > 0:
            print(f"    overlaps {ov / 1e3:7.1f} us of {(e2 - b2) / 1e3:7.1f} us  {n2[:70]}")
