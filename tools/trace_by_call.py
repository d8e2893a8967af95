"""Per-call-site durations of one kernel inside the training step, from a rocprofv3 kernel trace (csv):
    python tools/trace_by_call.py <kernel_trace.csv> <kernel name substring> <launches per step>
Launch i of a step (in start order) is one call site; prints its mean duration over the steps in the trace."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = int(sys.argv[3])
n = len(rows) // per
skip = len(rows) - n * per
rows = rows[skip:]
for i in range(per):
    d = [int(rows[s * per + i]["End_Timestamp"]) - int(rows[s * per + i]["Start_Timestamp"]) for s in range(n)]
    d = d[1:] if len(d) > 2 else d    # first step: cold caches, lazy allocations
    print(f"launch {i:3d}: {sum(d) / len(d) / 1e3:8.1f} us   (min {min(d) / 1e3:8.1f}, {len(d)} steps)")
