"""Average every counter of rocprofv3 --pmc CSVs per kernel (name, workgroups): `python tools/pmc_sum.py DIR [filter]`
walks DIR for *counter_collection.csv files."""
import collections, csv, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for root, _, files in os.walk(sys.argv[1]):
    for f in files:
        if f.endswith("counter_collection.csv"):
            for r in csv.DictReader(open(os.path.join(root, f))):
                name = r["Kernel_Name"]
                if flt and flt not in name:
                    continue
                key = (name[:70], int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in sorted(acc):
    print(key)
    for c, v in sorted(acc[key].items()):
        print(f"    {c:32s} {sum(v)/len(v):18.1f}   (n={len(v)})")
