"""Turn two rocprofv3 counter passes of `bench.py --steps 1 --warmup 0 --decode-tokens 0 --no-cpu-baseline`
(one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE, each --output-format csv) into profiles/rNN_pmc_traffic.json:
HBM-side bytes per launch for every MFMA / attention / cross-entropy kernel, grouped by kernel name and grid size.

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>

FETCH_SIZE is reported in KiB and tallies 16-byte-per-lane streaming reads at half their size on gfx950, so it is
doubled; WRITE_SIZE (KiB) is taken as it is.
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        if not any(k in name for k in ("gemm_kernel", "attn_", "ce_kernel", "ce_row_kernel")):
            continue
        wgs = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
        acc[(name, wgs)].append(float(r["Counter_Value"]))
    return acc


def main(fetch_csv, write_csv, out):
    f = per_kernel(fetch_csv, "FETCH_SIZE")
    w = per_kernel(write_csv, "WRITE_SIZE")
    kernels = []
    fwd_bytes, fwd_n = 0.0, 0
    for key in sorted(f):
        name, wgs = key
        fb = sum(f[key]) / len(f[key]) * 1024 * 2
        wb = sum(w[key]) / len(w[key]) * 1024 if key in w else None
        kernels.append({"kernel": name[:120], "workgroups": wgs, "launches_seen": len(f[key]),
                        "fetch_bytes": round(fb), "write_bytes": None if wb is None else round(wb)})
        if "gemm_kernel<false, false, 0" in name and wb is not None:   # x @ W^T, bf16 out: QKV, gate-up, lm_head fwd
            fwd_bytes += (fb + wb) * len(f[key])
            fwd_n += len(f[key])
    doc = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --steps 1 "
                      "--warmup 0 --decode-tokens 0 --no-cpu-baseline",
           "units": "bytes per launch; FETCH_SIZE (KiB) x 1024 x 2 (gfx950: 16 B/lane streaming reads are tallied at "
                    "half), WRITE_SIZE (KiB) x 1024",
           "kernels": kernels,
           "gemm_fwd_bf16_avg_bytes_per_launch": round(fwd_bytes / fwd_n) if fwd_n else None,
           "gemm_fwd_bf16_launches": fwd_n}
    json.dump(doc, open(out, "w"), indent=1)
    print(out, "kernels", len(kernels), "fwd launches", fwd_n, "avg bytes", doc["gemm_fwd_bf16_avg_bytes_per_launch"])


if __name__ == "__main__":
    main(*sys.argv[1:4])
