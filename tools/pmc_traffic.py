"""HBM traffic per kernel from rocprofv3 counter passes.

    python tools/pmc_traffic.py <out.json> <label>=<fetch.csv>,<write.csv>,<kernel_trace.csv> [<label>=...]

For each label (e.g. train, decode): one pass with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE and one
--kernel-trace pass of the same command (all --output-format csv). Per kernel name and grid size the output holds
the launches seen, average HBM-side bytes fetched and written per launch, the average duration and the resulting
GB/s. FETCH_SIZE is reported in KiB and tallies 16-byte-per-lane streaming reads at half their size on gfx950, so it is
doubled; WRITE_SIZE (KiB) is taken as it is. Kernels of the library only (names from the anonymous namespace).
"""
import collections
import csv
import json
import sys


def short(name):
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    if name.startswith("_ZN12_GLOBAL__N_1"):
        # a mangled name rocprofv3 left alone: keep the identifier part
        rest = name[len("_ZN12_GLOBAL__N_1"):]
        n = int("".join(c for c in rest[:3] if c.isdigit()))
        rest = rest[len(str(n)):]
        return rest[:n] + "<" + rest[n:n + 40] + ">"
    cut = name.find("(")
    return name[:cut] if cut > 0 else name[:100]


def ours(name):
    return "anonymous namespace" in name or "_GLOBAL__N_1" in name


def counters(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or not ours(r["Kernel_Name"]):
            continue
        wgs = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
        acc[(short(r["Kernel_Name"]), wgs)].append(float(r["Counter_Value"]))
    return acc


def durations(path):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if not ours(r["Kernel_Name"]):
            continue
        wgs = (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])) // max(
            1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
        acc[(short(r["Kernel_Name"]), wgs)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return acc


def decode_step_launches(path):
    """(kernel, workgroups) -> launches per decode step, counted over the launches after the last prefill attention
    kernel, from the first token-embedding launch on (a decode step starts with one)."""
    rows = [r for r in csv.DictReader(open(path)) if ours(r["Kernel_Name"])]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    pre = [i for i, r in enumerate(rows) if "attn_fwd_kernel" in r["Kernel_Name"]]
    if not pre:
        return {}
    dec = rows[pre[-1] + 1:]
    first = next((i for i, r in enumerate(dec) if "embedding_fwd" in r["Kernel_Name"]), None)
    if first is None:
        return {}
    dec = dec[first:]
    steps = sum(1 for r in dec if "embedding_fwd" in r["Kernel_Name"])
    cnt = collections.Counter()
    for r in dec:
        wgs = (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])) // max(
            1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
        cnt[(short(r["Kernel_Name"]), wgs)] += 1
    return {k: v / steps for k, v in cnt.items()}


def section(fetch_csv, write_csv, trace_csv, decode=False):
    f, w, d = counters(fetch_csv, "FETCH_SIZE"), counters(write_csv, "WRITE_SIZE"), durations(trace_csv)
    per_step = decode_step_launches(trace_csv) if decode else {}
    rows = []
    for key in sorted(f):
        fb = sum(f[key]) / len(f[key]) * 1024 * 2
        wb = sum(w[key]) / len(w[key]) * 1024 if key in w else 0.0
        us = sum(d[key]) / len(d[key]) / 1e3 if key in d else None
        rows.append({"kernel": key[0], "workgroups": key[1], "launches_seen": len(f[key]), "fetch_bytes": round(fb),
                     "write_bytes": round(wb), "avg_us": None if us is None else round(us, 2),
                     "GBps": None if not us else round((fb + wb) / us / 1e3, 1)})
        if key in per_step:
            rows[-1]["launches_per_decode_step"] = round(per_step[key], 2)
    return rows


def main(out, specs):
    doc = {"units": "bytes per launch; FETCH_SIZE (KiB) x 1024 x 2 (gfx950: 16 B/lane streaming reads are tallied at half), "
                    "WRITE_SIZE (KiB) x 1024; avg_us from a --kernel-trace pass of the same command; GBps = (fetch + write) / avg_us",
           "passes": "separate rocprofv3 runs per counter (--pmc FETCH_SIZE, --pmc WRITE_SIZE) and for the kernel trace"}
    streams = None
    for spec in specs:
        if spec.startswith("decode_streams="):   # streams of the decode run: a batch may be decoded in several parts
            streams = int(spec.split("=")[1])
    for spec in specs:
        if spec.startswith("decode_streams="):
            continue
        label, files = spec.split("=")
        doc[label] = section(*files.split(","), decode=label.startswith("decode"))
        if label.startswith("decode"):
            part_bytes = sum((r["fetch_bytes"] + r["write_bytes"]) * r["launches_per_decode_step"]
                             for r in doc[label] if "launches_per_decode_step" in r)
            # a "step" above runs from one token-embedding launch to the next, i.e. over ONE part of the batch; the
            # embedding kernel takes 4 rows per workgroup, which tells how many streams a part holds
            parts = 1
            emb = [r for r in doc[label] if r["kernel"].startswith("embedding_fwd") and "launches_per_decode_step" in r]
            if streams and emb:
                parts = max(1, round(streams / (4 * emb[0]["workgroups"])))
            doc[label + "_parts"] = parts
            doc[label + "_step_hbm_bytes"] = round(part_bytes * parts)
    # Sections named decode@<context>: HBM bytes per step are linear in the context (weights + logits + KV rows read),
    # so a least-squares line through them gives the run average of a decode run over contexts 1 .. 4095 (the
    # literal 256-stream x 4096-token benchmark run) at its mean context, 2048.
    # (a section decode@P is a run from a P-token prompt over 8 decode steps: contexts P + 1 .. P + 8, mean P + 4.5)
    pts = [(int(k.split("@")[1].split("_")[0]) + 4.5, v) for k, v in doc.items()
           if k.startswith("decode@") and k.endswith("_step_hbm_bytes")]
    if len(pts) >= 2:
        n = len(pts)
        mx, my = sum(p[0] for p in pts) / n, sum(p[1] for p in pts) / n
        b = sum((p[0] - mx) * (p[1] - my) for p in pts) / sum((p[0] - mx) ** 2 for p in pts)
        a = my - b * mx
        doc["decode_bytes_per_step_fit"] = {"constant": round(a), "per_context_token": round(b, 1),
                                            "points": sorted(pts)}
        doc["decode_literal_run_avg_step_hbm_bytes"] = round(a + b * 2048.0)
    # the kernel bench.py's roofline block times: the plain instance of the register-panel GEMM (K = 512 products: QKV,
    # lm_head logits, o_proj / down_proj input gradients)
    fwd = [r for r in doc.get("train", []) if r["kernel"].startswith("gemm_rs_kernel<false")]
    if fwd:
        n = sum(r["launches_seen"] for r in fwd)
        doc["roofline_kernel"] = "gemm_rs_kernel<false, 0>"
        doc["roofline_kernel_avg_bytes_per_launch"] = round(
            sum((r["fetch_bytes"] + r["write_bytes"]) * r["launches_seen"] for r in fwd) / n)
        doc["roofline_kernel_avg_us"] = round(sum(r["avg_us"] * r["launches_seen"] for r in fwd) / n, 2)
        doc["roofline_kernel_launches"] = n
    json.dump(doc, open(out, "w"), indent=1)
    print(out, {k: len(v) for k, v in doc.items() if isinstance(v, list)})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
