"""Decode-only kernel breakdown. Two modes:

  python tools/profile_decode.py run [streams] [prompt] [new_tokens]
      prefill + decode of the default decoder (run it under `rocprofv3 --kernel-trace --output-format csv`)
  python tools/profile_decode.py report <kernel_trace.csv>
      per-kernel totals of the launches AFTER the last prefill attention kernel, i.e. of the decode steps only
  python tools/profile_decode.py timeline <kernel_trace.csv> [launches]
      the last launches of the run in start order: queue, start (us), duration (us), kernel
"""
import csv
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(streams=256, prompt_len=1792, new_tokens=33):
    sys.path.insert(0, os.path.join(ROOT, "symbolic-music-generation_amd"))
    import torch
    from mgen import MidiLM, default_decoder, aria_style_llama
    dev = torch.device("cuda:0")
    cfg = aria_style_llama() if os.environ.get("MODEL") == "aria" else default_decoder()   # MODEL=aria: the 650M model
    model = MidiLM(cfg, device=dev, seed=0)
    g = torch.Generator().manual_seed(99)
    prompt = torch.randint(3, min(30000, cfg.vocab_size), (streams, prompt_len), generator=g).to(dev)
    out = model.generate(prompt, max_new_tokens=new_tokens, do_sample=True, temperature=0.97, top_k=0, top_p=0.95,
                         seed=1234)
    torch.cuda.synchronize()
    print("generated", tuple(out.shape))


def timeline(path, n=260):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-n:]
    t0 = int(rows[0]["Start_Timestamp"])
    short = {"attn_decode": "ATTN", "gemm_skinny_kernelILb1ELb1": "down", "gemm_skinny_kernelILb1ELb0": "o_proj",
             "gemm_skinny_kernel<true, true>": "down", "gemm_skinny_kernel<true, false>": "o_proj",
             "gemm_skinny": "qkv|gu", "slab_resid": "slabnorm", "sample": "SAMPLE", "embedding": "embed",
             "rmsnorm_fwd": "norm", "gemm_kernel": "lm_head"}
    for r in rows:
        name = next((v for k, v in short.items() if k in r["Kernel_Name"]), r["Kernel_Name"][:30])
        b, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print(f"q{r.get('Queue_Id', '?'):>3} {b / 1e3:9.1f} {(e - b) / 1e3:7.1f}  {name}")


def report(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    last_prefill = max(i for i, r in enumerate(rows) if "attn_fwd_kernel" in r["Kernel_Name"])
    dec = rows[last_prefill + 1:]
    # the prefill's tail (final norm, lm head, first sample) precedes the first decode step: start at the first
    # decode-attention launch's layer, i.e. drop everything before the first embedding launch after the prefill
    first = next(i for i, r in enumerate(dec) if "embedding_fwd" in r["Kernel_Name"])
    dec = dec[first:]
    steps = sum(1 for r in dec if "embedding_fwd" in r["Kernel_Name"])
    tot = collections.defaultdict(lambda: [0, 0])
    for r in dec:
        t = tot[r["Kernel_Name"][:90]]
        t[0] += 1
        t[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    span = int(dec[-1]["End_Timestamp"]) - int(dec[0]["Start_Timestamp"])
    busy = sum(v[1] for v in tot.values())
    # timeline: time with at least one / at least two kernels running (two-part decoding overlaps the parts)
    ev = []
    for r in dec:
        ev.append((int(r["Start_Timestamp"]), 1))
        ev.append((int(r["End_Timestamp"]), -1))
    ev.sort()
    depth, prev, one, two = 0, ev[0][0], 0, 0
    for ts, dlt in ev:
        if depth >= 1:
            one += ts - prev
        if depth >= 2:
            two += ts - prev
        depth += dlt
        prev = ts
    print("(a step here runs from one token-embedding launch to the next: ONE part of the batch when the batch is "
          "decoded in parts; the whole batch advances one token per `parts` such steps)")
    print(f"decode steps {steps}; wall {span / steps / 1e3:.1f} us/step; kernel time {busy / steps / 1e3:.1f} us/step; "
          f"some kernel running {one / steps / 1e3:.1f} us/step (idle {(span - one) / steps / 1e3:.1f}); "
          f">= 2 kernels running {two / steps / 1e3:.1f} us/step")
    for k, (n, ns) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"{ns / steps / 1e3:9.1f} us/step  {n / steps:6.1f} launches/step  {ns / n / 1e3:8.1f} us avg  {k}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(*[int(a) for a in sys.argv[2:]])
    elif sys.argv[1] == "timeline":
        timeline(sys.argv[2], *(int(a) for a in sys.argv[3:4]))
    else:
        report(sys.argv[2])
