"""Headline benchmark: training tokens/s of the default decoder (bf16-mixed) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one full optimizer step of the reference's training loop (src/train.py:63-72) on one synthetic batch per
GPU: zero_grad, forward, backward (+ overlapped RCCL gradient mean for N > 1), clip to max-norm 1.0, AdamW, LR
schedule. Workload = BASELINE.json configs[1]: default decoder (14 layers, d 512, 8 heads x 128, I 2048, V 151936,
tied), batch 32 x seq 2048 per GPU, token ids uniform in [3, 30000), labels = ids. Weak scaling: the per-GPU batch is
fixed. Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "symbolic-music-generation_amd"))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak of MI355X
PEAK_HBM_GBS = 8000.0


def train_flops_per_token(cfg, S):
    """Algorithmic training FLOPs per token: 3 x forward; forward = 2*(block params + d*V) + L*2*S*(H*Dh)
    (causal attention counted at half the square, tied head counted once as a GEMM, embedding gather excluded)."""
    d, I, H, Dh, KVH, L, V = (cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.head_dim,
                              cfg.num_key_value_heads, cfg.num_hidden_layers, cfg.vocab_size)
    mlp = 2 if getattr(cfg, "arch", "llama") == "gpt2" else 3      # c_fc + c_proj, or gate + up + down
    blocks = L * (d * (H + 2 * KVH) * Dh + H * Dh * d + mlp * d * I)
    fwd = 2 * (blocks + d * V) + L * 2 * S * (H * Dh)
    return 3 * fwd


def cpu_baseline(cfg, S, seconds_budget=25.0):
    """The reference's CPU path restated (oracle/ref_cpu.py, fp32) timed on the host cores: one training step
    (fwd + bwd + clip + AdamW) on a bounded sample of the workload: ONE sequence of S tokens (1/32 of the GPU batch)."""
    from oracle import ref_cpu as R
    ocfg = R.LMConfig(**{k: getattr(cfg, k) for k in R.LMConfig.__dataclass_fields__})
    cores = torch.get_num_threads()
    p = R.init_params(ocfg, seed=0)
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(x) for k, x in p.items()}
    g = torch.Generator().manual_seed(1234)
    ids = torch.randint(3, 30000, (1, S), generator=g)
    times = []
    t_start = time.time()
    for it in range(3):
        t0 = time.time()
        loss, grads = R.loss_and_grads(p, ocfg, ids, None, ids, "fp32")
        _, coef = R.clip_coef(list(grads.values()), 1.0)
        for k in p:
            R.adamw_step(p[k], grads[k] * coef, m[k], v[k], it + 1, 5e-5)
        times.append(time.time() - t0)
        if time.time() - t_start > seconds_budget:
            break
    best = min(times)
    return {"value": S / best, "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"1 sequence x {S} tokens (1/32 of the per-GPU batch), fp32, full step incl. clip+AdamW, "
                      f"best of {len(times)} runs, {best:.2f} s/step"}


PMC_TRAFFIC = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
PMC_DECODE_CONTEXTS = os.path.join(ROOT, "profiles", "r03_pmc_decode_contexts.json")


def pmc_decode_traffic(streams, prompt_len, new_tokens):
    """HBM bytes per decode step (every kernel of a step, fetch + write) from the committed counter passes. For 256
    streams at contexts 1792.. : the "decode" section of profiles/r03_pmc_traffic.json. For the literal run (one-token
    prompt to 4096 tokens): the run average from the line through the passes at three contexts
    (profiles/r03_pmc_decode_contexts.json; bytes per step are linear in the context). None for any other shape."""
    if streams != 256:
        return None
    if prompt_len == 1792 and os.path.exists(PMC_TRAFFIC):
        with open(PMC_TRAFFIC) as fh:
            return json.load(fh).get("decode_step_hbm_bytes")
    if prompt_len == 1 and new_tokens == 4095 and os.path.exists(PMC_DECODE_CONTEXTS):
        with open(PMC_DECODE_CONTEXTS) as fh:
            return json.load(fh).get("decode_literal_run_avg_step_hbm_bytes")
    return None


def bench_decode(model, cfg, streams, prompt_len, new_tokens, dev):
    """Batched sampling (BASELINE.json configs[3]: 256 streams, contexts up to 4096): `streams` concurrent KV-cache
    decode streams on a random prompt of `prompt_len` tokens, temperature 0.97, top-p 0.95, no top-k, EOS disabled,
    Philox noise seed 1234. The decode rate is (time of prefill + N steps) - (time of prefill + 1 step) over
    N - 1 steps; algorithmic bytes per step = bf16 weights + the [streams, V] bf16 logits + the KV cache of the
    context, summed over the run's growing contexts. Reported per GPU, with a roofline block (bound: hbm)."""
    g = torch.Generator().manual_seed(99)
    prompt = torch.randint(3, min(30000, cfg.vocab_size), (streams, prompt_len), generator=g).to(dev)
    kw = dict(do_sample=True, temperature=0.97, top_k=0, top_p=0.95, seed=1234)
    # a serving process sizes its KV cache once: both timed runs below use it (same row capacity), and neither pays the
    # allocation and first-touch page faults of 60 GB of fresh device memory inside its first step
    model.reserve_kv_cache(streams, prompt_len + new_tokens)
    model.generate(prompt, max_new_tokens=4, **kw)  # warm-up

    def run(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        o = model.generate(prompt, max_new_tokens=n, **kw)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, o

    t1, _ = run(1)
    tn, out = run(new_tokens)
    steps = new_tokens - 1
    dt = tn - t1
    L, KVH, Dh = cfg.num_hidden_layers, cfg.num_key_value_heads, cfg.head_dim
    kv_tok = 2 * L * KVH * Dh * 2
    weights = 2 * (sum(p.numel() for p in model.parameters()))
    ctx_sum = sum(range(prompt_len + 1, prompt_len + steps + 1))  # keys read by the decode steps
    bytes_total = steps * (weights + streams * cfg.vocab_size * 2) + streams * ctx_sum * kv_tok
    gbs = bytes_total / dt / 1e9
    model.release_kv_cache()
    return {"streams": streams, "prompt_len": prompt_len, "new_tokens": new_tokens,
            "tokens_per_s_per_gpu": round(streams * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 3),
            "prefill_tokens_per_s": round(streams * prompt_len / t1, 1),
            "algorithmic_GBps": round(gbs, 1), "hbm_frac": round(gbs / PEAK_HBM_GBS, 4),
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(gbs / PEAK_HBM_GBS, 4),
                         "algorithmic_bytes": round(bytes_total / steps),
                         "traffic": pmc_decode_traffic(streams, prompt_len, new_tokens),
                         "traffic_unit": "HBM bytes per decode step, all kernels of the step (PMC FETCH_SIZE + WRITE_SIZE, "
                                         "profiles/r03_pmc_traffic.json, r03_pmc_decode_contexts.json); "
                                         "algorithmic_bytes = run average per step"},
            "sampling": "T=0.97 top_p=0.95 top_k=off philox(seed 1234)", "out_shape": list(out.shape)}


def nccl_version():
    try:
        return ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:   # a report field must not cost the run its result line
        return f"unknown ({type(e).__name__})"


def launch_ranks_if_needed(args):
    """`python bench.py --gpus N` with N > 1 and no rendezvous in the environment: this process (which has not touched
    the GPU) starts N fresh rank processes through torch.distributed.run with the same arguments, relays their
    output (rank 0 prints the one JSON line) and exits with their return code. Under a launcher (WORLD_SIZE set) the
    flag must agree with it. Never returns in the parent case."""
    import socket
    import subprocess
    rehearse = os.environ.get("MGEN_REHEARSE_ON_ONE_GPU") == "1"
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" in os.environ:
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.gpus == 1 or "WORLD_SIZE" in os.environ:
        return   # (under a launcher the ranks' device visibility is the launcher's business: it may show each rank one GPU)
    have = torch.cuda.device_count()     # counts devices without creating a GPU context
    if args.gpus > have and not rehearse:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible "
                         "(MGEN_REHEARSE_ON_ONE_GPU=1 puts every rank on device 0 for a control-flow rehearsal)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs between processes here
    env.setdefault("OMP_NUM_THREADS", "4")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--decode-streams", type=int, default=256)
    ap.add_argument("--decode-tokens", type=int, default=256, help="0 disables the decode measurement")
    ap.add_argument("--decode-full", type=int, default=4096,
                    help="total length of the configs[3] decode run (one-token prompt, decode-full - 1 new tokens)")
    ap.add_argument("--decode-prompt", type=int, default=1792,
                    help="prompt length of the decode measurement (contexts run from here to +decode-tokens)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--model", choices=["default", "aria", "gpt2"], default="default",
                    help="aria: the ~650M Llama-architecture stand-in of the other benchmark configurations "
                         "(e.g. --batch 8 --seq 4096, --batch 1 --seq 16384); gpt2: the GPT-2 variant with its "
                         "dropout of 0.1 (--pdrop); decode and CPU legs are skipped for both")
    ap.add_argument("--pdrop", type=float, default=0.1, help="dropout probability of --model gpt2")
    args = ap.parse_args()
    launch_ranks_if_needed(args)

    from mgen import MidiLM, default_decoder, aria_style_llama, ops
    from mgen.config import gpt2_decoder
    from mgen.gpt2 import MidiGPT2LM
    from mgen.trainer import Trainer, init_distributed_from_env

    if os.environ.get("MGEN_BENCH_WATCHDOG"):   # debugging aid: dump every thread's stack after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["MGEN_BENCH_WATCHDOG"]), exit=True)
    rank, local, world = init_distributed_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    dev = torch.device("cuda", local)
    B, S = args.batch, args.seq
    if args.model == "aria":
        cfg = aria_style_llama()
        cfg.max_position_embeddings = max(cfg.max_position_embeddings, S)
        args.decode_tokens, args.no_cpu_baseline = 0, True
        wl = "Aria-style Llama stand-in (L16 d1536 H24xDh64 I6144 V17727 untied)"
    elif args.model == "gpt2":
        cfg = gpt2_decoder(pdrop=args.pdrop)
        cfg.max_position_embeddings = max(cfg.max_position_embeddings, S)
        args.decode_tokens, args.no_cpu_baseline = 0, True
        wl = f"GPT-2 variant (L6 d512 H8xDh64 I2048 V30000 tied, dropout {args.pdrop})"
    else:
        cfg = default_decoder()
        if os.environ.get("MGEN_LM_CHUNK"):   # lm_head chunk size in tokens (measurements)
            cfg.lm_head_chunk_tokens = int(os.environ["MGEN_LM_CHUNK"])
        wl = "default decoder (MidiQwen shape: L14 d512 H8xDh128 I2048 V151936 tied)"
    model = (MidiGPT2LM if args.model == "gpt2" else MidiLM)(cfg, device=dev, seed=0)
    trainer = Trainer(model, lr=5e-5, num_epochs=12, steps_per_epoch=1000, warmup=1000, max_grad_norm=1.0)
    g = torch.Generator().manual_seed(1234 + rank)
    ids = torch.randint(3, min(30000, cfg.vocab_size), (B, S), generator=g).to(dev)
    batch = {"input_ids": ids, "labels": ids}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = trainer.training_step(batch)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.training_step(batch)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    ms_per_step = dt / args.steps * 1e3
    tokens_per_s = world * B * S * args.steps / dt

    # ---- roofline of the dominant kernel: the bf16 MFMA GEMM for x @ W^T products with a 512-long reduction and a
    # bf16 output (QKV, lm_head logits, o_proj / down_proj input gradients through the transposed weight copies).
    # One extra step with a HIP event pair around every launch of that kernel on its stream.
    roofline = None
    # every rank runs the extra step (it contains collectives for N > 1); rank 0 reports
    recs = []
    orig = ops.gemm

    def timed_gemm(a, b, c, M, N, K, lda, ldb, ldc, trans_a=False, trans_b=False, out_mode=ops.OUT_BF16, *rest,
                   **kw):
        if not trans_a and not trans_b and out_mode == ops.OUT_BF16:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig(a, b, c, M, N, K, lda, ldb, ldc, trans_a, trans_b, out_mode, *rest, **kw)
            e1.record()
            recs.append((2.0 * M * N * K, e0, e1, 2.0 * (M * K + N * K + M * N)))
        else:
            orig(a, b, c, M, N, K, lda, ldb, ldc, trans_a, trans_b, out_mode, *rest, **kw)

    ops.gemm = timed_gemm
    # (the instrumented step issues the lm_head's launches from Python - the same kernels in the same order as the
    # library-side chunk loop of the timed steps - so that the event pairs see them)
    os.environ["MGEN_LMHEAD_C"] = "0"
    try:
        trainer.training_step(batch)
    finally:
        os.environ.pop("MGEN_LMHEAD_C", None)
    torch.cuda.synchronize()
    ops.gemm = orig
    fl = sum(r[0] for r in recs)
    ms = sum(r[1].elapsed_time(r[2]) for r in recs)
    achieved = fl / (ms * 1e-3) / 1e12
    # HBM traffic per launch of this kernel cannot be counted from inside the process: it comes from the committed
    # PMC passes of this same command (profiles/r03_pmc_traffic.json: separate FETCH_SIZE and WRITE_SIZE runs,
    # FETCH doubled per the gfx950 note), valid for the default workload only
    traffic = None
    pmc = PMC_TRAFFIC
    if (B, S) == (32, 2048) and args.model == "default" and os.path.exists(pmc):
        with open(pmc) as fh:
            traffic = json.load(fh).get("roofline_kernel_avg_bytes_per_launch")
    # (the other --model choices have no 512-long reductions: their x @ W^T products run on the tile-loop kernel)
    kname = ("gemm_rs_kernel<false, 0> (bf16 products with a 512-long reduction: QKV, lm_head logits, "
             "o_proj / down_proj input gradients)") if cfg.hidden_size == 512 and args.model == "default" else \
        "x @ W^T products with a bf16 output (all shapes of this model; kernel chosen per shape)"
    roofline = {"bound": "mfma", "kernel": kname,
                "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                "traffic_unit": "HBM bytes per launch (PMC, profiles/r03_pmc_traffic.json); algorithmic: "
                                "2*(M*K + N*K + M*N) bytes",
                "algorithmic_bytes_per_launch": round(sum(r[3] for r in recs) / len(recs)),
                "launches": len(recs), "avg_launch_ms": round(ms / len(recs), 4),
                "flops_per_launch_avg": fl / len(recs)}

    ranks_seen = 1
    if world > 1:
        # every rank contributes a 1 through the same collective path the gradients use: the sum is the number of
        # ranks the communicator really joined
        one = torch.ones(1, dtype=torch.float32, device=dev)
        if trainer.reducer._comm is not None:
            trainer.reducer._comm.allreduce_(one, average=False)
            trainer.reducer._comm.wait()
        elif dist.get_backend() == "nccl":
            dist.all_reduce(one, op=dist.ReduceOp.SUM)
        else:
            host = one.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            one = host
        ranks_seen = int(round(float(one)))
        dist.barrier()
    if rank == 0:
        ftok = train_flops_per_token(cfg, S)
        out = {
            "metric": "train tokens/s", "value": round(tokens_per_s, 1), "unit": "tokens/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{wl}, bf16-mixed full train step, per-GPU batch {B} x seq {S}",
                       "global_batch": B * world, "seq_len": S, "parallelism": f"dp{world}"},
            "final_loss": round(float(loss), 4),
            "train_flops_per_token": ftok,
            "model_tflops_per_gpu": round(tokens_per_s / world * ftok / 1e12, 1),
            "mfma_frac_whole_step": round(tokens_per_s / world * ftok / 1e12 / PEAK_BF16_TFLOPS, 4),
            "roofline": roofline,
        }
        if world > 1:
            log = list(trainer.reducer.bucket_log)
            out["comm"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                           "ranks_seen": ranks_seen,
                           "nccl_version": nccl_version() if dist.get_backend() == "nccl" else None,
                           "collective": "all_reduce(AVG)" if trainer.reducer._native_avg else "all_reduce(SUM) / N",
                           "path": "mgen_comm (RCCL through the C-ABI)" if trainer.reducer._comm is not None
                           else "torch.distributed",
                           "buckets": [{"bytes": b, "dtype": d} for b, d in log],
                           "bytes_per_step": sum(b for b, _ in log)}
        if world == 1 and args.decode_tokens > 0:
            del trainer, batch
            model.zero_grad()
            torch.cuda.empty_cache()
            # BASELINE.json configs[3] as written: 256 streams from a one-token (BOS) prompt to 4096 tokens
            out["decode"] = bench_decode(model, cfg, args.decode_streams, 1, args.decode_full - 1, dev)
            out["decode_long_context"] = bench_decode(model, cfg, args.decode_streams, args.decode_prompt,
                                                      args.decode_tokens, dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, S)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
