"""Generate golden fixtures under tests/golden/ from the stock `transformers` modules the reference builds.

Run ONLY in the build container (needs `transformers`; never runs on the GPU box, never imported by the
product). The reference constructs its networks with AutoModelForCausalLM.from_config / GPT2LMHeadModel
(/root/reference/src/model/model.py:65,136), i.e. the stock transformers classes; this script instantiates the
same classes from explicit configs, runs them on seeded synthetic token batches and stores inputs + outputs.

    python tools/make_golden.py            # writes tests/golden/*.npz

Weights are NOT stored: they are re-created from a seed with oracle.ref_cpu.init_params (torch CPU generator,
deterministic for a fixed torch build), and a checksum of them is stored so that a drift would be noticed.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_cpu as R  # noqa: E402

import transformers  # noqa: E402
from transformers import Qwen3Config, Qwen3ForCausalLM, LlamaConfig, LlamaForCausalLM  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def hf_qwen3(cfg: R.LMConfig, params):
    hc = Qwen3Config(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                     num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                     num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
                     max_position_embeddings=cfg.max_position_embeddings, rms_norm_eps=cfg.rms_norm_eps,
                     tie_word_embeddings=cfg.tie_word_embeddings,
                     rope_parameters={"rope_type": "default", "rope_theta": cfg.rope_theta},
                     bos_token_id=cfg.bos_token_id, eos_token_id=cfg.eos_token_id, pad_token_id=cfg.pad_token_id)
    m = Qwen3ForCausalLM(hc)
    missing, unexpected = m.load_state_dict(params, strict=False)
    assert not unexpected, unexpected
    assert all(k == "lm_head.weight" for k in missing), missing
    m.tie_weights()
    return m.train()


def hf_llama(cfg: R.LMConfig, params):
    hc = LlamaConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                     num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                     num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
                     max_position_embeddings=cfg.max_position_embeddings, rms_norm_eps=cfg.rms_norm_eps,
                     tie_word_embeddings=cfg.tie_word_embeddings,
                     rope_parameters={"rope_type": "default", "rope_theta": cfg.rope_theta},
                     bos_token_id=cfg.bos_token_id, eos_token_id=cfg.eos_token_id, pad_token_id=cfg.pad_token_id)
    m = LlamaForCausalLM(hc)
    missing, unexpected = m.load_state_dict(params, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    return m.train()


def slim(a):
    """Big 2-D tensors keep every 4th row, logits [B,S,V] every 4th position (tests apply the same slicing)."""
    a = np.asarray(a)
    if a.ndim == 2 and a.size > 65536:
        return a[::4]
    if a.ndim == 3:
        return a[:, ::4, :]
    return a


def checksum(params):
    return float(sum(v.double().abs().sum() for v in params.values()))


def grads_of(m):
    return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}


GRAD_KEYS = ["model.embed_tokens.weight", "model.layers.0.self_attn.q_proj.weight",
             "model.layers.0.self_attn.k_proj.weight", "model.layers.0.self_attn.v_proj.weight",
             "model.layers.0.self_attn.o_proj.weight", "model.layers.0.input_layernorm.weight",
             "model.layers.0.post_attention_layernorm.weight", "model.layers.0.mlp.gate_proj.weight",
             "model.layers.0.mlp.up_proj.weight", "model.layers.1.mlp.down_proj.weight", "model.norm.weight"]


def tiny_case(name, cfg, builder, seed_ids=1234, B=2, S=128, extra_keys=()):
    params = R.init_params(cfg, seed=0)
    # make norm gains non-trivial so that their role is exercised
    g = torch.Generator().manual_seed(7)
    for k in params:
        if "norm" in k:
            params[k] = 1.0 + 0.1 * torch.randn(params[k].shape, generator=g)
    m = builder(cfg, params)
    gen = torch.Generator().manual_seed(seed_ids)
    ids = torch.randint(3, cfg.vocab_size, (B, S), generator=gen)
    out = {"seed_params": 0, "seed_norm": 7, "seed_ids": seed_ids, "param_checksum": checksum(params),
           "input_ids": ids.numpy()}

    # ---- fp32, labels = ids, no mask
    m.zero_grad()
    o = m(input_ids=ids, labels=ids)
    o.loss.backward()
    out["fp32_loss"] = o.loss.item()
    out["fp32_logits"] = o.logits.detach().numpy()
    gr = grads_of(m)
    for k in list(GRAD_KEYS) + list(extra_keys):
        if k in gr:
            out["fp32_grad/" + k] = gr[k].numpy()
    gnorm = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g_) for g_ in gr.values()]))
    out["fp32_grad_norm"] = gnorm.item()

    # ---- one clip(1.0)+AdamW step at lr schedule step 0 and step `warmup`
    opt = torch.optim.AdamW(m.parameters(), lr=5e-5, weight_decay=0.1, betas=(0.9, 0.95), eps=1e-5)
    warm = torch.optim.lr_scheduler.LinearLR(opt, start_factor=1e-6, end_factor=1.0, total_iters=10)
    dec = torch.optim.lr_scheduler.LinearLR(opt, start_factor=1.0, end_factor=0.1, total_iters=12 * 20 - 10)
    sch = torch.optim.lr_scheduler.SequentialLR(opt, [warm, dec], milestones=[10])
    lrs = []
    losses = []
    for step in range(14):
        m.zero_grad()
        o = m(input_ids=ids, labels=ids)
        o.loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
        losses.append(o.loss.item())
    out["train_lrs"] = np.array(lrs)
    out["train_losses"] = np.array(losses)
    sd = m.state_dict()
    for k in GRAD_KEYS[:2] + GRAD_KEYS[-1:]:
        out["after14/" + k] = sd[k].detach().numpy().copy()

    # ---- restore, padded batch (right padding, labels -100 on pads), fp32
    m = builder(cfg, params)
    mask = torch.ones(B, S, dtype=torch.long)
    mask[1, S - 37:] = 0
    lab = ids.clone()
    lab[mask == 0] = -100
    lab[0, :11] = -100  # prompt-span masking as in train_aria.py:139-140
    m.zero_grad()
    o = m(input_ids=ids, attention_mask=mask, labels=lab)
    o.loss.backward()
    out["pad_mask"] = mask.numpy()
    out["pad_labels"] = lab.numpy()
    out["pad_loss"] = o.loss.item()
    out["pad_logits"] = o.logits.detach().numpy()
    gr = grads_of(m)
    out["pad_grad/" + GRAD_KEYS[1]] = gr[GRAD_KEYS[1]].numpy()
    out["pad_grad/" + GRAD_KEYS[0]] = gr[GRAD_KEYS[0]].numpy()

    # ---- bf16 autocast (precision="bf16-mixed"), labels = ids
    m.zero_grad()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        o = m(input_ids=ids, labels=ids)
    o.loss.backward()
    out["bf16_loss"] = o.loss.item()
    out["bf16_logits"] = o.logits.detach().float().numpy()
    gr = grads_of(m)
    for k in GRAD_KEYS:
        if k in gr:
            out["bf16_grad/" + k] = gr[k].float().numpy()

    # ---- greedy decode, 512 tokens from an 8-token prompt, KV cache on, EOS disabled
    m.eval()
    prompt = ids[:1, :8]
    with torch.no_grad():
        gtok = m.generate(prompt, max_new_tokens=512, do_sample=False, eos_token_id=None, pad_token_id=0,
                          use_cache=True)
    out["greedy_tokens"] = gtok.numpy()
    # teacher-forced logits margin for the greedy run (top1-top2 gap per step) for tie diagnostics
    with torch.no_grad():
        lg = m(input_ids=gtok[:, :-1]).logits[0, 7:, :]
    top2 = torch.topk(lg, 2, dim=-1)[0]
    out["greedy_margin"] = (top2[:, 0] - top2[:, 1]).numpy()

    # ---- sampled decode with injected noise: multinomial(p,1) == argmax(p/q), q ~ Exp(1)
    for tag, kw in (("t097_p095", dict(temperature=0.97, top_p=0.95)),
                    ("t1_k5_p095", dict(temperature=1.0, top_k=5, top_p=0.95))):
        steps, Bs = 64, 2
        ng = torch.Generator().manual_seed(4321)
        noise = torch.empty(steps, Bs, cfg.vocab_size).exponential_(1.0, generator=ng)
        it = iter(range(steps))
        orig = torch.multinomial

        def fake_multinomial(probs, num_samples, replacement=False, *, generator=None):
            t = next(it)
            return (probs / noise[t]).argmax(dim=-1, keepdim=True)

        torch.multinomial = fake_multinomial
        try:
            with torch.no_grad():
                stok = m.generate(ids[:Bs, :8], max_new_tokens=steps, do_sample=True, eos_token_id=None,
                                  pad_token_id=0, use_cache=True, **kw)
        finally:
            torch.multinomial = orig
        out[f"sample_{tag}/tokens"] = stok.numpy()
        out[f"sample_{tag}/noise_seed"] = 4321
    m.train()

    out = {k: slim(v) if isinstance(v, np.ndarray) and ("grad/" in k or "logits" in k or "after14/" in k) else v
           for k, v in out.items()}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "fp32 loss", out["fp32_loss"], "bf16 loss", out["bf16_loss"], "pad loss", out["pad_loss"])
    return params, out


def full_dq_case():
    """C1 shape: default decoder, fp32, B=4, S=1024; scalars and slices only."""
    cfg = R.LMConfig()
    params = R.init_params(cfg, seed=0)
    m = hf_qwen3(cfg, params)
    gen = torch.Generator().manual_seed(1234)
    ids = torch.randint(3, 30000, (4, 1024), generator=gen)
    o = m(input_ids=ids, labels=ids)
    o.loss.backward()
    gr = grads_of(m)
    out = {"param_checksum": checksum(params), "loss": o.loss.item(),
           "logits_slice": o.logits[:, ::97, ::1013].detach().numpy(),
           "grad_norm": torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g_) for g_ in gr.values()])).item(),
           "grad_q0_slice": gr["model.layers.0.self_attn.q_proj.weight"][::16, ::16].numpy(),
           "grad_down13_slice": gr["model.layers.13.mlp.down_proj.weight"][::16, ::16].numpy(),
           "grad_norm_final": gr["model.norm.weight"].numpy(),
           "grad_embed_rows": gr["model.embed_tokens.weight"][ids[0, :16]].numpy(),
           "ids_head": ids[:, :16].numpy()}
    m.eval()
    with torch.no_grad():
        gtok = m.generate(ids[:1, :8], max_new_tokens=256, do_sample=False, eos_token_id=None, pad_token_id=0)
    out["greedy_tokens"] = gtok.numpy()
    with torch.autocast("cpu", dtype=torch.bfloat16), torch.no_grad():
        ob = m(input_ids=ids[:1, :256], labels=ids[:1, :256])
    out["bf16_loss_b1s256"] = ob.loss.item()
    np.savez_compressed(os.path.join(OUT, "dq_full_c1.npz"), **out)
    print("dq_full_c1 loss", out["loss"], "gnorm", out["grad_norm"])


def leftpad_case(name="leftpad_qwen3"):
    """Batched generation from LEFT-padded prompts (the batched-prompt form of GenerationMixin.generate: position ids
    from the mask's running sum, padded keys masked for every later step), tiny Qwen3 in fp32: greedy tokens, the
    per-step logits margin, and a sampled run with injected noise."""
    cfg = R.LMConfig(vocab_size=1024, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                     num_attention_heads=2, num_key_value_heads=2, head_dim=128)
    params = R.init_params(cfg, seed=0)
    g = torch.Generator().manual_seed(7)
    for k in params:
        if "norm" in k:
            params[k] = 1.0 + 0.1 * torch.randn(params[k].shape, generator=g)
    m = hf_qwen3(cfg, params).eval()
    gen = torch.Generator().manual_seed(99)
    B, P, steps = 4, 12, 40
    ids = torch.randint(3, cfg.vocab_size, (B, P), generator=gen)
    mask = torch.ones(B, P, dtype=torch.long)
    for b, pad in enumerate((0, 5, 11, 2)):
        mask[b, :pad] = 0
        ids[b, :pad] = 0
    out = {"param_checksum": checksum(params), "input_ids": ids.numpy(), "attention_mask": mask.numpy()}
    with torch.no_grad():
        gtok = m.generate(ids, attention_mask=mask, max_new_tokens=steps, do_sample=False, eos_token_id=None,
                          pad_token_id=0, use_cache=True)
    out["greedy_tokens"] = gtok.numpy()
    # each row generated alone from its unpadded prompt must give the same tokens (what left padding means)
    for b in range(B):
        n = int(mask[b].sum())
        with torch.no_grad():
            alone = m.generate(ids[b:b + 1, P - n:], max_new_tokens=steps, do_sample=False, eos_token_id=None,
                               pad_token_id=0, use_cache=True)
        assert torch.equal(alone[0, n:], gtok[b, P:]), b
    # top1 - top2 gap of every chosen token (teacher-forced on the unpadded rows), for tie diagnostics
    margins = []
    for b in range(B):
        n = int(mask[b].sum())
        row = gtok[b:b + 1, P - n:]
        with torch.no_grad():
            lg = m(input_ids=row[:, :-1]).logits[0, n - 1:, :]
        top2 = torch.topk(lg, 2, dim=-1)[0]
        margins.append((top2[:, 0] - top2[:, 1]).numpy())
    out["greedy_margin"] = np.stack(margins)
    ng = torch.Generator().manual_seed(4321)
    noise = torch.empty(steps, B, cfg.vocab_size).exponential_(1.0, generator=ng)
    it = iter(range(steps))
    orig = torch.multinomial

    def fake_multinomial(probs, num_samples, replacement=False, *, generator=None):
        t = next(it)
        return (probs / noise[t]).argmax(dim=-1, keepdim=True)

    torch.multinomial = fake_multinomial
    try:
        with torch.no_grad():
            stok = m.generate(ids, attention_mask=mask, max_new_tokens=steps, do_sample=True, temperature=0.97,
                              top_p=0.95, eos_token_id=None, pad_token_id=0, use_cache=True)
    finally:
        torch.multinomial = orig
    out["sample_tokens"] = stok.numpy()
    out["noise_seed"] = 4321
    # EOS bookkeeping with ragged prompts: a token row 1 emits early becomes the EOS id
    eos = int(gtok[1, P + 3])
    with torch.no_grad():
        etok = m.generate(ids, attention_mask=mask, max_new_tokens=steps, do_sample=False, eos_token_id=eos,
                          pad_token_id=0, use_cache=True)
    out["eos_id"] = eos
    out["eos_tokens"] = etok.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "greedy", gtok.shape, "min margin", float(out["greedy_margin"].min()), "eos run", etok.shape)


def load_reference_pos4d():
    """Import the reference's own 4-D model module (the only in-tree arithmetic of the reference) and pull the two
    post-processing functions out of generate_music.py without importing that script (its top-level imports need
    packages that are absent here). `wandb` is absent too: a stub with a no-op log() takes its place."""
    import ast
    import importlib.machinery
    import types
    import transformers.activations  # noqa: F401
    import transformers.modeling_utils  # noqa: F401
    if "wandb" not in sys.modules:
        w = types.ModuleType("wandb")
        w.__spec__ = importlib.machinery.ModuleSpec("wandb", None)
        w.log = lambda *a, **k: None
        sys.modules["wandb"] = w
    sys.path.insert(0, "/root/reference")
    from src.model import modeling as ref_modeling
    from src.model import config as ref_config
    tree = ast.parse(open("/root/reference/generate_music.py").read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("f2", "process_last_logits")]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=keep, type_ignores=[]), "generate_music.py", "exec"), ns)
    return ref_modeling, ref_config, ns["process_last_logits"]


def synth_notes(gen, B, S, pad_tail=()):
    """Batches in the layout of MidiDataset4D (/root/reference/src/train_model_new.py:23-47,114-158): rows
    [start s, duration s, pitch, velocity], a [0,0,0,0] row first, labels = next row with the start replaced by the
    start delta, the last label = the last row with delta 0; padded tails repeat [last start, 0, 2, 0] and carry
    -100 labels."""
    pos = torch.zeros(B, S, 4)
    lab = torch.zeros(B, S, 4)
    mask = torch.ones(B, S, dtype=torch.long)
    for b in range(B):
        n_pad = pad_tail[b] if b < len(pad_tail) else 0
        n = S - n_pad
        delta = torch.empty(n - 1).exponential_(4.0, generator=gen)
        delta[torch.rand(n - 1, generator=gen) < 0.3] = 0.0   # chords: equal start times
        start = torch.cat([torch.zeros(1), delta.cumsum(0)])
        dur = 0.05 + 2.0 * torch.rand(n, generator=gen)
        pitch = torch.randint(21, 109, (n,), generator=gen).float()
        vel = torch.randint(1, 128, (n,), generator=gen).float()
        rows = torch.stack([start, dur, pitch, vel], dim=1)
        rows[0] = 0.0
        if n_pad:
            rows = torch.cat([rows, torch.tensor([float(rows[-1, 0]), 0.0, 2.0, 0.0]).repeat(n_pad, 1)], dim=0)
            mask[b, n:] = 0
        l = torch.cat([rows[1:].clone(), rows[-1:].clone()], dim=0)
        l[:, 0] = l[:, 0] - rows[:, 0]
        if n_pad:
            l[n:] = -100
        pos[b], lab[b] = rows, l
    return pos, lab, mask


POS4D_GRAD_KEYS = ["lm_head.weight", "model.layers.0.self_attn.q_proj.weight",
                   "model.layers.0.self_attn.k_proj.weight", "model.layers.0.self_attn.v_proj.weight",
                   "model.layers.0.self_attn.o_proj.weight", "model.layers.0.self_attn.q_norm.weight",
                   "model.layers.0.self_attn.k_norm.weight", "model.layers.0.input_layernorm.weight",
                   "model.layers.0.mlp.gate_proj.weight", "model.layers.1.mlp.down_proj.weight", "model.norm.weight"]


def pos4d_case(name="tiny_pos4d", B=2, S=128):
    from oracle import ref_pos4d as P
    ref_modeling, ref_config, ref_process_last = load_reference_pos4d()
    cfg = R.LMConfig(vocab_size=1, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                     num_attention_heads=2, num_key_value_heads=2, head_dim=128, tie_word_embeddings=False)
    params = P.init_params_pos4d(cfg, seed=0)
    g = torch.Generator().manual_seed(7)
    for k in params:
        if "norm" in k:
            params[k] = 1.0 + 0.1 * torch.randn(params[k].shape, generator=g)

    def build():
        hc = ref_config.Qwen3Config(vocab_size=1, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                                    num_hidden_layers=cfg.num_hidden_layers,
                                    num_attention_heads=cfg.num_attention_heads,
                                    num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
                                    max_position_embeddings=4096, rms_norm_eps=cfg.rms_norm_eps,
                                    pad_token_id=0, bos_token_id=0, eos_token_id=0)
        hc.pad_token_id = 0
        m = ref_modeling.Qwen3ForCausalLM(hc)
        missing, unexpected = m.load_state_dict(params, strict=True)
        return m.train()

    causal = {"full_attention": None}   # "already prepared" mask mapping: plain causal attention
    m = build()
    gen = torch.Generator().manual_seed(2468)
    pos, lab, mask = synth_notes(gen, B, S, pad_tail=(0, 37))
    out = {"seed_params": 0, "seed_norm": 7, "param_checksum": checksum(params), "position_tensors": pos.numpy(),
           "labels": lab.numpy(), "attention_mask": mask.numpy()}

    m.zero_grad()
    o = m(position_tensors=pos, labels=lab, attention_mask=causal)
    o.loss.backward()
    out["fp32_loss"] = o.loss.item()
    out["fp32_pred"] = o.logits.detach().numpy()
    gr = grads_of(m)
    for k in POS4D_GRAD_KEYS:
        out["fp32_grad/" + k] = gr[k].numpy()
    out["fp32_grad_norm"] = torch.linalg.vector_norm(
        torch.stack([torch.linalg.vector_norm(g_) for g_ in gr.values()])).item()
    out["encoding_of_labels"] = ref_modeling._compute_encoding(lab[:, :S - 37]).numpy()
    out["rope_angles"] = m.model.rot_pos_emb(pos).detach().numpy()[:, ::8]

    # all labels masked -> constant zero loss
    o = m(position_tensors=pos, labels=torch.full_like(lab, -100.0), attention_mask=causal)
    out["all_masked_loss"] = float(o.loss)

    # bf16 autocast on the CPU (torch.norm of the head output runs in bf16 there, in fp32 on a GPU)
    m.zero_grad()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        o = m(position_tensors=pos, labels=lab, attention_mask=causal)
    o.loss.backward()
    out["bf16_loss"] = o.loss.item()
    out["bf16_pred"] = o.logits.detach().float().numpy()
    gr = grads_of(m)
    for k in POS4D_GRAD_KEYS:
        out["bf16_grad/" + k] = gr[k].float().numpy()

    # 14 steps of clip(5.0) + AdamW + schedule (src/train_model_new.py:372-378: lr 5e-4, clip 5.0)
    m = build()
    opt = torch.optim.AdamW(m.parameters(), lr=5e-4, weight_decay=0.1, betas=(0.9, 0.95), eps=1e-5)
    warm = torch.optim.lr_scheduler.LinearLR(opt, start_factor=1e-6, end_factor=1.0, total_iters=10)
    dec = torch.optim.lr_scheduler.LinearLR(opt, start_factor=1.0, end_factor=0.1, total_iters=12 * 20 - 10)
    sch = torch.optim.lr_scheduler.SequentialLR(opt, [warm, dec], milestones=[10])
    lrs, losses = [], []
    for step in range(14):
        m.zero_grad()
        o = m(position_tensors=pos, labels=lab, attention_mask=causal)
        o.loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 5.0)
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
        losses.append(o.loss.item())
    out["train_lrs"], out["train_losses"] = np.array(lrs), np.array(losses)
    sd = m.state_dict()
    for k in ("lm_head.weight", "model.layers.0.self_attn.q_proj.weight", "model.norm.weight"):
        out["after14/" + k] = sd[k].detach().numpy().copy()

    # generate_music loop (generate_music.py:165-179) on the seeded weights: 48 notes, prefix re-run every step
    m = build().eval()
    generated = pos[0:1, 0:1, :]
    preds = []
    with torch.no_grad():
        for i in range(48):
            o = m(input_ids=None, position_tensors=generated, attention_mask=causal)
            preds.append(o.logits[0, i].clone())
            nxt = ref_process_last(o.logits[0, i, :, :], device="cpu")
            nxt[0, 0, 0] += generated[0, -1, 0]
            generated = torch.cat([generated, nxt], dim=1)
    out["gen_rows"] = generated[0].numpy()
    out["gen_preds"] = torch.stack(preds, 0).numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "fp32 loss", out["fp32_loss"], "bf16 loss", out["bf16_loss"], "last loss", losses[-1])


GPT2_GRAD_KEYS = ["transformer.wte.weight", "transformer.wpe.weight", "transformer.h.0.ln_1.weight",
                  "transformer.h.0.ln_1.bias", "transformer.h.0.attn.c_attn.weight", "transformer.h.0.attn.c_attn.bias",
                  "transformer.h.0.attn.c_proj.weight", "transformer.h.0.attn.c_proj.bias",
                  "transformer.h.0.mlp.c_fc.weight", "transformer.h.0.mlp.c_fc.bias",
                  "transformer.h.1.mlp.c_proj.weight", "transformer.h.1.mlp.c_proj.bias", "transformer.ln_f.weight",
                  "transformer.ln_f.bias"]


def gpt2_case(name="tiny_gpt2", B=2, S=128):
    """transformers' GPT2LMHeadModel (what /root/reference/src/model/model.py:54-65 builds) with dropout off."""
    from transformers import GPT2Config, GPT2LMHeadModel
    from oracle import ref_gpt2 as G
    cfg = R.LMConfig(vocab_size=1000, hidden_size=256, intermediate_size=1024, num_hidden_layers=2,
                     num_attention_heads=4, num_key_value_heads=4, head_dim=64, max_position_embeddings=1024,
                     tie_word_embeddings=True, qk_norm=False)
    params = G.init_params_gpt2(cfg, seed=0)
    g = torch.Generator().manual_seed(7)
    for k in params:   # non-trivial gains and biases
        if ".ln_" in k or k.endswith(".bias"):
            params[k] = params[k] + 0.1 * torch.randn(params[k].shape, generator=g)

    def build():
        hc = GPT2Config(vocab_size=cfg.vocab_size, n_positions=cfg.max_position_embeddings, n_embd=cfg.hidden_size,
                        n_layer=cfg.num_hidden_layers, n_head=cfg.num_attention_heads, resid_pdrop=0.0,
                        embd_pdrop=0.0, attn_pdrop=0.0, bos_token_id=1, eos_token_id=2, pad_token_id=0)
        m = GPT2LMHeadModel(hc)
        missing, unexpected = m.load_state_dict(params, strict=False)
        assert not unexpected and all(k == "lm_head.weight" for k in missing), (missing, unexpected)
        m.tie_weights()
        return m.train()

    m = build()
    gen = torch.Generator().manual_seed(1234)
    ids = torch.randint(3, cfg.vocab_size, (B, S), generator=gen)
    out = {"param_checksum": checksum(params), "input_ids": ids.numpy()}
    m.zero_grad()
    o = m(input_ids=ids, labels=ids)
    o.loss.backward()
    out["fp32_loss"] = o.loss.item()
    out["fp32_logits"] = o.logits.detach().numpy()
    gr = grads_of(m)
    for k in GPT2_GRAD_KEYS:
        out["fp32_grad/" + k] = gr[k].numpy()
    out["fp32_grad_norm"] = torch.linalg.vector_norm(
        torch.stack([torch.linalg.vector_norm(g_) for g_ in gr.values()])).item()

    # padded batch
    mask = torch.ones(B, S, dtype=torch.long)
    mask[1, S - 37:] = 0
    lab = ids.clone()
    lab[mask == 0] = -100
    m.zero_grad()
    o = m(input_ids=ids, attention_mask=mask, labels=lab)
    o.loss.backward()
    out["pad_mask"], out["pad_labels"], out["pad_loss"] = mask.numpy(), lab.numpy(), o.loss.item()
    out["pad_grad/" + GPT2_GRAD_KEYS[4]] = grads_of(m)[GPT2_GRAD_KEYS[4]].numpy()

    # bf16 autocast
    m.zero_grad()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        o = m(input_ids=ids, labels=ids)
    o.loss.backward()
    out["bf16_loss"] = o.loss.item()
    out["bf16_logits"] = o.logits.detach().float().numpy()

    # 14 optimizer steps, lr 3e-4 (MidiGPT2 default), clip 1.0
    opt = torch.optim.AdamW(m.parameters(), lr=3e-4, weight_decay=0.1, betas=(0.9, 0.95), eps=1e-5)
    warm = torch.optim.lr_scheduler.LinearLR(opt, start_factor=1e-6, end_factor=1.0, total_iters=10)
    dec = torch.optim.lr_scheduler.LinearLR(opt, start_factor=1.0, end_factor=0.1, total_iters=12 * 20 - 10)
    sch = torch.optim.lr_scheduler.SequentialLR(opt, [warm, dec], milestones=[10])
    lrs, losses = [], []
    for step in range(14):
        m.zero_grad()
        o = m(input_ids=ids, labels=ids)
        o.loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
        losses.append(o.loss.item())
    out["train_lrs"], out["train_losses"] = np.array(lrs), np.array(losses)

    # greedy decode from the seeded weights
    m = build().eval()
    with torch.no_grad():
        gtok = m.generate(ids[:1, :8], max_new_tokens=256, do_sample=False, eos_token_id=None, pad_token_id=0,
                          use_cache=True)
        lg = m(input_ids=gtok[:, :-1]).logits[0, 7:, :]
    out["greedy_tokens"] = gtok.numpy()
    top2 = torch.topk(lg, 2, dim=-1)[0]
    out["greedy_margin"] = (top2[:, 0] - top2[:, 1]).numpy()
    out = {k: slim(v) if isinstance(v, np.ndarray) and ("grad/" in k or "logits" in k) else v for k, v in out.items()}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "fp32 loss", out["fp32_loss"], "bf16 loss", out["bf16_loss"], "pad loss", out["pad_loss"],
          "last", losses[-1])


def gpt2_dropout_case(name="tiny_gpt2_dropout", B=2, S=128, seed=11, pdrop=0.1):
    """The same GPT2LMHeadModel in training mode with the GPT2Config default dropout of 0.1 (what the reference
    trains with). torch's own mask stream cannot be reproduced elsewhere, so torch.nn.functional.dropout is replaced
    for the duration of the forward by a function that applies the masks of oracle.ref_dropout in call order
    (embeddings; per block: attention probabilities, attention projection, MLP projection) with torch's arithmetic
    x * mask / (1 - p). What the fixture pins is WHERE transformers applies its dropouts and how they enter the loss
    and the gradients; the mask stream itself stays this build's own."""
    from transformers import GPT2Config, GPT2LMHeadModel
    from oracle import ref_gpt2 as G
    from oracle import ref_dropout as RD
    import torch.nn.functional as F
    cfg = R.LMConfig(vocab_size=1000, hidden_size=256, intermediate_size=1024, num_hidden_layers=2,
                     num_attention_heads=4, num_key_value_heads=4, head_dim=64, max_position_embeddings=1024,
                     tie_word_embeddings=True, qk_norm=False)
    params = G.init_params_gpt2(cfg, seed=0)
    g = torch.Generator().manual_seed(7)
    for k in params:
        if ".ln_" in k or k.endswith(".bias"):
            params[k] = params[k] + 0.1 * torch.randn(params[k].shape, generator=g)
    hc = GPT2Config(vocab_size=cfg.vocab_size, n_positions=cfg.max_position_embeddings, n_embd=cfg.hidden_size,
                    n_layer=cfg.num_hidden_layers, n_head=cfg.num_attention_heads, resid_pdrop=pdrop,
                    embd_pdrop=pdrop, attn_pdrop=pdrop, bos_token_id=1, eos_token_id=2, pad_token_id=0,
                    attn_implementation="eager")
    m = GPT2LMHeadModel(hc)
    missing, unexpected = m.load_state_dict(params, strict=False)
    assert not unexpected and all(k == "lm_head.weight" for k in missing), (missing, unexpected)
    m.tie_weights()
    m.train()
    plan = RD.DropPlan(seed, 0, cfg.num_hidden_layers, pdrop, pdrop, pdrop)
    calls = []

    def planned_dropout(x, p=0.5, training=True, inplace=False):
        site = len(calls)
        calls.append(tuple(x.shape))
        assert training and abs(p - pdrop) < 1e-12
        if x.dim() == 4:   # attention probabilities [B, H, S, S]
            keep = RD.keep_attn(plan.key(site), x.shape[0], x.shape[1], x.shape[2], p)
        else:
            keep = RD.keep_flat(plan.key(site), x.numel(), p).view(x.shape)
        return x * keep.to(x.dtype) * (1.0 / (1.0 - p))

    gen = torch.Generator().manual_seed(1234)
    ids = torch.randint(3, cfg.vocab_size, (B, S), generator=gen)
    real = F.dropout
    F.dropout = planned_dropout
    try:
        m.zero_grad()
        o = m(input_ids=ids, labels=ids)
        o.loss.backward()
    finally:
        F.dropout = real
    L = cfg.num_hidden_layers
    want = [(B, S, cfg.hidden_size)] + [(B, cfg.num_attention_heads, S, S), (B, S, cfg.hidden_size),
                                        (B, S, cfg.hidden_size)] * L
    assert calls == want, calls
    out = {"param_checksum": checksum(params), "input_ids": ids.numpy(), "seed": seed, "pdrop": pdrop,
           "fp32_loss": o.loss.item(), "fp32_logits": slim(o.logits.detach().numpy())}
    gr = grads_of(m)
    for k in GPT2_GRAD_KEYS:
        out["fp32_grad/" + k] = slim(gr[k].numpy())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "fp32 loss with dropout", out["fp32_loss"], "sites", len(calls))


def load_reference_collate():
    """Pull the batch-assembly functions out of the reference's training scripts without importing those scripts
    (their top-level imports need miditok / symusic / lightning, absent here): the function definitions are compiled
    from the parsed source with only `torch` in scope."""
    import ast
    ns = {"torch": torch}
    tree = ast.parse(open("/root/reference/src/train_aria.py").read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("collate_fn", "style_collate_fn")]
    exec(compile(ast.Module(body=keep, type_ignores=[]), "train_aria.py", "exec"), ns)
    tree = ast.parse(open("/root/reference/src/train_model_new.py").read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "custom_collate_fn"]
    for n in tree.body:
        if isinstance(n, ast.ClassDef) and n.name == "MidiDataset4D":
            keep += [m for m in n.body if isinstance(m, ast.FunctionDef) and m.name == "_create_chunks"]
    exec(compile(ast.Module(body=keep, type_ignores=[]), "train_model_new.py", "exec"), ns)
    return ns["collate_fn"], ns["style_collate_fn"], ns["_create_chunks"], ns["custom_collate_fn"]


def collate_case(name="collate"):
    """Ragged token batches through the reference's two in-tree collate functions, and a concatenated note-row
    sequence through MidiDataset4D._create_chunks + custom_collate_fn (last chunk short)."""
    import contextlib
    import io
    import types
    melody_fn, style_fn, create_chunks, stack_fn = load_reference_collate()
    gen = torch.Generator().manual_seed(99)
    lens = [137, 512, 1, 300, 511, 64]
    seqs = [torch.randint(3, 17000, (n,), generator=gen).tolist() for n in lens]
    mel = [40, 512, 0, 299, 100, 64]
    pad_id = 2
    out = {"tok_flat": np.concatenate([np.asarray(s, np.int64) for s in seqs]),
           "tok_offsets": np.cumsum([0] + lens).astype(np.int64), "tok_melody": np.asarray(mel, np.int32),
           "tok_pad_id": pad_id}
    o = melody_fn([{"input_ids": s, "melody_length": m} for s, m in zip(seqs, mel)], pad_id)
    out["melody_input_ids"], out["melody_labels"] = o["input_ids"].numpy(), o["labels"].numpy()
    o = style_fn([{"input_ids": s} for s in seqs], pad_id)
    out["style_input_ids"], out["style_labels"] = o["input_ids"].numpy(), o["labels"].numpy()

    S = 64
    pos, _, _ = synth_notes(gen, 1, 5 * S - 23)
    rows = pos[0]
    ds = types.SimpleNamespace(max_seq_len=S, chunks=[])
    with contextlib.redirect_stdout(io.StringIO()):   # the method prints two lines per chunk
        create_chunks(ds, rows)
    assert len(ds.chunks) == 5
    order = [4, 0, 2, 2, 1, 3]
    o = stack_fn([ds.chunks[c] for c in order])
    out["p4_rows"], out["p4_S"], out["p4_chunk_ids"] = rows.numpy(), S, np.asarray(order, np.int64)
    for k in ("input_ids", "attention_mask", "labels", "position_tensors"):
        out["p4_" + k] = o[k].numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, {k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    print("transformers", transformers.__version__, "torch", torch.__version__)
    if "--leftpad-only" in sys.argv:
        leftpad_case()
        sys.exit(0)
    if "--collate-only" in sys.argv:
        collate_case()
        sys.exit(0)
    if "--pos4d-only" in sys.argv:
        pos4d_case()
        sys.exit(0)
    if "--gpt2-only" in sys.argv:
        gpt2_case()
        sys.exit(0)
    if "--gpt2-dropout-only" in sys.argv:
        gpt2_dropout_case()
        sys.exit(0)
    tiny_q = R.LMConfig(vocab_size=1024, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                        num_attention_heads=2, num_key_value_heads=2, head_dim=128)
    tiny_case("tiny_qwen3", tiny_q, hf_qwen3)
    tiny_l = R.LMConfig(vocab_size=1031, hidden_size=256, intermediate_size=768, num_hidden_layers=2,
                        num_attention_heads=4, num_key_value_heads=4, head_dim=64, rms_norm_eps=1e-6,
                        rope_theta=10000.0, tie_word_embeddings=False, qk_norm=False)
    tiny_case("tiny_llama", tiny_l, hf_llama, extra_keys=("lm_head.weight",))
    collate_case()
    leftpad_case()
    if "--full" in sys.argv:
        full_dq_case()
