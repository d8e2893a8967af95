"""The Lightning-shaped wrapper classes: protocol, checkpoint round trip, GPT-2 position-table growth."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


class Tok:
    pad_token_id = 0
    vocab_size = 1000

    def __getitem__(self, k):
        return {"BOS_None": 1, "EOS_None": 2}[k]


def _small(kind):
    from mgen.config import LMConfig
    if kind == "qwen":
        return LMConfig(vocab_size=1000, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                        num_attention_heads=2, num_key_value_heads=2, head_dim=128)
    if kind == "aria":
        return LMConfig(vocab_size=1000, hidden_size=256, intermediate_size=768, num_hidden_layers=2,
                        num_attention_heads=4, num_key_value_heads=4, head_dim=64, rope_theta=10000.0,
                        tie_word_embeddings=False, qk_norm=False)
    if kind == "gpt2":
        return LMConfig(arch="gpt2", vocab_size=1000, hidden_size=256, intermediate_size=1024, num_hidden_layers=2,
                        num_attention_heads=4, num_key_value_heads=4, head_dim=64, max_position_embeddings=256,
                        tie_word_embeddings=True, qk_norm=False)
    return LMConfig(vocab_size=1, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                    num_key_value_heads=2, head_dim=128, tie_word_embeddings=False, pos4d=True)


@pytest.mark.parametrize("kind", ["qwen", "aria", "gpt2"])
def test_token_wrappers_follow_the_module_protocol(kind, tmp_path):
    from mgen import wrappers as W
    cls = {"qwen": W.MidiQwen, "aria": W.MidiAria, "gpt2": W.MidiGPT2}[kind]
    loader = list(range(20))                         # len(dataloader) = steps per epoch
    m = cls(Tok(), loader, warmup_steps=5, config=_small(kind))
    assert m.lr == cls.lr_default
    ids = torch.randint(3, 1000, (2, 64), generator=torch.Generator().manual_seed(1)).cuda()
    batch = {"input_ids": ids, "attention_mask": torch.ones_like(ids), "labels": ids}
    cfgd = m.configure_optimizers()
    opt, sched = cfgd["optimizer"], cfgd["lr_scheduler"]["scheduler"]
    assert cfgd["lr_scheduler"]["interval"] == "step"
    losses = []
    for step in range(8):
        opt.zero_grad()
        loss = m.training_step(batch, step)
        loss.backward()
        opt.step()
        sched.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0] and "train_loss" in m.logged
    v = m.validation_step(batch, 0)
    assert abs(float(v) - float(m.logged["val_loss"])) == 0.0 and not v.requires_grad
    # checkpoint in the Lightning layout: {"state_dict": {"model.<hf name>": tensor}}
    sd = m.state_dict()
    assert all(k.startswith("model.") for k in sd)
    path = os.path.join(tmp_path, "last.ckpt")
    torch.save({"state_dict": {k: t.detach().cpu().clone() for k, t in sd.items()}, "epoch": 3}, path)
    m2 = cls(Tok(), loader, config=_small(kind), seed=5)
    m2.load_checkpoint_expanding_pos_emb(path)
    with torch.no_grad():
        a = float(m.validation_step(batch, 0))
        b = float(m2.validation_step(batch, 0))
    assert a == b
    out = m2.model.generate(ids[:, :8], max_new_tokens=6, do_sample=False)
    assert tuple(out.shape) == (2, 14)


def test_gpt2_checkpoint_with_shorter_context_grows_position_table(tmp_path):
    from mgen import wrappers as W
    from mgen.config import LMConfig
    short_cfg = _small("gpt2")
    short_cfg.max_position_embeddings = 128
    short = W.MidiGPT2(Tok(), [0], config=short_cfg, seed=3)
    path = os.path.join(tmp_path, "short.ckpt")
    torch.save({"state_dict": {k: t.detach().cpu().clone() for k, t in short.state_dict().items()}}, path)
    full = W.MidiGPT2(Tok(), [0], config=_small("gpt2"), seed=4)
    before = full.model.transformer.wpe.weight.detach().clone()
    full.load_checkpoint_expanding_pos_emb(path)
    wpe = full.model.transformer.wpe.weight.detach()
    assert torch.equal(wpe[:128], short.model.transformer.wpe.weight.detach())
    assert torch.equal(wpe[128:], before[128:])
    assert torch.equal(full.model.transformer.h[0].mlp.c_fc.weight.detach(),
                       short.model.transformer.h[0].mlp.c_fc.weight.detach())


def test_pos4d_wrapper(tmp_path):
    from mgen import wrappers as W
    m = W.MidiQwenNew(Tok(), list(range(10)), lr=5e-4, warmup_steps=5, config=_small("pos4d"))
    g = torch.Generator().manual_seed(2)
    pos = torch.rand(2, 32, 4, generator=g).cumsum(1)
    lab = torch.cat([pos[:, 1:], pos[:, -1:]], 1).clone()
    lab[..., 0] = lab[..., 0] - pos[..., 0]
    batch = {"input_ids": torch.zeros(2, 32, dtype=torch.long).cuda(), "attention_mask": torch.ones(2, 32).long().cuda(),
             "labels": lab.cuda(), "position_tensors": pos.cuda()}
    cfgd = m.configure_optimizers(max_grad_norm=5.0)
    opt, sched = cfgd["optimizer"], cfgd["lr_scheduler"]["scheduler"]
    losses = []
    for step in range(8):
        opt.zero_grad()
        loss = m.training_step(batch, step)
        loss.backward()
        opt.step()
        sched.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0]
    # the eight per-pair losses the reference logs from its forward; their mean is the logged loss
    names = ["loss/s_0", "loss/s_1", "loss/d_0", "loss/d_1", "loss/p_0", "loss/p_1", "loss/v_0", "loss/v_1"]
    comp = torch.stack([m.logged[n] for n in names])
    assert abs(float(comp.mean()) - float(m.logged["train_loss"])) < 1e-5
    rows = m.model.generate_music(pos[:1, :1].cuda(), 5)
    assert tuple(rows.shape) == (1, 6, 4)
