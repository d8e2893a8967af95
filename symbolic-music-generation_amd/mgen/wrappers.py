"""Wrapper classes with the reference's module protocol around the MI355X models.

The reference wraps each network in a LightningModule (/root/reference/src/model/model.py: MidiGPT2 :49-116,
MidiQwen :120-177, MidiAria :185-249, MidiQwenNew :273-337) whose surface its scripts and checkpoints rely on:
`module(**batch)` -> outputs with `.loss`, `training_step(batch, idx)`, `validation_step(batch, idx)`,
`configure_optimizers()` -> {"optimizer", "lr_scheduler": {"scheduler", "interval": "step"}}, `.model` (the network,
whose `.generate` the sampling sites call), `state_dict()` keys under the `model.` prefix, and
`load_checkpoint_expanding_pos_emb(path)` reading a Lightning `.ckpt` (`checkpoint["state_dict"]`, strict=False; only
the GPT-2 variant grows its position table). Lightning itself is not required: these are plain torch modules driven by
`mgen.trainer.Trainer` or by any loop that follows the protocol.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .config import LMConfig, default_decoder, gpt2_decoder, aria_style_llama, pos4d_decoder
from .gpt2 import MidiGPT2LM
from .model import MidiLM
from .optim import get_optim

EPOCHS = 12          # src/train.py:15 (configure_optimizers of every wrapper imports this one)


def _tok(tokenizer, name, default):
    try:
        return int(tokenizer[name])
    except Exception:
        return default


class _Wrapper(nn.Module):
    lr_default, warmup_default = 5e-5, 1000

    def __init__(self, tokenizer, dataloader, lr=None, warmup_steps=None):
        super().__init__()
        self.tokenizer = tokenizer
        self.dataloader = dataloader
        self.lr = self.lr_default if lr is None else lr
        self.warmup_steps = self.warmup_default if warmup_steps is None else warmup_steps
        self.logged = {}

    # LightningModule.log stand-in: keeps the last value of every key
    def log(self, name, value, **unused):
        self.logged[name] = value.detach() if torch.is_tensor(value) else value

    def forward(self, input_ids=None, attention_mask=None, labels=None, **extra):
        return self.model(input_ids=input_ids, attention_mask=attention_mask, labels=labels, **extra)

    @staticmethod
    def _loss_of(outputs):
        return outputs[0] if isinstance(outputs, tuple) else outputs.loss

    def training_step(self, batch, batch_idx=0):
        loss = self._loss_of(self(**batch))
        self.log("train_loss", loss)
        return loss

    @torch.no_grad()
    def validation_step(self, batch, batch_idx=0):
        val_loss = self._loss_of(self(**batch))
        if torch.distributed.is_available() and torch.distributed.is_initialized():   # sync_dist=True: mean over ranks
            torch.distributed.all_reduce(val_loss)
            val_loss = val_loss / torch.distributed.get_world_size()
        self.log("val_loss", val_loss)
        return val_loss

    def configure_optimizers(self, max_grad_norm: float = 1.0):
        """`max_grad_norm` folds the Trainer's gradient_clip_val (1.0 in src/train.py:66; 5.0 in
        src/train_model_new.py:378) into the fused optimizer step."""
        optimizer, scheduler = get_optim(self.model, self.lr, EPOCHS, len(self.dataloader), warmup=self.warmup_steps,
                                         end_ratio=0.1, max_grad_norm=max_grad_norm)
        return {"optimizer": optimizer, "lr_scheduler": {"scheduler": scheduler, "interval": "step"}}

    # checkpoints -------------------------------------------------------------------------------------------------
    def state_dict(self, *args, **kw):
        return {"model." + k: v for k, v in self.model.state_dict().items()}

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        inner = {k[len("model."):]: v for k, v in state_dict.items() if k.startswith("model.")}
        extra = [k for k in state_dict if not k.startswith("model.")]
        if strict and extra:
            raise RuntimeError(f"load_state_dict: unexpected keys {extra}")
        res = self.model.load_state_dict(inner, strict=strict)
        return torch.nn.modules.module._IncompatibleKeys(["model." + k for k in res.missing_keys],
                                                         ["model." + k for k in res.unexpected_keys] + extra)

    @staticmethod
    def _read_checkpoint(path):
        """A Lightning checkpoint is a pickled dict; it is read with the tensors-only loader (nothing in the file is
        executed). Files holding arbitrary pickled objects are refused by that loader."""
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        if "state_dict" not in ckpt:
            raise KeyError(f"{path}: no 'state_dict' entry (not a Lightning checkpoint)")
        return ckpt["state_dict"]

    def load_checkpoint_expanding_pos_emb(self, checkpoint_path):
        return self.load_state_dict(self._read_checkpoint(checkpoint_path), strict=False)


class MidiQwen(_Wrapper):
    """Default decoder (Qwen3 blocks, d 512, 14 layers, 8 heads of 128, vocabulary 151 936, tied head)."""
    lr_default, warmup_default = 5e-5, 1000

    def __init__(self, tokenizer, dataloader, lr=None, warmup_steps=None, device="cuda", config: Optional[LMConfig] = None,
                 seed: Optional[int] = 0):
        super().__init__(tokenizer, dataloader, lr, warmup_steps)
        cfg = config or default_decoder()
        cfg.bos_token_id = _tok(tokenizer, "BOS_None", cfg.bos_token_id)
        cfg.eos_token_id = _tok(tokenizer, "EOS_None", cfg.eos_token_id)
        cfg.pad_token_id = getattr(tokenizer, "pad_token_id", cfg.pad_token_id)
        self.model = MidiLM(cfg, device=device, seed=seed)


class MidiAria(_Wrapper):
    """The reference fine-tunes a remote-code Llama-style checkpoint (loubb/aria-medium-base) that cannot be fetched
    here; this wrapper builds a Llama-architecture network of that size class with random weights."""
    lr_default, warmup_default = 1e-5, 100

    def __init__(self, tokenizer, dataloader, lr=None, warmup_steps=None, device="cuda", config: Optional[LMConfig] = None,
                 seed: Optional[int] = 0):
        super().__init__(tokenizer, dataloader, lr, warmup_steps)
        self.model = MidiLM(config or aria_style_llama(), device=device, seed=seed)

    def to_lora(self):
        """/root/reference/src/model/model.py:251-271: freeze the base, rank-32 adapters (alpha 64, dropout 0.05) on
        the fused QKV, attention output, gate, up and down projections. Call it before configure_optimizers(): the
        optimizer then covers the adapters only."""
        return self.model.to_lora(r=32, lora_alpha=64, lora_dropout=0.05)


class MidiGPT2(_Wrapper):
    lr_default, warmup_default = 3e-4, 1000

    def __init__(self, tokenizer, dataloader, lr=None, warmup_steps=None, device="cuda", config: Optional[LMConfig] = None,
                 seed: Optional[int] = 0):
        super().__init__(tokenizer, dataloader, lr, warmup_steps)
        cfg = config or gpt2_decoder(vocab_size=int(getattr(tokenizer, "vocab_size", 30000)))
        cfg.bos_token_id = _tok(tokenizer, "BOS_None", cfg.bos_token_id)
        cfg.eos_token_id = _tok(tokenizer, "EOS_None", cfg.eos_token_id)
        cfg.pad_token_id = getattr(tokenizer, "pad_token_id", cfg.pad_token_id)
        self.model = MidiGPT2LM(cfg, device=device, seed=seed)

    def load_checkpoint_expanding_pos_emb(self, checkpoint_path):
        """A checkpoint trained with a shorter context keeps its rows; the remaining positions keep their current
        values (src/model/model.py:101-116)."""
        sd = dict(self._read_checkpoint(checkpoint_path))
        key = "model.transformer.wpe.weight"
        cur = self.model.transformer.wpe.weight
        old = sd.get(key)
        if old is not None and old.shape[0] < cur.shape[0]:
            grown = cur.detach().clone().cpu()
            grown[:old.shape[0]] = old
            sd[key] = grown
        return self.load_state_dict(sd, strict=False)


class MidiQwenNew(_Wrapper):
    """4-D position model: `module(input_ids, attention_mask, labels, position_tensors)`; sampling through
    `module.model.generate_music(...)`. Its training script clips at 5.0 and uses lr 5e-4 / 100 warm-up steps
    (src/train_model_new.py:372-378)."""
    lr_default, warmup_default = 5e-5, 1000

    def __init__(self, tokenizer, dataloader, lr=None, warmup_steps=None, device="cuda", config: Optional[LMConfig] = None,
                 seed: Optional[int] = 0):
        super().__init__(tokenizer, dataloader, lr, warmup_steps)
        self.model = MidiLM(config or pos4d_decoder(), device=device, seed=seed)

    def forward(self, input_ids=None, attention_mask=None, labels=None, position_tensors=None):
        out = self.model(input_ids=input_ids, attention_mask=attention_mask, labels=labels,
                         position_tensors=position_tensors)
        if labels is not None:   # the reference logs the eight per-pair losses from inside this forward
            for name, value in self.model.component_losses().items():
                self.log(name, value)
        return out
