"""Fused clip + AdamW over MidiLM's flat parameter store, and the reference's optimizer factory.

Reference: `_get_optim` (/root/reference/src/model/model.py:11-46): AdamW(lr, weight_decay=0.1, betas=(0.9, 0.95),
eps=1e-5) over ALL parameters in one group, LinearLR warm-up (1e-6 -> 1 over `warmup` steps) then LinearLR decay
(1 -> end_ratio over num_epochs*steps_per_epoch - warmup steps) chained with SequentialLR and stepped every optimizer
step (:170). Gradient clipping to max-norm 1.0 is what the Trainer adds (src/train.py:66).
"""
import torch

from . import ops
from ._lib import lib


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics, executed as two kernel launches over the flat buffers:
    (1) global gradient norm + clip coefficient (kept on the device, no host sync), (2) fused update of
    p, m, v and of the bf16 shadow weights. torch LR schedulers drive `param_groups[0]['lr']` as usual."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=0.0):
        self.model = model
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        # with low-rank adapters (model.to_lora()) the base is frozen and the adapter store is what gets trained:
        # it has the same flat_param / flat_grad / flat_shadow protocol
        self.store = model.lora if getattr(model, "lora", None) is not None else model
        params = [self.store.anchor] if self.store is not model else [p for p in model.parameters()]
        super().__init__(params, defaults)
        dev = self.store.flat_param.device
        self.exp_avg = torch.zeros_like(self.store.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.store.flat_param)
        self.max_grad_norm = float(max_grad_norm)
        self._ws = torch.empty(lib().mgen_sumsq_partial_count(), dtype=torch.float64, device=dev)
        self.norm_and_coef = torch.ones(2, dtype=torch.float32, device=dev)  # {total norm, clip coefficient}
        self.step_count = 0

    @torch.no_grad()
    def clip_grad_norm_(self, max_norm=None):
        """Computes the total gradient L2 norm and the clip coefficient on the device; the scaling itself is
        folded into the update kernel. Returns the (device) norm tensor."""
        mn = self.max_grad_norm if max_norm is None else float(max_norm)
        ops.grad_norm_clip(self.store.flat_grad, mn, self._ws, self.norm_and_coef)
        return self.norm_and_coef[0]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if (getattr(self.model, "lora", None) is not None) != (self.store is not self.model):
            raise RuntimeError("the model's adapters changed after this optimizer was built (to_lora / merge_lora): "
                               "build the optimizer after them")
        g = self.param_groups[0]
        self.step_count += 1
        clip = None
        if self.max_grad_norm > 0:
            self.clip_grad_norm_()
            clip = self.norm_and_coef
        m = self.store
        ops.adamw_step(m.flat_param, m.flat_grad, self.exp_avg, self.exp_avg_sq, m.flat_shadow, clip,
                       float(g["lr"]), g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self.step_count)
        m.mark_shadow_fresh()
        return loss

    def zero_grad(self, set_to_none: bool = False):
        self.model.zero_grad()
        if self.store is not self.model:
            self.store.zero_grad()

    # The moments live in two flat buffers outside `self.state` (one kernel updates everything): a checkpoint has
    # to carry them, the step count (bias correction) and, with adapters, nothing else (the adapter weights are
    # part of the model's own state). Lightning checkpoints of the reference hold `optimizer_states` the same way.
    def state_dict(self):
        sd = super().state_dict()
        sd["fused"] = {"exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                       "step_count": self.step_count, "numel": self.exp_avg.numel()}
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        fused = state_dict.pop("fused", None)
        super().load_state_dict(state_dict)
        if fused is None:
            raise KeyError("FusedAdamW.load_state_dict: the 'fused' entry (moments and step count) is missing")
        if int(fused["numel"]) != self.exp_avg.numel():
            raise ValueError("FusedAdamW.load_state_dict: moment buffers of another parameter store "
                             f"({fused['numel']} elements, this one has {self.exp_avg.numel()})")
        self.exp_avg.copy_(fused["exp_avg"])
        self.exp_avg_sq.copy_(fused["exp_avg_sq"])
        self.step_count = int(fused["step_count"])


def get_optim(model, lr, num_epochs, steps_per_epoch, warmup=1000, end_ratio=0.1, max_grad_norm=1.0):
    """Mirror of the reference's `_get_optim`: returns (optimizer, scheduler); call scheduler.step() after every
    optimizer step."""
    opt = FusedAdamW(model, lr=lr, weight_decay=0.1, betas=(0.9, 0.95), eps=1e-5, max_grad_norm=max_grad_norm)
    total_steps = num_epochs * steps_per_epoch
    warm = torch.optim.lr_scheduler.LinearLR(opt, start_factor=1e-6, end_factor=1.0, total_iters=warmup)
    decay = torch.optim.lr_scheduler.LinearLR(opt, start_factor=1.0, end_factor=end_ratio,
                                              total_iters=max(total_steps - warmup, 1))
    sched = torch.optim.lr_scheduler.SequentialLR(opt, schedulers=[warm, decay], milestones=[warmup])
    return opt, sched
