"""Minimal trainer reproducing the arithmetic slice of the reference's `lightning.Trainer` configuration
(/root/reference/src/train.py:63-72): bf16-mixed forward/backward, gradient clipping to max-norm 1.0, AdamW with the
warm-up/decay schedule stepped every optimizer step, and data-parallel gradient averaging when launched with one
process per GPU (what Lightning's DDP strategy does implicitly).

Data parallelism on MI355X: each rank owns a full replica; the only exchange is the mean of the fp32 gradients.
MidiLM keeps all gradients in one flat buffer and finishes them in contiguous ranges (head, layers L-1..0, embedding),
so each range is handed to RCCL as ONE large all-reduce the moment backward has produced it (few, large collectives
suit point-to-point xGMI links), on RCCL's own stream, overlapping the rest of backward. The optimizer waits for
the outstanding reductions only.
"""
import os
from typing import Optional

import torch
import torch.distributed as dist

from .optim import get_optim


class GradReducer:
    """Bucketed, overlapped gradient mean over the data-parallel group."""

    def __init__(self, model, group=None, min_bucket_elems: int = 4 * 1024 * 1024):
        self.model = model
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.min_bucket = min_bucket_elems
        self._pending = []
        self._open: Optional[list] = None  # [start, end) of the bucket being grown (ranges arrive high -> low)
        self.bucket_log = []   # (bytes, dtype) of every collective of the last step, in issue order
        self._ranges = {tag: (s, e) for tag, s, e in model.grad_ranges()}
        # RCCL averages in the collective; gloo (CPU rehearsal of the same logic) has no AVG: sum, then divide
        self._native_avg = self.world > 1 and dist.get_backend(group) == "nccl"
        # MGEN_NATIVE_COMM=1: the buckets go through the library's own RCCL communicator (mgen_comm_*) instead of
        # torch.distributed's; same collective, same overlap (RCCL's stream beside the compute stream)
        self._comm = None
        if self._native_avg and os.environ.get("MGEN_NATIVE_COMM") == "1":
            from .comm import NativeComm
            self._comm = NativeComm(group=group)
        self._lora = getattr(model, "lora", None)   # adapters: one small buffer, reduced in wait()
        # Tied embedding / head: the head's weight gradient arrives as a buffer of its own at the START of backward
        # (first bucket, reduced under the whole backward); the embedding lookup's scatter-add stays the last bucket,
        # cut down to the rows [0, embed_rows) that any rank's batch can have touched (set_embed_rows).
        self._tied = []            # (reduced head-gradient buffer, destination view) of this step
        self._embed_rows = None    # () -> row count, or None: the whole table
        self._max_host = self._max_event = None
        if self.world > 1 and self._lora is None:
            model._grad_hook = self._on_range_done
            if hasattr(model, "_lm_head_name") and getattr(model.cfg, "tie_word_embeddings", False) \
                    and os.environ.get("MGEN_DDP_SPLIT_TIED", "1") != "0":
                model._head_grad_sink = self._on_head_grad
                model._head_grad_in_flight = lambda: bool(self._tied)

    def _flush(self):
        if self._open is None:
            return
        s, e = self._open
        self._open = None
        buf = self.model.flat_grad[s:e]
        self.bucket_log.append((buf.numel() * buf.element_size(), str(buf.dtype).replace("torch.", "")))
        self._pending.append((self._start(buf), buf))

    def _start(self, buf):
        """Begin the reduction of one slice; returns something with .wait()."""
        if self._comm is not None:
            self._comm.allreduce_(buf, average=True)
            return self._comm
        if self._native_avg:
            return dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        if buf.is_cuda:
            # gloo rehearsal on device tensors (several ranks sharing one GPU): staged through host memory, blocking.
            # Only the control flow is like the RCCL path.
            host = buf.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            buf.copy_(host)

            class _Done:
                def wait(self):
                    return None
            return _Done()
        return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _on_head_grad(self, buf, dest):
        flat = buf.view(-1)
        self.bucket_log.append((flat.numel() * flat.element_size(), str(flat.dtype).replace("torch.", "")))
        self._pending.append((self._start(flat), flat))
        self._tied.append((buf, dest))

    def note_token_ids(self, ids):
        """Call at the start of a step with this rank's input ids. With a tied head, the largest token id over all
        ranks bounds the embedding rows whose lookup gradient can be non-zero, so the last gradient bucket (the only
        one nothing overlaps) shrinks from the whole [V, d] table to those rows (the reference's default decoder
        keeps a 151 936-row table for token ids below 30 000). The maximum is reduced now and read when the bucket
        is sent."""
        if self.world <= 1 or ids is None or getattr(self.model, "_head_grad_sink", None) is None:
            return
        mx = ids.max().to(torch.int64).reshape(1)
        if self._native_avg:
            dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=self.group)
            if self._max_host is None:
                self._max_host = torch.zeros(1, dtype=torch.int64).pin_memory()
                self._max_event = torch.cuda.Event()
            self._max_host.copy_(mx, non_blocking=True)
            self._max_event.record()

            def rows():
                self._max_event.synchronize()
                return int(self._max_host) + 1
        else:   # gloo rehearsal: staged through the host
            host = mx.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.MAX, group=self.group)
            val = int(host) + 1

            def rows():
                return val
        self.set_embed_rows(rows)

    def set_embed_rows(self, fn):
        """fn() -> number of leading embedding rows that can hold a lookup gradient on ANY rank (the same number on
        every rank), or None. Called when the embedding bucket is about to be sent."""
        self._embed_rows = fn

    def _on_range_done(self, tag):
        s, e = self._ranges[tag]
        if tag == "embed" and self._tied and self._embed_rows is not None:
            rows = self._embed_rows()
            if rows is not None:
                d = self._tied[0][1].shape[1]
                e = min(e, s + int(rows) * d)
        if self._open is not None and self._open[0] == e:
            self._open[0] = s          # ranges are adjacent and arrive in descending address order
        else:
            self._flush()
            self._open = [s, e]
        if self._open[1] - self._open[0] >= self.min_bucket or tag == "embed":
            self._flush()

    def wait(self):
        if getattr(self.model, "lora", None) is not self._lora:
            raise RuntimeError("the model's adapters changed after this reducer was built (to_lora / merge_lora): "
                               "build the Trainer after them")
        if self._lora is not None and self.world > 1:
            buf = self._lora.flat_grad
            self.bucket_log.append((buf.numel() * buf.element_size(), str(buf.dtype).replace("torch.", "")))
            self._pending.append((self._start(buf), buf))
        self._flush()
        waited_comm = False
        for w, buf in self._pending:
            if w is self._comm:
                if not waited_comm:   # one stream-side wait covers every bucket enqueued so far
                    w.wait()
                    waited_comm = True
            else:
                w.wait()
            if not self._native_avg:
                buf.div_(self.world)
        self._pending.clear()
        for buf, dest in self._tied:   # reduced head gradient + reduced lookup gradient = the tied tensor's gradient
            dest.add_(buf)
        self._tied.clear()


class Trainer:
    """fit-loop pieces: `training_step(batch)` = zero_grad, forward, backward (+ overlapped all-reduce), clip,
    AdamW, scheduler step. `batch` is a dict with input_ids / labels (/ attention_mask) on the device."""

    def __init__(self, model, lr=5e-5, num_epochs=12, steps_per_epoch=1000, warmup=1000, max_grad_norm=1.0,
                 group=None):
        self.model = model
        self.opt, self.sched = get_optim(model, lr, num_epochs, steps_per_epoch, warmup, max_grad_norm=max_grad_norm)
        self.reducer = GradReducer(model, group)
        if hasattr(model, "defer_input_checks"):
            model.defer_input_checks = True   # out-of-range ids / labels / masks raise at the end of training_step
        for holder in (model, getattr(model, "lora", None)):   # every replica draws its own dropout masks
            if hasattr(holder, "dropout_seed"):
                holder.dropout_seed += 1000003 * self.reducer.rank

    def training_step(self, batch):
        self.model.train()
        self.reducer.bucket_log.clear()
        self.reducer.note_token_ids(batch.get("input_ids") if isinstance(batch, dict) else None)
        self.opt.zero_grad()
        loss = self.model(**batch).loss
        loss.backward()
        self.reducer.wait()
        self.opt.step()
        self.sched.step()
        if hasattr(self.model, "raise_pending_errors"):
            self.model.raise_pending_errors()   # (the optimizer step is queued behind backward by now)
        return loss.detach()

    @torch.no_grad()
    def validation_step(self, batch):
        self.model.eval()
        loss = self.model(**batch).loss
        if self.reducer.world > 1:  # self.log(..., sync_dist=True): mean over ranks
            if self.reducer._native_avg:
                dist.all_reduce(loss, op=dist.ReduceOp.AVG, group=self.reducer.group)
            else:   # gloo has no AVG
                loss = loss.clone()
                host = loss.cpu() if loss.is_cuda else loss
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.reducer.group)
                loss = (host / self.reducer.world).to(loss.device)
        return loss


def init_distributed_from_env():
    """One process per GPU, rendezvous from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # MGEN_REHEARSE_ON_ONE_GPU=1: every rank uses device 0 and the exchange goes through gloo (RCCL refuses two ranks
    # on one device) - for running the multi-rank code path on a one-GPU machine, not for measurements
    rehearse = os.environ.get("MGEN_REHEARSE_ON_ONE_GPU") == "1"
    if rehearse:
        local = 0
    if torch.cuda.is_available():
        if torch.cuda.device_count() == 1:
            local = 0   # a launcher that shows every rank only its own GPU
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        backend = "nccl" if torch.cuda.is_available() and not rehearse else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world
