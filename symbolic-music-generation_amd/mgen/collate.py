"""Batch assembly on the GPU, behind the names the reference's training scripts use.

The reference pads / masks / shifts every batch on the host inside DataLoader workers and then copies four tensors to
the device. Here the examples are shipped once as one flat buffer (or are already resident, e.g. a tokenised corpus kept
in HBM) and the padded ids, attention mask and labels are written by one kernel launch (csrc/collate.hip).

  DataCollator(pad_token_id, copy_inputs_as_labels=True)   miditok collator used at /root/reference/src/train.py:33-35
  collate_fn(batch, pad_token_id)                          /root/reference/src/train_aria.py:118-145
  style_collate_fn(batch, pad_token_id)                    /root/reference/src/train_aria.py:208-221
  MidiDataset4D(rows, max_seq_len) + custom_collate_fn     /root/reference/src/train_model_new.py:114-158,291-303
"""
import torch

from . import ops


def _flatten(batch, device):
    """list of {"input_ids": list | tensor} -> (flat int64 on device, offsets int64 [B+1] on device, longest)."""
    seqs = [torch.as_tensor(item["input_ids"], dtype=torch.long).reshape(-1) for item in batch]
    if not seqs:
        raise ValueError("empty batch")
    lens = [int(s.numel()) for s in seqs]
    if max(lens) == 0:
        raise ValueError("every example of the batch is empty")
    off = torch.zeros(len(seqs) + 1, dtype=torch.long)
    off[1:] = torch.tensor(lens, dtype=torch.long).cumsum(0)
    flat = torch.cat(seqs)
    if not flat.is_cuda:
        flat = flat.pin_memory().to(device, non_blocking=True)
    return flat, off.to(device), max(lens)


def collate_tokens(flat, offsets, seq_len, pad_token_id, mask_pads, prompt_lengths=None, want_mask=True):
    """Device-resident form: flat int64 [sum of lengths], offsets int64 [B+1] (both on the GPU), seq_len >= the longest
    example. Returns (input_ids, attention_mask | None, labels), each int64 [B, seq_len]."""
    B = offsets.numel() - 1
    dev = flat.device
    ids = torch.empty(B, seq_len, dtype=torch.long, device=dev)
    labels = torch.empty_like(ids)
    mask = torch.empty_like(ids) if want_mask else None
    pl = None
    if prompt_lengths is not None:
        pl = torch.as_tensor(prompt_lengths, dtype=torch.int32).to(dev)
    ops.collate_tokens(flat, offsets, pl, ids, mask, labels, pad_token_id, mask_pads)
    return ids, mask, labels


class DataCollator:
    """Right-pads to the longest example; attention_mask marks real tokens; with copy_inputs_as_labels the labels are
    the inputs with -100 on the padding (the shift by one happens inside the loss, as in the reference's models)."""

    def __init__(self, pad_token_id, copy_inputs_as_labels=False, device="cuda"):
        self.pad_token_id = int(pad_token_id)
        self.copy_inputs_as_labels = copy_inputs_as_labels
        self.device = torch.device(device)

    def __call__(self, batch):
        flat, off, S = _flatten(batch, self.device)
        ids, mask, labels = collate_tokens(flat, off, S, self.pad_token_id, mask_pads=True)
        out = {"input_ids": ids, "attention_mask": mask}
        if self.copy_inputs_as_labels:
            out["labels"] = labels
        return out


def collate_fn(batch, pad_token_id, device="cuda"):
    """Melody-conditioned batches: labels = padded inputs with the first melody_length positions ignored; the padding
    keeps its token id in the labels and no attention mask is returned."""
    dev = torch.device(device)
    flat, off, S = _flatten(batch, dev)
    ids, _, labels = collate_tokens(flat, off, S, pad_token_id, mask_pads=False,
                                    prompt_lengths=[int(item["melody_length"]) for item in batch], want_mask=False)
    return {"input_ids": ids, "labels": labels}


def style_collate_fn(batch, pad_token_id, device="cuda"):
    dev = torch.device(device)
    flat, off, S = _flatten(batch, dev)
    ids, _, labels = collate_tokens(flat, off, S, pad_token_id, mask_pads=False, want_mask=False)
    return {"input_ids": ids, "labels": labels}


class MidiDataset4D:
    """Fixed-size chunks of the concatenated note rows [total, 4] = (start s, duration s, pitch, velocity). The rows
    stay in HBM; a batch is cut, padded and turned into labels by one launch. `rows` replaces the MIDI file list of
    the reference's constructor: reading .mid files is outside this library."""

    def __init__(self, rows, max_seq_len, device="cuda"):
        rows = torch.as_tensor(rows, dtype=torch.float32)
        if rows.ndim != 2 or rows.shape[1] != 4 or rows.shape[0] == 0:
            raise ValueError("rows must be a non-empty [total, 4] array")
        self.rows = rows.to(device).contiguous()
        self.max_seq_len = int(max_seq_len)

    def __len__(self):
        return (self.rows.shape[0] + self.max_seq_len - 1) // self.max_seq_len

    def batch(self, chunk_ids):
        cid = torch.as_tensor(chunk_ids, dtype=torch.long)
        if cid.numel() == 0 or int(cid.min()) < 0 or int(cid.max()) >= len(self):
            raise IndexError("chunk index out of range")
        dev = self.rows.device
        B, S = cid.numel(), self.max_seq_len
        pos = torch.empty(B, S, 4, dtype=torch.float32, device=dev)
        labels = torch.empty_like(pos)
        mask = torch.empty(B, S, dtype=torch.long, device=dev)
        ids = torch.empty_like(mask)
        ops.collate_pos4d(self.rows, cid.to(dev), pos, labels, mask, ids)
        return {"input_ids": ids, "attention_mask": mask, "labels": labels, "position_tensors": pos}

    def __getitem__(self, idx):
        return {k: v[0] for k, v in self.batch([idx]).items()}


def custom_collate_fn(batch):
    """Stack per-chunk dicts (what MidiDataset4D.__getitem__ returns); MidiDataset4D.batch does both steps at once."""
    return {k: torch.stack([item[k] for item in batch]) for k in
            ("input_ids", "attention_mask", "labels", "position_tensors")}
