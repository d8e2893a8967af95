"""Batch assembly kernels (through the C-ABI) against the reference-generated fixture and the CPU oracle: bit-exact."""
import numpy as np
import pytest
import torch

from golden_util import load

pytestmark = pytest.mark.gpu


def _batch(g, with_melody):
    off = g["tok_offsets"]
    items = []
    for i in range(len(off) - 1):
        it = {"input_ids": g["tok_flat"][off[i]:off[i + 1]].tolist()}
        if with_melody:
            it["melody_length"] = int(g["tok_melody"][i])
        items.append(it)
    return items


def test_melody_and_style_collate_match_the_reference_fixture():
    from mgen import collate as K
    g = load("collate")
    pad = int(g["tok_pad_id"])
    o = K.collate_fn(_batch(g, True), pad)
    assert set(o) == {"input_ids", "labels"}
    assert np.array_equal(o["input_ids"].cpu().numpy(), g["melody_input_ids"])
    assert np.array_equal(o["labels"].cpu().numpy(), g["melody_labels"])
    o = K.style_collate_fn(_batch(g, False), pad)
    assert np.array_equal(o["input_ids"].cpu().numpy(), g["style_input_ids"])
    assert np.array_equal(o["labels"].cpu().numpy(), g["style_labels"])


def test_data_collator_matches_the_oracle():
    from mgen import collate as K
    from oracle import ref_collate as C
    g = load("collate")
    batch = _batch(g, False)
    pad = int(g["tok_pad_id"])
    ref = C.collate_miditok([b["input_ids"] for b in batch], pad)
    o = K.DataCollator(pad, copy_inputs_as_labels=True)(batch)
    for k in ("input_ids", "attention_mask", "labels"):
        assert o[k].dtype == torch.long
        assert np.array_equal(o[k].cpu().numpy(), ref[k]), k
    assert "labels" not in K.DataCollator(pad)(batch)


def test_collated_batch_trains():
    """The collator's output goes straight into the model call, as in the reference's training_step."""
    from mgen import collate as K
    from mgen.config import LMConfig
    from mgen.model import MidiLM
    cfg = LMConfig(vocab_size=17000, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                   num_attention_heads=2, num_key_value_heads=2, head_dim=128)
    m = MidiLM(cfg, device="cuda", seed=0)
    g = load("collate")
    o = K.DataCollator(int(g["tok_pad_id"]), copy_inputs_as_labels=True)(_batch(g, False))
    loss = m(**o).loss
    loss.backward()
    assert np.isfinite(float(loss))
    # same loss as the hand-made equivalent batch: pads ignored, real tokens counted
    ids = o["input_ids"].clone()
    lab = ids.clone()
    lab[o["attention_mask"] == 0] = -100
    with torch.no_grad():
        assert float(m(input_ids=ids, labels=lab).loss) == pytest.approx(float(loss), rel=1e-6)


def test_pos4d_chunks_match_the_reference_fixture():
    from mgen import collate as K
    g = load("collate")
    ds = K.MidiDataset4D(g["p4_rows"], int(g["p4_S"]))
    assert len(ds) == 5
    o = ds.batch(g["p4_chunk_ids"].tolist())
    for k in ("input_ids", "attention_mask", "labels", "position_tensors"):
        got = o[k].cpu().numpy()
        assert got.dtype == g["p4_" + k].dtype, k
        assert np.array_equal(got, g["p4_" + k]), k
    one = ds[4]
    assert np.array_equal(one["labels"].cpu().numpy(), g["p4_labels"][0])
    o2 = K.custom_collate_fn([ds[int(c)] for c in g["p4_chunk_ids"]])
    assert np.array_equal(o2["position_tensors"].cpu().numpy(), g["p4_position_tensors"])
    with pytest.raises(IndexError):
        ds.batch([5])


def test_large_ragged_batch_against_the_oracle():
    from mgen import collate as K
    from oracle import ref_collate as C
    rng = np.random.default_rng(5)
    lens = rng.integers(1, 2049, size=96)
    lens[7] = 2048
    seqs = [rng.integers(0, 17000, size=n).tolist() for n in lens]
    mel = [int(rng.integers(0, n + 1)) for n in lens]
    batch = [{"input_ids": s, "melody_length": m} for s, m in zip(seqs, mel)]
    ref = C.collate_melody(seqs, mel, 2)
    o = K.collate_fn(batch, 2)
    assert np.array_equal(o["input_ids"].cpu().numpy(), ref["input_ids"])
    assert np.array_equal(o["labels"].cpu().numpy(), ref["labels"])
    ref = C.collate_miditok(seqs, 2)
    o = K.DataCollator(2, copy_inputs_as_labels=True)(batch)
    for k in ref:
        assert np.array_equal(o[k].cpu().numpy(), ref[k]), k

    rows = np.cumsum(rng.random((100_003, 4), dtype=np.float32), axis=0)
    ds = K.MidiDataset4D(rows, 2048)
    ids = [len(ds) - 1, 0, 17, 48]
    ref = C.collate_pos4d(rows, ids, 2048)
    o = ds.batch(ids)
    for k in ref:
        assert np.array_equal(o[k].cpu().numpy(), ref[k]), k


def test_bad_arguments_fail_loudly():
    from mgen import collate as K
    with pytest.raises(ValueError):
        K.DataCollator(0)([])
    with pytest.raises(ValueError):
        K.MidiDataset4D(np.zeros((0, 4), np.float32), 16)
