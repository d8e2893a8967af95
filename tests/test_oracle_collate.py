"""The CPU restatement of the reference's batch assembly against the fixture made by running the reference's own
collate functions (tests/golden/collate.npz, tools/make_golden.py:collate_case)."""
import numpy as np

from oracle import ref_collate as C
from golden_util import load


def _seqs(g):
    off = g["tok_offsets"]
    return [g["tok_flat"][off[i]:off[i + 1]].tolist() for i in range(len(off) - 1)]


def test_melody_and_style_collate_match_the_reference():
    g = load("collate")
    seqs, pad = _seqs(g), int(g["tok_pad_id"])
    o = C.collate_melody(seqs, g["tok_melody"].tolist(), pad)
    assert np.array_equal(o["input_ids"], g["melody_input_ids"])
    assert np.array_equal(o["labels"], g["melody_labels"])
    o = C.collate_style(seqs, pad)
    assert np.array_equal(o["input_ids"], g["style_input_ids"])
    assert np.array_equal(o["labels"], g["style_labels"])


def test_pos4d_chunks_match_the_reference():
    g = load("collate")
    o = C.collate_pos4d(g["p4_rows"], g["p4_chunk_ids"].tolist(), int(g["p4_S"]))
    for k in ("input_ids", "attention_mask", "labels", "position_tensors"):
        assert o[k].dtype == g["p4_" + k].dtype, k
        assert np.array_equal(o[k], g["p4_" + k]), k


def test_miditok_layout_properties():
    # no fixture exists for the third-party collator (parity unpinned): check the documented layout only
    seqs = [[5, 6, 7], [8], [9, 10, 11, 12, 13]]
    o = C.collate_miditok(seqs, pad_id=0)
    assert o["input_ids"].shape == (3, 5)
    assert o["attention_mask"].sum() == 9
    assert np.array_equal(o["labels"][1], [8, -100, -100, -100, -100])
    assert np.array_equal(o["input_ids"][1], [8, 0, 0, 0, 0])
