"""Note rows -> MIDI events / file bytes through the C-ABI against the CPU oracle (bit-exact). The oracle itself is
"parity unpinned" for the byte layout (the reference writes through the absent `mido` package): the file is also parsed
back and compared event by event."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _notes(n, seed):
    rng = np.random.default_rng(seed)
    start = np.cumsum(rng.exponential(0.21, n)).astype(np.float32)
    start[rng.random(n) < 0.3] = start[0]                    # chords and ties in the sort key
    start = np.sort(start)
    dur = (0.03 + 2.5 * rng.random(n)).astype(np.float32)
    pitch = rng.uniform(-5, 140, n).astype(np.float32)       # out-of-range values exercise the clipping
    vel = rng.uniform(-3, 135, n).astype(np.float32)
    return np.stack([start, dur, pitch, vel], 1)


@pytest.mark.parametrize("n,tempo", [(1, 120), (257, 120), (5000, 96)])
def test_events_match_the_oracle(n, tempo):
    from mgen import midi
    from oracle import ref_midi as R
    notes = _notes(n, n)
    ref = R.notes_to_events(notes, tempo)
    t, s, d1, d2 = midi.note_events(torch.from_numpy(notes).cuda(), tempo)
    got = list(zip(t.tolist(), s.tolist(), d1.tolist(), d2.tolist()))
    assert got == ref


def test_file_bytes_and_round_trip(tmp_path):
    from mgen import midi
    from oracle import ref_midi as R
    notes = _notes(300, 7)
    path = tmp_path / "out.mid"
    midi.notes_to_midi(notes, str(path))                     # numpy input, as the reference passes it
    data = path.read_bytes()
    assert data == R.smf_bytes(R.notes_to_events(notes))
    tpb, us, events = midi.read_midi_events(data)
    assert tpb == 480 and us == 500000
    assert events == R.notes_to_events(notes)
    # overlapping notes of one pitch keep both note-offs (the reason the reference writes ON/OFF pairs itself)
    two = np.array([[0.0, 1.0, 60, 100], [0.5, 1.0, 60, 90]], np.float32)
    ev = midi.read_midi_events(midi.midi_bytes(two))[2]
    assert [(e[1], e[2]) for e in ev] == [(0x90, 60), (0x90, 60), (0x80, 60), (0x80, 60)]
    assert [e[0] for e in ev] == [0, 480, 960, 1440]


def test_generated_rows_go_straight_to_a_file(tmp_path):
    """generate_music's output tensor (on the GPU) is accepted as it is."""
    from mgen import midi
    rows = torch.tensor([[0.0, 0.5, 60.0, 80.0], [0.5, 0.25, 64.2, 70.9]], device="cuda")
    midi.notes_to_midi(rows, str(tmp_path / "g.mid"), tempo=100)
    tpb, us, ev = midi.read_midi_events((tmp_path / "g.mid").read_bytes())
    assert us == 600000 and len(ev) == 4 and ev[1][2] == 60 and ev[2][2] == 64 and ev[2][3] == 70


def test_bad_input_fails_loudly():
    from mgen import midi
    with pytest.raises(ValueError):
        midi.note_events(np.zeros((0, 4), np.float32))
    with pytest.raises(ValueError):
        midi.note_events(np.zeros((3, 5), np.float32))
