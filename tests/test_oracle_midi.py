"""The CPU restatement of the note rows -> MIDI file step against a hand-assembled Standard MIDI File (the byte layout
is fixed by the SMF specification; the reference's writer package is absent, so this path is "parity unpinned")."""
import numpy as np

from oracle import ref_midi as R


def test_one_note_known_answer():
    ev = R.notes_to_events(np.array([[0.0, 0.5, 60.0, 100.0]], np.float32), tempo=120)
    assert ev == [(0, 0x90, 60, 100), (480, 0x80, 60, 0)]
    want = bytes.fromhex(
        "4d546864 00000006 0001 0001 01e0"      # MThd: type 1, one track, 480 ticks per beat
        "4d54726b 00000014"                     # MTrk, 20 bytes
        "00 ff5103 07a120"                      # set_tempo 500000 us per beat
        "00 903c64"                             # note on  60 vel 100
        "8360 803c00"                           # delta 480, note off
        "00 ff2f00")                            # end of track
    assert R.smf_bytes(ev, tempo=120) == want


def test_float32_tick_arithmetic_and_stable_order():
    # 0.1 s * 960 in float32 is 96.00000143..., (0.1 + 0.2) * 960 truncates to 288; equal ticks keep input order
    notes = np.array([[0.1, 0.2, 61.9, 127.6], [0.1, 0.2, -4.0, 3.0]], np.float32)
    ev = R.notes_to_events(notes)
    assert ev == [(96, 0x90, 61, 127), (96, 0x90, 0, 3), (288, 0x80, 61, 0), (288, 0x80, 0, 0)]
    # running status: the second note-on and the second note-off omit their status byte
    data = R.smf_bytes(ev)
    assert data[22 + 7:] == bytes.fromhex("60 903d7f 00 0003 8140 803d00 00 0000 00 ff2f00".replace(" ", ""))
