"""Generated note rows -> Standard MIDI File, behind the reference's name (`notes_to_midi`,
/root/reference/generate_music.py:62-105).

The per-note arithmetic (seconds -> ticks in fp32, clipping, note-on / note-off pairs) runs in one kernel on the rows
`generate_music` left in HBM (csrc/midi.hip); only the 2n small integers come to the host, where they are ordered by
tick (stable, as the reference's list sort) and framed as bytes. The byte layout is what the reference's writer (the
`mido` package) emits: type 1, 480 ticks per beat, one track = set_tempo, the events with variable-length delta times
and running status, end of track."""
import numpy as np
import torch

from . import ops

TICKS_PER_BEAT = 480


def _vlq(n):
    out = [n & 0x7F]
    n >>= 7
    while n:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    return bytes(reversed(out))


def note_events(notes, tempo=120, device="cuda"):
    """[n, 4] rows (tensor on the GPU, or anything torch.as_tensor takes) -> int arrays (tick, status, data1, data2),
    each [2n], ordered by tick with ties in input order."""
    rows = torch.as_tensor(notes, dtype=torch.float32)
    if rows.ndim != 2 or rows.shape[1] != 4 or rows.shape[0] == 0:
        raise ValueError("notes must be a non-empty [n, 4] array of (start s, duration s, pitch, velocity)")
    if not rows.is_cuda:
        rows = rows.to(device)
    rows = rows.contiguous()
    n = rows.shape[0]
    ticks = torch.empty(2 * n, dtype=torch.int32, device=rows.device)
    msgs = torch.empty(2 * n, dtype=torch.int32, device=rows.device)
    ops.notes_to_events(rows, (tempo / 60.0) * TICKS_PER_BEAT, ticks, msgs)
    t = ticks.cpu().numpy()
    m = msgs.cpu().numpy()
    order = np.argsort(t, kind="stable")
    t, m = t[order], m[order]
    return t, m & 0xFF, (m >> 8) & 0xFF, (m >> 16) & 0xFF


def midi_bytes(notes, tempo=120, device="cuda"):
    tick, status, d1, d2 = note_events(notes, tempo, device)
    us_per_beat = int(round(60 * 1e6 / tempo))
    data = bytearray(b"\x00\xff\x51\x03" + us_per_beat.to_bytes(3, "big"))
    last, running = 0, None
    for k in range(len(tick)):
        if tick[k] < last:
            raise ValueError("negative start time in the note rows")
        data += _vlq(int(tick[k]) - last)
        last = int(tick[k])
        if status[k] != running:
            data.append(int(status[k]))
        running = status[k]
        data += bytes([int(d1[k]), int(d2[k])])
    data += b"\x00\xff\x2f\x00"
    head = b"MThd" + (6).to_bytes(4, "big") + (1).to_bytes(2, "big") + (1).to_bytes(2, "big") + \
        TICKS_PER_BEAT.to_bytes(2, "big")
    return head + b"MTrk" + len(data).to_bytes(4, "big") + bytes(data)


def notes_to_midi(notes_array, output_path, tempo=120, device="cuda"):
    """Same arguments as the reference's function; `notes_array` may also be the CUDA tensor `generate_music` returns."""
    with open(output_path, "wb") as fh:
        fh.write(midi_bytes(notes_array, tempo, device))


def read_midi_events(data):
    """Parse a one-track file written by notes_to_midi back into (ticks_per_beat, us_per_beat, [(abs tick, status, d1,
    d2)]): the inverse used to check a written file."""
    if data[:4] != b"MThd" or data[14:18] != b"MTrk":
        raise ValueError("not a single-track Standard MIDI File")
    tpb = int.from_bytes(data[12:14], "big")
    end = 22 + int.from_bytes(data[18:22], "big")
    i, now, running, tempo, events = 22, 0, None, None, []
    while i < end:
        delta = 0
        while True:
            b = data[i]
            i += 1
            delta = (delta << 7) | (b & 0x7F)
            if not b & 0x80:
                break
        now += delta
        if data[i] == 0xFF:
            kind, ln = data[i + 1], data[i + 2]
            if kind == 0x51:
                tempo = int.from_bytes(data[i + 3:i + 3 + ln], "big")
            i += 3 + ln
            running = None
            continue
        if data[i] & 0x80:
            running = data[i]
            i += 1
        events.append((now, running, data[i], data[i + 1]))
        i += 2
    return tpb, tempo, events
