"""CPU restatement of the reference's note rows -> MIDI file step. TEST INFRASTRUCTURE ONLY: nothing in the product path
may import this file.

PARITY UNPINNED: the reference writes the file through the third-party package `mido` (pinned 1.3.3 in the reference's
uv.lock), which is absent here, and ships no generated .mid file. The event arithmetic below follows the reference's
own lines; the byte layout follows the Standard MIDI File format as mido's writer emits it (type 1, 480 ticks per
beat, one track, running status for repeated channel-status bytes, an end-of-track meta event appended).
"""
import numpy as np

TICKS_PER_BEAT = 480          # mido.MidiFile() default


def notes_to_events(notes, tempo=120):
    """/root/reference/generate_music.py:75-94. `notes` float32 [n, 4]; numpy >= 2 keeps np.float32 * python float in
    float32. Returns the (tick, status, data1, data2) list sorted by tick, stably."""
    sec_to_ticks = (tempo / 60.0) * TICKS_PER_BEAT
    events = []
    for start, duration, pitch, velocity in np.asarray(notes, np.float32):
        start_ticks = int(np.float32(start) * np.float32(sec_to_ticks))
        end_ticks = int((np.float32(start) + np.float32(duration)) * np.float32(sec_to_ticks))
        p = int(np.clip(pitch, 0, 127))
        v = int(np.clip(velocity, 0, 127))
        events.append((start_ticks, 0x90, p, v))
        events.append((end_ticks, 0x80, p, 0))
    events.sort(key=lambda e: e[0])
    return events


def _vlq(n):
    out = [n & 0x7F]
    n >>= 7
    while n:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    return bytes(reversed(out))


def smf_bytes(events, tempo=120):
    """/root/reference/generate_music.py:68-73,96-104: set_tempo first, then the events with delta times."""
    us_per_beat = int(round(60 * 1e6 / tempo))          # mido.bpm2tempo
    data = bytearray(b"\x00\xff\x51\x03" + us_per_beat.to_bytes(3, "big"))
    last, running = 0, None
    for tick, status, d1, d2 in events:
        data += _vlq(tick - last)
        last = tick
        if status != running:
            data.append(status)
        running = status
        data += bytes([d1, d2])
    data += b"\x00\xff\x2f\x00"
    head = b"MThd" + (6).to_bytes(4, "big") + (1).to_bytes(2, "big") + (1).to_bytes(2, "big") + \
        TICKS_PER_BEAT.to_bytes(2, "big")
    return head + b"MTrk" + len(data).to_bytes(4, "big") + bytes(data)
