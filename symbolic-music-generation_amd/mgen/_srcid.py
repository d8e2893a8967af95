"""Identity of the kernel sources: the first 16 hex digits of a SHA-256 over the names and contents of
symbolic-music-generation_amd/csrc/{*.hip,*.h,Makefile}. The Makefile compiles it into the library
(mgen_source_id); the loader (mgen/_lib.py) and a test compare it with the sources beside the library, so that a library left over
from older sources (a failed or interrupted build) cannot pass for the current one."""
import hashlib
import os
import sys


def source_id(csrc: str) -> str:
    names = sorted(n for n in os.listdir(csrc) if n.endswith((".hip", ".h")) or n == "Makefile")
    h = hashlib.sha256()
    for n in names:
        h.update(n.encode() + b"\0")
        with open(os.path.join(csrc, n), "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    print(source_id(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "csrc")))
