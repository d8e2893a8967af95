"""Does issuing the launches of two batch parts from two host threads raise the launch rate? Each thread issues
mgen_decode_advance (a trivial kernel) on its own stream through ctypes (the GIL is released inside the call)."""
import os
import sys
import threading
import time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "symbolic-music-generation_amd"))
from mgen._lib import lib

l = lib()
dev = "cuda:0"
N = 20000
bufs = [(torch.zeros(256, dtype=torch.int32, device=dev), torch.zeros(256, dtype=torch.int32, device=dev)) for _ in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]


def issue(i, n):
    p, k = bufs[i]
    s = streams[i].cuda_stream
    f = l.mgen_decode_advance
    pp, kp = p.data_ptr(), k.data_ptr()
    for _ in range(n):
        f(pp, kp, 256, None, s)


issue(0, 100); issue(1, 100)
torch.cuda.synchronize()
t0 = time.perf_counter()
issue(0, N); issue(1, N)
t1 = time.perf_counter()
torch.cuda.synchronize()
t1s = time.perf_counter()
print(f"one thread : {2 * N} launches issued in {(t1 - t0) * 1e3:.1f} ms = {(t1 - t0) / (2 * N) * 1e6:.2f} us per launch "
      f"(GPU done after {(t1s - t0) * 1e3:.1f} ms)")
ths = [threading.Thread(target=issue, args=(i, N)) for i in range(2)]
t0 = time.perf_counter()
for t in ths:
    t.start()
for t in ths:
    t.join()
t1 = time.perf_counter()
torch.cuda.synchronize()
t1s = time.perf_counter()
print(f"two threads: {2 * N} launches issued in {(t1 - t0) * 1e3:.1f} ms = {(t1 - t0) / (2 * N) * 1e6:.2f} us per launch "
      f"(GPU done after {(t1s - t0) * 1e3:.1f} ms)")
