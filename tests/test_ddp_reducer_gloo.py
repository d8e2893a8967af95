"""world_size-2 rehearsal of the data-parallel gradient reducer on CPU (gloo): buckets are formed from the ranges
backward reports, reduced asynchronously, and every element ends as the mean over ranks exactly once."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _FakeModel:
    def __init__(self, rank):
        self.flat_grad = torch.arange(1000, dtype=torch.float32) * (rank + 1)

    def grad_ranges(self):
        return [("head", 900, 1000), (2, 600, 900), (1, 300, 600), (0, 100, 300), ("embed", 0, 100)]


def _worker(rank, world, port, min_bucket, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "symbolic-music-generation_amd"))
    from mgen.trainer import GradReducer
    m = _FakeModel(rank)
    red = GradReducer(m, min_bucket_elems=min_bucket)
    calls = []
    orig = dist.all_reduce

    def spy(t, *a, **k):
        calls.append(t.numel())
        return orig(t, *a, **k)

    dist.all_reduce = spy
    for tag, _, _ in m.grad_ranges():
        m._grad_hook(tag)
    red.wait()
    dist.all_reduce = orig
    expect = torch.arange(1000, dtype=torch.float32) * (1 + 2) / 2
    q.put((rank, bool(torch.allclose(m.flat_grad, expect)), calls))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(min_bucket):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, min_bucket, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    return res


def test_every_range_reduced_once_per_layer_buckets():
    for rank, ok, calls in _run(1):
        assert ok
        assert calls == [100, 300, 300, 200, 100]


def test_small_ranges_are_merged_into_large_buckets():
    for rank, ok, calls in _run(450):
        assert ok
        assert calls == [700, 300] or sum(calls) == 1000 and len(calls) < 5
