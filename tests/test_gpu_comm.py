"""The RCCL communicator behind the C-ABI (mgen_comm_*) on the one GPU of the test machine: a one-rank communicator is
created, a bucket is reduced (mean over one rank = identity, but the whole path runs: id, init, event ordering
between the compute stream and the communicator's stream, all-reduce, wait, destroy). The N > 1 logic of the reducer is
covered on the CPU with gloo (tests/test_ddp_reducer_gloo.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_reduces_in_stream_order():
    from mgen.comm import NativeComm, rccl_version
    assert rccl_version() >= 20000
    dev = torch.device("cuda:0")
    comm = NativeComm(rank=0, world=1, device=0)
    x = torch.zeros(5_000_003, dtype=torch.float32, device=dev)
    x.add_(3.0)                       # queued on the compute stream: the reduction must see it
    comm.allreduce_(x, average=True)
    y = torch.ones(70_001, dtype=torch.float32, device=dev) * 2
    comm.allreduce_(y, average=False)
    comm.wait()
    x.mul_(2.0)                       # ordered behind the reduction
    torch.cuda.synchronize()
    assert comm.last_buckets == 2 and comm.last_bytes == 4 * (5_000_003 + 70_001)
    assert float(x.min()) == 6.0 and float(x.max()) == 6.0
    assert float(y.min()) == 2.0 and float(y.max()) == 2.0
    with pytest.raises(ValueError):
        comm.allreduce_(x.to(torch.bfloat16))
    comm.close()
