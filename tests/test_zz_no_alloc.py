"""The entry point that allocates nothing (mgen_grad_norm_clip_static: partial sums in an array of the code object)
against the entry point it is a variant of (mgen_grad_norm_clip with a caller workspace), through the C-ABI: same
gradients, two calls in a row, the second with a smaller input. Both run the same two kernels in the same order, so
the results must be identical bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_no_alloc_grad_norm_matches_workspace_variant():
    from mgen._lib import call, lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    big = torch.randn(5_000_003 + 5, device=dev, generator=g)[: 5_000_000 + 3]   # more than 1024 workgroups' worth
    small = torch.randn(70_001, device=dev, generator=g) * 3.0                   # fewer partial sums than before
    ws = torch.empty(lib().mgen_sumsq_partial_count(), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for grad, max_norm in ((big, 1.0), (small, 5.0)):
        want = torch.full((2,), -1.0, device=dev)
        got = torch.full((2,), -2.0, device=dev)
        call("mgen_grad_norm_clip", grad.data_ptr(), grad.numel(), max_norm, ws.data_ptr(), want.data_ptr(), st)
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        call("mgen_grad_norm_clip_static", grad.data_ptr(), grad.numel(), max_norm, got.data_ptr(), st)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before          # the device's free memory did not move
        assert torch.equal(got, want), (got.tolist(), want.tolist())
        ref = float(torch.linalg.vector_norm(grad.double()))
        assert abs(float(got[0]) - ref) < 1e-5 * ref
        assert abs(float(got[1]) - min(1.0, max_norm / (ref + 1e-6))) < 1e-5
    from mgen._lib import MgenError
    with pytest.raises(MgenError, match="exceeds the supported maximum"):
        call("mgen_grad_norm_clip_static", big.data_ptr(), (1 << 36) + 4, 1.0, got.data_ptr(), st)
