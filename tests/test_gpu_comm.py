"""RCCL collectives behind the C ABI on one GPU: a communicator of one rank (RCCL wants one
GPU per rank, and the test box has one), so the sums and gathers are identities -- what is
checked is that RCCL binds at run time, the calls go onto the caller's stream, can be captured
into a HIP graph behind a kernel of this library, and that errors surface as exceptions.
Groups of several ranks are covered by the gloo tests of the same host code (test_tp_gloo.py)."""
import pytest
import torch

from baby_sglang_amd import _hip
from baby_sglang_amd import tp_runtime as tp

pytestmark = pytest.mark.gpu


@pytest.fixture()
def comm(gpu):
    torch.cuda.set_device(0)
    _hip.comm_init(0, 1, _hip.comm_unique_id())
    yield
    _hip.comm_destroy()
    assert _hip.comm_world() == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_single_rank_collectives(comm, dtype):
    assert _hip.comm_world() == 1
    x = torch.randn((64, 768), device="cuda").to(dtype)
    want = x.clone()
    assert _hip.all_reduce_sum(x) is x
    g = _hip.all_gather(x)
    torch.cuda.synchronize()
    assert torch.equal(x, want) and torch.equal(g, want) and g.data_ptr() != x.data_ptr()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        g2 = _hip.all_gather(x * 2)
    side.synchronize()
    assert torch.equal(g2, want * 2)


def test_collectives_inside_a_hip_graph(comm):
    """A normalisation launch of this library and the gather of its result, replayed from a graph."""
    x = torch.randn((8, 256), device="cuda", dtype=torch.bfloat16)
    w = torch.ones(256, device="cuda", dtype=torch.bfloat16)
    out = torch.empty((8, 256), device="cuda", dtype=torch.bfloat16)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        _hip.all_gather(_hip.rmsnorm(x, w, 1e-6)[0], out=out)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y = _hip.rmsnorm(x, w, 1e-6)[0]
        _hip.all_gather(y, out=out)
        _hip.all_reduce_sum(out)
    x.copy_(torch.randn((8, 256), device="cuda").to(torch.bfloat16))
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, _hip.rmsnorm(x, w, 1e-6)[0])


def test_errors(comm):
    with pytest.raises(ValueError):
        _hip.comm_init(0, 1, _hip.comm_unique_id())  # one communicator per process
    with pytest.raises(ValueError):
        _hip.comm_init(0, 1, b"short")
    with pytest.raises(NotImplementedError):
        _hip.all_reduce_sum(torch.zeros(4, device="cuda", dtype=torch.int32))
    with pytest.raises(ValueError):
        _hip.all_gather(torch.zeros(4, device="cuda"), out=torch.zeros(5, device="cuda"))


def test_wrappers_route_through_the_library(comm):
    """tp_all_gather / tp_all_reduce_sum over the chosen group use the library's communicator."""
    tp._CAPI_GROUP[0], tp._CAPI_GROUP[1] = None, True
    try:
        x = torch.arange(12, device="cuda", dtype=torch.float32).view(3, 4)
        # (no process group in this test: a world of one returns early, so call the routed branch directly)
        assert tp._capi(x, None) and not tp._capi(x.cpu(), None)
        assert torch.equal(_hip.all_gather(x), x)
    finally:
        tp._CAPI_GROUP[0], tp._CAPI_GROUP[1] = None, False
