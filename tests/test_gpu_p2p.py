"""The library's one-shot all-reduce (bsgl_p2p_*, behind tp_all_reduce_sum: tp_runtime.py:27-33) with
its ranks inside ONE process on one GPU: every rank has its own communicator, window and stream, and
the peers' windows are handed over as addresses.  What is checked is the protocol -- flags, slots,
call numbers, the bounded wait -- and the sum: every rank must hold the fp32 sum in rank order,
rounded once, bit for bit.  (Ranks in separate processes, windows mapped through IPC handles:
tests/test_gpu_tp_procs.py.)"""
import pytest
import torch

from baby_sglang_amd import _hip

pytestmark = pytest.mark.gpu


_STREAMS = []


def _make(world=2):
    """Two ranks: a call waits inside its kernel for the peer's kernel, so the two must be able to
    run side by side.  Streams of one process may share a hardware queue, in which case one launch
    sits behind the other; streams of different priority never do, which is why there are two ranks
    here and not more (ranks are processes where this is used: each has queues of its own)."""
    assert world == 2

    def pair():
        comms = [_hip.P2PComm(r, world) for r in range(world)]
        for c in comms:
            c.connect(windows=[p.window for p in comms])
        return comms

    if not _STREAMS:
        # found out once per session, with a short wait: do the two streams run side by side here?
        streams = [torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=0)]
        old = _hip.set_knob("BSGL_P2P_WAIT_MS", 1000)
        probe = pair()
        try:
            xs = [torch.ones(8, dtype=torch.bfloat16, device="cuda") for _ in probe]
            torch.cuda.synchronize()
            for c, s, x in zip(probe, streams, xs):
                with torch.cuda.stream(s):
                    c.all_reduce_sum(x)
            torch.cuda.synchronize()
            _STREAMS.append([c.status()[0] for c in probe] == [0, 0])
            _STREAMS.extend(streams)
        finally:
            _hip.set_knob("BSGL_P2P_WAIT_MS", old)
            for c in probe:
                c.destroy()
    if not _STREAMS[0]:
        pytest.skip("two streams of this process do not run side by side on this device: the ranks need processes of their own")
    return pair(), _STREAMS[1:]


def _expected(xs):
    acc = torch.zeros_like(xs[0], dtype=torch.float32)
    for x in xs:  # rank order, fp32, rounded once at the end
        acc = acc + x.float()
    return acc.to(xs[0].dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_two_ranks_every_size(gpu, dtype):
    comms, streams = _make()
    esz = torch.empty((), dtype=dtype).element_size()
    try:
        calls = 0
        # one 16-byte piece; a ragged number of pieces; 64 x 1536 rows of a 1.5B model; the largest message
        for nbytes in (16, 16 * 1031, 64 * 1536 * 2, _hip.P2P_MAX_BYTES):
            n = nbytes // esz
            for trial in range(3):  # (consecutive calls alternate between the two slots)
                g = torch.Generator(device="cpu").manual_seed(1000 * trial + n)
                xs = [torch.randn(n, generator=g).to(dtype).to(gpu) for _ in comms]
                want = _expected(xs)
                bufs = [x.clone() for x in xs]
                torch.cuda.synchronize()
                for c, s, b in zip(comms, streams, bufs):
                    with torch.cuda.stream(s):
                        assert c.all_reduce_sum(b) is b
                torch.cuda.synchronize()
                calls += 1
                for r, b in enumerate(bufs):
                    assert torch.equal(b.view(torch.uint8), want.view(torch.uint8)), (dtype, nbytes, trial, r)
        for c in comms:
            assert c.status() == (0, calls)
    finally:
        for c in comms:
            c.destroy()


def test_replayed_from_graphs(gpu):
    """A captured call holds no call number (it lives in the window): each rank's graph is replayed
    three times on the rank's stream, with a new addend every time."""
    comms, streams = _make()
    try:
        n = 64 * 1536
        bufs = [torch.zeros(n, dtype=torch.bfloat16, device=gpu) for _ in comms]
        graphs = []
        for c, b in zip(comms, bufs):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                c.all_reduce_sum(b)
            graphs.append(g)
        for trial in range(3):
            xs = [torch.randn(n, device=gpu).to(torch.bfloat16) for _ in comms]
            for b, x in zip(bufs, xs):
                b.copy_(x)
            torch.cuda.synchronize()
            for g, s in zip(graphs, streams):
                with torch.cuda.stream(s):
                    g.replay()
            torch.cuda.synchronize()
            want = _expected(xs)
            for b in bufs:
                assert torch.equal(b, want), trial
        assert [c.status() for c in comms] == [(0, 3), (0, 3)]
    finally:
        for c in comms:
            c.destroy()


def test_refusals(gpu):
    comms, _ = _make()
    try:
        c = comms[0]
        x = torch.zeros(1024, dtype=torch.bfloat16, device=gpu)
        with pytest.raises(NotImplementedError):  # 6 bytes: not a multiple of 16
            c.all_reduce_sum(x[:3])
        with pytest.raises(NotImplementedError):  # misaligned start
            c.all_reduce_sum(x[1:9])
        with pytest.raises(NotImplementedError):  # beyond the window's slots
            c.all_reduce_sum(torch.zeros(_hip.P2P_MAX_BYTES // 2 + 8, dtype=torch.bfloat16, device=gpu))
        with pytest.raises(NotImplementedError):
            c.all_reduce_sum(torch.zeros(16, dtype=torch.int32, device=gpu))
        assert not c.fits(x[:3]) and not c.fits(x[1:9]) and c.fits(x)
        with pytest.raises(ValueError):
            _hip.P2PComm(0, _hip.P2P_MAX_WORLD + 1)
        with pytest.raises(ValueError):
            _hip.P2PComm(2, 2)
        fresh = _hip.P2PComm(0, 2)
        try:
            with pytest.raises(ValueError):  # not connected yet
                fresh.all_reduce_sum(x)
            with pytest.raises(ValueError):
                fresh.connect()
        finally:
            fresh.destroy()
        assert c.status() == (0, 0)  # nothing above was launched
    finally:
        for c in comms:
            c.destroy()


def test_a_peer_that_never_arrives(gpu, knobs):
    """The wait for a peer is bounded: the call gives up (here after 300 ms instead of ten seconds),
    marks the communicator, and later calls return at once without touching their buffer."""
    import time

    knobs("BSGL_P2P_WAIT_MS", 300)
    comms, streams = _make()
    try:
        x = torch.ones(4096, dtype=torch.float32, device=gpu)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(streams[0]):
            comms[0].all_reduce_sum(x)  # rank 1 never calls
        torch.cuda.synchronize()
        waited = time.perf_counter() - t0
        assert 0.25 < waited < 5.0, waited
        assert comms[0].status() == (1, 1)
        y = torch.full((4096,), 3.0, dtype=torch.float32, device=gpu)
        t0 = time.perf_counter()
        with torch.cuda.stream(streams[0]):
            comms[0].all_reduce_sum(y)
        torch.cuda.synchronize()
        assert time.perf_counter() - t0 < 0.25
        assert bool((y == 3.0).all()) and comms[0].status() == (1, 2)
        # the wrapper tells its caller: a waited check raises, and so does the call that follows a
        # finished status probe (a probe goes out every `probe_every` calls and is never waited for)
        with pytest.raises(RuntimeError, match="a peer did not arrive"):
            comms[0].check()
        comms[0].probe_every = 2
        with torch.cuda.stream(streams[0]):
            comms[0].all_reduce_sum(y)
            comms[0].all_reduce_sum(y)  # the second call enqueues the probe behind itself
            torch.cuda.synchronize()
            with pytest.raises(RuntimeError, match="a peer did not arrive"):
                comms[0].all_reduce_sum(y)
    finally:
        for c in comms:
            c.destroy()


def test_eager_warm_up_then_capture_then_replay(gpu):
    """The usual way to a graph: eager calls on the current stream first, then the capture (on the
    capture's own stream), then replays.  The wrapper makes a new stream wait for the stream of the
    communicator's last call -- but not while that new stream is being captured (a wait for work
    outside the capture is no legal capture operation)."""
    comms, streams = _make()
    try:
        n = 64 * 1536
        bufs = [torch.zeros(n, dtype=torch.bfloat16, device=gpu) for _ in comms]

        def run_eager(xs):
            for b, x in zip(bufs, xs):
                b.copy_(x)
            torch.cuda.synchronize()
            for c, s, b in zip(comms, streams, bufs):
                with torch.cuda.stream(s):
                    c.all_reduce_sum(b)
            torch.cuda.synchronize()

        xs = [torch.randn(n, device=gpu).to(torch.bfloat16) for _ in comms]
        run_eager(xs)  # warm-up on the ranks' streams
        for b in bufs:
            assert torch.equal(b, _expected(xs))
        graphs = []
        for c, b in zip(comms, bufs):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):  # (a capture stream of torch's own: not the stream of the warm-up)
                c.all_reduce_sum(b)
            graphs.append(g)
        for trial in range(2):
            xs = [torch.randn(n, device=gpu).to(torch.bfloat16) for _ in comms]
            for b, x in zip(bufs, xs):
                b.copy_(x)
            torch.cuda.synchronize()
            for g, s in zip(graphs, streams):
                with torch.cuda.stream(s):
                    g.replay()
            torch.cuda.synchronize()
            for b in bufs:
                assert torch.equal(b, _expected(xs)), trial
        assert [c.status() for c in comms] == [(0, 3), (0, 3)]
        for c in comms:
            c.check()
    finally:
        for c in comms:
            c.destroy()
