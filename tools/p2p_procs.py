"""One rank of the one-shot all-reduce as a PROCESS of its own (several of them share the one GPU):
init_tp() from the torchrun-style environment with gloo as the store, use_p2p_all_reduce() -- windows
mapped through IPC handles --, then tp_all_reduce_sum (reference: tp_runtime.py:27-33) over three
dtypes, four message sizes and three calls each; every rank's result must be the fp32 sum in rank
order, rounded once, bit for bit.  Started by tests/conftest.py (tests/test_gpu_tp_procs.py waits
for it) or from a shell:  WORLD_SIZE=4 RANK=r MASTER_ADDR=127.0.0.1 MASTER_PORT=p BABYSGL_TP_BACKEND=gloo
python tools/p2p_procs.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import baby_sglang_amd as pkg  # noqa: E402
from baby_sglang_amd import tp_runtime  # noqa: E402


def main():
    rank, world, group = pkg.init_tp()
    assert world > 1, "WORLD_SIZE > 1, one process per rank"
    dev = torch.device("cuda", int(os.getenv("LOCAL_RANK", "0")))
    assert pkg.use_p2p_all_reduce(group), f"rank {rank}: no one-shot all-reduce: {tp_runtime._P2P_WHY[0]}"
    comm = tp_runtime._P2P[1]
    calls = 1  # (the probe of use_p2p_all_reduce)
    for d, dtype in enumerate((torch.bfloat16, torch.float16, torch.float32)):
        esz = torch.empty((), dtype=dtype).element_size()
        for nbytes in (16, 16 * 1031, 64 * 1536 * 2, pkg._hip.P2P_MAX_BYTES):
            n = nbytes // esz
            for trial in range(3):
                # (every rank makes every rank's addend from the seed: nothing but the sum is exchanged)
                xs = [torch.randn(n, generator=torch.Generator().manual_seed(((d * 8 + trial) * 8 + r) * 1000003 + n)).to(dtype)
                      for r in range(world)]
                acc = torch.zeros(n, dtype=torch.float32)
                for x in xs:
                    acc = acc + x.float()
                want = acc.to(dtype)
                mine = xs[rank].to(dev)
                assert comm.fits(mine)
                got = pkg.tp_all_reduce_sum(mine, group)
                assert got is mine
                calls += 1
                assert torch.equal(mine.cpu().view(torch.uint8), want.view(torch.uint8)), (rank, dtype, nbytes, trial)
    # what does not fit a window keeps the group's other path
    big = torch.full((1 << 20,), float(rank + 1), dtype=torch.float32, device=dev)
    assert not comm.fits(big)
    pkg.tp_all_reduce_sum(big, group)
    assert bool((big == world * (world + 1) / 2).all())
    assert comm.status() == (0, calls), comm.status()
    print(f"rank {rank}/{world}: {calls} one-shot sums through IPC windows equal the rank-ordered fp32 sum bit for bit", flush=True)
    # what a call costs when the ranks share one GPU (no link in between: the floor of the protocol, not of xGMI)
    for rows in (1, 64, 256):
        x = torch.zeros(rows * 1536, dtype=torch.bfloat16, device=dev)
        for _ in range(20):
            comm.all_reduce_sum(x)
        torch.cuda.synchronize()
        torch.distributed.barrier()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(200):
            comm.all_reduce_sum(x)
        t1.record()
        torch.cuda.synchronize()
        print(f"rank {rank}/{world}: {rows} x 1536 bf16 ({x.numel() * 2 // 1024} KiB): {t0.elapsed_time(t1) * 5:.1f} us per call", flush=True)
    assert comm.status()[0] == 0
    pkg.release_p2p_all_reduce()
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
