"""Can half of the CUs pull the whole HBM stream?  Times the decode call on streams whose
CU mask (hipExtStreamCreateWithCUMask) leaves all, every other, or a quarter of the CUs."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import baby_sglang_amd as pkg  # noqa: E402
from baby_sglang_amd import _hip  # noqa: E402


def masked_stream(hip, words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), len(words), arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(st.value)


def main():
    dev = torch.device("cuda:0")
    torch.cuda.init()
    hip = ctypes.CDLL("libamdhip64.so")
    B, S, hq, hkv, dim, layers = 64, 2048, 12, 2, 128, 8
    gen = torch.Generator(device=dev).manual_seed(1)
    size = B * S + 16
    kv = pkg.MHATokenToKVPool(size, torch.bfloat16, hkv, dim, layers, str(dev))
    kv.arena.normal_(generator=gen)
    table = torch.zeros((B, 4096), dtype=torch.int32, device=dev)
    table[:, :S] = (torch.randperm(B * S, generator=gen, device=dev) + 1).to(torch.int32).view(B, S)
    q = torch.randn((layers, B, hq, dim), generator=gen, device=dev).to(torch.bfloat16)
    req = torch.arange(B, device=dev)
    seq = torch.full((B,), S, dtype=torch.int64, device=dev)
    n_splits = _hip.decode_num_splits(B, hkv, 4096)
    out = torch.empty((B, hq * dim), dtype=torch.bfloat16, device=dev)
    masks = {
        "all CUs": [0xFFFFFFFF] * 8,
        "every other CU": [0x55555555] * 8,
        "every other pair": [0x33333333] * 8,
        "low half of each word": [0x0000FFFF] * 8,
        "first four words": [0xFFFFFFFF] * 4 + [0] * 4,
        "one CU in four": [0x11111111] * 8,
    }
    if os.environ.get("HSA_CU_MASK"):
        masks = {"HSA_CU_MASK=" + os.environ["HSA_CU_MASK"]: [0xFFFFFFFF] * 8}
    for name, words in masks.items():
        try:
            st = masked_stream(hip, words)
        except RuntimeError as e:
            print(f"{name:24s}: {e}", flush=True)
            continue
        ws = _hip.new_decode_workspace(_hip.decode_workspace_bytes(B, hq, dim, n_splits), dev)
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            def step():
                for l in range(layers):
                    k, v = kv.get_kv_buffer(l)
                    _hip.paged_decode(q[l], k, v, table, req, seq, n_splits=n_splits, variant=2, out=out, workspace=ws)
            step()
            st.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(5):
                step()
            e1.record(st)
            st.synchronize()
        print(f"{name:24s}: {e0.elapsed_time(e1) * 1e3 / (5 * layers):6.1f} us/layer", flush=True)
        torch.cuda.synchronize()
        hip.hipStreamDestroy(ctypes.c_void_p(st.cuda_stream))  # a masked stream owns a hardware queue


if __name__ == "__main__":
    main()
