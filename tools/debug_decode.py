"""Diagnostic: structured K/V patterns through both decode kernels (one GPU)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import baby_sglang_amd as pkg
hip = pkg._hip
dev = torch.device("cuda:0")
hq, hkv, dim = 12, 2, 128
dtype = torch.bfloat16
torch.manual_seed(0)
for seq in (1, 5, 16, 17, 40):
    n_slots = 64
    kv = pkg.MHATokenToKVPool(n_slots - 1, dtype, hkv, dim, 1, "cuda:0")
    k_slab, v_slab = kv.get_kv_buffer(0)
    table = torch.zeros((1, 64), dtype=torch.int32, device=dev)
    slots = (torch.randperm(n_slots - 1)[:seq] + 1).to(torch.int32).to(dev)
    table[0, :seq] = slots
    req = torch.zeros(1, dtype=torch.int64, device=dev)
    sl = torch.tensor([seq], dtype=torch.int64, device=dev)
    q = torch.randn((1, hq, dim), device=dev).to(dtype)
    for name in ("Kzero_Vtoken", "Kzero_Vdim", "Krand_Vdim"):
        k_slab.zero_(); v_slab.zero_()
        if name.startswith("Krand"):
            k_slab.copy_(torch.randn(k_slab.shape, device=dev).to(dtype))
        if name.endswith("Vtoken"):
            for t in range(seq):
                v_slab[slots[t].long()] = float(t + 1)
        else:
            v_slab[:] = (torch.arange(dim, device=dev).float() / 4).to(dtype).view(1, 1, dim) + torch.arange(hkv, device=dev).float().view(1, hkv, 1) * 64
        o1 = hip.paged_decode(q, k_slab, v_slab, table, req, sl, n_splits=1, variant=1).float().view(hq, dim)
        o2 = hip.paged_decode(q, k_slab, v_slab, table, req, sl, n_splits=1, variant=2).float().view(hq, dim)
        d = (o1 - o2).abs().max().item()
        print(f"seq={seq:3d} {name:14s} maxdiff={d:.4f}  valu[h0,:4]={o1[0,:4].tolist()} mfma[h0,:4]={o2[0,:4].tolist()} mfma[h6,:4]={o2[6,:4].tolist()}")
        if d > 0.05 and name == "Kzero_Vdim":
            print("   mfma h0 dims*4:", (o2[0] * 4).round().int().tolist())
