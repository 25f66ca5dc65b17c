"""ctypes binding of libbsgl_attn.so (C ABI declared in include/bsgl_attn.h).

The library is the only compute path of this package: importing this module
raises if the shared object has not been built, and every wrapper raises on a
non-zero return code.  PyTorch is used for device memory and streams only;
the wrappers pass raw device pointers, element strides and the caller's
current HIP stream, so calls are safe from any thread and on any stream
(the reference calls its backend from a side thread on a side stream,
overlap_worker.py:104-127).
"""

from __future__ import annotations

import ctypes
import math
import os
import struct
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbsgl_attn.so")

ABI_VERSION = 2

BSGL_OK = 0
BSGL_ERR_BAD_ARG = -1
BSGL_ERR_UNSUPPORTED = -2
BSGL_ERR_HIP = -3
BSGL_ERR_WORKSPACE = -4
BSGL_ERR_COMM = -5
COMM_ID_BYTES = 128

DECODE_AUTO, DECODE_VALU, DECODE_MFMA = 0, 1, 2
DECODE_FUSE_KV_WRITE = 0x100  # caller asserts: out_cache_loc slots belong to their own request only
DECODE_UNIFORM_SPLITS = 0x200  # exactly n_splits equal pieces per request instead of length-balanced ones
PREFILL_LIST_READY = 0x400  # paged_prefill: the workspace still holds this batch's work list
knob_epoch = 0  # bumped by set_knob: a cached work list was made under the knobs of its epoch

_DTYPE_CODE = {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}
_DTYPE_CODE_GET = _DTYPE_CODE.get

# every symbol include/bsgl_attn.h declares, with its ctypes signature
_vp, _i64, _i32, _f32, _sz = (
    ctypes.c_void_p,
    ctypes.c_int64,
    ctypes.c_int32,
    ctypes.c_float,
    ctypes.c_size_t,
)
class DecodeCall(ctypes.Structure):
    """``bsgl_decode_call`` of include/bsgl_attn.h: the arguments of ``bsgl_paged_decode`` as one
    struct, so that a caller fills the fields that never change once and passes one pointer."""

    _fields_ = [
        ("out", _vp), ("out_row_stride", _i64), ("lse", _vp),
        ("q", _vp), ("q_row_stride", _i64),
        ("k_new", _vp), ("v_new", _vp), ("k_new_row_stride", _i64), ("v_new_row_stride", _i64),
        ("out_cache_loc", _vp),
        ("k_slab", _vp), ("v_slab", _vp), ("slab_row_stride", _i64), ("n_slots", _i64),
        ("req_to_token", _vp), ("table_row_stride", _i64), ("table_rows", _i64), ("table_cols", _i64),
        ("req_pool_indices", _vp), ("seq_lens", _vp),
        ("batch", _i32), ("n_q_heads", _i32), ("n_kv_heads", _i32), ("head_dim", _i32),
        ("scale", _f32), ("dtype", _i32), ("out_dtype", _i32), ("n_splits", _i32), ("variant", _i32), ("max_seq_len", _i32),
        ("workspace", _vp), ("workspace_bytes", _sz), ("stream", _vp),
    ]  # fmt: skip


# out .. out_cache_loc: the ten 8-byte fields a caller rewrites on every call, as one packed write
DECODE_CALL_HEAD = struct.Struct("=10q")
assert DecodeCall.out.offset == 0 and DecodeCall.out_cache_loc.offset == 72 and DecodeCall.k_slab.offset == 80


SIGNATURES = {
    "bsgl_abi_version": (ctypes.c_int, []),
    "bsgl_last_error": (ctypes.c_char_p, []),
    "bsgl_debug_decode_trace": (None, [_vp]),
    "bsgl_debug_prefill_profile": (None, [_vp]),
    "bsgl_debug_set_knob": (ctypes.c_int, [ctypes.c_char_p, _i32, ctypes.POINTER(ctypes.c_int32)]),
    "bsgl_kv_scatter": (
        ctypes.c_int,
        [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp],
    ),
    "bsgl_decode_num_splits": (_i32, [_i32, _i32, _i32]),
    "bsgl_decode_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32]),
    "bsgl_decode_counter_bytes": (_sz, []),
    "bsgl_paged_decode": (
        ctypes.c_int,
        [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _vp]
        + [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp]
        + [_i32, _i32, _i32, _i32, _f32, _i32, _i32, _i32, _i32, _vp, _sz, _vp],
    ),
    "bsgl_paged_decode_call": (ctypes.c_int, [ctypes.POINTER(DecodeCall)]),
    "bsgl_paged_decode_static": (
        ctypes.c_int,
        [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _vp]
        + [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp]
        + [_i32, _i32, _i32, _i32, _f32, _i32, _i32, _i32, _i32, _vp],
    ),
    "bsgl_prefill_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "bsgl_paged_prefill": (
        ctypes.c_int,
        [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp]
        + [_i32, _i32, _i32, _i32, _i32, _f32, _i32, _i32, _vp, _sz, _vp],
    ),
    "bsgl_merge_attn_states": (
        ctypes.c_int,
        [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    ),
    "bsgl_rmsnorm": (ctypes.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _i32, _i32, _f32, _i32, _vp]),
    "bsgl_rope_kv_write": (
        ctypes.c_int,
        [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    ),
    "bsgl_comm_unique_id": (ctypes.c_int, [_vp]),
    "bsgl_comm_init": (ctypes.c_int, [_i32, _i32, _vp]),
    "bsgl_comm_world": (ctypes.c_int, []),
    "bsgl_all_reduce_sum": (ctypes.c_int, [_vp, _i64, _i32, _vp]),
    "bsgl_all_gather": (ctypes.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "bsgl_comm_destroy": (ctypes.c_int, []),
    "bsgl_p2p_create": (ctypes.c_int, [_i32, _i32, ctypes.POINTER(_vp), _vp, ctypes.POINTER(_vp)]),
    "bsgl_p2p_connect": (ctypes.c_int, [_vp, _vp, ctypes.POINTER(_vp)]),
    "bsgl_p2p_all_reduce_sum": (ctypes.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "bsgl_p2p_status": (ctypes.c_int, [_vp, ctypes.POINTER(_i32), ctypes.POINTER(_i64)]),
    "bsgl_p2p_status_async": (ctypes.c_int, [_vp, _vp, _vp]),
    "bsgl_p2p_destroy": (ctypes.c_int, [_vp]),
}


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
            "(or __graft_entry__.build()); this package has no non-HIP attention path"
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    got = lib.bsgl_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI version {got}, expected {ABI_VERSION}")
    return lib


lib = _load()


def _check(rc: int, what: str) -> None:
    if rc == BSGL_OK:
        return
    msg = lib.bsgl_last_error().decode("utf-8", "replace")
    if rc == BSGL_ERR_UNSUPPORTED:
        raise NotImplementedError(f"{what}: {msg}")
    if rc == BSGL_ERR_BAD_ARG:
        raise ValueError(f"{what}: {msg}")
    raise RuntimeError(f"{what} failed ({rc}): {msg}")


def set_knob(name: str, value: int) -> int:
    """Replace one of the library's tuning knobs (named like its environment variable, e.g.
    ``BSGL_MFMA_WAVES``); returns the value it had.  The library reads the environment once per
    process, so this is how a test or a tuning script changes a knob afterwards."""
    global knob_epoch
    prev = ctypes.c_int32(0)
    _check(lib.bsgl_debug_set_knob(name.encode(), int(value), ctypes.byref(prev)), "bsgl_debug_set_knob")
    knob_epoch += 1
    return int(prev.value)


def _dtype_code(t: torch.Tensor) -> int:
    try:
        return _DTYPE_CODE[t.dtype]
    except KeyError:
        raise NotImplementedError(f"dtype {t.dtype} has no HIP attention kernel") from None


def _need_gpu(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError(
                "baby-sglang_amd runs on MI355X only: got a tensor on "
                f"{t.device}; there is no CPU attention path in this package"
            )


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


try:  # the raw handle without building a torch.cuda.Stream object (a third of the cost)
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:  # pragma: no cover  (older / newer torch without the private hook)
    _raw_stream = None


def current_stream_handle(device_index: int) -> int:
    """hipStream_t of the caller's current stream on ``device_index`` as an int."""
    if _raw_stream is not None:
        return _raw_stream(device_index)
    return torch.cuda.current_stream(device_index).cuda_stream


def _rows(x: torch.Tensor) -> torch.Tensor:
    """[T, heads, dim] (or [T, heads*dim]) view whose inner dims are dense."""
    return _rows3(x)[0]


def _rows3(x: torch.Tensor):
    """``_rows`` plus the result's shape and strides (asked for once: a stride() call costs as
    much as the rest of the check)."""
    sh, st = x.shape, x.stride()
    n = len(sh)
    if n == 3:
        if st[2] != 1 or st[1] != sh[2]:
            x = x.contiguous()
            st = x.stride()
    elif n == 2:
        if st[1] != 1:
            x = x.contiguous()
            st = x.stride()
    else:
        raise ValueError(f"expected a 2-D or 3-D tensor, got shape {tuple(sh)}")
    return x, sh, st


def kv_scatter(k_slab, v_slab, loc, k_new, v_new) -> None:
    """k_slab[loc] = k_new; v_slab[loc] = v_new (memory_pool.py:152-171)."""
    _need_gpu(k_slab, v_slab, loc, k_new, v_new)
    n_tok = int(loc.shape[0])
    if n_tok == 0:
        return
    if loc.dtype != torch.int32:
        loc = loc.to(torch.int32)
    if not loc.is_contiguous():
        loc = loc.contiguous()
    heads, dim = k_slab.shape[1], k_slab.shape[2]
    k_new = _rows(k_new.view(n_tok, heads, dim) if k_new.dim() == 2 else k_new)
    v_new = _rows(v_new.view(n_tok, heads, dim) if v_new.dim() == 2 else v_new)
    if k_new.shape != (n_tok, heads, dim) or v_new.shape != (n_tok, heads, dim):
        raise ValueError(
            f"new K/V rows {tuple(k_new.shape)}/{tuple(v_new.shape)} do not match "
            f"loc [{n_tok}] and slab heads {heads} x dim {dim}"
        )
    if k_slab.stride(0) != v_slab.stride(0):
        raise ValueError("K and V slabs must share a row stride")
    rc = lib.bsgl_kv_scatter(
        k_slab.data_ptr(), v_slab.data_ptr(), k_slab.stride(0), k_slab.shape[0],
        loc.data_ptr(), k_new.data_ptr(), v_new.data_ptr(),
        k_new.stride(0), v_new.stride(0), n_tok, heads, dim,
        _dtype_code(k_slab), _stream_ptr(k_slab.device),
    )  # fmt: skip
    _check(rc, "bsgl_kv_scatter")


def decode_num_splits(batch: int, n_kv_heads: int, max_context_len: int) -> int:
    return lib.bsgl_decode_num_splits(batch, n_kv_heads, max_context_len)


def decode_workspace_bytes(
    batch: int, n_q_heads: int, head_dim: int, n_splits: int, out_is_f32: bool = False
) -> int:
    return lib.bsgl_decode_workspace_bytes(batch, n_q_heads, head_dim, n_splits, int(out_is_f32))


def new_decode_workspace(n_bytes: int, device) -> torch.Tensor:
    """Workspace for paged_decode: zero-filled, because its head holds the tickets
    of the in-kernel split merge, which every call expects (and leaves) at zero."""
    return torch.zeros(max(int(n_bytes), 16), dtype=torch.uint8, device=device)


def paged_decode(
    q: torch.Tensor,
    k_slab: torch.Tensor,
    v_slab: torch.Tensor,
    req_to_token: torch.Tensor,
    req_pool_indices: torch.Tensor,
    seq_lens: torch.Tensor,
    scale: Optional[float] = None,
    n_splits: Optional[int] = None,
    variant: int = DECODE_AUTO,
    out: Optional[torch.Tensor] = None,
    lse: Optional[torch.Tensor] = None,
    workspace: Optional[torch.Tensor] = None,
    out_dtype: Optional[torch.dtype] = None,
    k_new: Optional[torch.Tensor] = None,
    v_new: Optional[torch.Tensor] = None,
    out_cache_loc: Optional[torch.Tensor] = None,
    fuse_kv_write: bool = False,
    uniform_splits: bool = False,
    static_scratch: bool = False,
    max_seq_len: int = 0,
    _plan: Optional[tuple] = None,
) -> torch.Tensor:
    """softmax(q K^T * scale) V over page-table rows; returns [B, Hq * D].

    (``_plan``: ``(workspace bytes needed, stream handle)`` from a caller that has looked them
    up already for this shape and stream -- the per-layer call of the backend --, so that they
    are not asked for again.)

    ``out_dtype=torch.float32`` returns unrounded results (with ``lse``, the
    state a cross-rank merge needs); default is q's dtype.  ``k_new`` /
    ``v_new`` / ``out_cache_loc`` (together) are stored into the slabs before
    attention reads them; ``fuse_kv_write`` lets the library do that inside the
    decode kernel, which is only valid when each slot in ``out_cache_loc`` is
    referenced by its own request's page-table row alone.  A caller-provided
    ``workspace`` must have been created zero-filled (``new_decode_workspace``).
    ``n_splits`` bounds the pieces a request's sequence is cut into; pieces have
    one batch-wide length unless ``uniform_splits`` asks for exactly ``n_splits``
    equal pieces per request.  ``static_scratch`` takes the entry point that allocates
    nothing (``bsgl_paged_decode_static``: the split merge's workspace is an array inside the
    library, 16 MiB per GPU, shared by all such calls on that GPU); ``out`` must then be given.
    ``max_seq_len`` (0 = unknown) tells the library that no request is longer: a hint for the
    shape of the launch only, results never depend on it.
    """
    # (this wrapper runs once per layer and decode step beside a ~30 us kernel: every tensor's
    # shape and strides are asked for once)
    _need_gpu(q, k_slab, v_slab, req_to_token, req_pool_indices, seq_lens)
    q, q_shape, q_stride = _rows3(q)
    q_dtype = q.dtype
    batch, hq, dim = q_shape
    k_shape, k_stride0 = k_slab.shape, k_slab.stride(0)
    hkv = k_shape[1]
    if out_dtype is None:
        out_dtype = out.dtype if out is not None else q_dtype
    if out is None:
        out = torch.empty((batch, hq * dim), dtype=out_dtype, device=q.device)
    if batch == 0:
        return out
    if k_slab.dtype != q_dtype or v_slab.dtype != q_dtype:
        raise ValueError(f"q is {q_dtype} but the KV slabs are {k_slab.dtype}")
    if k_stride0 != v_slab.stride(0):
        raise ValueError("K and V slabs must share a row stride")
    if req_pool_indices.dtype != torch.int64:
        req_pool_indices = req_pool_indices.to(torch.int64)
    if seq_lens.dtype != torch.int64:
        seq_lens = seq_lens.to(torch.int64)
    t_shape, t_stride = req_to_token.shape, req_to_token.stride()
    if req_to_token.dtype != torch.int32 or t_stride[1] != 1:
        raise ValueError("req_to_token must be int32 with unit column stride")
    if scale is None:
        scale = 1.0 / math.sqrt(dim)
    if n_splits is None:
        n_splits = decode_num_splits(batch, hkv, t_shape[1])
    need = _plan[0] if _plan is not None else decode_workspace_bytes(batch, hq, dim, n_splits, out_dtype != q_dtype)
    ws_bytes = 0
    if need and not static_scratch:
        if workspace is not None:
            ws_bytes = workspace.numel() * workspace.element_size()
        if ws_bytes < need:
            workspace = torch.zeros(need, dtype=torch.uint8, device=q.device)
            ws_bytes = need
    if uniform_splits:
        variant |= DECODE_UNIFORM_SPLITS
    q_code = _DTYPE_CODE_GET(q_dtype)
    if q_code is None:
        raise NotImplementedError(f"dtype {q_dtype} has no HIP attention kernel")
    new = (None, None, 0, 0, None)
    if k_new is not None or v_new is not None or out_cache_loc is not None:
        if k_new is None or v_new is None or out_cache_loc is None:
            raise ValueError("k_new, v_new and out_cache_loc go together")
        k_new, kn_shape, kn_stride = _rows3(k_new.view(batch, hkv, dim) if k_new.dim() == 2 else k_new)
        v_new, vn_shape, vn_stride = _rows3(v_new.view(batch, hkv, dim) if v_new.dim() == 2 else v_new)
        if k_new.dtype != q_dtype or v_new.dtype != q_dtype:
            raise ValueError("new K/V rows must have q's dtype")
        want = (batch, hkv, dim)
        if tuple(kn_shape) != want or tuple(vn_shape) != want:
            raise ValueError("new K/V rows must be [batch, n_kv_heads, head_dim]")
        loc = out_cache_loc if out_cache_loc.dtype == torch.int32 else out_cache_loc.to(torch.int32)
        loc = loc if loc.is_contiguous() else loc.contiguous()
        if loc.shape[0] != batch:
            raise ValueError("out_cache_loc must have one slot per request")
        new = (k_new.data_ptr(), v_new.data_ptr(), kn_stride[0], vn_stride[0], loc.data_ptr())
        if fuse_kv_write:
            variant |= DECODE_FUSE_KV_WRITE
    if static_scratch:
        rc = lib.bsgl_paged_decode_static(
            out.data_ptr(), out.stride(0), lse.data_ptr() if lse is not None else None,
            q.data_ptr(), q_stride[0], *new,
            k_slab.data_ptr(), v_slab.data_ptr(), k_stride0, k_shape[0],
            req_to_token.data_ptr(), t_stride[0], t_shape[0], t_shape[1],
            req_pool_indices.data_ptr(), seq_lens.data_ptr(),
            batch, hq, hkv, dim, scale, q_code, _DTYPE_CODE[out_dtype], n_splits, variant,
            _plan[1] if _plan is not None else _stream_ptr(q.device),
        )  # fmt: skip
        _check(rc, "bsgl_paged_decode_static")
        return out
    if max_seq_len > 0:
        # (the hint travels in the call descriptor only)
        c = DecodeCall()
        c.out, c.out_row_stride, c.lse = out.data_ptr(), out.stride(0), lse.data_ptr() if lse is not None else None
        c.q, c.q_row_stride = q.data_ptr(), q_stride[0]
        c.k_new, c.v_new, c.k_new_row_stride, c.v_new_row_stride, c.out_cache_loc = new
        c.k_slab, c.v_slab, c.slab_row_stride, c.n_slots = k_slab.data_ptr(), v_slab.data_ptr(), k_stride0, k_shape[0]
        c.req_to_token, c.table_row_stride, c.table_rows, c.table_cols = req_to_token.data_ptr(), t_stride[0], t_shape[0], t_shape[1]
        c.req_pool_indices, c.seq_lens = req_pool_indices.data_ptr(), seq_lens.data_ptr()
        c.batch, c.n_q_heads, c.n_kv_heads, c.head_dim, c.scale = batch, hq, hkv, dim, scale
        c.dtype, c.out_dtype, c.n_splits, c.variant, c.max_seq_len = q_code, _DTYPE_CODE[out_dtype], n_splits, variant, int(max_seq_len)
        c.workspace, c.workspace_bytes = workspace.data_ptr() if need else None, ws_bytes
        c.stream = _plan[1] if _plan is not None else _stream_ptr(q.device)
        _check(lib.bsgl_paged_decode_call(ctypes.byref(c)), "bsgl_paged_decode_call")
        return out
    rc = lib.bsgl_paged_decode(
        out.data_ptr(), out.stride(0), lse.data_ptr() if lse is not None else None,
        q.data_ptr(), q_stride[0], *new,
        k_slab.data_ptr(), v_slab.data_ptr(), k_stride0, k_shape[0],
        req_to_token.data_ptr(), t_stride[0], t_shape[0], t_shape[1],
        req_pool_indices.data_ptr(), seq_lens.data_ptr(),
        batch, hq, hkv, dim, scale, q_code, _DTYPE_CODE[out_dtype], n_splits, variant,
        workspace.data_ptr() if need else None, ws_bytes,
        _plan[1] if _plan is not None else _stream_ptr(q.device),
    )  # fmt: skip
    _check(rc, "bsgl_paged_decode")
    return out


def prefill_workspace_bytes(batch: int, total_q: int, n_q_heads: int, head_dim: int) -> int:
    return lib.bsgl_prefill_workspace_bytes(batch, total_q, n_q_heads, head_dim)


def paged_prefill(
    q: torch.Tensor,
    k_slab: torch.Tensor,
    v_slab: torch.Tensor,
    req_to_token: torch.Tensor,
    req_pool_indices: torch.Tensor,
    seq_lens: torch.Tensor,
    prefix_lens: torch.Tensor,
    extended_lens: torch.Tensor,
    scale: Optional[float] = None,
    variant: int = DECODE_AUTO,
    out: Optional[torch.Tensor] = None,
    workspace: Optional[torch.Tensor] = None,
    list_ready: bool = False,
) -> torch.Tensor:
    """Extend attention over cached prefixes; returns [T, Hq * D] (T rows of q).  ``list_ready``: ``workspace``
    still holds the work list an earlier call made for this very batch (``BSGL_PREFILL_LIST_READY``)."""
    _need_gpu(q, k_slab, v_slab, req_to_token, req_pool_indices, seq_lens, prefix_lens, extended_lens)
    q = _rows(q)
    total_q, hq, dim = q.shape
    hkv = k_slab.shape[1]
    batch = int(req_pool_indices.shape[0])
    if out is None:
        out = torch.empty((total_q, hq * dim), dtype=q.dtype, device=q.device)
    if total_q == 0 or batch == 0:
        return out
    if k_slab.dtype != q.dtype or v_slab.dtype != q.dtype:
        raise ValueError(f"q is {q.dtype} but the KV slabs are {k_slab.dtype}")
    if req_to_token.dtype != torch.int32 or req_to_token.stride(1) != 1:
        raise ValueError("req_to_token must be int32 with unit column stride")
    i64 = [
        t if t.dtype == torch.int64 else t.to(torch.int64)
        for t in (req_pool_indices, seq_lens, prefix_lens, extended_lens)
    ]
    if scale is None:
        scale = 1.0 / math.sqrt(dim)
    need = prefill_workspace_bytes(batch, total_q, hq, dim)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=q.device)
    rc = lib.bsgl_paged_prefill(
        out.data_ptr(), out.stride(0), q.data_ptr(), q.stride(0),
        k_slab.data_ptr(), v_slab.data_ptr(), k_slab.stride(0), k_slab.shape[0],
        req_to_token.data_ptr(), req_to_token.stride(0), req_to_token.shape[0], req_to_token.shape[1],
        i64[0].data_ptr(), i64[1].data_ptr(), i64[2].data_ptr(), i64[3].data_ptr(),
        batch, total_q, hq, hkv, dim, scale, _dtype_code(q), variant | (PREFILL_LIST_READY if list_ready else 0),
        workspace.data_ptr(), workspace.numel() * workspace.element_size(),
        _stream_ptr(q.device),
    )  # fmt: skip
    _check(rc, "bsgl_paged_prefill")
    return out


def merge_attn_states(
    parts_o: torch.Tensor,
    parts_lse: torch.Tensor,
    out_dtype: torch.dtype,
    out: Optional[torch.Tensor] = None,
    lse_out: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """Merge [P, rows, H, D] fp32 partial outputs with their [P, rows, H] lse."""
    _need_gpu(parts_o, parts_lse)
    n_parts, rows, heads, dim = parts_o.shape
    parts_o = parts_o.contiguous()
    parts_lse = parts_lse.contiguous()
    if parts_o.dtype != torch.float32 or parts_lse.dtype != torch.float32:
        raise ValueError("partial states must be float32")
    if out is None:
        out = torch.empty((rows, heads * dim), dtype=out_dtype, device=parts_o.device)
    rc = lib.bsgl_merge_attn_states(
        out.data_ptr(), out.stride(0), lse_out.data_ptr() if lse_out is not None else None,
        parts_o.data_ptr(), parts_lse.data_ptr(), n_parts, rows, heads, dim,
        _DTYPE_CODE[out.dtype], _stream_ptr(parts_o.device),
    )  # fmt: skip
    _check(rc, "bsgl_merge_attn_states")
    return out


def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float, residual: Optional[torch.Tensor] = None):
    """``(weight * norm(x + residual), x + residual)`` for ``[rows, hidden]`` inputs
    (``residual`` optional: then the second result is ``None``), one launch."""
    _need_gpu(x, weight)
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("x must be [rows, hidden] with unit column stride")
    rows, hidden = x.shape
    if weight.dtype != x.dtype or weight.shape != (hidden,) or not weight.is_contiguous():
        raise ValueError("weight must be a dense [hidden] tensor of x's dtype")
    out = torch.empty((rows, hidden), dtype=x.dtype, device=x.device)
    res_out = None
    if residual is not None:
        _need_gpu(residual)
        if residual.shape != x.shape or residual.dtype != x.dtype:
            raise ValueError("residual must match x")
        residual = residual if residual.is_contiguous() else residual.contiguous()
        res_out = torch.empty((rows, hidden), dtype=x.dtype, device=x.device)
    rc = lib.bsgl_rmsnorm(
        out.data_ptr(), out.stride(0), res_out.data_ptr() if res_out is not None else None,
        x.data_ptr(), x.stride(0), residual.data_ptr() if residual is not None else None,
        weight.data_ptr(), rows, hidden, eps, _dtype_code(x), _stream_ptr(x.device),
    )  # fmt: skip
    _check(rc, "bsgl_rmsnorm")
    return out, res_out


def rope_kv_write(
    qkv: torch.Tensor,
    positions: torch.Tensor,
    inv_freq: torch.Tensor,
    n_q_heads: int,
    n_kv_heads: int,
    head_dim: int,
    k_slab: Optional[torch.Tensor] = None,
    v_slab: Optional[torch.Tensor] = None,
    out_cache_loc: Optional[torch.Tensor] = None,
    return_kv: bool = False,
):
    """Split a ``[T, (Hq + 2 Hkv) * D]`` projection row into q | k | v, rotate q and k,
    store rotated k and v at slab rows ``out_cache_loc`` (when given) and return
    ``(q [T, Hq, D], k or None, v or None)`` -- k and v only with ``return_kv``."""
    _need_gpu(qkv, positions, inv_freq)
    if qkv.dim() != 2 or qkv.stride(1) != 1 or qkv.shape[1] != (n_q_heads + 2 * n_kv_heads) * head_dim:
        raise ValueError("qkv must be [T, (n_q_heads + 2 n_kv_heads) * head_dim] with unit column stride")
    n_tok = qkv.shape[0]
    if positions.dtype != torch.int64 or not positions.is_contiguous() or positions.shape[0] != n_tok:
        positions = positions.to(torch.int64).contiguous()
    if inv_freq.dtype != torch.float32 or not inv_freq.is_contiguous():
        inv_freq = inv_freq.float().contiguous()
    if inv_freq.numel() != head_dim // 2:
        raise ValueError("inv_freq must have head_dim / 2 entries")
    q = torch.empty((n_tok, n_q_heads, head_dim), dtype=qkv.dtype, device=qkv.device)
    k = v = None
    if return_kv or out_cache_loc is None:
        k = torch.empty((n_tok, n_kv_heads, head_dim), dtype=qkv.dtype, device=qkv.device)
        v = torch.empty((n_tok, n_kv_heads, head_dim), dtype=qkv.dtype, device=qkv.device)
    loc = None
    slab = (None, None, 0, 0)
    if out_cache_loc is not None:
        if k_slab is None or v_slab is None:
            raise ValueError("out_cache_loc needs k_slab and v_slab")
        _need_gpu(k_slab, v_slab, out_cache_loc)
        if k_slab.dtype != qkv.dtype or v_slab.dtype != qkv.dtype or k_slab.stride(0) != v_slab.stride(0):
            raise ValueError("KV slabs must have qkv's dtype and share a row stride")
        loc = out_cache_loc if out_cache_loc.dtype == torch.int32 else out_cache_loc.to(torch.int32)
        loc = loc if loc.is_contiguous() else loc.contiguous()
        if loc.shape[0] != n_tok:
            raise ValueError("out_cache_loc must have one slot per token")
        slab = (k_slab.data_ptr(), v_slab.data_ptr(), k_slab.stride(0), k_slab.shape[0])
    rc = lib.bsgl_rope_kv_write(
        q.data_ptr(), q.stride(0), k.data_ptr() if k is not None else None, v.data_ptr() if v is not None else None,
        k.stride(0) if k is not None else 0, qkv.data_ptr(), qkv.stride(0), positions.data_ptr(), inv_freq.data_ptr(),
        *slab, loc.data_ptr() if loc is not None else None, n_tok, n_q_heads, n_kv_heads, head_dim,
        _dtype_code(qkv), _stream_ptr(qkv.device),
    )  # fmt: skip
    _check(rc, "bsgl_rope_kv_write")
    return q, k, v


# ---- tensor-parallel collectives on RCCL through the C ABI -------------------------------------


def comm_unique_id() -> bytes:
    """A fresh communicator id (rank 0 makes it; every rank of the group passes the same bytes
    to ``comm_init``)."""
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    _check(lib.bsgl_comm_unique_id(buf), "bsgl_comm_unique_id")
    return buf.raw


def comm_init(rank: int, world: int, unique_id: bytes) -> None:
    """This process's communicator, on the current HIP device; blocks until every rank has called."""
    if len(unique_id) != COMM_ID_BYTES:
        raise ValueError(f"unique id must be {COMM_ID_BYTES} bytes")
    _check(lib.bsgl_comm_init(rank, world, ctypes.c_char_p(unique_id)), "bsgl_comm_init")


def comm_world() -> int:
    return int(lib.bsgl_comm_world())


def comm_destroy() -> None:
    _check(lib.bsgl_comm_destroy(), "bsgl_comm_destroy")


def all_reduce_sum(x: torch.Tensor) -> torch.Tensor:
    """In-place sum over the communicator's ranks, on the caller's current stream."""
    _need_gpu(x)
    if not x.is_contiguous():
        raise ValueError("all_reduce_sum needs a contiguous tensor")
    _check(lib.bsgl_all_reduce_sum(x.data_ptr(), x.numel(), _dtype_code(x), _stream_ptr(x.device)), "bsgl_all_reduce_sum")
    return x


def all_gather(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``[world * x.shape[0], ...]``: every rank's ``x`` in rank order, on the caller's current stream."""
    _need_gpu(x)
    world = comm_world()
    if world < 1:
        raise ValueError("all_gather: no communicator (comm_init first)")
    x = x.contiguous()
    if out is None:
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    elif out.numel() != world * x.numel() or out.dtype != x.dtype or not out.is_contiguous():
        raise ValueError("all_gather: out must be contiguous, of x's dtype and world times its size")
    _check(lib.bsgl_all_gather(out.data_ptr(), x.data_ptr(), x.numel(), _dtype_code(x), _stream_ptr(x.device)), "bsgl_all_gather")
    return out


# ---- one-shot all-reduce through windows the ranks map from each other ------------------------

P2P_HANDLE_BYTES = 64
P2P_MAX_WORLD = 8
P2P_MAX_BYTES = 1 << 20


class P2PComm:
    """One rank's end of the library's one-shot all-reduce (``bsgl_p2p_*``): a window of device
    memory on the current device that the peers map -- by its IPC ``handle`` from other processes,
    by its address ``window`` inside this one.  ``connect`` takes the peers' handles or addresses
    in rank order; ``all_reduce_sum`` then sums a tensor of at most ``P2P_MAX_BYTES`` in place on
    the caller's current stream (every call of a communicator on the same stream)."""

    def __init__(self, rank: int, world: int):
        comm, window = _vp(), _vp()
        handle = ctypes.create_string_buffer(P2P_HANDLE_BYTES)
        _check(lib.bsgl_p2p_create(rank, world, ctypes.byref(comm), handle, ctypes.byref(window)), "bsgl_p2p_create")
        self._comm = comm
        self.rank, self.world = rank, world
        self.handle: bytes = handle.raw
        self.window: int = int(window.value)
        self._last_stream = None
        # a status probe in flight: (event behind the copy, pinned {status, calls}); see all_reduce_sum
        self._probe = None
        self._probe_words = None
        self._calls_since_probe = 0
        self.probe_every = 64

    def connect(self, handles: Optional[list] = None, windows: Optional[list] = None) -> None:
        if (handles is None) == (windows is None):
            raise ValueError("give the peers' windows either as IPC handles or as addresses")
        if handles is not None:
            if len(handles) != self.world or any(len(h) != P2P_HANDLE_BYTES for h in handles):
                raise ValueError(f"connect wants {self.world} handles of {P2P_HANDLE_BYTES} bytes")
            _check(lib.bsgl_p2p_connect(self._comm, ctypes.c_char_p(b"".join(handles)), None), "bsgl_p2p_connect")
        else:
            if len(windows) != self.world:
                raise ValueError(f"connect wants {self.world} window addresses")
            arr = (_vp * self.world)(*[int(w) for w in windows])
            _check(lib.bsgl_p2p_connect(self._comm, None, arr), "bsgl_p2p_connect")

    def fits(self, x: torch.Tensor) -> bool:
        """Whether ``all_reduce_sum`` takes this tensor (else the caller uses another collective)."""
        nbytes = x.numel() * x.element_size()
        return (x.is_cuda and x.is_contiguous() and x.dtype in _DTYPE_CODE and 0 < nbytes <= P2P_MAX_BYTES
                and nbytes % 16 == 0 and x.data_ptr() % 16 == 0)

    def all_reduce_sum(self, x: torch.Tensor) -> torch.Tensor:
        """In place on the caller's current stream.  Raises ``RuntimeError`` once a status probe has
        shown that a peer did not arrive (``check`` below): after that the library skips every sum,
        and a model that kept going would run on unsummed activations without a word."""
        _need_gpu(x)
        if not x.is_contiguous():
            raise ValueError("all_reduce_sum needs a contiguous tensor")
        stream = torch.cuda.current_stream(x.device)
        capturing = torch.cuda.is_current_stream_capturing()
        if self._last_stream is not None and stream != self._last_stream and not capturing:
            # the calls of a communicator run one after the other: a new stream waits for the last one
            # (not while the new stream is being captured: a wait for work outside the capture is no
            # legal capture operation, and the caller synchronises before it captures)
            stream.wait_stream(self._last_stream)
        if not capturing:
            self._last_stream = stream
            self.check(wait=False)
        _check(lib.bsgl_p2p_all_reduce_sum(self._comm, x.data_ptr(), x.numel(), _dtype_code(x), stream.cuda_stream),
               "bsgl_p2p_all_reduce_sum")
        if not capturing:
            self._calls_since_probe += 1
            if self._probe is None and self._calls_since_probe >= self.probe_every:
                self._start_probe(stream)
        return x

    def _start_probe(self, stream) -> None:
        if self._probe_words is None:
            self._probe_words = torch.zeros(2, dtype=torch.int32).pin_memory()
        _check(lib.bsgl_p2p_status_async(self._comm, self._probe_words.data_ptr(), stream.cuda_stream), "bsgl_p2p_status_async")
        ev = torch.cuda.Event()
        ev.record(stream)
        self._probe = ev
        self._calls_since_probe = 0

    def check(self, wait: bool = True) -> None:
        """Look at the communicator's status word and raise ``RuntimeError`` if a peer did not arrive.
        ``wait=False`` (what ``all_reduce_sum`` does on every eager call): only looks at a probe whose
        copy has already finished -- a probe is enqueued every ``probe_every`` calls, so a failure is
        noticed at most that many calls plus the depth of the stream late, and the hot path never
        waits.  ``wait=True`` (a serving loop at the end of a step, a test): enqueues a probe if none
        is in flight and waits for it."""
        if self._comm is None:
            return
        if wait and self._probe is None:
            self._start_probe(torch.cuda.current_stream())
        if self._probe is None:
            return
        if wait:
            self._probe.synchronize()
        elif not self._probe.query():
            return
        self._probe = None
        status, calls = int(self._probe_words[0]), int(self._probe_words[1])
        if status != 0:
            raise RuntimeError(
                f"one-shot all-reduce: a peer did not arrive within the wait (status {status} after {calls} completed calls); "
                "the buffer of that call is undefined and every later sum was skipped"
            )

    def status(self):
        """``(status, completed calls)`` -- waits for the device; status 0 = good, 1 = a peer did not arrive."""
        st, calls = _i32(0), _i64(0)
        _check(lib.bsgl_p2p_status(self._comm, ctypes.byref(st), ctypes.byref(calls)), "bsgl_p2p_status")
        return int(st.value), int(calls.value)

    def destroy(self) -> None:
        if self._comm is not None:
            comm, self._comm = self._comm, None
            _check(lib.bsgl_p2p_destroy(comm), "bsgl_p2p_destroy")
