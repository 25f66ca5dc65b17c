"""The C-ABI library loads without a GPU, exports every function the public
header declares, and rejects bad arguments before any launch (CPU only)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from baby_sglang_amd import _hip


def declared_functions():
    text = open(os.path.join(ROOT, "include", "bsgl_attn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bsgl_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    names = declared_functions()
    assert len(names) >= 9
    assert set(names) == set(_hip.SIGNATURES)


@pytest.mark.parametrize("name", declared_functions())
def test_symbol_exported(name):
    lib = ctypes.CDLL(_hip.LIB_PATH)
    assert getattr(lib, name) is not None


def test_host_arithmetic():
    assert _hip.lib.bsgl_abi_version() == _hip.ABI_VERSION
    # 64 requests x 2 kv heads x 8 pieces = 1024 workgroups, and twice that as the most
    # pieces one request may be cut into (length-balanced pieces)
    assert _hip.decode_num_splits(64, 2, 4096) == 16
    assert _hip.decode_num_splits(64, 1, 4096) == 32
    assert _hip.decode_num_splits(4096, 2, 4096) == 1
    assert _hip.decode_num_splits(1, 2, 100) == 1  # one short request: a single chunk
    assert _hip.decode_workspace_bytes(64, 12, 128, 1) == 0
    assert _hip.lib.bsgl_decode_counter_bytes() >= 64 * 2 * 4
    assert _hip.decode_workspace_bytes(64, 12, 128, 1, True) >= 64 * 12 * 129 * 4
    assert _hip.decode_workspace_bytes(64, 12, 128, 8) >= 8 * 64 * 12 * 129 * 4
    assert _hip.prefill_workspace_bytes(3, 100, 12, 128) >= (3 + 1) * 8 + 2 * 100 * 8


def test_bad_arguments_are_rejected_on_the_host():
    lib = _hip.lib
    rc = lib.bsgl_paged_decode(None, 0, None, None, 0, None, None, 0, 0, None,
                               None, None, 0, 0, None, 0, 0, 0, None, None,
                               4, 12, 2, 128, 0.1, 0, 0, 1, 0, None, 0, None)
    assert rc == _hip.BSGL_ERR_BAD_ARG and b"null" in lib.bsgl_last_error()
    rc = lib.bsgl_kv_scatter(None, None, 0, 0, None, None, None, 0, 0, 3, 2, 128, 7, None)
    assert rc == _hip.BSGL_ERR_UNSUPPORTED and b"dtype" in lib.bsgl_last_error()
    rc = lib.bsgl_merge_attn_states(None, 0, None, None, None, 0, 1, 1, 128, 0, None)
    assert rc == _hip.BSGL_ERR_BAD_ARG
    # empty work is a successful no-op
    assert lib.bsgl_kv_scatter(None, None, 0, 0, None, None, None, 0, 0, 0, 2, 128, 0, None) == 0


def test_collectives_reject_bad_arguments_without_a_communicator():
    lib = _hip.lib
    assert lib.bsgl_comm_world() == 0
    assert lib.bsgl_all_reduce_sum(None, 4, 0, None) == _hip.BSGL_ERR_BAD_ARG and b"null" in lib.bsgl_last_error()
    assert lib.bsgl_all_reduce_sum(None, 4, 9, None) == _hip.BSGL_ERR_UNSUPPORTED
    assert lib.bsgl_all_gather(None, None, -1, 0, None) == _hip.BSGL_ERR_BAD_ARG
    assert lib.bsgl_all_gather(None, None, 0, 0, None) == 0  # nothing to move
    assert lib.bsgl_comm_init(2, 2, None) == _hip.BSGL_ERR_BAD_ARG and b"rank" in lib.bsgl_last_error()
    assert lib.bsgl_comm_init(0, 1, None) == _hip.BSGL_ERR_BAD_ARG and b"id" in lib.bsgl_last_error()
    assert lib.bsgl_comm_unique_id(None) == _hip.BSGL_ERR_BAD_ARG
    assert lib.bsgl_comm_destroy() == 0  # no communicator: nothing to do
