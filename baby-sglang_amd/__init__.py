"""MI355X-native paged-attention backend for baby-sglang's attention hot path.

Host-side mirror of the reference interface (``SimpleAttentionBackend``,
``SimplifiedForwardBatch``, ``ReqToTokenPool``, ``MHATokenToKVPool``, the
``tp_runtime`` collectives) over hand-written gfx950 HIP kernels reached
through the C ABI of ``libbsgl_attn.so`` (include/bsgl_attn.h).  Importing the
package loads that library and raises if it has not been built.
"""

from . import _hip
from .attn_backend import SimpleAttentionBackend
from .forward_batch import SimplifiedForwardBatch
from .memory_pool import BaseTokenToKVPool, MHATokenToKVPool, ReqToTokenPool
from .qwen2_hip import BabyQwen2ForCausalLM, shard_state_dict
from .tp_runtime import (
    LocalGroup,
    TPLayout,
    init_tp,
    split_tp_groups,
    tp_all_gather,
    tp_all_reduce_sum,
    tp_attention_exchange,
    tp_broadcast,
    use_capi_collectives,
    use_p2p_all_reduce,
    release_p2p_all_reduce,
)

__all__ = [
    "_hip",
    "SimpleAttentionBackend",
    "BabyQwen2ForCausalLM",
    "shard_state_dict",
    "SimplifiedForwardBatch",
    "BaseTokenToKVPool",
    "MHATokenToKVPool",
    "ReqToTokenPool",
    "LocalGroup",
    "TPLayout",
    "init_tp",
    "split_tp_groups",
    "tp_all_gather",
    "tp_all_reduce_sum",
    "tp_attention_exchange",
    "tp_broadcast",
    "use_capi_collectives",
    "use_p2p_all_reduce",
    "release_p2p_all_reduce",
]
