"""Drop-in replacement of the reference's ``index_max`` extension module
(models/index_max_ext/index_max.cpp:154-159, imported at models/networks.py:17).

Same four functions, same argument meaning, same error behaviour (RuntimeError for non-CUDA or
non-contiguous inputs, index_max.cpp:119-121), same result: a NEW int32 tensor B x C x K on the
device of ``data``.  Underneath: one hand-written gfx950 kernel behind the C ABI
``sonet_index_max_f32`` / ``_bf16`` (include/sonet_hip.h, so-net_amd/csrc/index_max.hip).

``forward_cuda`` and ``forward_cuda_shared_mem`` are the same kernel (the reference's two CUDA
kernels compute the same function).  ``forward_cpu`` / ``forward_multi_thread_cpu`` keep the
reference signature for CPU tensors but there is no CPU implementation in this build: they stage
the tensors through the current MI355X, run the same kernel and copy the result back, and raise
if no GPU is present.
"""
import torch

from sonet_hip import ops as _ops
from sonet_hip._lib import SonetHipError


def forward_cuda(data, index, K):
    """CUDA(ROCm) tensors: data B x C x N' float32 (or bfloat16), index B x N' int32 -> B x C x K int32."""
    return _ops.index_max(data, index, int(K))


def forward_cuda_shared_mem(data, index, K):
    return _ops.index_max(data, index, int(K))


def _staged(data, index, K):
    if data.is_cuda or index.is_cuda:
        raise SonetHipError("forward_cpu expects CPU tensors (use forward_cuda for device tensors)")
    if not torch.cuda.is_available():
        raise SonetHipError("index_max: this build has no CPU implementation; an MI355X is required")
    dev = torch.device("cuda", torch.cuda.current_device())
    out = _ops.index_max(data.contiguous().to(dev), index.contiguous().to(dev), int(K))
    return out.cpu()


def forward_cpu(data, index, K):
    return _staged(data, index, K)


def forward_multi_thread_cpu(data, index, K, thread_num):
    """Reference signature (index_max.cpp:73-112).  ``thread_num`` is validated and otherwise has no meaning here: the staged
    tensors run through the one GPU kernel, there are no host worker threads to size."""
    if int(thread_num) < 1:
        raise SonetHipError("thread_num must be >= 1")
    return _staged(data, index, K)
