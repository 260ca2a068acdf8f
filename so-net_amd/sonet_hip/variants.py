"""Wrappers of the VARIANTS-build entry points (libsonet_hip_variants.so, ``make -C so-net_amd/csrc variants``): kernels that
measured slower than what the product dispatches, kept as tested records (tests/variants, tools/).  Importing this module needs
SONET_HIP_LIB to point at the variants library (tools/_variants.py does that); the product never imports it."""
import torch

from . import _lib
from ._lib import SonetHipError, check, ptr, stream_ptr
from .ops import _chk, _range_arm, _same_device, _timed


def _need(name):
    lib = _lib.load()
    if not hasattr(lib, name):
        raise SonetHipError("%s is exported by the variants build only: set SONET_HIP_LIB=%s (make -C so-net_amd/csrc variants)"
                            % (name, _lib.VARIANTS_PATH))
    return lib


# KNNModule: max over the neighbour planes by ordered-integer atomics in the last layer's epilogue.  Bit-identical, but measured SLOWER
# than layer + planes_max (110 vs 73 + 21 us at B = 64: 19 M L2 atomics cost more than the 75 MB store and re-read they replace): off.


def pointmlp_kmax(x1, wp, scale, shift, relu, Cout, M):
    """max over the K planes of the k-major output of pointmlp(x1, ...): x1 B x C x (K*M) f32 -> B x Cout x M f32, the B x Cout x K*M
    tensor is never written.  h3 packs, Cout % 128 == 0."""
    if wp.dtype != torch.int8:
        raise SonetHipError("pointmlp_kmax: an h3 pack")
    _chk(x1, "x", torch.float32, 3)
    B, C1, L = x1.shape
    if L % int(M) != 0 or Cout % 128 != 0:
        raise SonetHipError("pointmlp_kmax: L %% M == 0 and Cout %% 128 == 0")
    dev = _same_device(x1, wp, scale, shift)
    lib = _need("sonet_pointmlp_h3_kmax_f32")
    if wp.numel() != lib.sonet_pointmlp_x3_pack_size(C1, Cout):
        raise SonetHipError("packed weight does not match Cin=%d Cout=%d" % (C1, Cout))
    out = torch.empty((B, Cout, int(M)), dtype=torch.float32, device=dev)
    keys = torch.empty((B, Cout, int(M)), dtype=torch.int32, device=dev)
    name = "pointmlph3_kmax_%dx%d_L%d" % (C1, Cout, L)
    _range_arm(name)
    with torch.cuda.device(dev), _timed(name):
        check(lib.sonet_pointmlp_h3_kmax_f32(ptr(x1), C1, None, 0, ptr(wp), ptr(scale), ptr(shift), int(bool(relu)), ptr(out), ptr(keys),
                                             B, Cout, L, int(M), stream_ptr()), "sonet_pointmlp_h3_kmax_f32")
    return out



def chamfer_nn2(a, b):
    """a B x 3 x Na, b B x 3 x Nb -> (nn_ab B x Na, nn_ba B x Nb) i32: both nearest-neighbour directions of the Chamfer loss from
    one sweep of the distance matrix (``sonet_chamfer_nn2_f32``)."""
    _chk(a, "a", torch.float32, 3)
    _chk(b, "b", torch.float32, 3)
    dev = _same_device(a, b)
    B, _, Na = a.shape
    Nb = b.shape[2]
    if a.shape[1] != 3 or b.shape[1] != 3 or b.shape[0] != B:
        raise SonetHipError("chamfer_nn2: clouds must be B x 3 x N")
    lib = _need("sonet_chamfer_nn2_f32")
    nn_ab = torch.empty((B, Na), dtype=torch.int32, device=dev)
    nn_ba = torch.empty((B, Nb), dtype=torch.int32, device=dev)
    ws = torch.empty((lib.sonet_chamfer_nn2_ws_size(B, Na, Nb),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev), _timed("chamfer_nn2"):
        check(lib.sonet_chamfer_nn2_f32(ptr(a), ptr(b), ptr(nn_ab), ptr(nn_ba), ptr(ws), B, Na, Nb, stream_ptr()), "sonet_chamfer_nn2_f32")
    return nn_ab, nn_ba
