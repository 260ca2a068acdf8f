"""Imported first by the experiment tools: the SONET_* tuning / ablation knobs and the measured-slower kernels exist only in the
VARIANTS build (make -C so-net_amd/csrc variants -> so-net_amd/lib/libsonet_hip_variants.so); the product library reads no environment
variable.  This points sonet_hip at the variants library (unless SONET_HIP_LIB is already set)."""
import os

_V = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "so-net_amd", "lib", "libsonet_hip_variants.so")
if "SONET_HIP_LIB" not in os.environ:
    if not os.path.exists(_V):
        raise SystemExit("%s is not built: make -C so-net_amd/csrc variants" % _V)
    os.environ["SONET_HIP_LIB"] = _V
