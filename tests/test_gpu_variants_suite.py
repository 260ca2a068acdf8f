"""The variants build (tuning knobs + measured-slower kernels kept as records) is exercised in a SUBPROCESS that loads
libsonet_hip_variants.so through SONET_HIP_LIB -- the test process itself only ever maps the product library."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = os.path.join(ROOT, "so-net_amd", "lib", "libsonet_hip_variants.so")


@pytest.mark.gpu
def test_variants_build_suite():
    if not os.path.exists(VARIANTS):
        pytest.skip("libsonet_hip_variants.so is not built (make -C so-net_amd/csrc variants)")
    env = dict(os.environ, SONET_HIP_LIB=VARIANTS)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "variants", "variants_gpu.py"), "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, cwd=ROOT)
    out = r.stdout.decode()
    assert r.returncode == 0 and " passed" in out, out[-3000:]
