"""The C ABI from a plain C99 host (tests/cabi/cabi_hotpath.c): no torch, no Python in the loop.

CPU (``-m "not gpu"``): include/sonet_hip.h is valid C99 on its own, and the program compiles with gcc and links against
libsonet_hip.so + the HIP runtime -- i.e. every entry point it uses really is an ``extern "C"`` symbol with plain
pointer / integer arguments.  GPU (``-m gpu``): the binary runs som_assign -> som_group -> index_max_gather on its own
HIP stream and checks the results against the CPU oracle (bit-exact ids / counts / arg-max positions):
tests/test_gpu_parity_extended.py::test_c_host_program_matches_the_oracle_on_the_gpu."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cabi", "cabi_hotpath.c")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def build(out_dir):
    from oracle import build_ref
    oracle_so = build_ref.build_oracle()
    lib_dir = os.path.join(ROOT, "so-net_amd", "lib")
    assert os.path.exists(os.path.join(lib_dir, "libsonet_hip.so")), "run __graft_entry__.build() first"
    exe = os.path.join(str(out_dir), "cabi_hotpath")
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROCM, "include"),
           SRC, "-o", exe, "-L" + lib_dir, "-lsonet_hip", oracle_so, "-L" + os.path.join(ROCM, "lib"), "-lamdhip64", "-lm",
           "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + os.path.dirname(oracle_so), "-Wl,-rpath," + os.path.join(ROCM, "lib")]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    return exe


def test_header_is_self_contained_c99(tmp_path):
    src = tmp_path / "hdr.c"
    src.write_text('#include "sonet_hip.h"\nint main(void) { return sonet_abi_version() == 0; }\n')
    p = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]


def test_c_host_program_compiles_and_links(tmp_path):
    exe = build(tmp_path)
    assert os.access(exe, os.X_OK)
    # without a GPU the program must stop at its first check with a status and a message, not crash
    import torch
    if not torch.cuda.is_available():
        p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert p.returncode != 0 and p.returncode > 0, (p.returncode, p.stderr[-500:])
        assert "sonet_check_device" in p.stderr or "hip" in p.stderr.lower()
