"""oracle/build_ref.py -- TEST INFRASTRUCTURE: recipe that builds the checker binaries.

1. ``oracle/_build/libsonet_oracle.so``  <- oracle/sonet_oracle.c   (gcc, plain C restatement)
2. ``oracle/_ref/index_max.so``          <- /root/reference/models/index_max_ext/index_max.cpp
   compiled *from where it lies* (never copied) with oracle/ref_stub.cpp standing in for the CUDA
   translation unit.  Only possible where /root/reference is mounted (this container); the GPU box
   uses the prebuilt file, which travels with the gpurun snapshot (oracle/_ref/ is git-ignored but
   not gpurun-ignored).  The reference's own build system (setup.py / CUDAExtension) is NOT run.

Both outputs are loaded only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_CPP = "/root/reference/models/index_max_ext/index_max.cpp"


def _newer(src_list, out):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in src_list)


def build_oracle(verbose=False):
    src = os.path.join(HERE, "sonet_oracle.c")
    out_dir = os.path.join(HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "libsonet_oracle.so")
    if _newer([src], out):
        cmd = ["gcc", "-O2", "-std=c99", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC",
               src, "-o", out, "-lm"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return out


def build_ref(verbose=False):
    """Returns the path of oracle/_ref/index_max.so, or None when it cannot be (re)built."""
    out_dir = os.path.join(HERE, "_ref")
    out = os.path.join(out_dir, "index_max.so")
    stub = os.path.join(HERE, "ref_stub.cpp")
    if not os.path.exists(REF_CPP):
        return out if os.path.exists(out) else None
    if not _newer([REF_CPP, stub], out):
        return out
    os.makedirs(out_dir, exist_ok=True)
    import torch
    from torch.utils import cpp_extension as ce
    inc = ce.include_paths()
    libdirs = ce.library_paths()
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-w",
           "-DTORCH_EXTENSION_NAME=index_max", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    cmd += ["-I" + p for p in inc] + ["-I" + sysconfig.get_paths()["include"]]
    cmd += [REF_CPP, stub, "-o", out]
    cmd += ["-L" + p for p in libdirs] + ["-Wl,-rpath," + p for p in libdirs]
    cmd += ["-lc10", "-ltorch", "-ltorch_cpu", "-ltorch_python", "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build_oracle(verbose=True))
    print(build_ref(verbose=True))
