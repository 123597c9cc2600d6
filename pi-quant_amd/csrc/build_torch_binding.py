#!/usr/bin/env python3
"""Builds piquant/_piquant_torch.so (csrc/torch_binding.cpp) in-tree with g++ against the installed PyTorch-ROCm.

Host code only (no device code, no hipcc): include paths and libraries come from torch.utils.cpp_extension.  The extension
links libpiquant.so by name and finds it through an $ORIGIN rpath -- both live in the package directory.  Called by
__graft_entry__.build(); skipped (with a message) when the compiler or the PyTorch headers are missing, in which case
piquant.torch uses its ctypes path.
"""
import shutil
import subprocess
import sys
import sysconfig
from pathlib import Path

HERE = Path(__file__).resolve().parent
PKG = HERE.parent / "piquant"
OUT = PKG / "_piquant_torch.so"
SRC = HERE / "torch_binding.cpp"


def up_to_date() -> bool:
    deps = [SRC, HERE.parent.parent / "include" / "piquant.h", HERE.parent.parent / "include" / "piquant_hip.h", Path(__file__)]
    return OUT.exists() and all(OUT.stat().st_mtime >= d.stat().st_mtime for d in deps)


def build(verbose: bool = False) -> bool:
    if up_to_date():
        return True
    cxx = shutil.which("g++")
    if cxx is None:
        print("build_torch_binding: no g++, skipping the native torch front end", file=sys.stderr)
        return False
    import torch
    from torch.utils import cpp_extension as ce

    torch_lib = Path(torch.__file__).resolve().parent / "lib"
    rocm = Path("/opt/rocm")
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_piquant_torch", "-DTORCH_API_INCLUDE_EXTENSION_H",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    cmd += [f"-I{p}" for p in ce.include_paths()]
    cmd += [f"-I{rocm / 'include'}", f"-I{sysconfig.get_paths()['include']}", f"-I{HERE.parent.parent / 'include'}"]
    cmd += [str(SRC), "-o", str(OUT), f"-L{torch_lib}", f"-L{PKG}", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch", "-ltorch_hip", "-ltorch_python",
            "-lpiquant", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{torch_lib}"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        print("build_torch_binding: compilation failed, piquant.torch keeps its ctypes path\n" + r.stderr[-4000:], file=sys.stderr)
        if OUT.exists():
            OUT.unlink()
        return False
    return True


if __name__ == "__main__":
    sys.exit(0 if build(verbose=True) else 1)
