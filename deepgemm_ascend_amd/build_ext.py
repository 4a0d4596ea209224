"""Builds the `deep_gemm_cpp` torch extension in-tree (deepgemm_ascend_amd/deep_gemm_cpp*.so) from
csrc/python_api_amd.cpp -- the MI355X counterpart of the reference's pybind module
(/root/reference/deep_gemm_ascend/framework/csrc/python_api.cpp:30-36; its CMake build links torch_npu + ACL,
framework/CMakeLists.txt:44-47).  One hipcc invocation on host code: no hipify pass, no JIT cache outside the tree.
The module links libdga_hip.so next to it (rpath $ORIGIN)."""
from __future__ import annotations

import subprocess
import sys
import sysconfig
from pathlib import Path

PKG = Path(__file__).resolve().parent
SRC = PKG / "csrc" / "python_api_amd.cpp"


def ext_path() -> Path:
    return PKG / ("deep_gemm_cpp" + sysconfig.get_config_var("EXT_SUFFIX"))


def build(force: bool = False) -> Path:
    out = ext_path()
    deps = [SRC, PKG.parent / "include" / "dga_hip.h"]
    if not force and out.exists() and all(out.stat().st_mtime >= d.stat().st_mtime for d in deps):
        return out
    import torch
    from torch.utils import cpp_extension as ce
    incs = [str(PKG.parent / "include"), sysconfig.get_paths()["include"], "/opt/rocm/include"] + ce.include_paths()
    libdir = str(Path(torch.__file__).resolve().parent / "lib")
    cmd = ["/opt/rocm/bin/hipcc", "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-shared", "-w",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=deep_gemm_cpp",
           "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    cmd += [f"-I{i}" for i in incs]
    cmd += [str(SRC), "-o", str(out), f"-L{libdir}", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch",
            "-ltorch_python", f"-L{PKG}", "-ldga_hip", "-L/opt/rocm/lib", "-lamdhip64",
            "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
