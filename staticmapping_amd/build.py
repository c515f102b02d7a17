"""In-tree build of libsmhip.so (gfx950 only).  `python -m staticmapping_amd.build`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libsmhip.so")
SHARD_EXE = os.path.join(LIB_DIR, "smhip_shard")          # C++ sharded sequence driver (RCCL gather), csrc/shard_driver.cc

HIP_SOURCES = ["smhip_api.hip", "prep_normals.hip", "cloud_filters.hip", "smhip_mrvm.hip", "host_cloud.cc"]            # translation units (each may #include kernel files)
HIP_DEPS = ["icp_kernels.hip", "icp_one.hip", "nabo_kernels.hip", "kd_median_tree.h", "smhip_device.h", "host_cloud.cc", "prep_normals.h", "ndt_kernels.hip", "smhip_ndt_api.hip",
            "gicp_kernels.hip", "smhip_gicp_api.hip", "smhip_filter_api.hip", "cloud_filters.h",
            os.path.join("..", "..", "include", "smhip.h")]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libsmhip.so can only be built with the ROCm toolchain")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    for f in HIP_SOURCES + HIP_DEPS:
        p = os.path.normpath(os.path.join(CSRC, f))
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build_shard_driver(force: bool = False, verbose: bool = False) -> str:
    """The C++ host program of BASELINE config #4: links libsmhip.so + librccl.so (no device code of its own)."""
    src = os.path.join(CSRC, "shard_driver.cc")
    deps = [src, os.path.join(ROOT, "include", "smhip.h"), LIB_PATH]
    if not force and os.path.exists(SHARD_EXE) and all(os.path.getmtime(d) <= os.path.getmtime(SHARD_EXE) for d in deps):
        return SHARD_EXE
    cmd = [_hipcc(), "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), src, "-o", SHARD_EXE,
           "-L", LIB_DIR, "-lsmhip", "-L", "/opt/rocm/lib", "-lrccl", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return SHARD_EXE


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 into staticmapping_amd/lib/libsmhip.so (+ the smhip_shard driver)."""
    if not force and not needs_build():
        build_shard_driver(False, verbose)
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I", os.path.join(ROOT, "include"), "-o", LIB_PATH] + os.environ.get("SMHIP_EXTRA_HIPCC_FLAGS", "").split() + srcs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    build_shard_driver(True, verbose)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
