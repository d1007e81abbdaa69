"""Build driver for the in-tree HIP library (libmi355scan.so).  hipcc cross-compiles gfx950 without a GPU."""
from __future__ import annotations

import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libmi355scan.so")


def build(jobs: int = 8, verbose: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 and link shared_simd_scan_amd/libmi355scan.so."""
    cmd = ["make", "-C", CSRC, f"-j{jobs}"]
    res = subprocess.run(cmd, stdout=None if verbose else subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libmi355scan.so failed:\n" + (res.stdout or ""))
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} missing after build")
    return LIB_PATH


def kernel_sources_sha() -> str:
    """sha256 (first 16 hex digits) over the device-side sources: identifies the kernels a profile was taken on"""
    import glob
    import hashlib

    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "kernels", "*.hpp"))) + [os.path.join(CSRC, f) for f in
                                                                           ("kernels.hpp", "dispatch.hpp", "width_group.hip")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(build(verbose=True))
