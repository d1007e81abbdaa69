"""CPU: the C-ABI library loads and exports every symbol include/mi355_scan.h declares (no compute)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from shared_simd_scan_amd import build, lib

    if not os.path.exists(build.LIB_PATH):
        build.build()
    return lib()


def header_symbols():
    text = open(os.path.join(ROOT, "include", "mi355_scan.h")).read()
    return sorted(set(re.findall(r"^MI355_API [^;(]*?\b(mi355_\w+)\(", text, flags=re.M)))


def test_header_declares_what_python_binds():
    from shared_simd_scan_amd._capi import SYMBOLS

    assert sorted(s[0] for s in SYMBOLS) == header_symbols()


def test_library_exports_every_declared_symbol(L):
    for name in header_symbols():
        assert hasattr(L, name), name


def test_sizing_helpers_match_reference_formulas(L):
    # src/simd_scan.hpp:20-40
    assert L.mi355_compressed_buffer_size(9, 13) == 15 + 256
    assert L.mi355_compressed_buffer_size(9, 8) == 9 + 256
    assert L.mi355_decompression_output_buffer_size(10) == 72
    assert L.mi355_scan_output_buffer_size(12) == 34
    assert L.mi355_scan_output_buffer_size(16) == 34
    assert L.mi355_tile_values(9) == 8192 and L.mi355_tile_values(21) == 4096 and L.mi355_tile_values(33) == 0


def test_fails_loudly_without_a_gpu(L):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ctx = C.c_void_p()
    rc = L.mi355_ctx_create(0, None, C.byref(ctx))
    assert rc == -3 and b"no CPU fallback" in L.mi355_last_error()
    # compute entry points refuse too (default context cannot be created)
    buf = (C.c_uint8 * 512)()
    hits = C.c_uint64()
    assert L.mi355_scan_eq(None, buf, 12, 9, 3, buf, C.byref(hits)) != 0
    from shared_simd_scan_amd import Mi355Error, ScanEngine

    with pytest.raises(Mi355Error):
        ScanEngine()


def test_oracle_is_not_reachable_from_the_product():
    """The shipped package must not import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "shared_simd_scan_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.replace("no CPU fallback", ""), os.path.join(dirpath, f)
    assert "oracle" not in open(os.path.join(ROOT, "include", "mi355_scan.h")).read()


def test_header_is_plain_c99_and_links_from_c(L, tmp_path):
    """the boundary is a C ABI: include/mi355_scan.h compiles as strict C99 (no C++ types, no extensions) and a C program
    links against the library and calls the entry points that need no device (sizing helpers, shard arithmetic, errors)"""
    import subprocess

    from shared_simd_scan_amd import build

    src = tmp_path / "c_caller.c"
    src.write_text(r'''
#include "mi355_scan.h"
#include <stdio.h>
#include <string.h>
int main(void)
{
    uint64_t first = 0, count = 0;
    if (mi355_compressed_buffer_size(9, 13) != 15 + 256) return 1;
    if (mi355_scan_output_buffer_size(8) != 1 + 32) return 2;
    if (mi355_bitmap_stride(1000) != 256) return 3;
    if (mi355_shard_rows(1000000, 8, 7, &first, &count) != MI355_OK || first + count != 1000000 || first % 8192) return 4;
    if (mi355_shard_rows(10, 0, 0, &first, &count) == MI355_OK) return 5; /* world = 0: rejected */
    if (!mi355_last_error() || !strlen(mi355_last_error())) return 6;
    if (!mi355_kernel_name("scan_eq", 9) || mi355_kernel_name("nonsense", 9)) return 7;
    if (MI355_TUNE_ALL != (MI355_TUNE_SCAN | MI355_TUNE_COUNT | MI355_TUNE_MASK | MI355_TUNE_DECOMPRESS)) return 8;
    printf("c99 ok\n");
    return 0;
}
''')
    exe = tmp_path / "c_caller"
    libdir = os.path.dirname(build.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", libdir, "-lmi355scan", f"-Wl,-rpath,{libdir}"], check=True)
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert res.returncode == 0 and "c99 ok" in res.stdout, (res.returncode, res.stdout, res.stderr)
