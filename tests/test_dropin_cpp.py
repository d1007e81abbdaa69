"""include/simd_scan.hpp (the C++ drop-in with the reference's names) -- compiled on CPU, run on GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "dropin_tests.cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "build", "dropin_tests")
LIBDIR = os.path.join(ROOT, "shared_simd_scan_amd")


def build_binary():
    from shared_simd_scan_amd import build

    if not os.path.exists(build.LIB_PATH):
        build.build()
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < max(os.path.getmtime(SRC), os.path.getmtime(
            os.path.join(ROOT, "include", "simd_scan.hpp"))):
        subprocess.run(["g++", "-std=gnu++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), SRC, "-o", BIN,
                        "-L", LIBDIR, "-lmi355scan", f"-Wl,-rpath,{LIBDIR}"], check=True)
    return BIN


def test_dropin_header_compiles_and_links_with_plain_gxx():
    exe = build_binary()
    assert subprocess.run([exe, "--compile-check"]).returncode == 0


@pytest.mark.gpu
def test_reference_unit_tests_pass_through_the_dropin_header():
    exe = build_binary()
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "All tests passed" in res.stdout
    assert "not supported for 3 predicate keys!" in res.stderr
