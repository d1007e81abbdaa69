"""include/simd_scan.hpp (the C++ drop-in with the reference's names) -- compiled on CPU, run on GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "dropin_tests.cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "build", "dropin_tests")
LIBDIR = os.path.join(ROOT, "shared_simd_scan_amd")


THREADS_SRC = os.path.join(ROOT, "tests", "cpp", "threads_tests.cpp")
THREADS_BIN = os.path.join(ROOT, "tests", "cpp", "build", "threads_tests")


def _compile(src, exe, extra=()):
    from shared_simd_scan_amd import build

    if not os.path.exists(build.LIB_PATH):
        build.build()
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    newest = max(os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "include", "simd_scan.hpp")),
                 os.path.getmtime(os.path.join(ROOT, "include", "mi355_scan.h")))
    if not os.path.exists(exe) or os.path.getmtime(exe) < newest:
        subprocess.run(["g++", "-std=gnu++17", "-O1", "-Wall", *extra, "-I", os.path.join(ROOT, "include"), src, "-o", exe,
                        "-L", LIBDIR, "-lmi355scan", f"-Wl,-rpath,{LIBDIR}"], check=True)
    return exe


NEXT_SRC = os.path.join(ROOT, "tests", "cpp", "next_tests.cpp")
NEXT_BIN = os.path.join(ROOT, "tests", "cpp", "build", "next_tests")


def build_binary():
    _compile(THREADS_SRC, THREADS_BIN, extra=("-pthread",))
    _compile(NEXT_SRC, NEXT_BIN)
    return _compile(SRC, BIN)


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


@pytest.mark.gpu
def test_dropin_calls_from_eight_host_threads():
    """the reference's functions are re-entrant (src/simd_scan_shared.cpp:25-32 calls scan_128 from an OpenMP loop):
    8 threads x different keys through include/simd_scan.hpp, one explicit context shared by 8 threads, and threads
    mixing 16-key shared scans / scans / decompression -- every bitmap and hit count checked (tests/cpp/threads_tests.cpp)"""
    build_binary()
    res = subprocess.run([THREADS_BIN], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "All thread tests passed" in res.stdout


def test_next_row_host_compiles_with_plain_gxx():
    build_binary()
    assert subprocess.run([NEXT_BIN, "--compile-check"]).returncode == 0


@pytest.mark.gpu
def test_cpp_host_drives_the_device_pointer_api():
    """a plain C++ host (no Python, no HIP headers) on the entry points beyond the reference's surface: resident columns,
    count-only / fused-mask scans, two columns in one call, IN-list, fused selection with global row ids, shared scans of
    2 / 5 / 37 keys in both layouts, the tuning call -- against scalar loops over the generator's closed form
    (tests/cpp/next_tests.cpp)"""
    build_binary()
    res = subprocess.run([NEXT_BIN], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "All next-row tests passed" in res.stdout


CLI = os.path.join(ROOT, "cli", "shared_simd_scan_mi355")


def parse_output(output):
    """the parsing rule of the reference's scripts/prepare_shared_scan_results.py:14-20, restated"""
    rows = []
    for line in output.splitlines():
        if not line.startswith("*"):
            continue
        variant = line[2:].split(": ")[0]
        avg_runtime_ms = line.split(": ")[1].split("; ")[0][:-2]
        rows.append((variant, float(avg_runtime_ms)))
    return rows


@pytest.mark.gpu
@pytest.mark.parametrize("args", [["40", "3", "scan"], ["40", "3", "decompression"], ["40", "2", "sharedscan"],
                                  ["40", "2", "sharedscan", "3"], ["11", "1", "sharedscan", "100"],
                                  ["1", "1", "sharedscan", "512"]])
def test_cli_mirrors_reference_bench_harness(args):
    """cli/shared_simd_scan_mi355: the reference's command line (src/main.cpp:12-73), inputs, output format and
    self-checks (src/benchmark.cpp) on the GPU engine"""
    if not os.path.exists(CLI):
        subprocess.run(["make", "-C", os.path.join(ROOT, "shared_simd_scan_amd", "csrc"), "cli"], check=True)
    res = subprocess.run([CLI] + args, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "mismatch" not in res.stdout, res.stdout
    assert "finished benchmark" in res.stdout
    rows = parse_output(res.stdout)
    assert len(rows) >= 1 and all(ms > 0 for _, ms in rows)
    n = int(args[0]) * (1 << 20) * 8 // 9
    assert f"compressed input: {n} (" in res.stdout


# ---- the reference's OWN callers, built unchanged against the drop-in by oracle/Makefile (build container only; the
# binaries travel in oracle/_ref like the compiled reference itself), RUN here on the engine ---------------------------
REF_UNIT = os.path.join(ROOT, "oracle", "_ref", "ref_unit_tests_dropin")
REF_BENCH = os.path.join(ROOT, "oracle", "_ref", "ref_shared_simd_scan_dropin")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_UNIT), reason="oracle/_ref/ref_unit_tests_dropin not built (no /root/reference)")
def test_reference_own_catch_tests_run_on_the_engine():
    """test/simd_scan_tests.cpp + test/util_tests.cpp of the reference, byte for byte, linked against libmi355scan.so:
    its 6 test cases / 1175 assertions on the GPU path"""
    res = subprocess.run([REF_UNIT], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-4000:] + res.stderr[-2000:]
    assert "All tests passed (1175 assertions in 6 test cases)" in res.stdout, res.stdout[-2000:]


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_BENCH), reason="oracle/_ref/ref_shared_simd_scan_dropin not built")
@pytest.mark.parametrize("args", [["40", "2", "decompression"], ["40", "2", "scan"], ["40", "2", "sharedscan", "8"],
                                  ["11", "1", "sharedscan", "3"]])
def test_reference_own_benchmark_harness_runs_on_the_engine(args):
    """src/main.cpp + src/benchmark.cpp of the reference, unchanged, on the engine: every variant it times goes through
    include/simd_scan.hpp, and its own self-checks (`first mismatch at index`, src/benchmark.cpp:38-49,110-121) stay
    silent"""
    res = subprocess.run([REF_BENCH] + args, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-4000:] + res.stderr[-2000:]
    assert "mismatch" not in res.stdout, res.stdout[-4000:]
    rows = parse_output(res.stdout)
    assert len(rows) >= 4, res.stdout


@pytest.mark.gpu
def test_cli_runs_the_reference_predicate_sweep():
    """the reference's own sweep -- `<binary> 40 1 sharedscan P` for P = 1 .. 512
    (scripts/prepare_shared_scan_results.py:22-31), parsed with its rule -- through cli/shared_simd_scan_mi355, every run's
    self-checks silent.  tools/cli_sweep.py is that script restated; the full sweep's CSV is profiles/r03_cli_sweep_P1_512.csv
    (MI355_FULL_SWEEP=1 runs all 512 here; by default every 9th count plus both ends and the kernel boundaries)."""
    import io
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import cli_sweep
    finally:
        sys.path.pop(0)
    if not os.path.exists(CLI):
        subprocess.run(["make", "-C", os.path.join(ROOT, "shared_simd_scan_amd", "csrc"), "cli"], check=True)
    if os.environ.get("MI355_FULL_SWEEP") == "1":
        counts = list(range(1, 513))
    else:
        counts = sorted(set(range(1, 513, 9)) | {1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 32, 33, 40, 47, 63, 64, 65, 127, 128, 129, 255, 256,
                                                 257, 300, 500, 511, 512})
    buf = io.StringIO()
    bad = cli_sweep.sweep(CLI, 40, 1, counts, buf)
    assert not bad, bad[:5]
    rows = buf.getvalue().strip().splitlines()
    assert rows[0] == "data_size,predicate_count,variant,avg_runtime_ms"
    assert len(rows) == 1 + 2 * len(counts)  # per count: per-predicate and linear
