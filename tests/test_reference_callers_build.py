"""the reference's OWN callers -- test/*.cpp and src/{main,benchmark,benchmark_misc,util,profiling}.cpp --
compile and link UNCHANGED against include/simd_scan.hpp + libmi355scan.so.

The recipe is INTEGRATION.md section 1, executed: the reference's tree minus its scan library
(src/simd_scan.hpp, src/simd_scan_commons.hpp, src/simd_scan*.cpp -- the files the engine replaces), this repo's
include/ on the include path, -lmi355scan on the link line.  /root/reference is read-only, so "minus" is a farm of
symlinks in a tmp dir (gcc resolves a quoted include next to the file AS NAMED, so a symlinked main.cpp no longer sees
the reference's simd_scan.hpp beside it).  Nothing of the reference is copied anywhere; the binaries stay in the tmp
dir.  Skipped where /root/reference is absent (the GPU box): there tests/cpp/dropin_tests.cpp is the run-time half.
"""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
LIBDIR = os.path.join(ROOT, "shared_simd_scan_amd")
# the reference's own flags (CMakeLists.txt:30-38) + what its vendored Catch v1 needs on a modern glibc
CXXFLAGS = ["-std=gnu++17", "-msse3", "-msse4", "-msse4.1", "-mavx", "-mavx2", "-fopenmp", "-O1", "-w",
            "-DENABLE_PROFILING=0"]
REPLACED = ("simd_scan.hpp", "simd_scan_commons.hpp")  # + every src/simd_scan*.cpp

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="/root/reference not present")


def _farm(tmp_path):
    """<tmp>/src, <tmp>/test, <tmp>/lib: symlinks to every reference file except the scan library itself"""
    for sub in ("src", "test"):
        os.makedirs(tmp_path / sub)
        for f in sorted(os.listdir(os.path.join(REF, sub))):
            if sub == "src" and (f in REPLACED or (f.startswith("simd_scan") and f.endswith(".cpp"))):
                continue
            os.symlink(os.path.join(REF, sub, f), tmp_path / sub / f)
    os.symlink(os.path.join(REF, "lib"), tmp_path / "lib")
    return tmp_path


def _lib():
    from shared_simd_scan_amd import build

    if not os.path.exists(build.LIB_PATH):
        build.build()
    return build.LIB_PATH


def _gxx(sources, exe, includes, extra=()):
    cmd = ["g++", *CXXFLAGS, *extra]
    for inc in includes:
        cmd += ["-I", str(inc)]
    cmd += [str(s) for s in sources] + ["-o", str(exe), "-L", LIBDIR, "-lmi355scan", f"-Wl,-rpath,{LIBDIR}"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, " ".join(cmd) + "\n" + res.stdout + res.stderr
    return exe


def _undefined_symbols(exe):
    out = subprocess.run(["nm", "-D", "--undefined-only", str(exe)], capture_output=True, text=True, check=True).stdout
    return {line.split()[-1] for line in out.splitlines() if line.strip()}


def test_reference_unit_tests_build_against_the_dropin(tmp_path):
    """test/simd_scan_tests.cpp + test/util_tests.cpp + src/util.cpp (CMakeLists.txt:58-73 `unit_tests`)"""
    _lib()
    t = _farm(tmp_path)
    srcs = sorted(glob.glob(str(t / "test" / "*.cpp"))) + [t / "src" / "util.cpp"]
    assert len(srcs) == 3
    exe = _gxx(srcs, t / "unit_tests", [t / "test", t / "src", os.path.join(ROOT, "include"), t / "lib" / "catch"],
               extra=("-DCATCH_CONFIG_NO_POSIX_SIGNALS",))
    und = _undefined_symbols(exe)
    # the scan work is the engine's: the test binary binds the C ABI, not a CPU implementation
    for sym in ("mi355_pack_u16", "mi355_decompress", "mi355_scan_eq", "mi355_shared_scan_eq",
                "mi355_shared_scan_eq_linear"):
        assert sym in und, (sym, sorted(s for s in und if s.startswith("mi355")))
    # Catch's own command line works without a device: the six reference test cases are all there
    res = subprocess.run([str(exe), "--list-test-names-only"], capture_output=True, text=True, timeout=60)
    names = [l for l in res.stdout.splitlines() if l.strip()]
    assert names == ["Compress and decompress", "SIMD Scan", "Shared SIMD Scan", "Simple Shared SIMD Scan",
                     "Find next multiple", "Get bit in vector"], res.stdout + res.stderr
    # the util cases need no GPU and run here: the reference's util.hpp / util.cpp are the ones in the link
    res = subprocess.run([str(exe), "[util]"], capture_output=True, text=True, timeout=60)
    assert res.returncode == 0 and "All tests passed" in res.stdout, res.stdout + res.stderr


def test_reference_benchmark_executable_builds_against_the_dropin(tmp_path):
    """src/{main,benchmark,benchmark_misc,util,profiling}.cpp (CMakeLists.txt:15-19 `shared_simd_scan`)"""
    _lib()
    t = _farm(tmp_path)
    srcs = [t / "src" / f for f in ("main.cpp", "benchmark.cpp", "benchmark_misc.cpp", "util.cpp", "profiling.cpp")]
    assert sorted(os.path.basename(p) for p in glob.glob(str(t / "src" / "*.cpp"))) == sorted(s.name for s in srcs)
    exe = _gxx(srcs, t / "shared_simd_scan", [t / "src", os.path.join(ROOT, "include")])
    und = _undefined_symbols(exe)
    for sym in ("mi355_pack_u16", "mi355_decompress", "mi355_scan_eq", "mi355_shared_scan_eq",
                "mi355_shared_scan_eq_linear"):
        assert sym in und, sym
    # argument handling is host code of the reference and runs without a device (src/main.cpp:20-29)
    res = subprocess.run([str(exe), "1"], capture_output=True, text=True, timeout=60)
    assert res.returncode == 1 and res.stdout.startswith("Format: ./shared_simd_scan data_size repetitions"), res.stdout


def test_either_include_order_and_standalone(tmp_path):
    """util.hpp before or after the header inside the reference's tree; and no util.hpp at all (standalone)"""
    _lib()
    t = _farm(tmp_path)
    body = "int main(){ std::vector<uint8_t> v{5,5}; return (next_multiple(5,8)==8 && get_bit(v,0) && POPCNT(7)==3 && " \
           "scan_output_buffer_size(16)==34) ? 0 : 1; }\n"
    cases = {
        "before.cpp": '#include "util.hpp"\n#include "simd_scan.hpp"\n' + body,
        "after.cpp": '#include "simd_scan.hpp"\n#include "util.hpp"\n' + body,
    }
    for name, text in cases.items():
        (t / "src" / name).write_text(text)
        exe = _gxx([t / "src" / name, t / "src" / "util.cpp"], t / name.replace(".cpp", ""),
                   [t / "src", os.path.join(ROOT, "include")])
        assert subprocess.run([str(exe)]).returncode == 0
    alone = tmp_path / "alone"
    os.makedirs(alone)
    (alone / "alone.cpp").write_text('#include "simd_scan.hpp"\n' + body)
    exe = _gxx([alone / "alone.cpp"], alone / "alone", [os.path.join(ROOT, "include")])
    assert subprocess.run([str(exe)]).returncode == 0
