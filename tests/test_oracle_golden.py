"""CPU: pins oracle/oracle.c to the reference.

Golden vectors (tests/golden/ref_w*.npz) were produced by the reference's own functions compiled
from /root/reference (tests/golden/make_golden.py); the known-answer cases restate
test/simd_scan_tests.cpp and test/util_tests.cpp of the reference.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN_SIZES, GOLDEN_WIDTHS


def bits(buf, n):
    return np.unpackbits(np.ascontiguousarray(buf, dtype=np.uint8), bitorder="little")[:n]


@pytest.mark.parametrize("w", GOLDEN_WIDTHS)
@pytest.mark.parametrize("n", GOLDEN_SIZES)
def test_pack_matches_reference_bytes(O, golden, w, n):
    g = golden[w]
    values, packed = g[f"n{n}_values"], g[f"n{n}_packed"]
    mine = O.pack(values if w > 16 else values.astype(np.uint16), w)
    assert mine.shape == packed.shape  # compressed_buffer_size: ceil(c*n/8)+256 (src/simd_scan.hpp:20-26)
    assert np.array_equal(mine, packed)


@pytest.mark.parametrize("w", GOLDEN_WIDTHS)
@pytest.mark.parametrize("n", GOLDEN_SIZES)
def test_decompress_matches_reference(O, golden, w, n):
    g = golden[w]
    out = O.decompress(g[f"n{n}_packed"], n, w)
    assert np.array_equal(out, g[f"n{n}_decomp"])
    assert np.array_equal(out, g[f"n{n}_values"].astype(np.int32))


@pytest.mark.parametrize("w", GOLDEN_WIDTHS)
@pytest.mark.parametrize("n", GOLDEN_SIZES)
def test_scan_eq_matches_reference(O, golden, w, n):
    g = golden[w]
    packed = g[f"n{n}_packed"]
    for key, ref_buf, ref_hits in zip(g[f"n{n}_keys"], g[f"n{n}_scan128"], g[f"n{n}_scan128_hits"]):
        bm, hits = O.scan_eq(packed, n, w, int(key))
        assert np.array_equal(bits(bm, n), bits(ref_buf, n))
        if key != 0:
            # key 0 also "matches" the zero pad in the reference (SURVEY 8c hazard 1)
            assert hits == ref_hits
            assert np.array_equal(bm, ref_buf[: bm.shape[0]])
            assert not ref_buf[bm.shape[0]:].any()
        else:
            assert hits == int(bits(ref_buf, n).sum())
        # canonical tail: bits >= n are zero
        assert bits(bm, bm.shape[0] * 8)[n:].sum() == 0


@pytest.mark.parametrize("w", GOLDEN_WIDTHS)
@pytest.mark.parametrize("n", GOLDEN_SIZES)
@pytest.mark.parametrize("threads", [1, 3])
def test_avx2_restatement_matches_reference(O, golden, w, n, threads):
    """oracle_avx2.c (the CPU baseline of bench.py on hosts without oracle/_ref) against the reference's scan_128 vectors:
    first n bits always; the whole buffer and the hit count whenever the reference's variants agree (key != 0)."""
    if not O.avx2_available():
        pytest.skip("host CPU has no AVX2")
    g = golden[w]
    packed = g[f"n{n}_packed"]
    for key, ref_buf, ref_hits in zip(g[f"n{n}_keys"], g[f"n{n}_scan128"], g[f"n{n}_scan128_hits"]):
        out, hits, _ = O.scan_eq_avx2(packed, n, w, int(key), threads=threads)
        assert np.array_equal(bits(out, n), bits(ref_buf, n))
        assert hits == bits(ref_buf, n).sum()
        if key != 0:
            assert np.array_equal(out, ref_buf[: out.shape[0]]) and hits == ref_hits


@pytest.mark.parametrize("c", list(range(1, 33)))
def test_avx2_restatement_matches_scalar_oracle_all_widths(O, c):
    """every width (c > 25 takes the scalar path inside), ragged sizes, keys at the ends of the domain and outside it"""
    rng = np.random.default_rng(4000 + c)
    for n in (1, 8, 31, 33, 255, 257, 4101, 70_001):
        vals = rng.integers(0, 1 << c, size=n, dtype=np.uint64).astype(np.uint32)
        packed = O.pack(vals, c)
        for key in (int(vals[0]), int(vals[n // 2]), 0, (1 << c) - 1, (1 << c) + 1 if c < 31 else 5, -1):
            a, ha = O.scan_eq(packed, n, c, key)
            b, hb, _ = O.scan_eq_avx2(packed, n, c, key, threads=2)
            assert np.array_equal(a, b) and ha == hb, (c, n, key)


@pytest.mark.parametrize("w", GOLDEN_WIDTHS)
@pytest.mark.parametrize("n", GOLDEN_SIZES)
@pytest.mark.parametrize("P", [1, 3, 8])
def test_shared_scan_matches_reference(O, golden, w, n, P):
    g = golden[w]
    packed, keys = g[f"n{n}_packed"], g[f"n{n}_shared_keys_P{P}"]
    ref_std, ref_lin = g[f"n{n}_shared_std_P{P}"], g[f"n{n}_linear_std_P{P}"]
    pp, hits = O.shared_scan_eq(packed, n, w, keys, "per_predicate")
    lin, hits2 = O.shared_scan_eq(packed, n, w, keys, "linear")
    assert np.array_equal(hits, hits2)
    nb = (n + 7) // 8
    for k in range(P):
        assert np.array_equal(bits(pp[k], n), bits(ref_std[k], n))
        assert hits[k] == bits(ref_std[k], n).sum()
        # linear layout: byte of group g and key k at g*P+k (src/simd_scan_shared_linear.cpp:57)
        assert np.array_equal(bits(lin[k::P][:nb], n), bits(ref_lin[k::P][:nb], n))
    full = n // 8
    assert np.array_equal(lin[: full * P], ref_lin[: full * P])


def test_reference_tail_variants_agree_on_first_n_bits(O):
    """SURVEY 8c hazard 1: variants differ only past n; oracle = canonical zero tail."""
    import os

    from conftest import GOLDEN_DIR

    g = np.load(os.path.join(GOLDEN_DIR, "ref_tail_w9.npz"))
    n = 13
    bm, hits = O.scan_eq(g["packed"], n, 9, 0)
    assert hits == 2
    assert bm.tolist() == [0x09, 0x00]
    for v in ("scan_unvectorized", "scan_128", "scan_128_unrolled", "scan_256", "scan_256_unrolled"):
        assert np.array_equal(bits(g[v], n), bits(bm, n))
    # the variant-specific tails really are different (documents why parity is defined on [0,n))
    assert int(g["scan_128_hits"]) == 5 and int(g["scan_128_unrolled_hits"]) == 21


# ---- the reference's own known-answer tests, restated -------------------------------------

KAT = np.array([1, 2, 3, 3, 2, 1, 1, 2, 3, 1, 2, 3], dtype=np.uint16)


def test_kat_compress_decompress_roundtrip(O):
    # test/simd_scan_tests.cpp:6-43 -- n = 2^9-3 identity column
    n = (1 << 9) - 3
    values = np.arange(n, dtype=np.uint16)
    packed = O.pack(values, 9)
    assert np.array_equal(O.decompress(packed, n, 9), values.astype(np.int32))


def test_kat_pack_bytes(O):
    # {1..13}@9 -> 01 04 0c 20 50 c0 c0 01 04 09 14 2c 60 d0 00
    packed = O.pack(np.arange(1, 14, dtype=np.uint16), 9)
    assert packed[:15].tobytes().hex() == "01040c2050c0c0010409142c60d000"


def test_kat_scan(O):
    # test/simd_scan_tests.cpp:45-82
    packed = O.pack(KAT, 9)
    bm, hits = O.scan_eq(packed, 12, 9, 3)
    assert hits == 4
    for i in range(12):
        assert O.get_bit(bm, i) == (KAT[i] == 3)


def test_kat_shared_scan(O):
    # test/simd_scan_tests.cpp:84-106
    packed = O.pack(KAT, 9)
    out, hits = O.shared_scan_eq(packed, 12, 9, [1, 2, 3])
    for k, key in enumerate([1, 2, 3]):
        for i in range(12):
            assert O.get_bit(out[k], i) == (KAT[i] == key)
    assert hits.tolist() == [4, 4, 4]


def test_kat_linear_simple(O, golden):
    # test/simd_scan_tests.cpp:108-150
    packed = O.pack(KAT, 9)
    one, _ = O.shared_scan_eq(packed, 12, 9, [1], "linear")
    cmp1, h = O.scan_eq(packed, 12, 9, 1)
    assert h == 4 and np.array_equal(one, cmp1)
    two, _ = O.shared_scan_eq(packed, 12, 9, [2, 3], "linear")
    for k, key in enumerate([2, 3]):
        c, h = O.scan_eq(packed, 12, 9, key)
        assert h == 4 and np.array_equal(two[k::2], c)
    g = golden[9]
    assert np.array_equal(one, g["n12_linear_simple_P1"][:2])
    assert not g["n12_linear_simple_P1"][2:].any()  # untouched padding stays 0 (hazard 4)
    assert np.array_equal(two, g["n12_linear_simple_P2"][:4])


def test_kat_get_bit(O):
    # test/util_tests.cpp:15-36
    v = np.array([5, 5], dtype=np.uint8)
    expect = [1, 0, 1, 0, 0, 0, 0, 0] * 2
    assert [int(O.get_bit(v, i)) for i in range(16)] == expect


def test_sizing(O):
    # src/simd_scan.hpp:20-40
    assert O.compressed_buffer_size(9, 13) == 15 + 256
    assert O.compressed_buffer_size(9, 8) == 9 + 256
    assert O.decompression_output_buffer_size(10) == 72
    assert O.scan_output_buffer_size(12) == 2 + 32
    assert O.scan_output_buffer_size(16) == 2 + 32


def test_out_of_range_keys_never_match(O):
    # SURVEY 8c hazard 5
    vals = np.arange(16, dtype=np.uint16) + 500
    vals[vals > 511] = 511
    packed = O.pack(vals, 9)
    for key in (515, 1027, 65539, -1):
        bm, hits = O.scan_eq(packed, 16, 9, key)
        assert hits == 0 and not bm.any()
    bm, hits = O.scan_eq(packed, 16, 9, 511)
    assert hits == int((vals == 511).sum())


def test_range_scan_semantics(O):
    # src/simd_scan.hpp:76-84: predicate_low <= value <= predicate_high (inclusive)
    rng = np.random.default_rng(3)
    for c in (5, 9, 12, 21, 32):
        n = 1003
        vals = rng.integers(0, 1 << c, size=n, dtype=np.uint64).astype(np.uint32)
        packed = O.pack(vals, c)
        lo, hi = (1 << c) // 4, (1 << c) // 2
        bm, hits = O.scan_range(packed, n, c, lo, hi)
        expect = (vals >= lo) & (vals <= hi)
        assert np.array_equal(bits(bm, n), expect.astype(np.uint8))
        assert hits == expect.sum()


def test_generators_match_reference_bench_inputs(O):
    # src/benchmark.cpp:173 (i%5), :277 (i%P%512), :81 (i & 511)
    n = 1000
    i = np.arange(n, dtype=np.uint64)
    assert np.array_equal(O.gen_values("mod", n, 9, 5), (i % 5).astype(np.uint32))
    assert np.array_equal(O.gen_values("mod", n, 9, 8), (i % 8 % 512).astype(np.uint32))
    assert np.array_equal(O.gen_values("index", n, 9), (i & 511).astype(np.uint32))
    a = O.gen_values("splitmix", n, 9, 42)
    b = O.gen_values("splitmix", 100, 9, 42, first=900)
    assert np.array_equal(a[900:], b) and a.max() < 512 and len(np.unique(a)) > 400


def test_cfg1_reference_vs_oracle_1e7(O):
    """BASELINE config 1 (n = 1e7, c = 9, v = i % 5, key 3; src/benchmark.cpp:173,:150): when the reference itself is
    available (oracle/_ref, built from /root/reference by oracle/Makefile) run its AVX2 and scalar scans and hold the
    oracle against them on the full column; hits must be 2,000,000."""
    from oracle import RefLib, ref_available

    if not ref_available(9):
        pytest.skip("oracle/_ref/libref_w9.so not built (needs /root/reference)")
    R = RefLib(9)
    n, c = 10_000_000, 9
    vals = O.gen_values("mod", n, c, 5)
    packed = O.pack(vals, c)
    assert np.array_equal(packed, R.compress(vals.astype(np.uint16)))
    obm, ohits = O.scan_eq(packed, n, c, 3)
    assert ohits == 2_000_000
    for variant in ("scan_256_unrolled", "scan_256", "scan_128_unrolled", "scan_128", "scan_unvectorized"):
        rbuf, rhits = R.scan(variant, 3, packed, n)
        assert rhits == 2_000_000, variant
        assert np.array_equal(rbuf[: n // 8], obm), variant  # n is a multiple of 32: whole-buffer equality
    dec = R.decompress("decompress_256_avx2", packed, n)[:n]
    assert np.array_equal(dec, O.decompress(packed, n, c))
    keys = list(range(8))
    vals8 = O.gen_values("mod", n, c, 8)
    packed8 = O.pack(vals8, c)
    ref = R.shared_scan("shared_scan_128_standard", keys, packed8, n)
    mine, hits = O.shared_scan_eq(packed8, n, c, keys)
    assert np.array_equal(ref[:, : n // 8], mine) and hits.tolist() == [n // 8] * 8


def test_committed_fixtures_are_what_the_reference_produces(tmp_path):
    """tests/golden/*.npz are data produced by the reference itself: regenerating them with the committed script
    (tests/golden/make_golden.py, which drives oracle/_ref = the reference compiled from its own sources) gives
    array-for-array the committed files.  Needs oracle/_ref (built by __graft_entry__.build() where /root/reference
    exists); skipped elsewhere."""
    import glob
    import subprocess
    import sys

    from oracle import ref_available

    if not all(ref_available(w) for w in (5, 7, 9, 12, 17, 21)):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    subprocess.run([sys.executable, os.path.join(here, "make_golden.py"), str(tmp_path)], check=True, capture_output=True)
    names = sorted(os.path.basename(f) for f in glob.glob(os.path.join(here, "*.npz")))
    assert names == sorted(os.path.basename(f) for f in glob.glob(os.path.join(str(tmp_path), "*.npz")))
    for name in names:
        a, b = np.load(os.path.join(here, name)), np.load(os.path.join(str(tmp_path), name))
        assert set(a.files) == set(b.files), name
        for k in a.files:
            assert np.array_equal(a[k], b[k]), (name, k)
