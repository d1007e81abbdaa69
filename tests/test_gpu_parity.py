"""GPU (-m gpu): the HIP path through the C ABI vs the oracle and vs the reference's golden vectors.

Bit-exact everywhere: this is integer / bit work, there is no tolerance.  Parity is defined on bits
[0, n) plus the canonical zero tail (DESIGN.md "tail rule"); hits = popcount over [0, n).
"""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN_SIZES, GOLDEN_WIDTHS

pytestmark = pytest.mark.gpu


def bits(buf, n):
    return np.unpackbits(np.ascontiguousarray(buf, dtype=np.uint8), bitorder="little")[:n]


def vp(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def L():
    from shared_simd_scan_amd import lib

    return lib()


@pytest.fixture(scope="module")
def eng():
    import torch

    from shared_simd_scan_amd import ScanEngine

    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    return ScanEngine()


def ok(L, rc):
    assert rc == 0, L.mi355_last_error()


# ------------------------------------------------------------------------------------------------
# host-pointer C ABI (the drop-in flavour) against the reference's golden vectors
# ------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("w", GOLDEN_WIDTHS)
@pytest.mark.parametrize("n", GOLDEN_SIZES)
def test_golden_pack_decompress(L, golden, w, n):
    g = golden[w]
    values, ref_packed = g[f"n{n}_values"], g[f"n{n}_packed"]
    packed = np.full(L.mi355_compressed_buffer_size(w, n), 0xAA, dtype=np.uint8)
    if w <= 16:
        v16 = np.ascontiguousarray(values.astype(np.uint16))
        ok(L, L.mi355_pack_u16(None, vp(v16), n, w, vp(packed)))
    else:
        v32 = np.ascontiguousarray(values.astype(np.uint32))
        ok(L, L.mi355_pack_u32(None, vp(v32), n, w, vp(packed)))
    assert np.array_equal(packed, ref_packed)  # byte-identical incl. the zero pad
    out = np.full(n, -1, dtype=np.int32)
    ok(L, L.mi355_decompress(None, vp(ref_packed), n, w, vp(out)))
    assert np.array_equal(out, g[f"n{n}_decomp"])


@pytest.mark.parametrize("w", GOLDEN_WIDTHS)
@pytest.mark.parametrize("n", GOLDEN_SIZES)
def test_golden_scan_eq(L, golden, w, n):
    g = golden[w]
    packed = np.ascontiguousarray(g[f"n{n}_packed"])
    sobs = L.mi355_scan_output_buffer_size(n)
    nb = (n + 7) // 8
    for key, ref_buf, ref_hits in zip(g[f"n{n}_keys"], g[f"n{n}_scan128"], g[f"n{n}_scan128_hits"]):
        out = np.zeros(sobs, dtype=np.uint8)
        hits = C.c_uint64(123)
        ok(L, L.mi355_scan_eq(None, vp(packed), n, w, int(key), vp(out), C.byref(hits)))
        assert np.array_equal(bits(out, n), bits(ref_buf, n))
        assert not out[nb:].any()  # untouched padding stays 0 (SURVEY 8c hazard 4)
        if key != 0:
            assert np.array_equal(out, ref_buf)  # whole padded buffer identical to scan_128's
            assert hits.value == ref_hits
        else:
            assert hits.value == bits(ref_buf, n).sum()
            assert bits(out, nb * 8)[n:].sum() == 0


@pytest.mark.parametrize("w", GOLDEN_WIDTHS)
@pytest.mark.parametrize("n", GOLDEN_SIZES)
@pytest.mark.parametrize("P", [1, 3, 8])
def test_golden_shared_scan(L, golden, w, n, P):
    g = golden[w]
    packed = np.ascontiguousarray(g[f"n{n}_packed"])
    keys = np.ascontiguousarray(g[f"n{n}_shared_keys_P{P}"].astype(np.int32))
    ref_std, ref_lin = g[f"n{n}_shared_std_P{P}"], g[f"n{n}_linear_std_P{P}"]
    sobs = L.mi355_scan_output_buffer_size(n)
    nb, full = (n + 7) // 8, n // 8
    outs = [np.zeros(sobs, dtype=np.uint8) for _ in range(P)]
    ptrs = (C.c_void_p * P)(*[o.ctypes.data for o in outs])
    hits = np.zeros(P, dtype=np.uint64)
    ok(L, L.mi355_shared_scan_eq(None, vp(packed), n, w, vp(keys), P, ptrs, vp(hits)))
    lin = np.zeros(P * sobs, dtype=np.uint8)
    hits2 = np.zeros(P, dtype=np.uint64)
    ok(L, L.mi355_shared_scan_eq_linear(None, vp(packed), n, w, vp(keys), P, vp(lin), vp(hits2)))
    assert np.array_equal(hits, hits2)
    for k in range(P):
        assert np.array_equal(bits(outs[k], n), bits(ref_std[k], n))
        assert hits[k] == bits(ref_std[k], n).sum()
        assert not outs[k][nb:].any()
    assert np.array_equal(lin[: full * P], ref_lin[: full * P])
    for k in range(P):
        assert np.array_equal(bits(lin[k::P][:nb], n), bits(ref_lin[k::P][:nb], n))
    assert not lin[nb * P:].any()


def test_kat_reference_unit_tests(L, golden):
    """test/simd_scan_tests.cpp restated against the C ABI."""
    kat = np.array([1, 2, 3, 3, 2, 1, 1, 2, 3, 1, 2, 3], dtype=np.uint16)
    packed = np.zeros(L.mi355_compressed_buffer_size(9, 12), dtype=np.uint8)
    ok(L, L.mi355_pack_u16(None, vp(kat), 12, 9, vp(packed)))
    sobs = L.mi355_scan_output_buffer_size(12)
    # "SIMD Scan" (:45-82)
    out = np.zeros(sobs, dtype=np.uint8)
    hits = C.c_uint64()
    ok(L, L.mi355_scan_eq(None, vp(packed), 12, 9, 3, vp(out), C.byref(hits)))
    assert hits.value == 4
    assert bits(out, 12).tolist() == (kat == 3).astype(int).tolist()
    # "Simple Shared SIMD Scan" (:108-150): outputs1 == compare_output, whole padded vector
    keys1 = np.array([1], dtype=np.int32)
    lin1 = np.zeros(sobs, dtype=np.uint8)
    ok(L, L.mi355_shared_scan_eq_linear(None, vp(packed), 12, 9, vp(keys1), 1, vp(lin1), None))
    cmp1 = np.zeros(sobs, dtype=np.uint8)
    ok(L, L.mi355_scan_eq(None, vp(packed), 12, 9, 1, vp(cmp1), C.byref(hits)))
    assert hits.value == 4 and np.array_equal(lin1, cmp1)
    assert np.array_equal(lin1, golden[9]["n12_linear_simple_P1"])
    keys2 = np.array([2, 3], dtype=np.int32)
    lin2 = np.zeros(2 * sobs, dtype=np.uint8)
    ok(L, L.mi355_shared_scan_eq_linear(None, vp(packed), 12, 9, vp(keys2), 2, vp(lin2), None))
    assert np.array_equal(lin2, golden[9]["n12_linear_simple_P2"])
    # "Compress and decompress" (:6-43)
    n = (1 << 9) - 3
    ident = np.arange(n, dtype=np.uint16)
    packed = np.zeros(L.mi355_compressed_buffer_size(9, n), dtype=np.uint8)
    ok(L, L.mi355_pack_u16(None, vp(ident), n, 9, vp(packed)))
    dec = np.zeros(n, dtype=np.int32)
    ok(L, L.mi355_decompress(None, vp(packed), n, 9, vp(dec)))
    assert np.array_equal(dec, ident.astype(np.int32))


def test_invalid_arguments(L):
    buf = np.zeros(600, dtype=np.uint8)
    hits = C.c_uint64()
    assert L.mi355_scan_eq(None, vp(buf), 12, 0, 3, vp(buf), C.byref(hits)) == -1
    assert L.mi355_scan_eq(None, vp(buf), 12, 33, 3, vp(buf), C.byref(hits)) == -1
    assert L.mi355_scan_eq(None, None, 12, 9, 3, vp(buf), C.byref(hits)) == -1
    keys = np.zeros(4, dtype=np.int32)
    assert L.mi355_shared_scan_eq_linear(None, vp(buf), 12, 9, vp(keys), 0, vp(buf), None) == -1
    assert L.mi355_shared_scan_eq_linear(None, vp(buf), 12, 9, vp(keys), 1025, vp(buf), None) == -1
    # n == 0 is fine and writes nothing
    out = np.full(40, 7, dtype=np.uint8)
    ok(L, L.mi355_scan_eq(None, vp(buf), 0, 9, 3, vp(out), C.byref(hits)))
    assert hits.value == 0 and (out == 7).all()


# ------------------------------------------------------------------------------------------------
# device-resident path (ScanEngine) vs the oracle on seeded random columns
# ------------------------------------------------------------------------------------------------

def make_column(O, eng, n, c, seed):
    import torch

    rng = np.random.default_rng(seed)
    vals = rng.integers(0, 1 << c, size=n, dtype=np.uint64).astype(np.uint32)
    col = eng.compress(torch.from_numpy(vals.view(np.int32)).cuda(), c)
    return vals, col


@pytest.mark.parametrize("c", list(range(1, 33)))
def test_all_widths_vs_oracle(O, eng, c):
    """every width 1..32, n chosen to leave a ragged tail in the last tile"""
    import torch

    n = 3 * 8192 + 1237
    vals, col = make_column(O, eng, n, c, 100 + c)
    packed_host = col.data.cpu().numpy()
    assert np.array_equal(packed_host, O.pack(vals, c))  # device packer == reference format
    dec = eng.decompress(col)
    assert np.array_equal(dec.cpu().numpy().view(np.uint32), vals)
    for key in (int(vals[17]), 0, (1 << c) - 1):
        bm, hits = eng.scan(key, col)
        obm, ohits = O.scan_eq(packed_host, n, c, key if key < 2 ** 31 else key - 2 ** 32)
        assert np.array_equal(bm.cpu().numpy(), obm), (c, key)
        assert int(hits.item()) == ohits
    lo, hi = (1 << c) // 4, (1 << c) // 2
    bm, hits = eng.scan_range(lo, hi, col)
    obm, ohits = O.scan_range(packed_host, n, c, lo, hi)
    assert np.array_equal(bm.cpu().numpy(), obm) and int(hits.item()) == ohits
    keys = [int(vals[(31 * k + 5) % n]) for k in range(8)]
    keys = [k if k < 2 ** 31 else k - 2 ** 32 for k in keys]
    out, hits = eng.shared_scan(keys, col)
    oout, ohits = O.shared_scan_eq(packed_host, n, c, keys)
    nb = (n + 7) // 8
    assert np.array_equal(out.cpu().numpy()[:, :nb], oout)
    assert np.array_equal(hits.cpu().numpy().astype(np.uint64), ohits)
    torch.cuda.synchronize()


@pytest.mark.parametrize("n", [1, 7, 8, 9, 63, 64, 65, 4095, 4096, 4097, 8191, 8192, 8193, 16384, 100003])
@pytest.mark.parametrize("c", [9, 12, 21])
def test_ragged_sizes(O, eng, n, c):
    vals, col = make_column(O, eng, n, c, 7 * n + c)
    packed_host = col.data.cpu().numpy()
    nb = (n + 7) // 8
    key = int(vals[n // 2])
    # sentinel-filled outputs: exactly ceil(n/8) bytes / n ints may be written
    import torch

    bm = torch.full((nb + 64,), 0x5A, dtype=torch.uint8, device="cuda")
    _, hits = eng.scan(key, col, bitmap=bm)
    obm, ohits = O.scan_eq(packed_host, n, c, key)
    host = bm.cpu().numpy()
    assert np.array_equal(host[:nb], obm) and (host[nb:] == 0x5A).all()
    assert int(hits.item()) == ohits
    dec = torch.full((n + 64,), -7, dtype=torch.int32, device="cuda")
    eng.decompress(col, out=dec)
    host = dec.cpu().numpy()
    assert np.array_equal(host[:n].view(np.uint32), vals) and (host[n:] == -7).all()
    # key 0 on a ragged tail: pad values decode as 0 in the reference; canonical tail is zero
    bm0, hits0 = eng.scan(0, col)
    obm0, oh0 = O.scan_eq(packed_host, n, c, 0)
    assert np.array_equal(bm0.cpu().numpy(), obm0) and int(hits0.item()) == oh0 == int((vals == 0).sum())


@pytest.mark.parametrize("P", [1, 2, 3, 4, 5, 6, 7, 8, 9, 16, 37, 64, 65, 128, 300, 1024])
@pytest.mark.parametrize("layout", ["per_predicate", "linear"])
def test_shared_scan_predicate_counts(O, eng, P, layout):
    n, c = 2 * 8192 + 77, 9
    vals, col = make_column(O, eng, n, c, 500 + P)
    packed_host = col.data.cpu().numpy()
    keys = [int(v) for v in np.random.default_rng(P).integers(0, 1 << c, size=P)]
    keys[0] = 0  # the reference's own shared-scan bench uses keys 0..P-1 (src/benchmark.cpp:205-209)
    out, hits = eng.shared_scan(keys, col, layout=layout)
    oout, ohits = O.shared_scan_eq(packed_host, n, c, keys, layout)
    nb = (n + 7) // 8
    got = out.cpu().numpy()
    if layout == "per_predicate":
        got = got[:, :nb]
    assert np.array_equal(got, oout)
    assert np.array_equal(hits.cpu().numpy().astype(np.uint64), ohits)


@pytest.mark.parametrize("P", [2, 4, 8, 9, 16, 37, 64, 128, 191, 192, 300])
@pytest.mark.parametrize("layout", ["per_predicate", "linear"])
@pytest.mark.parametrize("c", [9, 13])
def test_shared_scan_without_hit_counts(O, eng, c, P, layout):
    """the reference's shared scans return no counts (src/simd_scan.hpp:102-120): with hits = NULL the launcher may
    pick other kernels (byte-entry tables for linear rows below 192 keys) -- same bitmaps required"""
    n = 3 * 8192 + 2049
    vals, col = make_column(O, eng, n, c, 77 + c + P)
    packed_host = col.data.cpu().numpy()
    keys = [int(v) for v in np.random.default_rng(P + c).integers(0, 1 << c, size=P)]
    keys[0] = int(vals[5])
    out, hits = eng.shared_scan(keys, col, layout=layout, hits=False)
    assert hits is None
    oout, _ = O.shared_scan_eq(packed_host, n, c, keys, layout)
    got = out.cpu().numpy()
    if layout == "per_predicate":
        got = got[:, :(n + 7) // 8]
    assert np.array_equal(got, oout)


@pytest.mark.parametrize("nts", [0, 1, 2])
@pytest.mark.parametrize("c", [7, 9, 16, 21])
def test_store_policy_variants_agree(O, eng, c, nts):
    """the launcher picks plain or non-temporal result stores by output size (option scan_nt_stores: -1 auto);
    both code paths are forced here on the same ragged column: eq, range, shared P = 3 / 8 / 24 in both layouts"""
    n = 5 * 8192 + 4099
    vals, col = make_column(O, eng, n, c, 4242 + c)
    packed_host = col.data.cpu().numpy()
    nb = (n + 7) // 8
    eng.set_option("scan_nt_stores", nts)
    try:
        key = int(vals[99])
        bm, hits = eng.scan(key, col)
        obm, ohits = O.scan_eq(packed_host, n, c, key)
        assert np.array_equal(bm.cpu().numpy(), obm) and int(hits.item()) == ohits
        lo, hi = (1 << c) // 3, (1 << c) // 2
        bm, hits = eng.scan_range(lo, hi, col)
        obm, ohits = O.scan_range(packed_host, n, c, lo, hi)
        assert np.array_equal(bm.cpu().numpy(), obm) and int(hits.item()) == ohits
        for P in (3, 8, 24):
            keys = [int(vals[(131 * k + 7) % n]) for k in range(P)]
            for layout in ("per_predicate", "linear"):
                out, hits = eng.shared_scan(keys, col, layout=layout)
                oout, ohits = O.shared_scan_eq(packed_host, n, c, keys, layout)
                got = out.cpu().numpy()
                if layout == "per_predicate":
                    got = got[:, :nb]
                assert np.array_equal(got, oout), (c, nts, P, layout)
                assert np.array_equal(hits.cpu().numpy().astype(np.uint64), ohits)
    finally:
        eng.set_option("scan_nt_stores", -1)


@pytest.mark.parametrize("c,P", [(5, 40), (10, 520), (12, 200), (17, 129), (25, 72), (32, 16), (32, 600)])
@pytest.mark.parametrize("layout", ["per_predicate", "linear"])
def test_shared_scan_wide_and_many_keys(O, eng, c, P, layout):
    """digit tables (c > 10), tables that do not fit in LDS (c=32, P=600 -> compare-chain kernel), duplicate and
    out-of-range keys"""
    n = 8192 + 4096 + 333
    vals, col = make_column(O, eng, n, c, 900 + c + P)
    packed_host = col.data.cpu().numpy()
    rng = np.random.default_rng(c * 1000 + P)
    keys = [int(vals[int(i)]) for i in rng.integers(0, n, size=P)]
    keys[1] = keys[0]                      # duplicate key
    if c < 31:
        keys[2] = (1 << c) + 5             # out of range: matches nothing
    keys[3] = -1
    keys = [k if k < 2 ** 31 else k - 2 ** 32 for k in keys]
    out, hits = eng.shared_scan(keys, col, layout=layout)
    oout, ohits = O.shared_scan_eq(packed_host, n, c, keys, layout)
    nb = (n + 7) // 8
    got = out.cpu().numpy()
    if layout == "per_predicate":
        got = got[:, :nb]
    assert np.array_equal(got, oout)
    assert np.array_equal(hits.cpu().numpy().astype(np.uint64), ohits)


@pytest.mark.parametrize("c", [5, 9, 10, 11, 16, 17, 20, 21, 25, 29, 30, 32])
@pytest.mark.parametrize("count", [True, False])
def test_shared_scan_round3_kernels(O, eng, c, count):
    """the round-3 shared-scan kernels at the key counts and widths that select their variants: per-predicate one word at a
    time (shared_wide3_kernel: hit counts in registers for one round / two packed rounds / none; byte digits and the wider
    digits of c = 17 .. 20, 25 .. 30), linear rows through the LDS stage (32 .. 40 keys), through the aligned output image
    (41 .. 63 keys, c <= 10), with the short last table in steps of its own (below 32 keys without hit counts) and on the lane
    of the row's last full piece (65 keys and more, where launch_width's attach_short() says so) -- several tiles, a ragged
    tail, duplicate and out-of-range keys"""
    n = 3 * 4096 + 2048 + 77
    vals, col = make_column(O, eng, n, c, 4200 + c)
    packed_host = col.data.cpu().numpy()
    nb = (n + 7) // 8
    for layout, counts in (("per_predicate", (9, 16, 31, 32, 33, 48, 63, 64, 65)), ("linear", (12, 24, 32, 36, 40, 41, 47, 52, 63, 64, 65, 80, 95, 97, 129, 159, 200, 225, 300, 513))):
        for P in counts:
            rng = np.random.default_rng(c * 977 + P)
            keys = [int(vals[int(i)]) for i in rng.integers(0, n, size=P)]
            keys[1] = keys[0]                      # duplicate key
            if c < 31:
                keys[2] = (1 << c) + 5             # out of range: matches nothing
            keys[3] = -1
            keys = [k if k < 2 ** 31 else k - 2 ** 32 for k in keys]
            out, hits = eng.shared_scan(keys, col, layout=layout, hits=None if count else False)
            oout, ohits = O.shared_scan_eq(packed_host, n, c, keys, layout)
            got = out.cpu().numpy()
            if layout == "per_predicate":
                got = got[:, :nb]
            assert np.array_equal(got, oout), (layout, P, c)
            if count:
                assert np.array_equal(hits.cpu().numpy().astype(np.uint64), ohits), (layout, P, c)


@pytest.mark.parametrize("c,P", [(32, 1024), (25, 900), (31, 705)])
@pytest.mark.parametrize("layout", ["per_predicate", "linear"])
@pytest.mark.parametrize("count", [True, False])
def test_shared_scan_compare_chain_fallback(O, eng, L, c, P, layout, count):
    """key counts whose lookup tables do not fit in LDS run on shared_general_kernel (compare chain): full tiles, the
    linear transpose / store path and the ragged tail, with and without hit counts"""
    lay = 1 if layout == "linear" else 0
    assert L.mi355_shared_scan_kernel(eng._ctx, c, P, lay, 1 if count else 0) == b"shared_general_kernel"
    assert L.mi355_shared_scan_kernel(eng._ctx, 9, 64, 0, 1) == b"shared_wide_kernel"
    assert L.mi355_shared_scan_kernel(eng._ctx, 9, 8, 1, 0) == b"shared_lut_kernel"
    n = 2 * 4096 + 77
    vals, col = make_column(O, eng, n, c, 5100 + c + P)
    rng = np.random.default_rng(P + c)
    keys = [int(vals[int(i)]) for i in rng.integers(0, n, size=P)]
    keys[7] = keys[3]
    keys32 = [k if k < 2 ** 31 else k - 2 ** 32 for k in keys]
    out, hits = eng.shared_scan(keys32, col, layout=layout, hits=None if count else False)
    v = vals.astype(np.int64)
    nb = (n + 7) // 8
    per_key = np.stack([np_bitmap(v == k) for k in keys])
    got = out.cpu().numpy()
    if layout == "per_predicate":
        assert np.array_equal(got[:, :nb], per_key)
    else:
        assert np.array_equal(got.reshape(nb, P), per_key.T)
    if count:
        assert np.array_equal(hits.cpu().numpy(), np.array([int((v == k).sum()) for k in keys]))


@pytest.mark.parametrize("P", [3, 8, 40, 64])
def test_shared_scan_accepts_any_16_byte_stride(O, eng, L, P):
    """per-predicate bitmaps may sit at any 16-byte-multiple stride >= ceil(n/8) (mi355_bitmap_stride, whole 128-byte
    lines, is the fast choice the engine's own callers use): results are the same bytes"""
    import torch

    n, c = 8192 * 6 + 999, 9
    vals, col = make_column(O, eng, n, c, 5400 + P)
    keys = np.ascontiguousarray(vals[:P].astype(np.int32))
    nb = (n + 7) // 8
    assert L.mi355_bitmap_stride(n) % 256 == 0 and L.mi355_bitmap_stride(n) >= nb
    oout, ohits = O.shared_scan_eq(col.data.cpu().numpy(), n, c, [int(k) for k in keys])
    for stride in ((nb + 15) // 16 * 16, (nb + 15) // 16 * 16 + 48, L.mi355_bitmap_stride(n)):
        out = torch.full((P * stride + 64,), 0xEE, dtype=torch.uint8, device="cuda")
        hits = torch.zeros(P, dtype=torch.int64, device="cuda")
        ok(L, L.mi355_shared_scan_eq_dev(eng._ctx, col.data.data_ptr(), n, c, vp(keys), P, 0, out.data_ptr(), stride, hits.data_ptr()))
        got = out.cpu().numpy()
        for k in range(P):
            assert np.array_equal(got[k * stride: k * stride + nb], oout[k]), (P, stride, k)
            assert (got[k * stride + nb: (k + 1) * stride] == 0xEE).all()  # nothing written between the bitmaps
        assert np.array_equal(hits.cpu().numpy().astype(np.uint64), ohits)


@pytest.mark.parametrize("c", [1, 4, 8, 9, 12, 16, 17, 21, 24, 32])
@pytest.mark.parametrize("P", [16, 32, 64, 256, 1024])
def test_linear_layout_lanes_in_memory_order(O, eng, L, c, P):
    """shared_linear_kernel (linear rows of 16 .. 1024 keys, a power of two): every width class, columns ending inside
    a row (n % 8 != 0), on a row boundary inside a tile, on a tile boundary and inside the first piece; with hit counts
    (packed byte counters) and without; duplicate and out-of-range keys"""
    # (wide widths with many keys stay on the per-group kernels -- width_group.hip lin_pays -- and tables that do not fit
    # LDS on the compare chain: the same bytes are required of whichever kernel runs)
    rng = np.random.default_rng(c * 100 + P)
    for n in (4096 * 3 + 77, 4096 * 2 + 16, 4096 * 2, 5):
        vals, col = make_column(O, eng, n, c, 6100 + c + P + n)
        v = vals.astype(np.int64)
        keys = [int(vals[int(i)]) for i in rng.integers(0, n, size=P)]
        keys[3] = keys[1]
        if c < 31:
            keys[5] = (1 << c) + 2
        keys32 = [k if k < 2 ** 31 else k - 2 ** 32 for k in keys]
        nb = (n + 7) // 8
        expect = np.stack([np_bitmap((v == k) & (k < (1 << c))) for k in keys]).T.reshape(-1)  # [nb, P] row-major
        for count in (True, False):
            out, hits = eng.shared_scan(keys32, col, layout="linear", hits=None if count else False)
            assert np.array_equal(out.cpu().numpy(), expect), (c, P, n, count)
            if count:
                assert np.array_equal(hits.cpu().numpy(), np.array([int(((v == k) & (k < (1 << c))).sum()) for k in keys]))


@pytest.mark.parametrize("c", [3, 9, 12, 17, 32])
@pytest.mark.parametrize("P", [9, 10, 11, 13, 15, 17, 23, 31, 33, 47, 63, 65, 95, 96, 97, 100, 127, 129, 160, 255, 257, 500, 777, 1023])
def test_linear_layout_any_key_count(O, eng, L, c, P):
    """shared_linear_kernel with key counts that are not powers of two (the reference's own sweep is P = 1 .. 512 in
    steps of one, scripts/prepare_shared_scan_results.py:28-31): rows of P bytes start at any byte, the last piece of a row
    is 1 .. 31 bytes, lanes of tables that do not exist idle.  Whole output compared with numpy, guard bytes behind the
    output untouched, columns ending inside a row / on a row boundary / on a tile boundary / inside the first piece."""
    import torch

    # (wide widths with many keys stay on the per-group kernels -- width_group.hip lin_pays -- and tables that do not fit
    # LDS on the compare chain: the same bytes are required of whichever kernel runs)
    rng = np.random.default_rng(c * 10000 + P)
    for n in (4096 * 3 + 77, 4096 * 2 + 16, 4096 * 2, 5):
        vals, col = make_column(O, eng, n, c, 7100 + c + P + n)
        v = vals.astype(np.int64)
        keys = [int(vals[int(i)]) for i in rng.integers(0, n, size=P)]
        keys[3] = keys[1]
        if c < 31:
            keys[5] = (1 << c) + 2
        keys[P - 1] = int(vals[0])  # the last key of the short piece matches something
        keys32 = [k if k < 2 ** 31 else k - 2 ** 32 for k in keys]
        nb = (n + 7) // 8
        expect = np.stack([np_bitmap((v == k) & (k < (1 << c))) for k in keys]).T.reshape(-1)  # [nb, P] row-major
        for count in (True, False):
            buf = torch.full((nb * P + 256,), 0xEE, dtype=torch.uint8, device="cuda")
            out, hits = eng.shared_scan(keys32, col, layout="linear", out=buf[:nb * P], hits=None if count else False)
            got = buf.cpu().numpy()
            assert np.array_equal(got[:nb * P], expect), (c, P, n, count)
            assert (got[nb * P:] == 0xEE).all(), (c, P, n, count)
            if count:
                assert np.array_equal(hits.cpu().numpy(), np.array([int(((v == k) & (k < (1 << c))).sum()) for k in keys]))


@pytest.mark.parametrize("c", [2, 7, 9, 12, 17, 32])
@pytest.mark.parametrize("P", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("vpl", [0, 128])
def test_linear_layout_up_to_8_keys(O, eng, c, P, vpl):
    """linear rows of 1 .. 8 bytes (one-pass LUT kernel; P = 3, 5, 6, 7 pack the lane's rows back to back in registers):
    whole output against numpy, guard bytes behind it untouched, ragged columns, both tile geometries where they exist"""
    import torch

    if vpl == 128 and c > 12:
        pytest.skip("128 values per lane exist for c <= 12")
    rng = np.random.default_rng(c * 100 + P)
    try:
        eng.set_option("shared_vpl", vpl)
        for n in (8192 * 5 + 77, 8192 * 2 + 16, 8192 * 2, 5):
            vals, col = make_column(O, eng, n, c, 8100 + c + P + n)
            v = vals.astype(np.int64)
            keys = [int(vals[int(i)]) for i in rng.integers(0, n, size=P)]
            if P > 2 and c < 31:
                keys[1] = (1 << c) + 2  # out of range: a zero byte in every row
            nb = (n + 7) // 8
            expect = np.stack([np_bitmap((v == k) & (k < (1 << c))) for k in keys]).T.reshape(-1)  # [nb, P] row-major
            for count in (True, False):
                buf = torch.full((nb * P + 256,), 0xEE, dtype=torch.uint8, device="cuda")
                out, hits = eng.shared_scan(keys, col, layout="linear", out=buf[:nb * P], hits=None if count else False)
                got = buf.cpu().numpy()
                assert np.array_equal(got[:nb * P], expect), (c, P, n, count, vpl)
                assert (got[nb * P:] == 0xEE).all(), (c, P, n, count, vpl)
                if count:
                    assert np.array_equal(hits.cpu().numpy(), np.array([int(((v == k) & (k < (1 << c))).sum()) for k in keys]))
    finally:
        eng.set_option("shared_vpl", 0)


@pytest.mark.parametrize("c", [1, 5, 9, 12, 17, 31, 32])
def test_gather_values_at_row_ids(O, eng, c):
    """mi355_gather_dev ("take"): values at arbitrary row ids (unsorted, duplicates, the last row, ids outside the column ->
    -1), the count read on the device and capped by the capacity; then the pipeline it is for: predicate on one column ->
    ascending row ids in one launch -> the values of ANOTHER column at those rows, nothing passing through the host"""
    import torch

    n = 8192 * 7 + 4099
    vals, col = make_column(O, eng, n, c, 9900 + c)
    rng = np.random.default_rng(c)
    first = 5_000_000_000  # global row ids (a shard far into a table)
    local = rng.integers(0, n, size=20_000)
    local[:4] = [n - 1, 0, n - 1, 12345 % n]
    ids = torch.from_numpy((local + first).astype(np.int64)).cuda()
    ids[7] = first - 1      # below the shard
    ids[8] = first + n      # behind it
    got = eng.gather(col, ids, ids.numel(), first_row=first).cpu().numpy()
    want = vals[local].astype(np.int64).astype(np.uint32).view(np.int32).copy()
    want[7] = want[8] = -1
    assert np.array_equal(got, want)
    # count on the device smaller than the capacity: the rest of `out` stays untouched
    out = torch.full((ids.numel(),), -7, dtype=torch.int32, device="cuda")
    eng.gather(col, ids, torch.tensor([100], dtype=torch.int64, device="cuda"), first_row=first, out=out)
    assert np.array_equal(out[:100].cpu().numpy(), want[:100]) and bool((out[100:] == -7).all().item())
    # select on one column, take from another
    vals2, col2 = make_column(O, eng, n, 12, 9950 + c)
    key = int(vals[77])
    expect_rows = np.nonzero(vals == key)[0]
    rowids, cnt = eng.scan_select("==", key, col, capacity=n, first_row=first)
    taken = eng.gather(col2, rowids, cnt, first_row=first)
    k = int(cnt.item())
    assert k == expect_rows.shape[0]
    assert np.array_equal(taken[:k].cpu().numpy(), vals2[expect_rows].astype(np.int32))


@pytest.mark.parametrize("c", [1, 2, 5, 7, 9, 12, 16, 17, 24, 25, 31, 32])
def test_aggregate_under_a_bitmap(O, eng, c):
    """mi355_aggregate_dev: sum / count / min / max of a column over the rows of a bitmap (or all rows) in one pass --
    against numpy on the oracle's values; ragged columns, an empty selection, the widths on both sides of the 32 / 64-bit
    partial sums (c = 24 / 25), 32-bit values"""
    import torch

    rng = np.random.default_rng(4400 + c)
    for n in (8192 * 6 + 4099, 8192 * 2, 777, 1):
        vals, col = make_column(O, eng, n, c, 8800 + c + n)
        v = vals.astype(np.uint64)
        got = eng.aggregate(col).cpu().numpy().view(np.uint64)
        assert got.tolist() == [int(v.sum()), n, int(v.min()), int(v.max())], (c, n)
        # a mask from a predicate on another column
        vals2, col2 = make_column(O, eng, n, 9, 8900 + c + n)
        a = int(rng.integers(0, 512))
        mask, hits = eng.scan_where("<", a, col2)
        sel = vals2 < a
        got = eng.aggregate(col, mask=mask).cpu().numpy().view(np.uint64)
        if sel.any():
            assert got.tolist() == [int(v[sel].sum()), int(sel.sum()), int(v[sel].min()), int(v[sel].max())], (c, n, a)
        else:
            assert got.tolist() == [0, 0, 2 ** 64 - 1, 0], (c, n, a)
        empty = torch.zeros_like(mask)
        assert eng.aggregate(col, mask=empty).cpu().numpy().view(np.uint64).tolist() == [0, 0, 2 ** 64 - 1, 0]
    # the whole pipeline on the device: predicate on one column -> sum of another, 2e7 rows (sums beyond 32 bits)
    n = 20_000_000 + 77
    col_a = eng.generate("mod", n, 9, 7)
    col_b = eng.generate("index", n, c)
    mask, hits = eng.scan(3, col_a)
    rows = np.arange(3, n, 7, dtype=np.uint64)
    vb = rows & np.uint64((1 << c) - 1)
    got = eng.aggregate(col_b, mask=mask).cpu().numpy().view(np.uint64)
    assert got.tolist() == [int(vb.sum()), rows.shape[0], int(vb.min()), int(vb.max())], c


@pytest.mark.parametrize("c", [1, 3, 7, 9, 12, 13, 14])
def test_histogram_under_a_bitmap(O, eng, c):
    """mi355_histogram_dev: per-value row counts (GROUP BY over a dictionary-coded column), all rows or the rows of a bitmap,
    against numpy.bincount; ragged columns, skewed data (every lane adding to the same few counters), widths above 14 refused"""
    import torch

    for n in (8192 * 9 + 4099, 8192 * 4, 77):
        vals, col = make_column(O, eng, n, c, 6600 + c + n)
        got = eng.histogram(col).cpu().numpy()
        assert np.array_equal(got, np.bincount(vals, minlength=1 << c)), (c, n)
        vals2, col2 = make_column(O, eng, n, 9, 6700 + c + n)
        mask, _ = eng.scan_where(">=", 300, col2)
        got = eng.histogram(col, mask=mask).cpu().numpy()
        assert np.array_equal(got, np.bincount(vals[vals2 >= 300], minlength=1 << c)), (c, n)
    n = 30_000_000 + 5
    col = eng.generate("mod", n, c, 3 if c > 1 else 2)  # three hot counters
    m = 3 if c > 1 else 2
    want = np.zeros(1 << c, dtype=np.int64)
    for k in range(m):
        want[k] = (n - 1 - k) // m + 1
    assert np.array_equal(eng.histogram(col).cpu().numpy(), want)
    with pytest.raises(Exception):
        eng.histogram(eng.generate("mod", 1000, 15, 5), out=torch.empty(1 << 15, dtype=torch.int64, device="cuda"))


def test_out_of_range_keys_never_match(O, eng):
    """SURVEY 8c hazard 5: keys 515, 1027, 65539, -1 on a 9-bit column -> no hits, zero bitmap."""
    import torch

    vals = (np.arange(4096, dtype=np.uint32) * 7) % 512
    col = eng.compress(torch.from_numpy(vals.view(np.int32)).cuda(), 9)
    for key in (515, 1027, 65539, -1):
        bm, hits = eng.scan(key, col)
        assert int(hits.item()) == 0 and not bm.any().item()
    bm, hits = eng.scan(511, col)
    assert int(hits.item()) == int((vals == 511).sum())
    out, hits = eng.shared_scan([515, 511, -1], col)
    assert hits.cpu().tolist() == [0, int((vals == 511).sum()), 0]


def test_generators_match_oracle(O, eng):
    for kind, c, param in (("mod", 9, 5), ("mod", 9, 8), ("splitmix", 9, 42), ("splitmix", 21, 42), ("index", 9, 0),
                           ("splitmix", 12, 7)):
        n = 50021
        col = eng.generate(kind, n, c, param, first_row=123456789)
        vals = O.gen_values(kind, n, c, param, first=123456789)
        assert np.array_equal(col.data.cpu().numpy(), O.pack(vals, c)), (kind, c)


# ------------------------------------------------------------------------------------------------
# BASELINE sizes: size-independent properties + oracle on sampled windows
# ------------------------------------------------------------------------------------------------

def test_cfg1_1e7_matches_oracle_exactly(O, eng):
    """BASELINE config 1: n=1e7, c=9, v=i%5, key 3 (src/benchmark.cpp:173,:150) -> hits 2,000,000."""
    n, c = 10_000_000, 9
    col = eng.generate("mod", n, c, 5)
    bm, hits = eng.scan(3, col)
    assert int(hits.item()) == 2_000_000
    packed_host = col.data.cpu().numpy()
    obm, ohits = O.scan_eq(packed_host, n, c, 3)
    assert ohits == 2_000_000 and np.array_equal(bm.cpu().numpy(), obm)
    dec = eng.decompress(col)
    assert np.array_equal(dec.cpu().numpy(), O.decompress(packed_host, n, c))


def test_cfg2_1e9_properties(O, eng):
    """BASELINE config 2: n=1e9 x 9 bit, equality.  Checks: hit count (i%5 -> exactly n/5), bitmap
    periodicity (period lcm(5,8)=40 bits = 5 bytes), popcount(bitmap) == hits, and oracle equality on
    windows sampled at tile-aligned offsets (start, middle, ragged end)."""
    import torch

    n, c = 1_000_000_000, 9
    col = eng.generate("mod", n, c, 5)
    bm, hits = eng.scan(3, col)
    assert int(hits.item()) == 200_000_000
    nb = n // 8
    pat = torch.tensor(np.packbits((np.arange(40) % 5 == 3).astype(np.uint8), bitorder="little"), device="cuda")
    assert torch.equal(bm[:nb].view(-1, 5), pat.expand(nb // 5, 5))
    # oracle on windows: rows [a, a+len) with a a multiple of 8192 (byte-aligned in both streams)
    for a, ln in ((0, 300_000), (8192 * 61_000, 250_000), (n - 123_456 - (n - 123_456) % 8192, None)):
        ln = n - a if ln is None else ln
        pk = col.data[a * c // 8: a * c // 8 + (ln * c + 7) // 8].cpu().numpy()
        obm, _ = O.scan_eq(pk, ln, c, 3)
        assert np.array_equal(bm[a // 8: a // 8 + (ln + 7) // 8].cpu().numpy(), obm)
    # random column, key = v[12345]: popcount of the bitmap == hits, windows == oracle
    col = eng.generate("splitmix", n, c, 42)
    key = int(O.gen_values("splitmix", 1, c, 42, first=12345)[0])
    bm, hits = eng.scan(key, col)
    h = int(hits.item())
    assert abs(h - n / 512) < 6 * (n / 512) ** 0.5
    lut = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int64, device="cuda")
    assert int(lut[bm.long()].sum().item()) == h
    for a, ln in ((0, 200_000), (8192 * 100_003, 200_000)):
        pk = col.data[a * c // 8: a * c // 8 + (ln * c + 7) // 8].cpu().numpy()
        assert np.array_equal(pk, O.pack(O.gen_values("splitmix", ln, c, 42, first=a), c)[: pk.shape[0]])
        obm, _ = O.scan_eq(pk, ln, c, key)
        assert np.array_equal(bm[a // 8: a // 8 + (ln + 7) // 8].cpu().numpy(), obm)


@pytest.mark.parametrize("c", [3, 9])
def test_more_than_2_pow_32_rows(O, eng, c):
    """row indices, tile indices and byte offsets are 64-bit all the way: 2^32 + 3*8192 + 77 rows (a ragged tail tile),
    generator v[i] = i % 7.  Hit counts exact; oracle windows at the start, across row 2^32 and over the ragged end;
    decompress and the selection vector (row ids > 2^32) checked on the last rows."""
    import torch

    n = (1 << 32) + 3 * 8192 + 77
    col = eng.generate("mod", n, c, 7)
    key = 5
    bm, hits = eng.scan(key, col)
    assert int(hits.item()) == (n - 1 - key) // 7 + 1
    lo, hi = 2, 4
    bmr, hitsr = eng.scan_range(lo, hi, col)
    assert int(hitsr.item()) == sum((n - 1 - k) // 7 + 1 for k in range(lo, hi + 1))
    assert int(eng.bitmap_count(bm, n).item()) == int(hits.item())
    a0 = (1 << 32) - 8192 * 2
    for a, ln in ((0, 100_000), (a0, 8192 * 4 + 96), (n - 77 - 8192, None)):
        ln = n - a if ln is None else ln
        pk = col.data[a * c // 8: a * c // 8 + (ln * c + 7) // 8].cpu().numpy()
        vals = ((np.arange(a, a + ln, dtype=np.uint64)) % 7).astype(np.uint32)
        assert np.array_equal(pk, O.pack(vals, c)[: pk.shape[0]])
        obm, _ = O.scan_eq(pk, ln, c, key)
        assert np.array_equal(bm[a // 8: a // 8 + (ln + 7) // 8].cpu().numpy(), obm)
        obm, _ = O.scan_range(pk, ln, c, lo, hi)
        assert np.array_equal(bmr[a // 8: a // 8 + (ln + 7) // 8].cpu().numpy(), obm)
    del bmr
    # selection vector of the tail: rows >= 2^32 with v == key
    tail_rows = n - (1 << 32)
    sub = bm[(1 << 32) // 8:].clone()
    ids, cnt = eng.bitmap_to_rowids(sub, tail_rows, capacity=tail_rows, first_row=1 << 32)
    k = int(cnt.item())
    expect = np.arange(1 << 32, n, dtype=np.uint64)
    expect = expect[expect % 7 == key]
    assert k == expect.shape[0] and np.array_equal(ids[:k].cpu().numpy().astype(np.uint64), expect)
    del bm, sub, col
    torch.cuda.empty_cache()


def test_more_than_2_pow_32_rows_round2_kernels(O, eng):
    """the kernels added after the first >2^32-row test, on the same column shape (2^32 + 3*8192 + 77 rows of v[i] = i % 7,
    9 bit): count-only and fused-mask scans, two columns in one launch, IN-list, the fused selection (row ids > 2^32),
    shared scans of 8 / 12 / 16 / 40 keys in both layouts (LUT, register-count, lanes-in-memory-order kernels) and
    decompress -- exact hit counts from the generator's period, oracle windows across row 2^32 and over the ragged end"""
    import torch

    c = 9
    n = (1 << 32) + 3 * 8192 + 77
    col = eng.generate("mod", n, c, 7)
    per_key = [(n - 1 - k) // 7 + 1 for k in range(7)]
    nb = (n + 7) // 8
    windows = [((1 << 32) - 8192 * 2, 8192 * 4 + 96), (n - 77 - 8192, 77 + 8192)]

    def window_packed(a, ln):
        return col.data[a * c // 8: a * c // 8 + (ln * c + 7) // 8].cpu().numpy()

    # count-only, fused masks, two columns, IN-list
    bm5, h5 = eng.scan(5, col)
    assert int(h5.item()) == per_key[5]
    _, hc = eng.scan_combine("==", 5, col, count_only=True)
    assert int(hc.item()) == per_key[5]
    bor, hor = eng.scan_combine("==", 2, col, mask=bm5, mask_op="or")
    assert int(hor.item()) == per_key[5] + per_key[2]
    bin_, hin = eng.scan_in([2, 5, 300], col)
    assert int(hin.item()) == per_key[5] + per_key[2] and torch.equal(bin_, bor)
    b2, h2 = eng.scan2(col, ">=", 2, col, "<=", 4, combine="and")
    assert int(h2.item()) == per_key[2] + per_key[3] + per_key[4]
    for a, ln in windows:
        pk = window_packed(a, ln)
        obm, _ = O.scan_range(pk, ln, c, 2, 4)
        assert np.array_equal(b2[a // 8: a // 8 + (ln + 7) // 8].cpu().numpy(), obm)
        o2, _ = O.scan_eq(pk, ln, c, 2)
        o5, _ = O.scan_eq(pk, ln, c, 5)
        assert np.array_equal(bor[a // 8: a // 8 + (ln + 7) // 8].cpu().numpy(), o2 | o5)
    del bor, bin_, b2

    # fused selection: every id, the tail against numpy (row ids above 2^32)
    ids, cnt = eng.scan_select("==", 5, col, capacity=per_key[5])
    assert int(cnt.item()) == per_key[5]
    step = 1 << 28
    for j0 in range(0, per_key[5], step):
        j1 = min(per_key[5], j0 + step)
        assert torch.equal(ids[j0:j1], torch.arange(j0, j1, dtype=torch.int64, device="cuda") * 7 + 5), j0
    assert int(ids[per_key[5] - 1].item()) > (1 << 32)
    del ids, bm5
    torch.cuda.empty_cache()

    # shared scans: per-key counts from the period, windows against the oracle
    for P in (8, 12, 16, 40):
        keys = [(3 * k + 1) % 11 for k in range(P)]  # values 0..10: 7..10 match nothing, duplicates from k = 11 on
        want = [per_key[k] if k < 7 else 0 for k in keys]
        for layout in ("per_predicate", "linear"):
            out, hits = eng.shared_scan(keys, col, layout=layout)
            assert hits.cpu().tolist() == want, (P, layout)
            for a, ln in windows:
                oout, _ = O.shared_scan_eq(window_packed(a, ln), ln, c, keys, layout)
                wb = (ln + 7) // 8
                if layout == "per_predicate":
                    assert np.array_equal(out[:, a // 8: a // 8 + wb].cpu().numpy(), oout), (P, layout, a)
                else:
                    assert np.array_equal(out[(a // 8) * P: (a // 8 + wb) * P].cpu().numpy(), oout.reshape(-1)), (P, layout, a)
            assert nb * P <= out.numel()
            del out
            torch.cuda.empty_cache()

    # decompress: all n values against the generator, on the device in slices
    dec = eng.decompress(col)
    step = 1 << 29
    for i0 in range(0, n, step):
        i1 = min(n, i0 + step)
        assert torch.equal(dec[i0:i1], (torch.arange(i0, i1, dtype=torch.int64, device="cuda") % 7).to(torch.int32)), i0
    del dec, col
    torch.cuda.empty_cache()


@pytest.mark.parametrize("c,n", [(5, 100_000_000), (9, 100_000_000), (17, 100_000_000),
                                 (5, 1_000_000_000), (7, 1_000_000_000), (9, 1_000_000_000), (12, 1_000_000_000),
                                 (17, 1_000_000_000), (21, 1_000_000_000)])
def test_cfg3_width_sweep_range_properties(O, eng, c, n):
    """BASELINE config 3: bit-width sweep {5,7,9,12,17,21}, inclusive range [2^c/4, 2^c/2] on the random column at the
    full 1e9 rows for every width (plus three 1e8-row cases).  decompress -> compare on the device gives an
    independent full-size check of all n/8 bitmap bytes."""
    import torch

    col = eng.generate("splitmix", n, c, 42)
    lo, hi = (1 << c) // 4, (1 << c) // 2
    bm, hits = eng.scan_range(lo, hi, col)
    dec = eng.decompress(col)
    expect = ((dec >= lo) & (dec <= hi))
    assert int(hits.item()) == int(expect.sum().item())
    # pack the boolean vector little-endian on the device and compare all n/8 bytes
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device="cuda")
    packed_expect = (expect.view(-1, 8).to(torch.int32) * w).sum(dim=1).to(torch.uint8)
    assert torch.equal(bm, packed_expect)
    a, ln = 8192 * 5000, 150_000
    pk = col.data[a * c // 8: a * c // 8 + (ln * c + 7) // 8].cpu().numpy()
    obm, _ = O.scan_range(pk, ln, c, lo, hi)
    assert np.array_equal(bm[a // 8: a // 8 + (ln + 7) // 8].cpu().numpy(), obm)
    assert np.array_equal(dec[a: a + ln].cpu().numpy(), O.decompress(pk, ln, c))


def test_cfg4_shared_8_predicates_1e9_properties(O, eng):
    """BASELINE config 4: P=8 over 1e9 x 9 bit, v=i%8, keys 0..7 (src/benchmark.cpp:277,:205-209).
    Properties: each row matches exactly one key -> the 8 bitmaps partition the rows (OR = all ones,
    pairwise AND = 0), every hit count = n/8, bitmap k = 0x01<<k repeated; oracle on a window."""
    import torch

    n, c, P = 1_000_000_000, 9, 8
    col = eng.generate("mod", n, c, 8)
    out, hits = eng.shared_scan(list(range(P)), col)
    assert hits.cpu().tolist() == [n // 8] * P
    nb = n // 8
    for k in range(P):
        assert bool((out[k, :nb] == (1 << k)).all().item())
    lin, hits2 = eng.shared_scan(list(range(P)), col, layout="linear")
    assert hits2.cpu().tolist() == [n // 8] * P
    expect = torch.tensor([1 << k for k in range(P)], dtype=torch.uint8, device="cuda")
    assert torch.equal(lin.view(-1, P), expect.expand(nb, P))
    a, ln = 8192 * 77, 100_000
    pk = col.data[a * c // 8: a * c // 8 + (ln * c + 7) // 8].cpu().numpy()
    oout, _ = O.shared_scan_eq(pk, ln, c, list(range(P)))
    assert np.array_equal(out[:, a // 8: a // 8 + (ln + 7) // 8].cpu().numpy(), oout)


def test_cfg4_random_column_1e9_both_layouts(O, eng):
    """BASELINE config 4, second half: P=8 over the 1e9 x 9 bit RANDOM column, keys v[31k+5], both output
    layouts.  Full-size, size-independent checks: bitmap k == the single-predicate scan of key k (a different kernel)
    over all n/8 bytes, popcount == hits, linear == byte-transposition of the per-predicate bitmaps; oracle on windows
    (start, middle, the ragged last tile)."""
    import torch

    n, c, P = 1_000_000_000, 9, 8
    col = eng.generate("splitmix", n, c, 42)
    keys = [int(O.gen_values("splitmix", 1, c, 42, first=31 * k + 5)[0]) for k in range(P)]
    nb = n // 8
    out, hits = eng.shared_scan(keys, col)
    h = hits.cpu().tolist()
    for k in range(P):
        # duplicates among the keys are legal: every predicate is evaluated on its own
        bm1, h1 = eng.scan(keys[k], col)
        assert int(h1.item()) == h[k]
        assert torch.equal(out[k, :nb], bm1[:nb])
        assert int(eng.bitmap_count(out[k], n).item()) == h[k]
        assert abs(h[k] - n / 512) < 6 * (n / 512) ** 0.5
    del bm1
    lin, hits2 = eng.shared_scan(keys, col, layout="linear")
    assert hits2.cpu().tolist() == h
    assert torch.equal(lin.view(nb, P).t(), out[:, :nb])
    for a, ln in ((0, 150_000), (8192 * 60_007, 150_000), (n - 2560 - 8192 * 3, None)):
        ln = n - a if ln is None else ln
        pk = col.data[a * c // 8: a * c // 8 + (ln * c + 7) // 8].cpu().numpy()
        assert np.array_equal(pk, O.pack(O.gen_values("splitmix", ln, c, 42, first=a), c)[: pk.shape[0]])
        oout, ohits = O.shared_scan_eq(pk, ln, c, keys)
        assert np.array_equal(out[:, a // 8: a // 8 + (ln + 7) // 8].cpu().numpy(), oout)
        olin, _ = O.shared_scan_eq(pk, ln, c, keys, layout="linear")
        assert np.array_equal(lin[a // 8 * P: (a // 8 + (ln + 7) // 8) * P].cpu().numpy(), olin)


def test_cfg5_shard_1e9x12_at_first_row_7e9(O, eng):
    """BASELINE config 5, the per-GPU workload of rank 7: rows [7e9, 8e9) of the 8e9 x 12 bit random column, equality
    on key v[12345] of the GLOBAL column.  Full-size: popcount == hits, bitmap == decompress -> compare on
    the device over all n/8 bytes; oracle windows at the start, in the middle and over the ragged last tile (1e9 is not
    a multiple of the 8192-row tile), each with packed bytes == the oracle's generator at that global row."""
    import torch

    n, c, first = 1_000_000_000, 12, 7_000_000_000
    col = eng.generate("splitmix", n, c, 42, first_row=first)
    key = int(O.gen_values("splitmix", 1, c, 42, first=12345)[0])
    bm, hits = eng.scan(key, col)
    h = int(hits.item())
    assert abs(h - n / 4096) < 6 * (n / 4096) ** 0.5
    assert int(eng.bitmap_count(bm, n).item()) == h
    dec = eng.decompress(col)
    expect = dec == key
    assert int(expect.sum().item()) == h
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device="cuda")
    assert torch.equal(bm, (expect.view(-1, 8).to(torch.int32) * w).sum(dim=1).to(torch.uint8))
    del expect
    for a, ln in ((0, 200_000), (8192 * 50_021, 200_000), (n - 2560 - 8192 * 2, None)):
        ln = n - a if ln is None else ln
        pk = col.data[a * c // 8: a * c // 8 + (ln * c + 7) // 8].cpu().numpy()
        assert np.array_equal(pk, O.pack(O.gen_values("splitmix", ln, c, 42, first=first + a), c)[: pk.shape[0]])
        obm, _ = O.scan_eq(pk, ln, c, key)
        assert np.array_equal(bm[a // 8: a // 8 + (ln + 7) // 8].cpu().numpy(), obm)
        assert np.array_equal(dec[a: a + ln].cpu().numpy(), O.decompress(pk, ln, c))
    del dec, bm, col
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------
# beyond the reference: comparisons, conjunctions, bitmap consumers (checked against numpy on the values)
# ------------------------------------------------------------------------------------------------

def np_bitmap(mask_bool):
    return np.packbits(mask_bool.astype(np.uint8), bitorder="little")


@pytest.mark.parametrize("c", [1, 5, 9, 12, 21, 32])
@pytest.mark.parametrize("n", [8192 * 2 + 77, 1000])
def test_scan_where_all_comparisons(O, eng, c, n):
    import torch

    vals, col = make_column(O, eng, n, c, 4000 + c + n)
    v = vals.astype(np.int64)
    vmax = (1 << c) - 1
    a = int(vals[3])
    b = min(vmax, a + max(1, vmax // 3))
    cases = [("==", a, 0, v == a), ("!=", a, 0, v != a), ("<", a, 0, v < a), ("<=", a, 0, v <= a), (">", a, 0, v > a),
             (">=", a, 0, v >= a), ("between", a, b, (v >= a) & (v <= b)), ("not_between", a, b, (v < a) | (v > b)),
             ("<", 0, 0, v < 0), (">", vmax, 0, v > vmax), ("==", vmax + 7, 0, v == vmax + 7), ("!=", -3, 0, v != -3),
             (">=", 0, 0, v >= 0), ("between", 5, 2, np.zeros(n, bool)), ("not_between", 5, 2, np.ones(n, bool))]
    for op, x, y, expect in cases:
        bm, hits = eng.scan_where(op, x, col, b=y)
        assert np.array_equal(bm.cpu().numpy(), np_bitmap(expect)), (op, x, y, c)
        assert int(hits.item()) == int(expect.sum())
    # conjunction fused into the scan: (v >= a) AND previous bitmap
    prev, _ = eng.scan_where("<=", b, col)
    bm, hits = eng.scan_where(">=", a, col, and_mask=prev)
    expect = (v >= a) & (v <= b)
    assert np.array_equal(bm.cpu().numpy(), np_bitmap(expect)) and int(hits.item()) == int(expect.sum())
    torch.cuda.synchronize()


@pytest.mark.parametrize("c", [1, 3, 5, 7, 9, 12, 16, 17, 21, 32])
@pytest.mark.parametrize("n", [8192 * 9 + 77, 4096 * 41 + 5, 1000, 8192 * 64])
def test_fused_mask_ops_and_count_only(O, eng, c, n):
    """mi355_scan_combine_dev: the earlier bitmap combined inside the scan by AND / OR / XOR / ANDNOT, and the count-only
    form (no bitmap stored), against numpy on the decoded values; every width class (table decode, 128 and 64 values
    per lane), sizes with several chunks per wave and a ragged last tile.  Repeated, so that an ordering bug between the
    mask's LDS-DMA and its use would show (it did, once: c <= 3, 29 launches of 30)."""
    vals, col = make_column(O, eng, n, c, 9100 + c + n)
    v = vals.astype(np.int64)
    vmax = (1 << c) - 1
    a = int(vals[5])
    first = v <= (vmax // 2)
    mask, _ = eng.scan_where("<=", vmax // 2, col)
    p = v >= a
    for rep in range(4):
        for mop, expect in (("and", p & first), ("or", p | first), ("xor", p ^ first), ("andnot", first & ~p)):
            bm, hits = eng.scan_combine(">=", a, col, mask=mask, mask_op=mop)
            assert np.array_equal(bm.cpu().numpy(), np_bitmap(expect)), (mop, c, n, rep)
            assert int(hits.item()) == int(expect.sum())
            none, hits = eng.scan_combine(">=", a, col, mask=mask, mask_op=mop, count_only=True)
            assert none is None and int(hits.item()) == int(expect.sum()), (mop, c, n, "count only")
        none, hits = eng.scan_combine("!=", a, col, count_only=True)
        assert int(hits.item()) == int((v != a).sum())
        none, hits = eng.scan_combine("between", a, col, b=vmax, count_only=True)
        assert int(hits.item()) == int(p.sum())


@pytest.mark.parametrize("c", [5, 6, 9, 12, 16])
def test_one_tile_per_burst_option_agrees(O, eng, c):
    """widths whose default is 4 tiles per store burst (burst_k): the K = 1 build of the same kernel gives the same bytes"""
    n = 8192 * 37 + 123
    vals, col = make_column(O, eng, n, c, 9300 + c)
    key = int(vals[11])
    try:
        eng.set_option("scan_burst", 0)
        bm4, h4 = eng.scan(key, col)
        r4, hr4 = eng.scan_range(1, (1 << c) // 2, col)
        eng.set_option("scan_burst", 1)
        bm1, h1 = eng.scan(key, col)
        r1, hr1 = eng.scan_range(1, (1 << c) // 2, col)
    finally:
        eng.set_option("scan_burst", 0)
    obm, ohits = O.scan_eq(col.data.cpu().numpy(), n, c, key)
    assert np.array_equal(bm4.cpu().numpy(), obm) and np.array_equal(bm1.cpu().numpy(), obm)
    assert int(h4.item()) == ohits == int(h1.item())
    orb, orh = O.scan_range(col.data.cpu().numpy(), n, c, 1, (1 << c) // 2)
    assert np.array_equal(r4.cpu().numpy(), orb) and np.array_equal(r1.cpu().numpy(), orb) and int(hr4.item()) == orh == int(hr1.item())


@pytest.mark.parametrize("c", [1, 9, 21, 32])
def test_load_time_tuning_never_changes_results(O, c):
    """mi355_tune_dev measures 1 / 2 / 4 blocks per CU on a resident column of >= 5e7 rows and keeps the fastest per
    kind; every kind gives the bytes it gave before tuning, and a small column is left alone"""
    import torch

    from shared_simd_scan_amd import ScanEngine

    eng = ScanEngine(0)  # own context: what it learns must not leak into the other tests' engine
    n = 50_000_000 + 12_345
    col = eng.generate("splitmix", n, c, 600 + c)
    key, lo, hi = 1, 0, max(1, ((1 << c) - 1) // 2)
    before = (eng.scan(key, col), eng.scan_range(lo, hi, col), eng.scan_combine("<=", hi, col, count_only=True)[1])
    mask = before[1][0]
    before += (eng.scan_combine("==", key, col, mask=mask, mask_op="or"), eng.decompress(col))
    assert eng.tuned(c) == {}
    small = eng.generate("splitmix", 1_000_000, c, 1)
    assert eng.tune(small) == {}
    kept = eng.tune(col)
    assert set(kept) == {"scan_eq", "scan_range", "count", "mask", "decompress"} and all(v in (1, 2, 4) for v in kept.values()), kept
    after = (eng.scan(key, col), eng.scan_range(lo, hi, col), eng.scan_combine("<=", hi, col, count_only=True)[1],
             eng.scan_combine("==", key, col, mask=mask, mask_op="or"), eng.decompress(col))
    for b, a in zip(before, after):
        if isinstance(b, tuple):
            assert torch.equal(b[0], a[0]) and int(b[1].item()) == int(a[1].item())
        else:
            assert torch.equal(b, a)
    obm, ohits = O.scan_eq(col.data.cpu().numpy(), n, c, key)
    assert np.array_equal(after[0][0].cpu().numpy(), obm) and int(after[0][1].item()) == ohits
    with pytest.raises(Exception):
        lib_tune_bad(eng, col)


def lib_tune_bad(eng, col):
    from shared_simd_scan_amd._capi import check, lib

    check(lib().mi355_tune_dev(eng._ctx, col.data.data_ptr(), col.n, col.c, 0))  # what = 0: invalid


@pytest.mark.parametrize("c", [1, 4, 5, 7, 9, 13, 14, 16, 17, 21, 32])
@pytest.mark.parametrize("n", [1, 77, 8192, 8192 * 16 + 5, 8192 * 16 * 5 + 4097, 1_000_003])
def test_scan_select_matches_numpy(O, eng, c, n):
    """mi355_scan_select_dev: predicate -> ascending row ids in one launch (no bitmap in HBM), against numpy: every
    width class, one tile / one chunk / several chunks with a ragged tail, dense and sparse predicates, a fused mask,
    a row offset, and a capacity smaller than the result"""
    vals, col = make_column(O, eng, n, c, 9500 + c + n)
    v = vals.astype(np.int64)
    vmax = (1 << c) - 1
    a = int(vals[n // 2])
    # both kernels behind the entry point whatever the predicate's expected selectivity would pick: option select_kernel
    # 1 = single-role (select_kernel), 2 = decoder / expander roles (select2_kernel); 0 = the engine's rule again at the end
    for forced in (1, 2):
        eng.set_option("select_kernel", forced)
        for op, x, y, expect in (("==", a, 0, v == a), ("<=", vmax // 2, 0, v <= vmax // 2), ("<", vmax // 8 + 2, 0, v < vmax // 8 + 2)):
            ids, cnt = eng.scan_select(op, x, col, capacity=n, b=y, first_row=7)
            want = np.nonzero(expect)[0].astype(np.int64) + 7
            k = int(cnt.item())
            assert k == want.shape[0] and np.array_equal(ids[:k].cpu().numpy(), want), (forced, op, c, n)
    eng.set_option("select_kernel", 0)
    # (selectivities 1/2 and ~1: tiles expanded 64 rows at a time from the registers; ~1/8 .. 1/64: through the LDS
    # stage; a single key: mostly sparse tiles written straight from the lanes)
    for op, x, y, expect in (("==", a, 0, v == a), ("<=", vmax // 2, 0, v <= vmax // 2), ("!=", a, 0, v != a),
                             ("<", vmax // 8 + 2, 0, v < vmax // 8 + 2), ("<=", vmax // 64, 0, v <= vmax // 64),
                             ("between", 5, 2, np.zeros(n, bool))):
        ids, cnt = eng.scan_select(op, x, col, capacity=n, b=y, first_row=10_000_000_000)
        want = np.nonzero(expect)[0].astype(np.int64) + 10_000_000_000
        k = int(cnt.item())
        assert k == want.shape[0], (op, c, n)
        assert np.array_equal(ids[:k].cpu().numpy(), want), (op, c, n)
    mask, _ = eng.scan_where(">=", vmax // 4, col)
    first = v >= vmax // 4
    for mop, expect in (("and", (v <= a) & first), ("or", (v <= a) | first), ("andnot", first & ~(v <= a))):
        ids, cnt = eng.scan_select("<=", a, col, capacity=n, mask=mask, mask_op=mop)
        want = np.nonzero(expect)[0].astype(np.int64)
        k = int(cnt.item())
        assert k == want.shape[0] and np.array_equal(ids[:k].cpu().numpy(), want), (mop, c, n)
    # capacity smaller than the result: the count is still the total, ids beyond the capacity are not written
    import torch

    expect = np.nonzero(v <= vmax // 2)[0].astype(np.int64)
    cap = max(1, expect.shape[0] // 3)
    ids, cnt = eng.scan_select("<=", vmax // 2, col, capacity=cap)
    assert int(cnt.item()) == expect.shape[0]
    assert np.array_equal(ids[: min(cap, expect.shape[0])].cpu().numpy(), expect[:cap])
    torch.cuda.synchronize()


def test_scan_select_beside_another_contexts_long_kernels(O):
    """Contexts are independent (include/mi355_scan.h): a fused selection on context A / stream A while context B /
    stream B keeps the device busy with back-to-back 1e9-row shared scans.  The selection's blocks then start at
    different times (some only when B's blocks retire); chunks are claimed from a ticket counter, so the look-back never
    waits for a block that has not started: count and ids exact, never the give-up flag UINT64_MAX (round 2 dealt
    chunks out by block index and could return it here)."""
    import torch

    from shared_simd_scan_amd import ScanEngine

    n, c = 1_000_000_000, 9
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    A, B = ScanEngine(0, stream=sA), ScanEngine(0, stream=sB)
    refs = []
    with torch.cuda.stream(sA):  # (allocations and .item() follow the current stream: make it A's)
        col = A.generate("splitmix", n, c, 42)
        key = int(O.gen_values("splitmix", 1, c, 42, first=12345)[0])
        for op, x in (("==", key), ("<", 256)):
            bm, hits = A.scan_where(op, x, col)
            h = int(hits.item())
            ids_ref, _ = A.bitmap_to_rowids(bm, n, capacity=h)
            refs.append((op, x, h, ids_ref))
            sA.synchronize()
            del bm
    torch.cuda.synchronize()
    P = 64
    keys = [(37 * k + 3) % 512 for k in range(P)]
    stride = ((n + 7) // 8 + 255) // 256 * 256
    out = torch.empty((P, stride), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for rep in range(4):
        for op, x, h, ids_ref in refs:
            with torch.cuda.stream(sB):
                for _ in range(3):  # ~3 ms each on stream B: the selection below is enqueued while these run
                    B.shared_scan(keys, col, out=out, hits=False)
            with torch.cuda.stream(sA):
                ids, cnt = A.scan_select(op, x, col, capacity=h)
            with torch.cuda.stream(sB):
                for _ in range(2):
                    B.shared_scan(keys, col, out=out, hits=False)
            sA.synchronize()
            with torch.cuda.stream(sA):
                k = int(cnt.item())
                assert k == h, (op, rep, k, h)
                assert torch.equal(ids[:h], ids_ref[:h]), (op, rep)
            sA.synchronize()
            del ids
    torch.cuda.synchronize()
    del out, col, refs
    torch.cuda.empty_cache()


def test_scan_select_1e9_chain(O, eng):
    """1e9 x 9 bit: the fused select against the chain scan -> bitmap -> row ids (7630 chunks: the look-back crosses
    many generations of the persistent grid), sparse (1/512) and dense (1/2) predicates, repeated"""
    import torch

    n, c = 1_000_000_000, 9
    col = eng.generate("splitmix", n, c, 42)
    key = int(O.gen_values("splitmix", 1, c, 42, first=12345)[0])
    for rep in range(3):
        bm, hits = eng.scan(key, col)
        h = int(hits.item())
        ids_ref, cnt_ref = eng.bitmap_to_rowids(bm, n, capacity=h)
        ids, cnt = eng.scan_select("==", key, col, capacity=h)
        assert int(cnt.item()) == h == int(cnt_ref.item())
        assert torch.equal(ids[:h], ids_ref[:h])
    del ids, ids_ref
    bm, hits = eng.scan_where("<", 256, col)
    h = int(hits.item())
    ids_ref, cnt_ref = eng.bitmap_to_rowids(bm, n, capacity=h)
    ids, cnt = eng.scan_select("<", 256, col, capacity=h)
    assert int(cnt.item()) == h and torch.equal(ids[:h], ids_ref[:h])
    # conjunction over "two columns": second predicate fused with the first one's bitmap, straight to row ids
    ids2, cnt2 = eng.scan_select(">=", 200, col, capacity=h, mask=bm, mask_op="and")
    dec = eng.decompress(col)
    want = torch.nonzero((dec >= 200) & (dec < 256)).flatten()
    assert int(cnt2.item()) == want.numel() and torch.equal(ids2[: want.numel()], want)
    del dec, ids, ids_ref, ids2, col
    torch.cuda.empty_cache()


@pytest.mark.parametrize("c1,c2", [(9, 9), (1, 1), (5, 5), (16, 16), (21, 21), (32, 32), (9, 12), (17, 5)])
@pytest.mark.parametrize("n", [8192 * 9 + 77, 1000, 4096 * 33 + 1])
def test_two_column_predicates(O, eng, c1, c2, n):
    """mi355_scan2_dev: predicates over two columns combined in one call (same width: one launch, no intermediate
    bitmap; different widths: two launches, the combination fused into the second) against numpy"""
    v1, col1 = make_column(O, eng, n, c1, 9700 + c1 + n)
    v2, col2 = make_column(O, eng, n, c2, 9800 + c2 + n)
    a1, a2 = int(v1[3]), int(v2[7])
    p1 = v1.astype(np.int64) <= a1
    p2 = v2.astype(np.int64) != a2
    for comb, expect in (("and", p1 & p2), ("or", p1 | p2), ("xor", p1 ^ p2), ("andnot", p1 & ~p2)):
        bm, hits = eng.scan2(col1, "<=", a1, col2, "!=", a2, combine=comb)
        assert np.array_equal(bm.cpu().numpy(), np_bitmap(expect)), (comb, c1, c2, n)
        assert int(hits.item()) == int(expect.sum())
        none, hits = eng.scan2(col1, "<=", a1, col2, "!=", a2, combine=comb, count_only=True)
        assert none is None and int(hits.item()) == int(expect.sum())
    lo, hi = int(min(v2[1], v2[2])), int(max(v2[1], v2[2]))
    bm, hits = eng.scan2(col1, "==", a1, col2, "between", lo, b2=hi, combine="or")
    expect = (v1.astype(np.int64) == a1) | ((v2.astype(np.int64) >= lo) & (v2.astype(np.int64) <= hi))
    assert np.array_equal(bm.cpu().numpy(), np_bitmap(expect)) and int(hits.item()) == int(expect.sum())


@pytest.mark.parametrize("n", [1, 63, 64, 127, 128, 1000, 16384, 16385, 100_003, 1_000_003])
def test_bitmap_consumers(O, eng, n):
    import torch

    rng = np.random.default_rng(n)
    x = rng.random(n) < 0.3
    y = rng.random(n) < 0.5
    bx, by = torch.from_numpy(np_bitmap(x)).cuda(), torch.from_numpy(np_bitmap(y)).cuda()
    # torch tensors from numpy are not necessarily 16-byte aligned on the device: clone into fresh allocations
    bx, by = bx.clone(), by.clone()
    for op, expect in (("and", x & y), ("or", x | y), ("xor", x ^ y), ("andnot", x & ~y)):
        out, cnt = eng.bitmap_combine(op, bx, by, n)
        assert np.array_equal(out.cpu().numpy(), np_bitmap(expect)), op
        assert int(cnt.item()) == int(expect.sum())
    assert int(eng.bitmap_count(bx, n).item()) == int(x.sum())
    ids, cnt = eng.bitmap_to_rowids(bx, n, capacity=n, first_row=10_000_000_000)
    k = int(cnt.item())
    assert k == int(x.sum())
    assert np.array_equal(ids.cpu().numpy()[:k], np.flatnonzero(x) + 10_000_000_000)
    # capacity smaller than the number of hits: the first `capacity` ids, the full count
    cap = max(1, k // 3)
    ids, cnt = eng.bitmap_to_rowids(bx, n, capacity=cap)
    assert int(cnt.item()) == k and np.array_equal(ids.cpu().numpy()[:min(cap, k)], np.flatnonzero(x)[:cap])


def test_scan_then_select_1e9(O, eng):
    """the full chain at BASELINE size: scan 1e9 x 9 bit -> bitmap -> row ids"""
    n, c = 1_000_000_000, 9
    col = eng.generate("splitmix", n, c, 42)
    key = int(O.gen_values("splitmix", 1, c, 42, first=12345)[0])
    bm, hits = eng.scan(key, col)
    h = int(hits.item())
    ids, cnt = eng.bitmap_to_rowids(bm, n, capacity=h)
    assert int(cnt.item()) == h
    ids = ids.cpu().numpy()[:h]
    assert (np.diff(ids) > 0).all() and ids[0] >= 0 and ids[-1] < n
    # every selected row really holds the key (oracle generator), spot-check 2000 of them + the known row 12345
    assert 12345 in ids[np.searchsorted(ids, 12345): np.searchsorted(ids, 12345) + 1]
    pick = ids[:: max(1, h // 2000)]
    for r in pick[:50]:
        assert int(O.gen_values("splitmix", 1, c, 42, first=int(r))[0]) == key
    assert int(eng.bitmap_count(bm, n).item()) == h


def test_interleaved_launches_keep_scratch_clean(O, eng):
    """the kernels share per-context scratch (hit-count replicas, tickets): interleave every kind of launch and check
    each result -- a kernel that left the scratch dirty would corrupt the next one's counts"""
    import torch

    n, c = 5 * 8192 + 4321, 9
    vals, col = make_column(O, eng, n, c, 31337)
    packed_host = col.data.cpu().numpy()
    v = vals.astype(np.int64)
    keys8 = [int(x) for x in vals[:8]]
    keys100 = [int(x) for x in vals[100:200]]
    exp8 = O.shared_scan_eq(packed_host, n, c, keys8)[1]
    exp100 = O.shared_scan_eq(packed_host, n, c, keys100)[1]
    for rnd in range(6):
        key = int(vals[rnd])
        _, h = eng.scan(key, col)
        assert int(h.item()) == int((v == key).sum())
        _, h8 = eng.shared_scan(keys8, col, layout="linear" if rnd % 2 else "per_predicate")
        assert np.array_equal(h8.cpu().numpy().astype(np.uint64), exp8)
        bm, hr = eng.scan_range(10, 200, col)
        assert int(hr.item()) == int(((v >= 10) & (v <= 200)).sum())
        assert int(eng.bitmap_count(bm, n).item()) == int(hr.item())
        _, h100 = eng.shared_scan(keys100, col, layout="per_predicate" if rnd % 2 else "linear")
        assert np.array_equal(h100.cpu().numpy().astype(np.uint64), exp100)
        _, hw = eng.scan_where("!=", key, col, and_mask=bm)
        assert int(hw.item()) == int(((v != key) & (v >= 10) & (v <= 200)).sum())
        ids, cnt = eng.bitmap_to_rowids(bm, n, capacity=16)
        assert int(cnt.item()) == int(hr.item())
    torch.cuda.synchronize()
    # a stream switch in the middle of queued work: the scratch belongs to the context, so mi355_ctx_set_stream orders the
    # new stream behind what the old one still has to run (no host synchronisation in between)
    original = eng.stream
    side = torch.cuda.Stream()
    try:
        for rnd in range(4):
            key = int(vals[50 + rnd])
            _, h_a = eng.shared_scan(keys100, col)          # long-running launch on the current stream
            eng.use_stream(side if rnd % 2 == 0 else original)
            _, h_b = eng.scan(key, col)                     # same scratch, other stream
            _, h_c = eng.shared_scan(keys8, col)
            torch.cuda.synchronize()
            assert np.array_equal(h_a.cpu().numpy().astype(np.uint64), exp100)
            assert int(h_b.item()) == int((v == key).sum())
            assert np.array_equal(h_c.cpu().numpy().astype(np.uint64), exp8)
    finally:
        eng.use_stream(original)
    torch.cuda.synchronize()


def test_two_contexts_two_streams(O):
    """independent contexts on their own streams do not share scratch"""
    import torch

    from shared_simd_scan_amd import ScanEngine

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    e1, e2 = ScanEngine(0, stream=s1), ScanEngine(0, stream=s2)
    n, c = 2_000_000, 9
    with torch.cuda.stream(s1):
        c1 = e1.generate("mod", n, c, 5)
    with torch.cuda.stream(s2):
        c2 = e2.generate("mod", n, c, 7)
    torch.cuda.synchronize()
    outs = []
    for _ in range(20):
        with torch.cuda.stream(s1):
            outs.append(("a", e1.scan(3, c1)[1]))
        with torch.cuda.stream(s2):
            outs.append(("b", e2.scan(3, c2)[1]))
    torch.cuda.synchronize()
    for tag, h in outs:
        assert int(h.item()) == (n // 5 if tag == "a" else (n - 3 + 6) // 7)


@pytest.mark.parametrize("c,P", [(1, 1), (5, 7), (9, 1), (9, 40), (12, 300), (16, 1024), (17, 5), (21, 33), (32, 4)])
def test_scan_in_list(O, eng, c, P):
    n = 2 * 8192 + 1001
    vals, col = make_column(O, eng, n, c, 7000 + c + P)
    rng = np.random.default_rng(P * 31 + c)
    keys = [int(vals[int(i)]) for i in rng.integers(0, n, size=P)]
    if P > 3:
        keys[1] = keys[0]
        keys[2] = -5
        if c < 31:
            keys[3] = (1 << c) + 1
    keys = [k if k < 2 ** 31 else k - 2 ** 32 for k in keys]
    member = np.isin(vals.astype(np.int64), np.array([k for k in keys if 0 <= k < (1 << c)] +
                                                     [k + 2 ** 32 for k in keys if k < 0 and c == 32], dtype=np.int64))
    bm, hits = eng.scan_in(keys, col)
    assert np.array_equal(bm.cpu().numpy(), np_bitmap(member)) and int(hits.item()) == int(member.sum())
    bm, hits = eng.scan_in(keys, col, negate=True)
    assert np.array_equal(bm.cpu().numpy(), np_bitmap(~member)) and int(hits.item()) == int((~member).sum())
    prev, _ = eng.scan_where(">=", int(vals[7]), col)
    bm, hits = eng.scan_in(keys, col, and_mask=prev)
    expect = member & (vals.astype(np.int64) >= int(vals[7]))
    assert np.array_equal(bm.cpu().numpy(), np_bitmap(expect)) and int(hits.item()) == int(expect.sum())


@pytest.mark.parametrize("seed", list(range(12)))
def test_fuzz_random_shapes_and_predicates(O, eng, seed):
    """seeded random (width, row count, predicate) combinations against numpy on the decoded values: 25 columns per
    seed; row counts cluster around tile boundaries, keys around the ends of the value range"""
    rng = np.random.default_rng(987_000 + seed)
    for _ in range(25):
        c = int(rng.integers(1, 33))
        tile = 8192 if c <= 16 else 4096
        n = int(rng.choice([rng.integers(1, 300), tile * rng.integers(1, 6) + rng.integers(-70, 70), rng.integers(1, 70_000)]))
        n = max(1, n)
        vmax = (1 << c) - 1
        vals, col = make_column(O, eng, n, c, int(rng.integers(0, 1 << 30)))
        v = vals.astype(np.int64)

        def pick():
            cand = [0, vmax, int(vals[int(rng.integers(0, n))]), int(rng.integers(0, vmax + 1))]
            if c < 31:  # keys outside [0, 2^c) match nothing; at c >= 31 they are not expressible as distinct int32 keys
                cand += [vmax + 3, -1]
            return int(cand[int(rng.integers(0, len(cand)))])

        assert np.array_equal(eng.decompress(col).cpu().numpy().view(np.uint32), vals), (c, n)
        a, b = pick(), pick()
        for op, expect in (("==", v == a), ("!=", v != a), ("<", v < a), (">=", v >= a),
                           ("between", (v >= a) & (v <= b)), ("not_between", ~((v >= a) & (v <= b)))):
            bm, hits = eng.scan_where(op, a, col, b=b)
            assert np.array_equal(bm.cpu().numpy(), np_bitmap(expect)) and int(hits.item()) == int(expect.sum()), (c, n, op, a, b)
        key = a if a < 2 ** 31 else a - 2 ** 32
        bm, hits = eng.scan(key, col)
        assert np.array_equal(bm.cpu().numpy(), np_bitmap(v == a)) and int(hits.item()) == int((v == a).sum()), (c, n, a)
        P = int(rng.choice([1, 2, 3, 4, 7, 8, 9, 17, 32, 33, 70]))
        keys = [pick() for _ in range(P)]
        keys32 = [k if k < 2 ** 31 else k - 2 ** 32 for k in keys]
        layout = str(rng.choice(["per_predicate", "linear"]))
        count = bool(rng.integers(0, 2))
        out, hits = eng.shared_scan(keys32, col, layout=layout, hits=None if count else False)
        nb = (n + 7) // 8
        got = out.cpu().numpy()
        per_key = np.stack([np_bitmap(v == k) for k in keys])  # [P, nb]
        if layout == "per_predicate":
            assert np.array_equal(got[:, :nb], per_key), (c, n, P, layout)
        else:
            assert np.array_equal(got.reshape(nb, P), per_key.T), (c, n, P, layout)
        if count:
            assert np.array_equal(hits.cpu().numpy(), np.array([int((v == k).sum()) for k in keys])), (c, n, P)
        member = np.isin(v, np.array([k for k in keys if 0 <= k <= vmax], dtype=np.int64))
        bm, hits = eng.scan_in(keys32, col)
        assert np.array_equal(bm.cpu().numpy(), np_bitmap(member)) and int(hits.item()) == int(member.sum()), (c, n, P)


FUZZ2_SEEDS = int(os.environ.get("MI355_FUZZ_SEEDS", "10"))  # (a one-off stress run sets it to a few hundred)


@pytest.mark.parametrize("seed", list(range(FUZZ2_SEEDS)))
def test_fuzz_round2_entry_points(O, eng, seed):
    """seeded random columns through the round-2 entry points, against numpy on the oracle-decoded values: shared scans of
    ANY key count (biased to the kernel boundaries 2, 8 / 9, 16, 32 / 33, 64, 128, 256 / 257, 320 / 321) in both layouts
    with and without hit counts, count-only and fused-mask scans, two columns in one call, the fused selection (with a
    capacity below the count now and then), take-by-row-id.  20 columns per seed."""
    import torch

    rng = np.random.default_rng(555_000 + seed)
    pool = [2, 3, 5, 6, 7, 8, 9, 10, 15, 16, 17, 31, 32, 33, 40, 63, 64, 65, 96, 127, 128, 129, 160, 255, 256, 257, 300, 320, 321, 400, 512, 700, 1024]
    for _ in range(20):
        c = int(rng.integers(1, 33))
        tile = 8192 if c <= 16 else 4096
        n = max(1, int(rng.choice([rng.integers(1, 300), tile * rng.integers(1, 20) + rng.integers(-70, 70), rng.integers(1, 200_000)])))
        vmax = (1 << c) - 1
        vals, col = make_column(O, eng, n, c, int(rng.integers(0, 1 << 30)))
        v = vals.astype(np.int64)
        nb = (n + 7) // 8

        def pick():
            cand = [0, vmax, int(vals[int(rng.integers(0, n))]), int(vals[int(rng.integers(0, n))]), int(rng.integers(0, vmax + 1))]
            if c < 31:
                cand += [vmax + 3, -1]
            return int(cand[int(rng.integers(0, len(cand)))])

        # shared scan, any key count
        P = int(rng.choice(pool)) if rng.integers(0, 4) else int(rng.integers(1, 1025))
        keys = [pick() for _ in range(P)]
        keys32 = [k if k < 2 ** 31 else k - 2 ** 32 for k in keys]
        layout = str(rng.choice(["per_predicate", "linear"]))
        count = bool(rng.integers(0, 2))
        out, hits = eng.shared_scan(keys32, col, layout=layout, hits=None if count else False)
        per_key = np.stack([np_bitmap(v == k) for k in keys])  # [P, nb]
        got = out.cpu().numpy()
        if layout == "per_predicate":
            assert np.array_equal(got[:, :nb], per_key), (seed, c, n, P, layout, count)
        else:
            assert np.array_equal(got.reshape(nb, P), per_key.T), (seed, c, n, P, layout, count)
        if count:
            assert np.array_equal(hits.cpu().numpy(), np.array([int((v == k).sum()) for k in keys])), (seed, c, n, P, layout)
        # count-only, fused masks, two columns
        a, b = sorted((pick(), pick()))
        p1 = (v >= a) & (v <= b)
        bm1, h1 = eng.scan_where("between", a, col, b=b)
        assert np.array_equal(bm1.cpu().numpy(), np_bitmap(p1)) and int(h1.item()) == int(p1.sum()), (seed, c, n, a, b)
        _, hc = eng.scan_combine("between", a, col, b=b, count_only=True)
        assert int(hc.item()) == int(p1.sum())
        k2 = pick()
        for mop, fn in (("and", lambda p, m: p & m), ("or", lambda p, m: p | m), ("xor", lambda p, m: p ^ m), ("andnot", lambda p, m: m & ~p)):
            bmm, hm = eng.scan_combine("==", k2, col, mask=bm1, mask_op=mop)
            e = fn(v == k2, p1)
            assert np.array_equal(bmm.cpu().numpy(), np_bitmap(e)) and int(hm.item()) == int(e.sum()), (seed, c, n, mop)
        vals2, col2 = make_column(O, eng, n, c, int(rng.integers(0, 1 << 30)))
        v2 = vals2.astype(np.int64)
        b2, h2 = eng.scan2(col, "<=", b, col2, ">", a, combine="or")
        e2 = (v <= b) | (v2 > a)
        assert np.array_equal(b2.cpu().numpy(), np_bitmap(e2)) and int(h2.item()) == int(e2.sum()), (seed, c, n, a, b)
        # fused selection (sometimes with too little room) and take-by-row-id from the second column
        rows = np.nonzero(p1)[0].astype(np.int64)
        cap = n if rng.integers(0, 3) else max(1, rows.shape[0] // 2)
        first = int(rng.choice([0, 1 << 33]))
        ids, cnt = eng.scan_select("between", a, col, capacity=cap, b=b, first_row=first)
        assert int(cnt.item()) == rows.shape[0], (seed, c, n, a, b)
        k = min(cap, rows.shape[0])
        assert np.array_equal(ids[:k].cpu().numpy(), rows[:k] + first), (seed, c, n, a, b, cap)
        if c <= 14:
            assert np.array_equal(eng.histogram(col2, mask=bm1).cpu().numpy(), np.bincount(vals2[p1], minlength=1 << c)), (seed, c, n)
        agg = eng.aggregate(col2, mask=bm1).cpu().numpy().view(np.uint64).tolist()
        u2 = vals2.astype(np.uint64)
        assert agg == ([int(u2[p1].sum()), int(p1.sum()), int(u2[p1].min()), int(u2[p1].max())] if p1.any() else [0, 0, 2 ** 64 - 1, 0]), (seed, c, n)
        if k:
            taken = eng.gather(col2, ids[:k], k, first_row=first)
            assert np.array_equal(taken[:k].cpu().numpy().view(np.uint32), vals2[rows[:k]]), (seed, c, n)


@pytest.mark.parametrize("c", [5, 9, 12, 21])
def test_slice_rows_scans_equal_slices_of_the_full_scan(O, eng, c):
    """ShardedColumn.scan_pipelined scans row ranges of a resident column through views (ScanEngine.slice_rows):
    interior slices are followed by more rows instead of the zero pad; their bitmaps must equal the matching
    bytes of the full scan, ragged last slice included"""
    n = 8192 * 7 + 3001
    vals, col = make_column(O, eng, n, c, 31337 + c)
    key = int(vals[11])
    full, hits = eng.scan(key, col)
    full = full.cpu().numpy()
    total = 0
    for a, b in ((0, 8192), (8192, 8192 * 4), (8192 * 4, 8192 * 4 + 128), (8192 * 4 + 128, n), (n - n % 128, n), (128 * 5, 128 * 5)):
        bm, h = eng.scan(key, eng.slice_rows(col, a, b))
        if a % 8 == 0:
            expect = np_bitmap((vals[a:b].astype(np.int64) == key))
            assert np.array_equal(bm.cpu().numpy()[: (b - a + 7) // 8], expect), (c, a, b)
            if b < n and (b - a) % 8 == 0 and b > a:
                assert np.array_equal(expect, full[a // 8: b // 8])
        assert int(h.item()) == int((vals[a:b] == key).sum())
        total += int(h.item()) if (a, b) in ((0, 8192), (8192, 8192 * 4), (8192 * 4, 8192 * 4 + 128), (8192 * 4 + 128, n)) else 0
    assert total == int(hits.item())
    with pytest.raises(ValueError):
        eng.slice_rows(col, 100, 200)


@pytest.mark.parametrize("c", [1, 7, 9, 13, 16])
@pytest.mark.parametrize("dtype", ["u16", "u32"])
@pytest.mark.parametrize("shift", [0, 1])
def test_device_packer_full_tiles_and_unaligned_sources(O, eng, c, dtype, shift):
    """the tiled packer stages 8192-value tiles through LDS with 16-byte loads when the source is 16-byte aligned and
    falls back to element loads otherwise (shift = 1: the source starts one element into an allocation); ragged
    last tile, u16 (the reference's compress_9bit_input type) and u32 sources"""
    import torch

    n = 3 * 8192 + 77
    rng = np.random.default_rng(55 + c)
    vals = rng.integers(0, 1 << c, size=n + shift, dtype=np.uint64).astype(np.uint32)
    if dtype == "u16":
        t = torch.from_numpy(vals.astype(np.uint16).view(np.int16)).cuda()
    else:
        t = torch.from_numpy(vals.view(np.int32)).cuda()
    src = t[shift:]
    assert (src.data_ptr() % 16 == 0) == (shift == 0)
    col = eng.compress(src, c)
    assert np.array_equal(col.data.cpu().numpy(), O.pack(vals[shift:], c)), (c, dtype, shift)


def test_many_key_lists_in_flight_without_synchronising(O, eng):
    """P > 8 key lists are uploaded asynchronously through a ring of 8 pinned slots: 40 shared scans / IN-list scans
    with DIFFERENT key lists are enqueued back to back (no host synchronisation in between) and checked afterwards --
    a slot reused too early would hand a scan another call's keys"""
    import torch

    n, c = 8192 * 3 + 333, 9
    vals, col = make_column(O, eng, n, c, 2468)
    v = vals.astype(np.int64)
    rng = np.random.default_rng(99)
    jobs = []
    for i in range(40):
        P = int(rng.integers(9, 60))
        keys = [int(k) for k in rng.integers(0, 1 << c, size=P)]
        if i % 2:
            out, hits = eng.shared_scan(keys, col)
            jobs.append(("shared", keys, out, hits))
        else:
            bm, hits = eng.scan_in(keys, col)
            jobs.append(("in", keys, bm, hits))
    torch.cuda.synchronize()
    nb = (n + 7) // 8
    for kind, keys, out, hits in jobs:
        if kind == "shared":
            expect = np.stack([np_bitmap(v == k) for k in keys])
            assert np.array_equal(out.cpu().numpy()[:, :nb], expect)
            assert np.array_equal(hits.cpu().numpy(), np.array([int((v == k).sum()) for k in keys]))
        else:
            member = np.isin(v, np.array(keys, dtype=np.int64))
            assert np.array_equal(out.cpu().numpy(), np_bitmap(member)) and int(hits.item()) == int(member.sum())


def test_scan_can_be_captured_in_a_hip_graph(O, eng):
    """the *_dev scan entry points only enqueue work (no allocation, no sync): capture one into a HIP graph on a
    side stream and replay it"""
    import torch

    n, c = 3_000_000, 9
    vals, col = make_column(O, eng, n, c, 2024)
    bitmap = eng.alloc_bitmap(n)
    hits = torch.zeros(1, dtype=torch.int64, device="cuda")
    key = int(vals[5])
    eng.scan(key, col, bitmap=bitmap, hits=hits)  # warm-up outside capture (module load)
    torch.cuda.synchronize()
    original = eng.stream
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        eng.use_stream(side)
        with torch.cuda.graph(g, stream=side):
            eng.use_stream(torch.cuda.current_stream())
            eng.scan(key, col, bitmap=bitmap, hits=hits)
    eng.use_stream(original)
    bitmap.zero_()
    hits.zero_()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    obm, ohits = O.scan_eq(col.data.cpu().numpy(), n, c, key)
    assert np.array_equal(bitmap.cpu().numpy(), obm) and int(hits.item()) == ohits
