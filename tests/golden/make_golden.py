#!/usr/bin/env python3
"""Generate tests/golden/ref_w<W>.npz from the REAL reference (oracle/_ref/libref_w<W>.so).

Run in the build container only (needs /root/reference to have been compiled by
`make -C oracle ref`).  The fixtures are DATA: seeded inputs and the outputs the reference's own
functions produced for them.  Nothing of the reference's source text is stored.

For each width W in WIDTHS and each n in SIZES:
  values      uint32[n]      seeded random in [0, 2^W)  (n=12 at W>=2: the reference's own KAT
                             column {1,2,3,3,2,1,1,2,3,1,2,3}, test/simd_scan_tests.cpp:47-48;
                             n=509 at W=9: the identity column of test/simd_scan_tests.cpp:10-15)
  packed      uint8[...]     reference compress_9bit_input for W<=16 (whole padded buffer);
                             for W>16 the reference cannot pack (uint16_t input): bytes produced by
                             oracle.pack and accepted only because reference decompress_128 returns
                             `values` from them (asserted here)
  decomp      int32[n]       reference decompress_128 (src/simd_scan_decompression.cpp:237)
  keys        int32[K]       scan keys: a present key, another, 2^W-1, 0 and an out-of-range key
  scan128     uint8[K, sobs] reference scan_128 whole padded output buffers
  scan128_hits int32[K]
  shared_keys_P<P>, shared_std_P<P>  uint8[P, sobs]   reference shared_scan_128_standard
  linear_std_P<P>                    uint8[P*sobs]    reference shared_scan_128_linear_standard
  (W=9 only) linear_simple_P{1,2}: reference shared_scan_128_linear_simple on the KAT column
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import RefLib, oracle  # noqa: E402

WIDTHS = [5, 7, 9, 12, 17, 21]
SIZES = [12, 13, 509, 1000, 4101]
KAT = np.array([1, 2, 3, 3, 2, 1, 1, 2, 3, 1, 2, 3], dtype=np.uint32)


def main():
    O = oracle()
    # output directory: this one, or argv[1] (tests/test_oracle_golden.py regenerates into a scratch directory and
    # compares with the committed files)
    here = sys.argv[1] if len(sys.argv) > 1 else os.path.dirname(os.path.abspath(__file__))
    for W in WIDTHS:
        R = RefLib(W)
        rng = np.random.default_rng(1000 + W)
        d = {"width": np.int32(W), "sizes": np.array(SIZES, dtype=np.int64)}
        for n in SIZES:
            if n == 12:
                values = KAT.copy()
            elif n == 509 and W == 9:
                values = np.arange(n, dtype=np.uint32)
            else:
                values = rng.integers(0, 1 << W, size=n, dtype=np.uint32)
            if W <= 16:
                packed = R.compress(values.astype(np.uint16))
            else:
                packed = O.pack(values, W)
            decomp = R.decompress("decompress_128", packed, n)[:n]
            assert np.array_equal(decomp, values.astype(np.int32)), (W, n)
            keys = np.array([int(values[0]), int(values[n // 2]), (1 << W) - 1, 0, (1 << W) + 3], dtype=np.int32)
            bufs, hits = [], []
            for k in keys:
                b, h = R.scan("scan_128", int(k), packed, n)
                bufs.append(b)
                hits.append(h)
            p = f"n{n}_"
            d[p + "values"] = values
            d[p + "packed"] = packed
            d[p + "decomp"] = decomp
            d[p + "keys"] = keys
            d[p + "scan128"] = np.stack(bufs)
            d[p + "scan128_hits"] = np.array(hits, dtype=np.int32)
            for P in (1, 3, 8):
                if n == 12 and P == 3:
                    ks = np.array([1, 2, 3], dtype=np.int32)  # test/simd_scan_tests.cpp:92
                else:
                    ks = np.array([int(values[(31 * k + 5) % n]) for k in range(P)], dtype=np.int32)
                d[p + f"shared_keys_P{P}"] = ks
                d[p + f"shared_std_P{P}"] = R.shared_scan("shared_scan_128_standard", ks, packed, n)
                d[p + f"linear_std_P{P}"] = R.shared_scan_linear("shared_scan_128_linear_standard", ks, packed, n)
            if W == 9 and n == 12:
                # test/simd_scan_tests.cpp:115-148
                d[p + "linear_simple_P1"] = R.shared_scan_linear("shared_scan_128_linear_simple", [1], packed, n)
                d[p + "linear_simple_P2"] = R.shared_scan_linear("shared_scan_128_linear_simple", [2, 3], packed, n)
        out = os.path.join(here, f"ref_w{W}.npz")
        np.savez_compressed(out, **d)
        print(out, os.path.getsize(out), "bytes")

    # tail-behaviour fixture (SURVEY 8c hazard 1): n=13, key 0 at W=9 for every scan variant
    R = RefLib(9)
    values = np.array([0, 5, 7, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9], dtype=np.uint32)
    packed = R.compress(values.astype(np.uint16))
    d = {"values": values, "packed": packed}
    for v in ("scan_unvectorized", "scan_128", "scan_128_unrolled", "scan_256", "scan_256_unrolled"):
        b, h = R.scan(v, 0, packed, 13)
        d[v] = b
        d[v + "_hits"] = np.int32(h)
    np.savez_compressed(os.path.join(here, "ref_tail_w9.npz"), **d)


if __name__ == "__main__":
    main()
