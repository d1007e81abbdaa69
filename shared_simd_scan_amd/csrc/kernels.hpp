// kernels.hpp -- CDNA4 (gfx950) kernels of the bit-packed column engine.  gfx950 only.
//
// Data flow of every kernel that reads a packed column (DESIGN.md "kernels"):
//
//   HBM --global_load_lds_dwordx4 (1 KiB / wave-instruction, coalesced)--> per-wave LDS tile
//       --ds_read_b128/b64 (each lane fetches the 16C / 8C bytes that hold ITS OWN run of
//         128 / 64 consecutive values: the LDS does the gather a CPU does with pshufb)-->
//   VGPRs --v_bfe_u32 / v_alignbit_b32 at COMPILE-TIME bit offsets (the run starts on a dword
//         boundary, so every shift is a constant)--> value
//       --v_cmp + v_addc_co_u32 (acc = 2*acc + match: one VALU op appends a result bit)-->
//   one 32-bit bitmap word per 32 values in the lane, written back as 16 B / lane (1 KiB / wave).
//
// Kernels (one header per family under kernels/):
//   scan_burst_kernel      equality / range scan (+ negation, + AND / OR / XOR / ANDNOT with an earlier bitmap, count-only),
//                          one bitmap; widths <= 7 evaluate several values per LDS table lookup instead of the compare chain
//   scan2_kernel           predicates over two columns of one width in one launch
//   shared_lut_kernel      shared multi-predicate scan (P <= 8, and linear rows below 192 keys) through byte-entry LDS
//                          lookup tables + 8x8 bit transposes
//   shared_pair_kernel     two keys by compares
//   shared_wide_kernel, shared_wide2_kernel, shared_linear_kernel
//                          shared scan for P > 8: dword-entry tables, 32 predicates per lookup
//   shared_general_kernel  shared scan by compare chain, for key counts whose tables do not fit in LDS
//   in_kernel              IN-list scan (one bitmap for a key set)
//   select2_kernel         predicate -> ascending row ids in one launch (decoder + expander waves, decoupled look-back over chunk
//                          counts), no bitmap; select_kernel: round 2's single-role form, kept for A/B
//   decompress_kernel      packed -> int32, lane per value
//   pack_kernel            packer and synthetic column generators
//   bitmap_kernel, rowid_* bitmap combine / count, selection vector
// Only kernels libmi355scan.so launches live here; ablation / experiment kernels are under tools/ (scan_kernel_r1.hpp).
//
// What this replaces in the reference (RRr89/Shared_SIMD_Scan): the pshufb byte-gather + pmulld /
// psrld shift + pcmpeqd + movmskps chains of src/simd_scan.cpp:103-306, src/simd_scan_shared.cpp:34-151,
// src/simd_scan_shared_linear.cpp:9-62 and src/simd_scan_decompression.cpp:237-470.  No MFMA: this is
// integer/bit work bound by HBM bandwidth.
#pragma once

#include "kernels/tile.hpp"
#include "kernels/scan.hpp"
#include "kernels/shared.hpp"
#include "kernels/in_list.hpp"
#include "kernels/select.hpp"
#include "kernels/select2.hpp"
#include "kernels/bitmap.hpp"
#include "kernels/decompress.hpp"
#include "kernels/pack.hpp"
