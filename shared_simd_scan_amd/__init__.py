"""shared_simd_scan_amd -- MI355X-native bit-packed column decompress / predicate scan engine.

Drop-in for the hot path of RRr89/Shared_SIMD_Scan (src/simd_scan.hpp): compress, decompress,
equality / range scan and shared multi-predicate scan as hand-written gfx950 HIP kernels behind a C
ABI (include/mi355_scan.h, shared_simd_scan_amd/libmi355scan.so).  See DESIGN.md.
"""
from ._capi import Mi355Error, lib  # noqa: F401
from .engine import (PackedColumn, ScanEngine, compressed_buffer_size, decompression_output_buffer_size,  # noqa: F401
                     kernel_name, scan_output_buffer_size, tile_values)

__all__ = ["Mi355Error", "PackedColumn", "ScanEngine", "compressed_buffer_size", "decompression_output_buffer_size",
           "scan_output_buffer_size", "kernel_name", "tile_values", "lib"]
