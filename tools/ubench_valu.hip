// ubench_valu.hip -- per-instruction VALU cost on gfx950 for the compare/accumulate idioms (tuning aid).
// One block of 256*W threads on one CU (W waves per SIMD); each variant runs ITER x 64 instruction groups;
// prints shader cycles per instruction group per wave and per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

template <int V> __device__ __forceinline__ void body(uint32_t &a0, uint32_t &a1, uint32_t &a2, uint32_t &a3, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t key, uint32_t vkey)
{
    if constexpr (V == 0) { // cmp(vcc)+addc(vcc), single chain, 4 pairs
        asm volatile("v_cmp_eq_u32_e32 vcc, %4, %5\n v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n"
                     "v_cmp_eq_u32_e32 vcc, %4, %6\n v_addc_co_u32_e32 %1, vcc, %1, %1, vcc\n"
                     "v_cmp_eq_u32_e32 vcc, %4, %7\n v_addc_co_u32_e32 %2, vcc, %2, %2, vcc\n"
                     "v_cmp_eq_u32_e32 vcc, %4, %8\n v_addc_co_u32_e32 %3, vcc, %3, %3, vcc\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(key), "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "vcc");
    } else if constexpr (V == 1) { // 4 cmp e64 to 4 sgpr pairs then 4 addc e64
        unsigned long long m0, m1, m2, m3;
        asm volatile("v_cmp_eq_u32_e64 %4, %12, %8\n v_cmp_eq_u32_e64 %5, %12, %9\n v_cmp_eq_u32_e64 %6, %12, %10\n v_cmp_eq_u32_e64 %7, %12, %11\n"
                     "v_addc_co_u32_e64 %0, %4, %0, %0, %4\n v_addc_co_u32_e64 %1, %5, %1, %1, %5\n v_addc_co_u32_e64 %2, %6, %2, %2, %6\n v_addc_co_u32_e64 %3, %7, %3, %3, %7\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "s"(key));
    } else if constexpr (V == 2) { // xad + alignbit (no SGPR traffic), 4 pairs
        uint32_t t0, t1, t2, t3;
        asm volatile("v_xad_u32 %4, %8, %12, -1\n v_xad_u32 %5, %9, %12, -1\n v_xad_u32 %6, %10, %12, -1\n v_xad_u32 %7, %11, %12, -1\n"
                     "v_alignbit_b32 %0, %0, %4, 31\n v_alignbit_b32 %1, %1, %5, 31\n v_alignbit_b32 %2, %2, %6, 31\n v_alignbit_b32 %3, %3, %7, 31\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "s"(key));
    } else if constexpr (V == 3) { // 8 plain v_add (baseline full-rate VALU)
        asm volatile("v_add_u32_e32 %0, %4, %0\n v_add_u32_e32 %1, %5, %1\n v_add_u32_e32 %2, %6, %2\n v_add_u32_e32 %3, %7, %3\n"
                     "v_add_u32_e32 %0, %5, %0\n v_add_u32_e32 %1, %6, %1\n v_add_u32_e32 %2, %7, %2\n v_add_u32_e32 %3, %4, %3\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
    } else if constexpr (V == 4) { // 8 v_cmp e64 only (SGPR writers)
        unsigned long long m0, m1, m2, m3;
        asm volatile("v_cmp_eq_u32_e64 %0, %8, %4\n v_cmp_eq_u32_e64 %1, %8, %5\n v_cmp_eq_u32_e64 %2, %8, %6\n v_cmp_eq_u32_e64 %3, %8, %7\n"
                     "v_cmp_eq_u32_e64 %0, %8, %5\n v_cmp_eq_u32_e64 %1, %8, %6\n v_cmp_eq_u32_e64 %2, %8, %7\n v_cmp_eq_u32_e64 %3, %8, %4\n"
                     : "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "s"(key));
        a0 += (uint32_t)m0; 
    } else if constexpr (V == 5) { // 8 v_bfe
        asm volatile("v_bfe_u32 %0, %4, 3, 9\n v_bfe_u32 %1, %5, 5, 9\n v_bfe_u32 %2, %6, 7, 9\n v_bfe_u32 %3, %7, 9, 9\n"
                     "v_bfe_u32 %0, %0, 1, 30\n v_bfe_u32 %1, %1, 1, 30\n v_bfe_u32 %2, %2, 1, 30\n v_bfe_u32 %3, %3, 1, 30\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
    } else if constexpr (V == 6) { // 8 v_alignbit
        asm volatile("v_alignbit_b32 %0, %0, %4, 31\n v_alignbit_b32 %1, %1, %5, 31\n v_alignbit_b32 %2, %2, %6, 31\n v_alignbit_b32 %3, %3, %7, 31\n"
                     "v_alignbit_b32 %0, %0, %5, 31\n v_alignbit_b32 %1, %1, %6, 31\n v_alignbit_b32 %2, %2, %7, 31\n v_alignbit_b32 %3, %3, %4, 31\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
    } else if constexpr (V == 7) { // 8 v_xad
        asm volatile("v_xad_u32 %0, %4, %8, %0\n v_xad_u32 %1, %5, %8, %1\n v_xad_u32 %2, %6, %8, %2\n v_xad_u32 %3, %7, %8, %3\n"
                     "v_xad_u32 %0, %5, %8, %0\n v_xad_u32 %1, %6, %8, %1\n v_xad_u32 %2, %7, %8, %2\n v_xad_u32 %3, %4, %8, %3\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "s"(key));
    } else if constexpr (V == 8) { // 4 x (v_cmp vcc + v_cndmask) + 4 v_lshl_or : 12 instr
        uint32_t t0, t1, t2, t3;
        asm volatile("v_cmp_eq_u32_e32 vcc, %8, %4\n s_nop 1\n v_cndmask_b32_e64 %9, 0, 1, vcc\n v_cmp_eq_u32_e32 vcc, %8, %5\n s_nop 1\n v_cndmask_b32_e64 %10, 0, 1, vcc\n"
                     "v_cmp_eq_u32_e32 vcc, %8, %6\n s_nop 1\n v_cndmask_b32_e64 %11, 0, 1, vcc\n v_cmp_eq_u32_e32 vcc, %8, %7\n s_nop 1\n v_cndmask_b32_e64 %12, 0, 1, vcc\n"
                     "v_lshl_or_b32 %0, %0, 1, %9\n v_lshl_or_b32 %1, %1, 1, %10\n v_lshl_or_b32 %2, %2, 1, %11\n v_lshl_or_b32 %3, %3, 1, %12\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "s"(key), "v"(t0), "v"(t1), "v"(t2), "v"(t3) : "vcc");
    } else if constexpr (V == 9) { // 8 v_and_b32 (e32)
        asm volatile("v_and_b32_e32 %0, %4, %0\n v_and_b32_e32 %1, %5, %1\n v_and_b32_e32 %2, %6, %2\n v_and_b32_e32 %3, %7, %3\n"
                     "v_or_b32_e32 %0, %5, %0\n v_or_b32_e32 %1, %6, %1\n v_or_b32_e32 %2, %7, %2\n v_or_b32_e32 %3, %4, %3\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
    } else if constexpr (V == 10) { // 8 v_cmp e32 vcc only
        asm volatile("v_cmp_eq_u32_e32 vcc, %1, %2\n v_cmp_eq_u32_e32 vcc, %1, %3\n v_cmp_eq_u32_e32 vcc, %1, %4\n v_cmp_eq_u32_e32 vcc, %1, %5\n"
                     "v_cmp_eq_u32_e32 vcc, %1, %3\n v_cmp_eq_u32_e32 vcc, %1, %4\n v_cmp_eq_u32_e32 vcc, %1, %5\n v_cmp_eq_u32_e32 vcc, %1, %2\n v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n"
                     : "+v"(a0) : "s"(key), "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "vcc");
    } else if constexpr (V == 11) { // 8 independent v_addc e32 (carry via vcc chain)
        asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %4, vcc\n v_addc_co_u32_e32 %1, vcc, %1, %5, vcc\n v_addc_co_u32_e32 %2, vcc, %2, %6, vcc\n v_addc_co_u32_e32 %3, vcc, %3, %7, vcc\n"
                     "v_addc_co_u32_e32 %0, vcc, %0, %5, vcc\n v_addc_co_u32_e32 %1, vcc, %1, %6, vcc\n v_addc_co_u32_e32 %2, vcc, %2, %7, vcc\n v_addc_co_u32_e32 %3, vcc, %3, %4, vcc\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "vcc");
    }
}

template <int V> __global__ void k(uint32_t *out, unsigned long long *cyc, uint32_t key, int iters)
{
    uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    uint32_t x0 = a0 ^ 0x55, x1 = a0 ^ 0x33, x2 = a0 ^ 0x0f, x3 = a0 ^ 0xff;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) body<V>(a0, a1, a2, a3, x0, x1, x2, x3, key, a0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int V, int UNROLL> __global__ void kbig(uint32_t *out, unsigned long long *cyc, uint32_t key, int iters)
{
    uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    uint32_t x0 = a0 ^ 0x55, x1 = a0 ^ 0x33, x2 = a0 ^ 0x0f, x3 = a0 ^ 0xff;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) body<V>(a0, a1, a2, a3, x0, x1, x2, x3, key, a0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 4095] = a0 ^ a1 ^ a2 ^ a3;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x / 64] = t1 - t0;
}

template <int V, int UNROLL> void runbig(const char *name, int ninstr, uint32_t *out, unsigned long long *cyc, int blocks)
{
    for (int waves_per_simd : {1, 2, 4}) {
        int threads = 256 * waves_per_simd;
        const int iters = 32768 / UNROLL;
        hipLaunchKernelGGL((kbig<V, UNROLL>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 77u, iters);
        hipLaunchKernelGGL((kbig<V, UNROLL>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 77u, iters);
        CK(hipDeviceSynchronize());
        unsigned long long h[16];
        CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
        double c = (double)h[0] / (iters * (double)UNROLL);
        printf("%-40s blocks=%3d unroll=%4d waves/SIMD=%d  %6.2f cyc/instr/wave  = %5.2f cyc/instr/SIMD\n", name, blocks, UNROLL,
               waves_per_simd, c / ninstr, c / ninstr / waves_per_simd);
    }
}

template <int V> void run(const char *name, int ninstr, uint32_t *out, unsigned long long *cyc)
{
    for (int waves_per_simd : {1, 2, 4}) {
        int threads = 256, blocks = waves_per_simd; // all blocks land on... not guaranteed one CU; use 1 block of up to 1024 threads
        threads = 256 * waves_per_simd; blocks = 1;
        if (threads > 1024) continue;
        const int iters = 2000;
        hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 77u, iters);
        hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 77u, iters);
        CK(hipDeviceSynchronize());
        unsigned long long h[16];
        CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
        double c = (double)h[0] / (iters * 16.0);
        printf("%-44s waves/SIMD=%d  %7.2f cyc per group of %2d instr  = %5.2f cyc/instr/wave  = %5.2f cyc/instr/SIMD\n", name,
               waves_per_simd, c, ninstr, c / ninstr, c / ninstr / waves_per_simd);
    }
}

template <int V, int UNROLL> void runwall(const char *name, uint32_t *out, unsigned long long *cyc)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 32768 / UNROLL;
    for (int threads : {512, 1024}) {
        for (int blocks : {512, 1024}) {
            hipLaunchKernelGGL((kbig<V, UNROLL>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 77u, iters);
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((kbig<V, UNROLL>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 77u, iters);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double winstr = (double)blocks * (threads / 64) * 32768.0 * 8; // wave-instructions
            printf("%-28s unroll=%4d threads=%4d blocks=%4d: %8.3f ms  -> %6.2f shader-cycles(2.4GHz)/instr/SIMD\n", name, UNROLL, threads, blocks, ms,
                   ms * 1e-3 * 2.4e9 / (winstr / 1024.0));
        }
    }
}

int main()
{
    uint32_t *out; unsigned long long *cyc;
    CK(hipMalloc(&out, 4096 * 4)); CK(hipMalloc(&cyc, 64 * 8));
    runwall<1, 16>("cmp e64 + addc e64", out, cyc);
    runwall<0, 16>("cmp e32 vcc + addc e32 vcc", out, cyc);
    runwall<2, 16>("v_xad + v_alignbit", out, cyc);
    runwall<5, 16>("v_bfe_u32", out, cyc);
    runwall<6, 16>("v_alignbit_b32", out, cyc);
    runwall<7, 16>("v_xad_u32", out, cyc);
    runwall<4, 16>("v_cmp e64 only", out, cyc);
    runwall<11, 16>("v_addc e32 only", out, cyc);
    runwall<3, 16>("v_add e32", out, cyc);
    runwall<9, 16>("v_and/v_or e32", out, cyc);
    return 0;
    runbig<1, 16>("cmp e64+addc e64, 1 CU", 8, out, cyc, 1);
    runbig<1, 16>("cmp e64+addc e64, all CUs", 8, out, cyc, 256);
    runbig<1, 1024>("cmp e64+addc e64, 64KB body, 1 CU", 8, out, cyc, 1);
    runbig<1, 1024>("cmp e64+addc e64, 64KB body, all CUs", 8, out, cyc, 256);
    runbig<1, 128>("cmp e64+addc e64, 8KB body, all CUs", 8, out, cyc, 256);
    runbig<0, 256>("cmp e32+addc e32, 8KB body, all CUs", 8, out, cyc, 256);
    runbig<3, 256>("v_add e32, 8KB body, all CUs", 8, out, cyc, 256);
    runbig<3, 16>("v_add e32, small body, all CUs", 8, out, cyc, 256);
    run<3>("8 v_add_u32 (baseline)", 8, out, cyc);
    run<9>("4 v_and + 4 v_or", 8, out, cyc);
    run<5>("8 v_bfe_u32", 8, out, cyc);
    run<6>("8 v_alignbit_b32", 8, out, cyc);
    run<7>("8 v_xad_u32", 8, out, cyc);
    run<10>("8 v_cmp_eq_u32_e32 vcc (+1 addc)", 9, out, cyc);
    run<4>("8 v_cmp_eq_u32_e64 -> 4 sgpr pairs", 8, out, cyc);
    run<11>("8 v_addc_co_u32_e32 (vcc chain)", 8, out, cyc);
    run<0>("4 x (cmp vcc + addc vcc) one chain", 8, out, cyc);
    run<1>("4 cmp e64 + 4 addc e64 (4 sgpr pairs)", 8, out, cyc);
    run<2>("4 v_xad + 4 v_alignbit", 8, out, cyc);
    run<8>("4 x (cmp + nop + cndmask) + 4 lshl_or", 12, out, cyc);
    return 0;
}
