// Helpers shared by the fused NNConv kernels (nnconv_mfma.hip: H = 64 tuned; nnconv_gen.hip: other widths):
// the fragment-grouped LDS operand tile, the 4-MFMA group, the LDS-only barrier, the XCD-aware persistent tile walk.
#pragma once
#include "common.hpp"

namespace qot {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// LDS operand tile, "fragment-grouped": k = 8*g + 2*r + hi  ->  float4 slot
//   At4[(2*g + hi)*32 + (i ^ (g & 7))] component r
// so the 4 A fragments a lane feeds to 4 consecutive MFMAs are ONE ds_read_b128, and the
// gather's 8 channels per lane (= one group g) are TWO ds_write_b128 (hi = 0 / 1).  The XOR
// spreads the 8 lanes of a destination (g & 7 = 0..7) over all 32 banks: both sides
// conflict-free.
__device__ __forceinline__ int at4_slot(int g, int hi, int i) { return (2 * g + hi) * 32 + (i ^ (g & 7)); }

// One 4-MFMA group g: A fragments = one float4 from the LDS tile, B fragments = one float4 of Wp.
__device__ __forceinline__ f32x16 mfma_group(const float4* __restrict__ At4, int g, int hi, int r31, float4 b,
                                             f32x16 c) {
    const float4 a = At4[at4_slot(g, hi, r31)];
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, c, 0, 0, 0);
    return c;
}

// Accumulators in AGPRs.  At <= 256 registers per lane and no other AGPR use the compiler selects the MFMAs' VGPR form
// (dst / srcC in the architectural file); one inline-asm operand with an "a" constraint makes it select the AGPR form for
// the whole kernel: the accumulator traffic of the matrix pipe then stays off the ports the LDS / global load returns and
// the VALU use (gemm.hip, tools/bench_gemm_k.py: +1..3 %).  Call once at the top of a kernel.  Only for kernels that fit
// 128 architectural registers besides the accumulators: under __launch_bounds__(256, 2) the AGPR form splits the 256
// registers 128 / 128, and the fused NNConv kernels (180-254 VGPRs) then spill into AGPR copies -- measured with the
// hint in nnconv_gen.hip: cfg4 28.6 -> 32.6 ms, cfg5 13.9 -> 15.7 ms; nnconv_mfma64 98.5 -> 101 us.  They keep the VGPR form.
__device__ __forceinline__ void mfma_acc_in_agprs() {
#ifndef QOT_NO_AGPR_HINT
    float az = 0.f;
    asm volatile("; accumulators in AGPRs" : "+a"(az));
#endif
}

// relu(v * s + t): BatchNorm(+ReLU) of the previous layer applied while a GEMM operand is loaded (gemm.hip, gemm256.hip)
template <bool AFFINE>
__device__ __forceinline__ float4 affine_relu4(float4 v, float4 s, float4 t) {
    if (!AFFINE) return v;
    return make_float4(fmaxf(fmaf(v.x, s.x, t.x), 0.f), fmaxf(fmaf(v.y, s.y, t.y), 0.f),
                       fmaxf(fmaf(v.z, s.z, t.z), 0.f), fmaxf(fmaf(v.w, s.w, t.w), 0.f));
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also emits vmcnt(0), i.e. it
// drains the epilogue's global stores (measured: ~30 % of the tile time went there).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// XCD-aware persistent tile walk.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8
// labels the XCD group -- speed only, never correctness), and each XCD has a private L2.  A graph's
// rows are gathered by the 3-4 consecutive tiles that hold its destinations, so every XCD group
// walks its own contiguous eighth of the tiles: the rows a tile gathers are then already in that
// XCD's L2 from the neighbouring tile.  Returns the tile of iteration `it` (or -1 when done).
__device__ __forceinline__ int64_t xcd_tile(int64_t it, int64_t ntiles) {
    const int nx = 8;
    if ((int)gridDim.x % nx != 0) {                        // small grids: plain strided walk
        const int64_t t = (int64_t)blockIdx.x + it * gridDim.x;
        return t < ntiles ? t : -1;
    }
    const int xcd = blockIdx.x % nx, slot = blockIdx.x / nx, per_x = gridDim.x / nx;
    const int64_t chunk = (ntiles + nx - 1) / nx;
    const int64_t local = slot + it * per_x;
    const int64_t t = xcd * chunk + local;
    return (local < chunk && t < ntiles) ? t : -1;
}

}  // namespace qot

static inline int num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

