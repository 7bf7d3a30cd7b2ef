// Shared device helpers for libqot_gnn (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qot_gnn.h"

#define QOT_LAUNCH_CHECK()                         \
    do {                                           \
        hipError_t _e = hipGetLastError();         \
        if (_e != hipSuccess) return (int)_e;      \
    } while (0)

#define QOT_HIP(call)                              \
    do {                                           \
        hipError_t _e = (call);                    \
        if (_e != hipSuccess) return (int)_e;      \
    } while (0)

namespace qot {

constexpr int kWave = 64;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float dot4(float4 a, float4 b) {
    return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}
__device__ __forceinline__ float4 fma4(float s, float4 a, float4 c) {
    return make_float4(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y), fmaf(s, a.z, c.z), fmaf(s, a.w, c.w));
}
__device__ __forceinline__ float4 scale4(float s, float4 a) {
    return make_float4(s * a.x, s * a.y, s * a.z, s * a.w);
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) {
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 sub4(float4 a, float4 b) {
    return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
}

// Sum over an aligned group of G consecutive lanes (G power of two <= 64).  Only lanes of
// the same group exchange data, so a wave may hold groups with different trip counts as
// long as each group is uniformly active.
// lane permutation inside a row of 16 lanes on the DPP path of the VALU (no LDS crossbar, no s_waitcnt)
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Value of lane SRC (compile-time, 0..7) of every aligned 8-lane group, in all lanes of the group: two DPP moves
// (`row_newbcast` of lane SRC into the lower half of the row -- bank mask 0x3 -- and of lane 8 + SRC into the upper
// half -- bank mask 0xC -- of the same register), no select, no ds_bpermute_b32 round trip.  `upper` (= (lane & 4) != 0)
// is no longer needed and kept for the call sites.
template <int SRC>
__device__ __forceinline__ int group8_bcast(int v, bool /*upper*/ = false) {
    const int lo = __builtin_amdgcn_mov_dpp(v, 0x150 + SRC, 0xF, 0x3, false);     // upper half: don't care yet
    return __builtin_amdgcn_update_dpp(lo, v, 0x150 + 8 + SRC, 0xF, 0xC, false);
}
template <int SRC>
__device__ __forceinline__ float group8_bcast(float v, bool /*upper*/ = false) {
    return __builtin_bit_cast(float, group8_bcast<SRC>(__builtin_bit_cast(int, v)));
}

// Value of lane SRC (compile-time, 0..15) of every DPP row of 16 lanes, in all lanes of the row: ONE VALU move
// (`row_newbcast:SRC`, gfx90a and later).  Written as __shfl(v, src, 16) with a runtime lane this is a
// ds_bpermute_b32 + s_waitcnt lgkmcnt round trip through the LDS crossbar.
template <int SRC>
__device__ __forceinline__ float row16_bcast(float v) { return dpp_move<0x150 + SRC>(v); }
template <int SRC>
__device__ __forceinline__ int row16_bcast(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0x150 + SRC, 0xF, 0xF, true);
}

// Sum over an aligned group of G consecutive lanes, result in every lane of the group.  Groups of up to 16
// lanes stay inside a DPP row: quad_perm [1,0,3,2] and [2,3,0,1] pair lanes inside a quad, row_half_mirror
// pairs the quads of an 8-lane half, row_mirror the two halves -- each step adds two partial sums of
// disjoint lane sets, and a + b == b + a makes the paired lanes agree bit for bit.  (Written with
// __shfl_xor every step was a ds_bpermute_b32 + s_waitcnt lgkmcnt: four LDS round trips per edge in the
// TransformerConv kernels.)  Wider groups keep the shuffle butterfly for the steps that cross rows.
template <int G>
__device__ __forceinline__ float group_sum(float v) {
    if constexpr (G > 16) {
#pragma unroll
        for (int o = G / 2; o >= 16; o >>= 1) v += __shfl_xor(v, o);
    }
    if constexpr (G >= 16) v += dpp_move<0x140>(v);      // row_mirror: lane i <-> 15 - i
    if constexpr (G >= 8) v += dpp_move<0x141>(v);       // row_half_mirror: i <-> 7 - i
    if constexpr (G >= 4) v += dpp_move<0x4E>(v);        // quad_perm [2,3,0,1]
    if constexpr (G >= 2) v += dpp_move<0xB1>(v);        // quad_perm [1,0,3,2]
    return v;
}

__host__ __device__ constexpr bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// counter-based dropout RNG: one 32-bit draw per element
__device__ __forceinline__ uint32_t rng32(uint64_t seed, uint64_t step, uint64_t idx) {
    uint64_t z = seed ^ (step * 0x9E3779B97F4A7C15ull) ^ (idx + 0xD1B54A32D192ED03ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}

// out[t] = sum_b partials[b*n + t]: one wave per output element, lanes stride over the blocks
// (coalescing does not matter here -- latency does: a serial loop over 512 partials costs ~120 us).
// Fixed order: lane-strided partial sums, then a fixed shuffle tree -> bitwise reproducible.
__device__ __forceinline__ float wave_sum_partials(const float* __restrict__ partials, int nblk, int n, int t) {
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int b = lane; b < nblk; b += 64) s += partials[(int64_t)b * n + t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    return s;
}

// leaky_relu + dropout epilogue shared by qot_act_* and the conv kernels that fuse it.
// One 64-bit hash per float4 of the [N, H] output (index = flat element index / 4), four 16-bit
// draws out of it; keep = draw >= thr16.  qot_act_bwd regenerates the same mask from
// (seed, *step, element), so forward kernels only have to use the same indexing.
struct ActParams {
    int enabled;            // 0: identity epilogue
    float slope;            // leaky_relu negative slope
    uint32_t thr16;         // round(p * 65536), 0 = no dropout
    float keep_scale;       // 1 / (1 - p)
    uint64_t seed;
    const int64_t* step;    // device-side step counter (graph-replay safe), may be NULL when thr16 == 0
};

// r04: three 32 x 32 -> 64 multiplies with hi ^ lo folds instead of SplitMix64's three 64 x 64 multiplies (nine quarter-rate
// 32-bit multiplies on this part; the hash was 15 % of the TransformerConv forward).  The key (seed, step) is wave-uniform in
// every caller.  Checked on 2 M consecutive indices for several keys (keep rates at p = 0.1 / 0.5 / 0.9 of all four 16-bit
// draws, per-bit bias, correlation between the draws, between neighbouring indices at strides 1 .. 25 600 and between
// consecutive steps): every deviation <= 2.2e-3, the sampling noise of the check (SplitMix64 reads the same).
__device__ __forceinline__ uint64_t act_hash64(uint64_t seed, uint64_t step, uint64_t idx4) {
    const uint64_t k = seed ^ (step * 0x9E3779B97F4A7C15ull);
    const uint32_t k0 = (uint32_t)k, k1 = (uint32_t)(k >> 32);
    const uint64_t p = (uint64_t)((uint32_t)idx4 ^ k0) * 0x9E3779B1u;
    const uint32_t a = (uint32_t)(p >> 32) ^ (uint32_t)p ^ (uint32_t)(idx4 >> 32) ^ k1;
    const uint64_t q = (uint64_t)a * 0x85EBCA77u;
    const uint64_t r = (uint64_t)(a ^ k0 ^ 0x68E31DA4u) * 0xC2B2AE3Du;
    const uint32_t w0 = (uint32_t)(q >> 32) ^ (uint32_t)q, w1 = (uint32_t)(r >> 32) ^ (uint32_t)r;
    return ((uint64_t)w1 << 32) | w0;
}

__device__ __forceinline__ float act_apply1(float v, const ActParams& a, uint64_t flat) {
    if (!a.enabled) return v;
    float y = v > 0.f ? v : a.slope * v;
    if (a.thr16) {
        const uint64_t z = act_hash64(a.seed, (uint64_t)a.step[0], flat >> 2);
        const bool keep = ((uint32_t)(z >> (16 * (flat & 3))) & 0xFFFFu) >= a.thr16;
        y = keep ? y * a.keep_scale : 0.f;
    }
    return y;
}

__device__ __forceinline__ float4 act_apply4(float4 v, const ActParams& a, uint64_t idx4) {
    if (!a.enabled) return v;
    float r[4] = {v.x, v.y, v.z, v.w};
    uint64_t z = 0;
    if (a.thr16) z = act_hash64(a.seed, (uint64_t)a.step[0], idx4);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float y = r[c] > 0.f ? r[c] : a.slope * r[c];
        if (a.thr16) y = (((uint32_t)(z >> (16 * c)) & 0xFFFFu) >= a.thr16) ? y * a.keep_scale : 0.f;
        r[c] = y;
    }
    return make_float4(r[0], r[1], r[2], r[3]);
}

// as act_apply4 with the step counter's value already in a register (kernels that apply the epilogue many times per
// thread read the counter once: a load per use is a global round trip in front of every store)
__device__ __forceinline__ float4 act_apply4s(float4 v, const ActParams& a, uint64_t step, uint64_t idx4) {
    if (!a.enabled) return v;
    float r[4] = {v.x, v.y, v.z, v.w};
    uint64_t z = 0;
    if (a.thr16) z = act_hash64(a.seed, step, idx4);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float y = r[c] > 0.f ? r[c] : a.slope * r[c];
        if (a.thr16) y = (((uint32_t)(z >> (16 * c)) & 0xFFFFu) >= a.thr16) ? y * a.keep_scale : 0.f;
        r[c] = y;
    }
    return make_float4(r[0], r[1], r[2], r[3]);
}

inline ActParams make_act(int act, float slope, float p, uint64_t seed, const int64_t* step) {
    ActParams a;
    a.enabled = act; a.slope = slope; a.thr16 = 0; a.keep_scale = 1.0f; a.seed = seed; a.step = step;
    if (act && p > 0.f && step) {
        uint32_t thr = (uint32_t)(p * 65536.0f + 0.5f);
        a.thr16 = thr > 65535u ? 65535u : thr;
        a.keep_scale = 1.0f / (1.0f - p);
    }
    return a;
}

// XCD-aware block remap (bijective for any grid size).  Blocks are dealt round-robin over the 8
// XCDs, each with a private L2; consecutive row blocks share gathered source rows (same graph),
// so give every XCD a CONTIGUOUS range of row blocks.  Speed only -- any placement is correct.
__device__ __forceinline__ int xcd_block(int orig, int nwg) {
    const int nx = 8;
    const int q = nwg / nx, r = nwg % nx;
    const int xcd = orig % nx, idx = orig / nx;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

inline int grid_for(int64_t work, int per_block) { return (int)((work + per_block - 1) / per_block); }

// Dynamic LDS above 64 KB has to be allowed per kernel AND per device (hipFuncAttributeMaxDynamicSharedMemorySize is a
// per-device attribute).  `allowed` is the caller's static table, one entry per device ordinal, zero-initialised (= the
// 64 KB default).  The first call for a size happens outside any stream capture (every captured step has been through
// an eager step).  Returns QOT_OK, QOT_ERR_UNSUPPORTED when the runtime refuses the size (the launch would otherwise
// fail later with a less telling error), or the HIP error of hipGetDevice.
constexpr int kMaxDevices = 64;
inline int ensure_dyn_lds(const void* func, size_t bytes, size_t (&allowed)[kMaxDevices]) {
    if (bytes <= 64 * 1024) return QOT_OK;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= kMaxDevices) return QOT_ERR_UNSUPPORTED;
    if (bytes <= allowed[dev]) return QOT_OK;
    e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return QOT_ERR_UNSUPPORTED;
    }
    allowed[dev] = bytes;
    return QOT_OK;
}

}  // namespace qot

#define QOT_DISPATCH_H(H, ...)                  \
    switch (H) {                                \
        case 16:  { constexpr int kH = 16;  __VA_ARGS__; } break;  \
        case 32:  { constexpr int kH = 32;  __VA_ARGS__; } break;  \
        case 64:  { constexpr int kH = 64;  __VA_ARGS__; } break;  \
        case 128: { constexpr int kH = 128; __VA_ARGS__; } break;  \
        case 256: { constexpr int kH = 256; __VA_ARGS__; } break;  \
        default: return QOT_ERR_UNSUPPORTED;    \
    }

#define QOT_DISPATCH_D(D, ...)                  \
    switch (D) {                                \
        case 1: { constexpr int kD = 1; __VA_ARGS__; } break;  \
        case 2: { constexpr int kD = 2; __VA_ARGS__; } break;  \
        case 3: { constexpr int kD = 3; __VA_ARGS__; } break;  \
        case 4: { constexpr int kD = 4; __VA_ARGS__; } break;  \
        case 5: { constexpr int kD = 5; __VA_ARGS__; } break;  \
        case 6: { constexpr int kD = 6; __VA_ARGS__; } break;  \
        case 7: { constexpr int kD = 7; __VA_ARGS__; } break;  \
        case 8: { constexpr int kD = 8; __VA_ARGS__; } break;  \
        default: return QOT_ERR_UNSUPPORTED;    \
    }
