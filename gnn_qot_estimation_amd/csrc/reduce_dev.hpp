// Fixed-order row sums of per-workgroup partials, as a device body (roles.hip).
#pragma once
#include "common.hpp"

namespace qot {

constexpr int kSumCols = 16;            // output columns (floats, or float4s) per 256-thread workgroup
constexpr int kSumSubs = 256 / kSumCols;

// out[g * n + t] = sum over the rows b of group g of partials[b * n + t], groups of `per` consecutive rows (the last one
// may be short).  One workgroup = kSumCols consecutive columns x kSumSubs row sub-ranges of one group: every load is a
// contiguous 64-B (256-B with float4 columns) segment, eight independent loads in flight per thread, sub-ranges meet in
// LDS in index order -> bitwise reproducible.  `lds`: 256 float4.  (The one-wave-per-output form read each partial
// with a stride of n floats: 12.8 us for the 9 MB of the read-out's partials.)
__device__ __forceinline__ void sum_rows_body(const float* __restrict__ partials, float* __restrict__ out, int64_t nblk,
                                              int64_t n, int64_t per, int v4, int vb, float* __restrict__ lds) {
    const int c = threadIdx.x % kSumCols, sg = threadIdx.x / kSumCols;
    const int64_t cols = v4 ? n / 4 : n;
    const int64_t cblocks = (cols + kSumCols - 1) / kSumCols;
    const int64_t g = vb / cblocks;
    const int64_t t = (vb % cblocks) * kSumCols + c;
    const int64_t b0 = g * per;
    const int64_t cnt = (b0 + per <= nblk) ? per : nblk - b0;
    const int64_t chunk = (cnt + kSumSubs - 1) / kSumSubs;
    const int64_t p0 = b0 + sg * chunk, p1 = (sg * chunk + chunk < cnt) ? p0 + chunk : b0 + cnt;
    float4* red = reinterpret_cast<float4*>(lds);
    float4 acc = f4zero();
    if (t < cols) {
        int64_t p = p0;
        if (v4) {
            for (; p + 8 <= p1; p += 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = ld4(partials + (p + u) * n + 4 * t);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = add4(acc, v[u]);
            }
            for (; p < p1; ++p) acc = add4(acc, ld4(partials + p * n + 4 * t));
        } else {
            for (; p + 8 <= p1; p += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = partials[(p + u) * n + t];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc.x += v[u];
            }
            for (; p < p1; ++p) acc.x += partials[p * n + t];
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (sg == 0 && t < cols) {
        float4 s = red[c];
#pragma unroll
        for (int q = 1; q < kSumSubs; ++q) s = add4(s, red[q * kSumCols + c]);
        if (v4) st4(out + g * n + 4 * t, s);
        else out[g * n + t] = s.x;
    }
}

}  // namespace qot
