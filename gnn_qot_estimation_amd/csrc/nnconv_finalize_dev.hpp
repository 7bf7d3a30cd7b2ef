// Second-stage sums of the H = 64 NNConv backward as a device body (used by roles.hip).
#pragma once
#include "common.hpp"

namespace qot {

constexpr int kAdjBlocksPerCu = 1;      // workgroups per CU of nnconv_adjoint_dw64 (= slabs per CU)

// One launch for both second-stage sums of the NNConv backward (H = 64): blocks [0, nb1) sum the per-workgroup
// weight-gradient slabs of nnconv_adjoint_dw64 (16 lanes per float4 stride over the slabs, fixed butterfly) and write the
// parameters' own layouts; the remaining blocks sum the grad-h partials (one wave per output).  Separately these were
// three dependent launches of ~5 us each behind kernels that had long finished.
// (256-thread virtual block vb of nb1 + ceil(hn / 4).)
__device__ __forceinline__ void nnconv_bwd_finalize64_body(const float* __restrict__ slabs, int nslabs, int64_t elems,
                                                           float* __restrict__ dst, int K, int nb1,
                                                           const float* __restrict__ hpart, int hblk, int hn, int KD,
                                                           float* __restrict__ gw1, float* __restrict__ gb1, int vb) {
    if (vb >= nb1) {
        const int t = (vb - nb1) * 4 + (threadIdx.x >> 6);
        if (t >= hn) return;
        const float s = wave_sum_partials(hpart, hblk, hn, t);
        if ((threadIdx.x & 63) == 0) { if (t < KD) gw1[t] = s; else gb1[t - KD] = s; }
        return;
    }
    const int64_t gt = vb * (int64_t)256 + threadIdx.x;
    const int64_t t = gt >> 4;             // float4 index
    const int sub = (int)(gt & 15);
    const bool live = t * 4 < elems;
    float4 acc = f4zero();
    if (live) {
        // eight slabs requested before the first is added (one load -> wait -> add per trip was 16 dependent round trips)
        int sidx = sub;
        for (; sidx + 7 * 16 < nslabs; sidx += 8 * 16) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ld4(slabs + (int64_t)(sidx + 16 * u) * elems + 4 * t);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = add4(acc, v[u]);
        }
        for (; sidx < nslabs; sidx += 16) acc = add4(acc, ld4(slabs + (int64_t)sidx * elems + 4 * t));
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        acc.x += __shfl_xor(acc.x, o); acc.y += __shfl_xor(acc.y, o);
        acc.z += __shfl_xor(acc.z, o); acc.w += __shfl_xor(acc.w, o);
    }
    if (!live || sub) return;
    const int64_t e = 4 * t;
    const int a0 = (int)(e & 63), o_ = (int)((e >> 6) & 63), kb = (int)(e >> 12);
    const float v[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int a = a0 + c;
        int64_t idx;
        if (kb < K) idx = ((int64_t)a * 64 + o_) * K + kb;
        else if (kb == K) idx = (int64_t)4096 * K + a * 64 + o_;
        else idx = (int64_t)4096 * (K + 1) + o_ * 64 + a;
        dst[idx] = v[c];
    }
}

}  // namespace qot
