// C[KT, 64] = A[N, KT]^T @ G[N, 64]   (fp32, reduction over the N rows -- "split-K" over nodes)
//
// This is the weight-gradient GEMM of NNConv (gWcat = A^T g; reference site: autograd of
// [PyG-ext] NNConv's nn.2 / root weights, topological_training/models.py:23,57,115) and of
// every node-level Linear.  The library GEMM runs this shape (tiny output, K = 10^5) at
// 13-28 TFLOP/s; here both MFMA operands are loaded from global memory directly in fragment
// order (lane l reads A[row 2s + (l>>5)][32*kb + (l&31)]: two 128-B row segments per
// wave-instruction), no LDS, every workgroup owns a contiguous slice of rows and keeps the
// whole [KT, 64] partial product in accumulators (8 waves x 5 tiles of 32x32).  Partials go
// to a slab per workgroup (plain stores) and a second pass sums the slabs in a fixed order:
// bitwise reproducible, no float atomics.
#include "common.hpp"

namespace qot {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTnBlocks = 256;     // one 512-thread workgroup per CU

// TPW = 32x32 output tiles per wave = KT/32 * 2 / 8
template <int TPW>
__global__ __launch_bounds__(512, 2) void gemm_tn_kernel(const float* __restrict__ A, int lda,
                                                         const float* __restrict__ G, int ldg, int64_t N,
                                                         float* __restrict__ slabs) {
    constexpr int KT = TPW * 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nh = wave & 1, kb0 = wave >> 1;           // k-blocks kb0, kb0+4, ...
    const int r31 = lane & 31, hi = lane >> 5;
    // rows of this workgroup: multiples of 8 so every 4-step group is full or absent
    const int64_t per = ((N + gridDim.x - 1) / gridDim.x + 15) & ~int64_t(15);
    const int64_t row0 = (int64_t)blockIdx.x * per;
    const int64_t row1 = (row0 + per < N) ? row0 + per : N;

    f32x16 c[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[t][r] = 0.f;

    constexpr int S = 8;               // k-steps (2 rows each) per prefetch group
    float a_cur[S][TPW], b_cur[S], a_nxt[S][TPW], b_nxt[S];
    auto load = [&](int64_t rbase, float (&a)[S][TPW], float (&b)[S]) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            // loads are unconditional (row clamped into the slice) and masked afterwards: a
            // "load or zero" select makes hipcc branch around every load and drain vmcnt each time
            const int64_t row = rbase + 2 * s + hi;
            const bool ok = row < row1;
            const int64_t rc = ok ? row : row1 - 1;
            const float* ar = A + rc * lda + r31;
            const float bv = G[rc * ldg + nh * 32 + r31];
            b[s] = ok ? bv : 0.f;
#pragma unroll
            for (int t = 0; t < TPW; ++t) a[s][t] = ar[(kb0 + 4 * t) * 32];
        }
    };
    if (row0 < row1) load(row0, a_cur, b_cur);
    for (int64_t rb = row0; rb < row1; rb += 2 * S) {
        if (rb + 2 * S < row1) load(rb + 2 * S, a_nxt, b_nxt);
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int t = 0; t < TPW; ++t)
                c[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s][t], b_cur[s], c[t], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            b_cur[s] = b_nxt[s];
#pragma unroll
            for (int t = 0; t < TPW; ++t) a_cur[s][t] = a_nxt[s][t];
        }
    }
    // slab[blk][k][n]
    float* slab = slabs + (int64_t)blockIdx.x * KT * 64;
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
            slab[((kb0 + 4 * t) * 32 + row) * 64 + nh * 32 + r31] = c[t][r];
        }
}

// Sums the slabs of group blockIdx.y (per_group consecutive slabs, slab_stride floats apart)
// into dst + blockIdx.y*dst_stride, 8 loads in flight per thread.  Two launches:
// 256 slabs -> 16 partials (written over the head of each group's first slab) -> C.
// Fixed summation order.
__global__ void slab_reduce_kernel(const float* __restrict__ slabs, int nslabs, int per_group, int64_t elems,
                                   int64_t slab_stride, float* __restrict__ dst, int64_t dst_stride) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;   // float4 index
    if (t * 4 >= elems) return;
    const int s0 = blockIdx.y * per_group;
    int s1 = s0 + per_group;
    if (s1 > nslabs) s1 = nslabs;
    float4 acc = f4zero();
    int s = s0;
    for (; s + 8 <= s1; s += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ld4(slabs + (int64_t)(s + u) * slab_stride + 4 * t);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = add4(acc, v[u]);
    }
    for (; s < s1; ++s) acc = add4(acc, ld4(slabs + (int64_t)s * slab_stride + 4 * t));
    st4(dst + blockIdx.y * dst_stride + 4 * t, acc);
}

}  // namespace qot

using namespace qot;

extern "C" size_t qot_gemm_tn_workspace_floats(int KT) {
    if (KT <= 0) return 0;
    return (size_t)kTnBlocks * (size_t)KT * 64;
}

extern "C" int qot_gemm_tn(const float* A, int lda, const float* G, int ldg, int64_t N, int KT, float* C,
                           float* workspace, qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N < 0 || KT <= 0) return QOT_ERR_BADARG;
    if (KT % 128 != 0 || KT > 1280) return QOT_ERR_UNSUPPORTED;
    if (!C || !workspace || (N > 0 && (!A || !G))) return QOT_ERR_BADARG;
    int blocks = kTnBlocks;
    if (N < (int64_t)blocks * 16) blocks = (int)((N + 15) / 16 > 0 ? (N + 15) / 16 : 1);
    switch (KT / 128) {
#define QOT_TN_CASE(T) case T: gemm_tn_kernel<T><<<blocks, 512, 0, stream>>>(A, lda, G, ldg, N, workspace); break;
        QOT_TN_CASE(1) QOT_TN_CASE(2) QOT_TN_CASE(3) QOT_TN_CASE(4) QOT_TN_CASE(5)
        QOT_TN_CASE(6) QOT_TN_CASE(7) QOT_TN_CASE(8) QOT_TN_CASE(9) QOT_TN_CASE(10)
#undef QOT_TN_CASE
        default: return QOT_ERR_UNSUPPORTED;
    }
    QOT_LAUNCH_CHECK();
    const int64_t elems = (int64_t)KT * 64;
    const int per_group = 16;
    const int groups = (blocks + per_group - 1) / per_group;
    // level 1: each group's sum overwrites the group's first slab (a group only reads itself)
    slab_reduce_kernel<<<dim3(grid_for(elems / 4, 256), groups), 256, 0, stream>>>(
        workspace, blocks, per_group, elems, elems, workspace, (int64_t)per_group * elems);
    QOT_LAUNCH_CHECK();
    // level 2: the `groups` partials sit per_group slabs apart
    slab_reduce_kernel<<<dim3(grid_for(elems / 4, 256), 1), 256, 0, stream>>>(
        workspace, groups, groups, elems, (int64_t)per_group * elems, C, 0);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}
