// NNConv (aggr = "mean"), factorised -- aggregation side.
// Replaces [PyG-ext] NNConv.message/aggregate reached from
// topological_training/models.py:57.  PyG materialises
// theta_e = nn(ea_e) as [E, H*H] (6.7 GB at B=1024,n=100,e=400,H=64) and runs E tiny GEMVs.
// Here the edge MLP's second layer is linear, so with h_e = relu(W1 ea_e + b1) in R^K:
//     sum_e x_j^T theta_e = sum_k (sum_e h_e[k] x_j)^T W2_k + (sum_e x_j)^T B2
// i.e. aggregate K+1 weighted copies of the 64-wide input rows FIRST (this kernel), then
// one dense GEMM [N,(K+2)H] x [(K+2)H,H] on the matrix cores (root weight folded in as the
// (K+2)-th block).  Gather traffic is H floats per edge instead of H*H.
//
// The same kernel run over the CSC with the scale taken at the gathered end is the exact
// adjoint (grad_x = U @ Wcat^T), so forward and backward-data share one code path.
#include "common.hpp"

namespace qot {

template <int H, int D, bool TRANSPOSE>
__global__ __launch_bounds__(256) void nnconv_agg_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ ea,
    const float* __restrict__ w1, const float* __restrict__ b1,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const int32_t* __restrict__ idx1, const int32_t* __restrict__ idx2,
    const float* __restrict__ invdeg, float* __restrict__ A, int64_t N) {
    constexpr int K = 2 * D;
    constexpr int TPR = H / 4;
    constexpr int RPB = 256 / TPR;
    constexpr int LDA = (K + 2) * H;
    const int sub = threadIdx.x % TPR;
    const int64_t i = (int64_t)xcd_block(blockIdx.x, gridDim.x) * RPB + threadIdx.x / TPR;
    if (i >= N) return;
    const int c0 = 4 * sub;

    float w[K][D], b[K];
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
        b[kk] = b1[kk];
#pragma unroll
        for (int d = 0; d < D; ++d) w[kk][d] = w1[kk * D + d];
    }
    float4 acc[K + 1];
#pragma unroll
    for (int kk = 0; kk <= K; ++kk) acc[kk] = f4zero();

    const int beg = rowptr[i], end = rowptr[i + 1];
    for (int p = beg; p < end; ++p) {
        const int64_t j = col[p];
        const int64_t e = TRANSPOSE ? (int64_t)idx2[idx1[p]] : (int64_t)idx1[p];
        float4 xj = ld4(x + j * ldx + c0);
        if (TRANSPOSE) xj = scale4(invdeg[j], xj);
        float ee[D];
#pragma unroll
        for (int d = 0; d < D; ++d) ee[d] = ea[e * D + d];
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            float h = b[kk];
#pragma unroll
            for (int d = 0; d < D; ++d) h = fmaf(w[kk][d], ee[d], h);
            h = fmaxf(h, 0.f);
            acc[kk] = fma4(h, xj, acc[kk]);
        }
        acc[K] = add4(acc[K], xj);
    }
    const float sc = TRANSPOSE ? 1.0f : invdeg[i];
    float* Ai = A + i * LDA + c0;
#pragma unroll
    for (int kk = 0; kk <= K; ++kk) st4(Ai + kk * H, scale4(sc, acc[kk]));
    st4(Ai + (K + 1) * H, ld4(x + i * ldx + c0));
}

// grad wrt the first edge-MLP layer.  GA_i[k,:] = (g_i Wcat_k^T) is held in registers by
// the destination's lane group; per in-edge the K dot products <GA_i[k,:], x_j> give
// dL/dh_e[k] (times invdeg_i), masked by relu'(pre_k) and accumulated against ea_e.
template <int H, int D>
__global__ __launch_bounds__(256) void nnconv_bwd_edge_kernel(
    const float* __restrict__ GA, int ldga, const float* __restrict__ x, int ldx,
    const float* __restrict__ ea, const float* __restrict__ w1, const float* __restrict__ b1,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const int32_t* __restrict__ eid, const float* __restrict__ invdeg, float* __restrict__ gw1,
    float* __restrict__ gb1, int64_t N) {
    constexpr int K = 2 * D;
    constexpr int TPR = H / 4;
    constexpr int RPB = 256 / TPR;
    constexpr int NOUT = K * D + K;
    __shared__ float red[NOUT];
    if (threadIdx.x < NOUT) red[threadIdx.x] = 0.f;
    __syncthreads();

    const int sub = threadIdx.x % TPR;
    const int64_t i = (int64_t)xcd_block(blockIdx.x, gridDim.x) * RPB + threadIdx.x / TPR;
    const int c0 = 4 * sub;
    if (i < N) {
        float w[K][D], b[K];
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            b[kk] = b1[kk];
#pragma unroll
            for (int d = 0; d < D; ++d) w[kk][d] = w1[kk * D + d];
        }
        float4 ga[K];
#pragma unroll
        for (int kk = 0; kk < K; ++kk) ga[kk] = ld4(GA + i * ldga + kk * H + c0);
        float aw[K][D], ab[K];
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            ab[kk] = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) aw[kk][d] = 0.f;
        }
        const float sc = invdeg[i];
        const int beg = rowptr[i], end = rowptr[i + 1];
        for (int p = beg; p < end; ++p) {
            const int64_t j = col[p];
            const int64_t e = eid[p];
            float4 xj = ld4(x + j * ldx + c0);
            float ee[D];
#pragma unroll
            for (int d = 0; d < D; ++d) ee[d] = ea[e * D + d];
            float dk[K];
#pragma unroll
            for (int kk = 0; kk < K; ++kk) dk[kk] = dot4(ga[kk], xj);
#pragma unroll
            for (int o = TPR / 2; o > 0; o >>= 1)
#pragma unroll
                for (int kk = 0; kk < K; ++kk) dk[kk] += __shfl_xor(dk[kk], o);
#pragma unroll
            for (int kk = 0; kk < K; ++kk) {
                float pre = b[kk];
#pragma unroll
                for (int d = 0; d < D; ++d) pre = fmaf(w[kk][d], ee[d], pre);
                const float gh = (pre > 0.f) ? dk[kk] * sc : 0.f;
                ab[kk] += gh;
#pragma unroll
                for (int d = 0; d < D; ++d) aw[kk][d] = fmaf(gh, ee[d], aw[kk][d]);
            }
        }
        if (sub == 0 && beg < end) {
#pragma unroll
            for (int kk = 0; kk < K; ++kk) {
#pragma unroll
                for (int d = 0; d < D; ++d) atomicAdd(&red[kk * D + d], aw[kk][d]);
                atomicAdd(&red[K * D + kk], ab[kk]);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < K * D) atomicAdd(&gw1[threadIdx.x], red[threadIdx.x]);
    else if (threadIdx.x < NOUT) atomicAdd(&gb1[threadIdx.x - K * D], red[threadIdx.x]);
}

}  // namespace qot

using namespace qot;

extern "C" int qot_nnconv_agg(const float* x, int ld_x, const float* edge_attr, const float* w1,
                              const float* b1, const int32_t* rowptr, const int32_t* col,
                              const int32_t* eid_or_pos, const int32_t* eid_of_pos, const float* invdeg,
                              int transpose, float* A, int64_t N, int H, int D, qot_stream_t stream) {
    if (N < 0 || !rowptr) return QOT_ERR_BADARG;
    if (N == 0) return QOT_OK;
    if (!x || !w1 || !b1 || !invdeg || !A || (ld_x & 3)) return QOT_ERR_BADARG;
    if (transpose && !eid_of_pos) return QOT_ERR_BADARG;
    QOT_DISPATCH_H(H, QOT_DISPATCH_D(D, {
        constexpr int RPB = 256 / (kH / 4);
        if (transpose)
            nnconv_agg_kernel<kH, kD, true><<<grid_for(N, RPB), 256, 0, (hipStream_t)stream>>>(
                x, ld_x, edge_attr, w1, b1, rowptr, col, eid_or_pos, eid_of_pos, invdeg, A, N);
        else
            nnconv_agg_kernel<kH, kD, false><<<grid_for(N, RPB), 256, 0, (hipStream_t)stream>>>(
                x, ld_x, edge_attr, w1, b1, rowptr, col, eid_or_pos, eid_of_pos, invdeg, A, N);
    }));
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_nnconv_bwd_edge(const float* GA, int ld_ga, const float* x, int ld_x,
                                   const float* edge_attr, const float* w1, const float* b1,
                                   const int32_t* rowptr, const int32_t* col, const int32_t* eid,
                                   const float* invdeg, float* gw1, float* gb1, int64_t N, int H, int D,
                                   qot_stream_t stream) {
    if (N < 0 || !rowptr) return QOT_ERR_BADARG;
    if (N == 0) return QOT_OK;
    if (!GA || !x || !w1 || !b1 || !invdeg || !gw1 || !gb1 || (ld_ga & 3) || (ld_x & 3))
        return QOT_ERR_BADARG;
    QOT_DISPATCH_H(H, QOT_DISPATCH_D(D, {
        constexpr int RPB = 256 / (kH / 4);
        nnconv_bwd_edge_kernel<kH, kD><<<grid_for(N, RPB), 256, 0, (hipStream_t)stream>>>(
            GA, ld_ga, x, ld_x, edge_attr, w1, b1, rowptr, col, eid, invdeg, gw1, gb1, N);
    }));
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}
