// GATConv (concat heads, self loops, LeakyReLU(0.2) logits) -- fused edge-softmax-aggregate.
// Replaces [PyG-ext] GATConv.edge_updater/propagate reached from
// lightpath_training/models.py:30.  The self-loop rewrite of the edge
// list is done once by qot_csr_build(gat_self_loops=1); logits are scalars per (edge, head)
// (a_src[j,h] + a_dst[i,h]) so the softmax needs no cross-lane traffic at all: every lane
// tracks the running (max, sum) of the head its channels belong to.
//
// Mapping: TPR = min(64, HC/4) lanes per destination, NV = HC/(4*TPR) float4 per lane;
// float4 number v of lane `sub` covers channels 4*(sub + TPR*v) .. +3, all in one head.
#include "common.hpp"

namespace qot {

template <int HEADS, int C>
struct GatCfg {
    static constexpr int HC = HEADS * C;
    static constexpr int TPR = (HC / 4 < 64) ? HC / 4 : 64;
    static constexpr int NV = HC / (4 * TPR);
    static constexpr int RPB = 256 / TPR;
    static constexpr int LPH = (C / 4 < TPR) ? C / 4 : TPR;  // lanes that share one head
};

template <int HEADS, int C>
__global__ __launch_bounds__(256) void gat_fwd_kernel(
    const float* __restrict__ z, const float* __restrict__ a_src, const float* __restrict__ a_dst,
    const float* __restrict__ bias, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ col, float* __restrict__ out, float* __restrict__ stats, int64_t N,
    float ns) {
    using G = GatCfg<HEADS, C>;
    const int sub = threadIdx.x % G::TPR;
    const int64_t i = (int64_t)xcd_block(blockIdx.x, gridDim.x) * G::RPB + threadIdx.x / G::TPR;
    if (i >= N) return;
    int hh[G::NV];
    float ad[G::NV], m[G::NV], l[G::NV];
    float4 acc[G::NV];
#pragma unroll
    for (int v = 0; v < G::NV; ++v) {
        hh[v] = (4 * (sub + G::TPR * v)) / C;
        ad[v] = a_dst[i * HEADS + hh[v]];
        m[v] = -INFINITY;
        l[v] = 0.f;
        acc[v] = f4zero();
    }
    const int beg = rowptr[i], end = rowptr[i + 1];
    for (int p = beg; p < end; ++p) {
        const int64_t j = col[p];
#pragma unroll
        for (int v = 0; v < G::NV; ++v) {
            float4 zj = ld4(z + j * G::HC + 4 * (sub + G::TPR * v));
            float raw = a_src[j * HEADS + hh[v]] + ad[v];
            float s = raw > 0.f ? raw : ns * raw;
            float mn = fmaxf(m[v], s);
            float sc = __expf(m[v] - mn);
            float pe = __expf(s - mn);
            l[v] = fmaf(l[v], sc, pe);
            acc[v] = fma4(pe, zj, scale4(sc, acc[v]));
            m[v] = mn;
        }
    }
#pragma unroll
    for (int v = 0; v < G::NV; ++v) {
        const int c = 4 * (sub + G::TPR * v);
        const float denom = l[v] + 1e-16f;
        st4(out + i * G::HC + c, add4(scale4(1.0f / denom, acc[v]), ld4(bias + c)));
        if (c % C == 0) {
            stats[(i * HEADS + hh[v]) * 2] = (beg < end) ? m[v] : 0.f;
            stats[(i * HEADS + hh[v]) * 2 + 1] = denom;
        }
    }
}

// destination pass of the backward: per (edge, head) alpha and dalpha = <g_i,h, z_j,h>,
// delta_{i,h} = sum_e alpha dalpha, grad_a_dst[i,h] = sum_e alpha (dalpha - delta) lrelu'(raw).
template <int HEADS, int C>
__global__ __launch_bounds__(256) void gat_bwd_dst_kernel(
    const float* __restrict__ g, const float* __restrict__ z, const float* __restrict__ a_src,
    const float* __restrict__ a_dst, const float* __restrict__ stats,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, float* __restrict__ gad,
    float* __restrict__ escr, float* __restrict__ delta, int64_t N, float ns) {
    using G = GatCfg<HEADS, C>;
    const int sub = threadIdx.x % G::TPR;
    const int64_t i = (int64_t)xcd_block(blockIdx.x, gridDim.x) * G::RPB + threadIdx.x / G::TPR;
    if (i >= N) return;
    int hh[G::NV];
    float ad[G::NV], m[G::NV], inv[G::NV], sada[G::NV], sal[G::NV], salk[G::NV];
    float4 gi[G::NV];
#pragma unroll
    for (int v = 0; v < G::NV; ++v) {
        const int c = 4 * (sub + G::TPR * v);
        hh[v] = c / C;
        ad[v] = a_dst[i * HEADS + hh[v]];
        m[v] = stats[(i * HEADS + hh[v]) * 2];
        inv[v] = 1.0f / stats[(i * HEADS + hh[v]) * 2 + 1];
        gi[v] = ld4(g + i * G::HC + c);
        sada[v] = sal[v] = salk[v] = 0.f;
    }
    const int beg = rowptr[i], end = rowptr[i + 1];
    for (int p = beg; p < end; ++p) {
        const int64_t j = col[p];
#pragma unroll
        for (int v = 0; v < G::NV; ++v) {
            const int c = 4 * (sub + G::TPR * v);
            float4 zj = ld4(z + j * G::HC + c);
            float da = group_sum<G::LPH>(dot4(gi[v], zj));
            float raw = a_src[j * HEADS + hh[v]] + ad[v];
            float s = raw > 0.f ? raw : ns * raw;
            float lk = raw > 0.f ? 1.0f : ns;
            float a = __expf(s - m[v]) * inv[v];
            sada[v] = fmaf(a, da, sada[v]);
            sal[v] = fmaf(a * lk, da, sal[v]);
            salk[v] = fmaf(a, lk, salk[v]);
            if (c % C == 0) {
                escr[((int64_t)p * HEADS + hh[v]) * 2] = a;
                escr[((int64_t)p * HEADS + hh[v]) * 2 + 1] = da;
            }
        }
    }
#pragma unroll
    for (int v = 0; v < G::NV; ++v) {
        const int c = 4 * (sub + G::TPR * v);
        if (c % C == 0) {
            delta[i * HEADS + hh[v]] = sada[v];
            gad[i * HEADS + hh[v]] = sal[v] - sada[v] * salk[v];
        }
    }
}

// source pass over the CSC: grad_z_j = sum_{e: j->i} alpha_e g_i ; grad_a_src[j,h] = sum_e ds_e.
template <int HEADS, int C>
__global__ __launch_bounds__(256) void gat_bwd_src_kernel(
    const float* __restrict__ g, const float* __restrict__ a_src, const float* __restrict__ a_dst,
    const float* __restrict__ escr, const float* __restrict__ delta,
    const int32_t* __restrict__ rowptr_t, const int32_t* __restrict__ col_t,
    const int32_t* __restrict__ pos_t, float* __restrict__ gz, float* __restrict__ gas, int64_t N,
    float ns) {
    using G = GatCfg<HEADS, C>;
    const int sub = threadIdx.x % G::TPR;
    const int64_t j = (int64_t)xcd_block(blockIdx.x, gridDim.x) * G::RPB + threadIdx.x / G::TPR;
    if (j >= N) return;
    int hh[G::NV];
    float as[G::NV], sds[G::NV];
    float4 acc[G::NV];
#pragma unroll
    for (int v = 0; v < G::NV; ++v) {
        hh[v] = (4 * (sub + G::TPR * v)) / C;
        as[v] = a_src[j * HEADS + hh[v]];
        sds[v] = 0.f;
        acc[v] = f4zero();
    }
    const int beg = rowptr_t[j], end = rowptr_t[j + 1];
    for (int t = beg; t < end; ++t) {
        const int64_t i = col_t[t];
        const int64_t p = pos_t[t];
#pragma unroll
        for (int v = 0; v < G::NV; ++v) {
            const float a = escr[(p * HEADS + hh[v]) * 2];
            const float da = escr[(p * HEADS + hh[v]) * 2 + 1];
            const float raw = as[v] + a_dst[i * HEADS + hh[v]];
            const float lk = raw > 0.f ? 1.0f : ns;
            sds[v] += a * (da - delta[i * HEADS + hh[v]]) * lk;
            acc[v] = fma4(a, ld4(g + i * G::HC + 4 * (sub + G::TPR * v)), acc[v]);
        }
    }
#pragma unroll
    for (int v = 0; v < G::NV; ++v) {
        const int c = 4 * (sub + G::TPR * v);
        st4(gz + j * G::HC + c, acc[v]);
        if (c % C == 0) gas[j * HEADS + hh[v]] = sds[v];
    }
}

}  // namespace qot

using namespace qot;

#define QOT_DISPATCH_GAT(heads, C, ...)                                    \
    if ((heads) != 4) return QOT_ERR_UNSUPPORTED;                          \
    switch (C) {                                                           \
        case 4:   { constexpr int kC = 4;   __VA_ARGS__; } break;          \
        case 8:   { constexpr int kC = 8;   __VA_ARGS__; } break;          \
        case 16:  { constexpr int kC = 16;  __VA_ARGS__; } break;          \
        case 32:  { constexpr int kC = 32;  __VA_ARGS__; } break;          \
        case 64:  { constexpr int kC = 64;  __VA_ARGS__; } break;          \
        case 128: { constexpr int kC = 128; __VA_ARGS__; } break;          \
        case 256: { constexpr int kC = 256; __VA_ARGS__; } break;          \
        default: return QOT_ERR_UNSUPPORTED;                               \
    }

extern "C" int qot_gat_fwd(const float* z, const float* a_src, const float* a_dst, const float* bias,
                           const int32_t* rowptr, const int32_t* col, float* out, float* stats,
                           int64_t N, int heads, int C, float neg_slope, qot_stream_t stream) {
    if (N < 0 || !rowptr) return QOT_ERR_BADARG;
    if (N == 0) return QOT_OK;
    if (!z || !a_src || !a_dst || !bias || !col || !out || !stats) return QOT_ERR_BADARG;
    QOT_DISPATCH_GAT(heads, C, {
        using G = GatCfg<4, kC>;
        gat_fwd_kernel<4, kC><<<grid_for(N, G::RPB), 256, 0, (hipStream_t)stream>>>(
            z, a_src, a_dst, bias, rowptr, col, out, stats, N, neg_slope);
    });
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_gat_bwd_dst(const float* grad_out, const float* z, const float* a_src,
                               const float* a_dst, const float* stats, const int32_t* rowptr,
                               const int32_t* col, float* grad_a_dst, float* escr, float* delta,
                               int64_t N, int heads, int C, float neg_slope, qot_stream_t stream) {
    if (N < 0 || !rowptr) return QOT_ERR_BADARG;
    if (N == 0) return QOT_OK;
    if (!grad_out || !z || !a_src || !a_dst || !stats || !col || !grad_a_dst || !escr || !delta)
        return QOT_ERR_BADARG;
    QOT_DISPATCH_GAT(heads, C, {
        using G = GatCfg<4, kC>;
        gat_bwd_dst_kernel<4, kC><<<grid_for(N, G::RPB), 256, 0, (hipStream_t)stream>>>(
            grad_out, z, a_src, a_dst, stats, rowptr, col, grad_a_dst, escr, delta, N, neg_slope);
    });
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_gat_bwd_src(const float* grad_out, const float* a_src, const float* a_dst,
                               const float* escr, const float* delta, const int32_t* rowptr_t,
                               const int32_t* col_t, const int32_t* pos_t, float* grad_z,
                               float* grad_a_src, int64_t N, int heads, int C, float neg_slope,
                               qot_stream_t stream) {
    if (N < 0 || !rowptr_t) return QOT_ERR_BADARG;
    if (N == 0) return QOT_OK;
    if (!grad_out || !a_src || !a_dst || !escr || !delta || !col_t || !pos_t || !grad_z || !grad_a_src)
        return QOT_ERR_BADARG;
    QOT_DISPATCH_GAT(heads, C, {
        using G = GatCfg<4, kC>;
        gat_bwd_src_kernel<4, kC><<<grid_for(N, G::RPB), 256, 0, (hipStream_t)stream>>>(
            grad_out, a_src, a_dst, escr, delta, rowptr_t, col_t, pos_t, grad_z, grad_a_src, N,
            neg_slope);
    });
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}
