// First-layer projection of LightpathGNN: z = x W^T with a handful of input features (lightpath_training/models.py:13,30:
// GATConv(in_channels = 5, ...).lin; F = len(NODE_FEATURES), lightpath_training/dataset.py).  A [N, F] x [F, 4C] product
// is pure bandwidth (write 4 N C floats); the library ran it as a padded 16 x 32 x 512 GEMM: 0.95 ms at cfg3 (N = 707 k,
// 4C = 512) against 0.3 ms of HBM time, and its weight gradient g^T x as another 0.31 ms.
#include "common.hpp"

namespace qot {

constexpr int kSkinnyMaxF = 8;

// out[n, c] = sum_f x[n, f] w[c, f] (+ 0): thread = (float4 column group, row slot); rows walked grid-stride.
// LOGITS: C = heads * 128; the 32 lanes that hold a head's 128 columns of a row also form GATConv's attention logits
// a_src[n, h] = <out[n, h, :], att_src[h, :]>, a_dst likewise (as gemm_nt_kernel<., LOGITS>).
template <bool LOGITS>
__global__ __launch_bounds__(256) void skinny_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         float* __restrict__ out, int64_t N, int F, int C,
                                                         const float* __restrict__ att_src, const float* __restrict__ att_dst,
                                                         float* __restrict__ a_src, float* __restrict__ a_dst) {
    const int C4 = C / 4;
    const int cg = threadIdx.x % C4, slot = threadIdx.x / C4, nslot = 256 / C4;
    float wr[4][kSkinnyMaxF];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int f = 0; f < kSkinnyMaxF; ++f) wr[c][f] = (f < F) ? w[(int64_t)(4 * cg + c) * F + f] : 0.f;
    if (slot >= nslot) return;
    float4 as4 = f4zero(), ad4 = f4zero();
    if (LOGITS) { as4 = ld4(att_src + 4 * cg); ad4 = ld4(att_dst + 4 * cg); }
    for (int64_t n = (int64_t)blockIdx.x * nslot + slot; n < N; n += (int64_t)gridDim.x * nslot) {
        float xv[kSkinnyMaxF];
#pragma unroll
        for (int f = 0; f < kSkinnyMaxF; ++f) xv[f] = (f < F) ? x[n * F + f] : 0.f;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < kSkinnyMaxF; ++f)
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = fmaf(xv[f], wr[c][f], o[c]);
        st4(out + n * C + 4 * cg, make_float4(o[0], o[1], o[2], o[3]));
        if (LOGITS) {                                  // (C4 a multiple of 32: a head's lanes are an aligned half wave)
            const float4 ov = make_float4(o[0], o[1], o[2], o[3]);
            const float ps = group_sum<32>(dot4(ov, as4)), pd = group_sum<32>(dot4(ov, ad4));
            if ((cg & 31) == 0) { a_src[n * (C4 / 32) + cg / 32] = ps; a_dst[n * (C4 / 32) + cg / 32] = pd; }
        }
    }
}

// partials[blk][c * F + f] = sum over the block's rows of g[n, c] x[n, f]; rows of a block: a contiguous chunk
__global__ __launch_bounds__(256) void skinny_dw_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                        float* __restrict__ partials, int64_t N, int F, int C) {
    extern __shared__ float red[];                     // [nslot][C * F]
    const int C4 = C / 4;
    const int cg = threadIdx.x % C4, slot = threadIdx.x / C4, nslot = 256 / C4;
    const int64_t chunk = (N + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * chunk, r1 = (r0 + chunk < N) ? r0 + chunk : N;
    float acc[4][kSkinnyMaxF];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int f = 0; f < kSkinnyMaxF; ++f) acc[c][f] = 0.f;
    if (slot < nslot) {
        for (int64_t n = r0 + slot; n < r1; n += nslot) {
            const float4 gv = ld4(g + n * C + 4 * cg);
            const float gg[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int f = 0; f < kSkinnyMaxF; ++f) {
                const float xf = (f < F) ? x[n * F + f] : 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c][f] = fmaf(gg[c], xf, acc[c][f]);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int f = 0; f < kSkinnyMaxF; ++f)
                if (f < F) red[(int64_t)slot * C * F + (4 * cg + c) * F + f] = acc[c][f];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < C * F; e += 256) {
        float s = red[e];
        for (int k = 1; k < nslot; ++k) s += red[(int64_t)k * C * F + e];
        partials[(int64_t)blockIdx.x * C * F + e] = s;
    }
}

}  // namespace qot

using namespace qot;

// out[N, C] = x[N, F] . w[C, F]^T;  F <= 8, C multiple of 4, 4 <= C <= 1024
extern "C" int qot_skinny_linear_fwd(const float* x, const float* w, float* out, int64_t N, int F, int C,
                                     qot_stream_t stream) {
    if (N < 0 || F <= 0 || C <= 0) return QOT_ERR_BADARG;
    if (F > kSkinnyMaxF || (C & 3) || C > 1024) return QOT_ERR_UNSUPPORTED;
    if (N == 0) return QOT_OK;
    if (!x || !w || !out) return QOT_ERR_BADARG;
    const int nslot = 256 / (C / 4);
    int64_t blocks = (N + nslot - 1) / nslot;
    if (blocks > 8192) blocks = 8192;
    skinny_fwd_kernel<false><<<(int)blocks, 256, 0, (hipStream_t)stream>>>(x, w, out, N, F, C, nullptr, nullptr, nullptr, nullptr);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

// ... with GATConv's attention logits: C = heads * 128 (heads = 1 or 2 per 256-thread row slot: C in {128, 256, 512, 1024}),
// att_src / att_dst [C], a_src / a_dst [N, C / 128]
extern "C" int qot_skinny_linear_fwd_logits(const float* x, const float* w, float* out, int64_t N, int F, int C,
                                            const float* att_src, const float* att_dst, float* a_src, float* a_dst,
                                            qot_stream_t stream) {
    if (N < 0 || F <= 0 || C <= 0) return QOT_ERR_BADARG;
    if (F > kSkinnyMaxF || (C % 128) || C > 1024 || (256 % (C / 4))) return QOT_ERR_UNSUPPORTED;
    if (N == 0) return QOT_OK;
    if (!x || !w || !out || !att_src || !att_dst || !a_src || !a_dst) return QOT_ERR_BADARG;
    if (((uintptr_t)att_src & 15) || ((uintptr_t)att_dst & 15)) return QOT_ERR_UNSUPPORTED;
    const int nslot = 256 / (C / 4);
    int64_t blocks = (N + nslot - 1) / nslot;
    if (blocks > 8192) blocks = 8192;
    skinny_fwd_kernel<true><<<(int)blocks, 256, 0, (hipStream_t)stream>>>(x, w, out, N, F, C, att_src, att_dst, a_src, a_dst);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_skinny_linear_dw_blocks(int64_t N) {
    int64_t b = (N + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

// partials[qot_skinny_linear_dw_blocks(N)][C * F]: per-workgroup sums of g[n, c] x[n, f] (the caller sums the rows in order:
// QOT_ROLE_SUM_ROWS) -- the weight gradient g^T x of the projection above
extern "C" int qot_skinny_linear_dw(const float* g, const float* x, float* partials, int64_t N, int F, int C,
                                    qot_stream_t stream) {
    if (N <= 0 || F <= 0 || C <= 0) return QOT_ERR_BADARG;
    if (F > kSkinnyMaxF || (C & 3) || C > 1024) return QOT_ERR_UNSUPPORTED;
    if (!g || !x || !partials) return QOT_ERR_BADARG;
    const int nslot = 256 / (C / 4);
    const size_t lds = (size_t)nslot * C * F * sizeof(float);
    if (lds > 64 * 1024) return QOT_ERR_UNSUPPORTED;
    skinny_dw_kernel<<<qot_skinny_linear_dw_blocks(N), 256, lds, (hipStream_t)stream>>>(g, x, partials, N, F, C);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}
