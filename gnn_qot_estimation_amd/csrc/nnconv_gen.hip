// Width-generic fused NNConv (H in {16, 32, 128, 256}; H = 64 has its own tuned kernels in nnconv_mfma.hip).
//
// Reference site: topological_training/models.py:20-30,57 (edge MLP + [PyG-ext] NNConv(aggr="mean")); algebra in
// nnconv.hip: out_i = bias + A_i @ Wcat with A_i = [inv_i sum_e h_e[k] x_j (k < K) | inv_i sum_e x_j | x_i].
// No width materialises A ([N, (K+2)H]: 5.2 GB per layer at BASELINE configs[3]) in HBM any more:
//
//   nnconv_gen_kernel      forward, and (TRANSPOSE) the adjoint grad_x = U @ WcatT over the CSC.  A 32-row operand
//                          tile lives in LDS only, one PASS per 64 input channels (H >= 128: 2 or 4 passes; the
//                          [32 x H] accumulator tile stays in registers across the passes), fp32 MFMA 32x32x2.
//                          H >= 128: a wave owns H/128 column blocks and the whole inner dimension (no cross-wave
//                          reduction); H <= 32: the four waves split the inner dimension and meet through LDS.
//   nnconv_dw_gen_kernel   weight gradient dWcat = A^T g.  The [(K+2)H, H] result does not fit any register file, so
//                          a workgroup owns a SLICE of it (all K+2 blocks x 16 input channels x <= 128 output
//                          columns, 80 accumulator registers per lane) and a strided share of the 32-node tiles; it
//                          gathers only its 16 channels of the operand.  Slabs are summed in a fixed order by
//                          nnconv_dw_final_kernel straight into the parameters' own layouts.
//   nnconv_gradh_gen_kernel  grad of the edge MLP's first layer: per tile GA = g_tile @ Wk^T for 32 input channels at
//                          a time on the matrix cores, per-edge dots against the source rows.  The tile's edges are
//                          staged in LDS once and dealt evenly over the lane groups (the results are sums over all
//                          lanes), every pass adds its share straight into those sums (the gradient is linear in the dots).
// All sums have a fixed order: bitwise reproducible; per-node results (forward, adjoint) do not depend on where a graph
// sits in the batch.
#include "common.hpp"
#include "mfma_tile.hpp"

namespace qot {

template <int CPL>
__device__ __forceinline__ void ldv(const float* __restrict__ p, float (&v)[CPL]) {
    if constexpr (CPL == 8) {
        const float4 a = ld4(p), b = ld4(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else if constexpr (CPL == 4) {
        const float4 a = ld4(p);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    } else {
        const float2 a = *reinterpret_cast<const float2*>(p);
        v[0] = a.x; v[1] = a.y;
    }
}

// Gather of one destination's operand blocks by its 8-lane group, CPL channels per lane starting at channel `cbase` of the
// rows (x + row*ldx + cbase).  Lane `sub` prefetches in-edge `sub` of a batch of eight (source, edge id, features) and
// evaluates that edge's h = relu(W1 ea + b1) once; four source rows in flight.  acc[kk] (kk < K): sum_e h_e[kk] x_j,
// acc[K]: sum_e x_j, scaled by the destination's 1/deg (forward) or with 1/deg of the gathered end folded into h
// (TRANSPOSE); root = the node's own row.  i >= N gives zeros.
// The batch's per-edge values (source row, scale, h[K]) are exchanged through LDS (r03; r02 broadcast them with DPP moves).
// Every lane of an 8-lane group still prefetches one in-edge of a batch of eight and evaluates
// its h once; it then WRITES {j, scale, h[0..K)} as float4 chunks to the group's exchange slots and the group reads
// edge u's chunks back as LDS broadcasts.  The DPP form costs 2 VALU moves per value and edge -- 20 values, 40 moves
// against 36 (CPL = 4) or 72 (CPL = 8) FMAs -- and vector instructions of a gathering wave take issue cycles away
// from the MFMA wave of the same SIMD one for one (section 4.4 of DESIGN.md), LDS reads do not.
//   xch: float4 slots of THIS WAVE; chunk c of group gi, edge u at xch[(NCH * gi + c) * S + u]   (S = chunk stride)
template <int D>
struct XchW {
    static constexpr int K = 2 * D;
    static constexpr int NCH = (K + 2 + 3) / 4;     // float4 chunks per edge: {j, scale, h0, h1}, {h2..h5}, {h6, h7, -, -}
};

template <int D, int CPL, bool TRANSPOSE, int S>
__device__ __forceinline__ void gen_gather_xch(const float* __restrict__ x, int ldx, int cbase, const float* __restrict__ ea,
                                               const float* __restrict__ w1, const float* __restrict__ b1,
                                               const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                               const int32_t* __restrict__ eidx, const float* __restrict__ invdeg, int64_t i,
                                               int64_t N, float4* __restrict__ xch, float (&acc)[2 * D + 1][CPL],
                                               float (&root)[CPL]) {
    constexpr int K = 2 * D;
    constexpr int NCH = XchW<D>::NCH;
    const int sub = threadIdx.x & 7, gi = (threadIdx.x & 63) >> 3;
#pragma unroll
    for (int kk = 0; kk <= K; ++kk)
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc[kk][c] = 0.f;
#pragma unroll
    for (int c = 0; c < CPL; ++c) root[c] = 0.f;
    int beg = 0, end = 0;
    float srow = 0.f;
    if (i < N) {
        beg = rowptr[i]; end = rowptr[i + 1];
        if (!TRANSPOSE) srow = invdeg[i];
        ldv<CPL>(x + i * ldx + cbase, root);
    }
    float4* mine = xch + NCH * gi * S;
    for (int base = beg; base < end; base += 8) {
        const int p = base + sub;
        // dead slots of a batch (lane p >= end) carry h = 0 and scale = 0, so their terms vanish without per-use
        // selects; their row load is pointed at the destination's own row (always a valid address)
        float w[4 * NCH];
#pragma unroll
        for (int z = 0; z < 4 * NCH; ++z) w[z] = 0.f;
        w[0] = __builtin_bit_cast(float, (int)i);
        if (p < end) {
            const int myj = col[p];
            const int64_t e = eidx[p];
            float ee[D];
#pragma unroll
            for (int d = 0; d < D; ++d) ee[d] = ea[e * D + d];
            const float mysc = TRANSPOSE ? invdeg[myj] : 1.0f;
            w[0] = __builtin_bit_cast(float, myj);
            w[1] = mysc;
#pragma unroll
            for (int kk = 0; kk < K; ++kk) {
                float h = b1[kk];
#pragma unroll
                for (int d = 0; d < D; ++d) h = fmaf(w1[kk * D + d], ee[d], h);
                w[2 + kk] = fmaxf(h, 0.f) * mysc;
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) mine[c * S + sub] = make_float4(w[4 * c], w[4 * c + 1], w[4 * c + 2], w[4 * c + 3]);
        const int cnt = (end - base < 8) ? end - base : 8;
#define QOT_XCH_EDGE4(U0)                                                                                 \
        {                                                                                                 \
            float xv[4][CPL];                                                                             \
            float4 q0[4];                                                                                 \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) q0[u] = mine[U0 + u];                           \
            _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                 \
                ldv<CPL>(x + (int64_t)__builtin_bit_cast(int, q0[u].x) * ldx + cbase, xv[u]);             \
            float hh[4][4 * NCH];                                                                         \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                               \
                hh[u][0] = q0[u].x; hh[u][1] = q0[u].y; hh[u][2] = q0[u].z; hh[u][3] = q0[u].w;           \
                _Pragma("unroll") for (int c = 1; c < NCH; ++c) {                                         \
                    const float4 v = mine[c * S + U0 + u];                                                \
                    hh[u][4 * c] = v.x; hh[u][4 * c + 1] = v.y; hh[u][4 * c + 2] = v.z; hh[u][4 * c + 3] = v.w; \
                }                                                                                         \
            }                                                                                             \
            _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                               \
                _Pragma("unroll") for (int kk = 0; kk < K; ++kk)                                          \
                    _Pragma("unroll") for (int c = 0; c < CPL; ++c) acc[kk][c] = fmaf(hh[u][2 + kk], xv[u][c], acc[kk][c]); \
                _Pragma("unroll") for (int c = 0; c < CPL; ++c) acc[K][c] = fmaf(hh[u][1], xv[u][c], acc[K][c]); \
            }                                                                                             \
            /* four rows stay in flight, the vector work is skipped in pairs */                           \
            if (cnt > U0 + 2) {                                                                           \
                _Pragma("unroll") for (int u = 2; u < 4; ++u) {                                           \
                    _Pragma("unroll") for (int kk = 0; kk < K; ++kk)                                      \
                        _Pragma("unroll") for (int c = 0; c < CPL; ++c) acc[kk][c] = fmaf(hh[u][2 + kk], xv[u][c], acc[kk][c]); \
                    _Pragma("unroll") for (int c = 0; c < CPL; ++c) acc[K][c] = fmaf(hh[u][1], xv[u][c], acc[K][c]); \
                }                                                                                         \
            }                                                                                             \
        }
        QOT_XCH_EDGE4(0)
        if (cnt > 4) QOT_XCH_EDGE4(4)
#undef QOT_XCH_EDGE4
    }
    if (!TRANSPOSE) {
#pragma unroll
        for (int kk = 0; kk <= K; ++kk)
#pragma unroll
            for (int c = 0; c < CPL; ++c) acc[kk][c] *= srow;
    }
}

template <int H>
struct GenW {
    static constexpr int CW = H < 64 ? H : 64;          // input channels per pass
    static constexpr int NP = H / CW;                   // passes
    static constexpr int CPL = CW / 8;                  // channels per lane in the gather (8 lanes per destination)
    static constexpr int NCB = (H + 31) / 32;           // 32-column blocks of the output (H = 16: half a block)
    static constexpr int CBW = NCB >= 4 ? NCB / 4 : 1;  // column blocks per wave
    static constexpr int KS = NCB >= 4 ? 1 : 4 / NCB;   // ways the inner dimension is split over waves
    static constexpr bool ROOT_LDS = H < 64;            // root block kept in the LDS tile (small tiles only)
};

// channels [CPL*sub, CPL*sub + CPL) of block kk of one tile row -> fragment-grouped LDS tile (local k = kk*CW + ch)
template <int CW, int CPL>
__device__ __forceinline__ void frag_store(float* __restrict__ At, int kk, int sub, int il, const float (&v)[CPL]) {
    float4* At4 = reinterpret_cast<float4*>(At);
    if constexpr (CPL == 8) {
        const int g = kk * (CW / 8) + sub;
        At4[at4_slot(g, 0, il)] = make_float4(v[0], v[2], v[4], v[6]);
        At4[at4_slot(g, 1, il)] = make_float4(v[1], v[3], v[5], v[7]);
    } else if constexpr (CPL == 4) {                   // half a group: components 2*(sub&1), 2*(sub&1)+1
        const int g = kk * (CW / 8) + (sub >> 1);
        float2* s0 = reinterpret_cast<float2*>(&At4[at4_slot(g, 0, il)]) + (sub & 1);
        float2* s1 = reinterpret_cast<float2*>(&At4[at4_slot(g, 1, il)]) + (sub & 1);
        *s0 = make_float2(v[0], v[2]);
        *s1 = make_float2(v[1], v[3]);
    } else {                                           // a quarter of a group: component sub & 3
        const int g = kk * (CW / 8) + (sub >> 2);
        At[at4_slot(g, 0, il) * 4 + (sub & 3)] = v[0];
        At[at4_slot(g, 1, il) * 4 + (sub & 3)] = v[1];
    }
}

// Wp layout (functional.nnconv_gen_perm_index): for pass p, column block cb, group g of 8 local k (all K+2 blocks
// of the pass: local k = kk*CW + channel - p*CW), lane l, r:
//   Wp[(((p*NCB + cb)*GALL + g)*64 + l)*4 + r] = Wcat[kk*H + p*CW + c][cb*32 + (l & 31)],  8g + 2r + (l>>5) = kk*CW + c
// (zero where the column is >= H).
template <int H, int D, bool TRANSPOSE>
__global__ __launch_bounds__(256, 2) void nnconv_gen_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ ea, const float* __restrict__ w1,
    const float* __restrict__ b1, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const int32_t* __restrict__ eidx, const float* __restrict__ invdeg, const float* __restrict__ Wp,
    const float* __restrict__ bias, float* __restrict__ out, int64_t N, ActParams act, int variant) {
    // variant (diagnostic builds only, tools/ablate_gen.py): 1 = no gather (constant operand tile), 2 = weight
    // fragments loaded once (no L2 stream), 3 = gather only (no MFMA), 4 = 1 + 2
    using G = GenW<H>;
    constexpr int K = 2 * D;
    constexpr int CW = G::CW, NP = G::NP, CPL = G::CPL, NCB = G::NCB, CBW = G::CBW, KS = G::KS;
    constexpr bool ROOT_LDS = G::ROOT_LDS;
    constexpr int NBLK = ROOT_LDS ? K + 2 : K + 1;        // blocks resident in LDS
    constexpr int GMAIN = NBLK * CW / 8;                  // groups in LDS
    constexpr int GROOT = ROOT_LDS ? 0 : CW / 8;          // root groups multiplied after the main part
    constexpr int GALL = (K + 2) * CW / 8;                // groups per (pass, column block) in Wp
    static_assert(GMAIN % KS == 0 && GROOT % KS == 0, "inner split");
    constexpr int GS = GMAIN / KS, GRS = GROOT / KS;
    constexpr int CH = (GS % 4 == 0) ? 4 : ((GS % 2 == 0) ? 2 : 1);
    constexpr int NCH = GS / CH;
    constexpr int TILE_FLOATS = GMAIN * 8 * 32;
    static_assert(2 * GMAIN >= 8 * XchW<D>::NCH, "exchange slots inside the tile");
    constexpr int RED_FLOATS = (KS > 1) ? 4 * 16 * 64 : 0;
    __shared__ __attribute__((aligned(16))) float At[TILE_FLOATS > RED_FLOATS ? TILE_FLOATS : RED_FLOATS];
    const float4* At4 = reinterpret_cast<const float4*>(At);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r31 = lane & 31, hi = lane >> 5;
    const int sub = threadIdx.x & 7, il = threadIdx.x >> 3;
    const int cb0 = (KS == 1) ? wave * CBW : wave % NCB;
    const int ks = (KS == 1) ? 0 : wave / NCB;
    const int64_t ntiles = (N + 31) / 32;

#pragma unroll 1
    for (int64_t it = 0;; ++it) {
        const int64_t tile = xcd_tile(it, ntiles);
        if (tile < 0) break;
        const int64_t tile0 = tile * 32;
        const int64_t i = tile0 + il;
        f32x16 c[CBW];
#pragma unroll
        for (int q = 0; q < CBW; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) c[q][r] = 0.f;
#pragma unroll 1
        for (int p = 0; p < NP; ++p) {
            float root[CPL];
            {
                float acc[K + 1][CPL];
#ifdef QOT_DIAG
                if (variant == 1 || variant == 4) {
#pragma unroll
                    for (int kk = 0; kk <= K; ++kk)
#pragma unroll
                        for (int c_ = 0; c_ < CPL; ++c_) acc[kk][c_] = 1.0f + kk;
#pragma unroll
                    for (int c_ = 0; c_ < CPL; ++c_) root[c_] = 1.0f;
                } else
#endif
                // (the exchange slots of the gather are float4 slots of the operand tile that only this wave's rows map
                // to -- at4_slot keeps a wave's 8 rows inside its aligned 8 slots of every (group, half) line -- and the
                // tile is free between the barrier that ended the last MFMA phase and this wave's own frag_store)
                gen_gather_xch<D, CPL, TRANSPOSE, 32>(x, ldx, p * CW + CPL * sub, ea, w1, b1, rowptr, col, eidx, invdeg, i, N,
                                                      reinterpret_cast<float4*>(At) + 8 * wave, acc, root);
#pragma unroll
                for (int kk = 0; kk <= K; ++kk) frag_store<CW, CPL>(At, kk, sub, il, acc[kk]);
                if (ROOT_LDS) frag_store<CW, CPL>(At, K + 1, sub, il, root);
            }
#ifdef QOT_DIAG
            const int64_t gstep = (variant == 2 || variant == 4) ? 0 : 1;      // 0: every fragment load hits the same line
#else
            constexpr int64_t gstep = 1;
#endif
            // ---- main part: this wave's groups [ks*GS, (ks+1)*GS) against its column blocks.  The first chunk of weight
            // fragments is requested in FRONT of the barrier that publishes the operand tile, and that barrier orders LDS
            // traffic only: behind a __syncthreads() their L2 round trip was exposed once per pass.
            const float4* wp = reinterpret_cast<const float4*>(Wp) + ((int64_t)(p * NCB + cb0) * GALL + ks * GS) * 64 + lane;
            float4 bc[CBW][CH], bn[CBW][CH];
#pragma unroll
            for (int q = 0; q < CBW; ++q)
#pragma unroll
                for (int u = 0; u < CH; ++u) bc[q][u] = wp[((int64_t)q * GALL + u) * 64 * gstep];
            lds_barrier();
#ifdef QOT_DIAG
            if (variant == 3) { lds_barrier(); continue; }
#endif
            int ch = 0;
#pragma unroll 1
            for (; ch + 1 < NCH; ch += 2) {
#pragma unroll
                for (int q = 0; q < CBW; ++q)
#pragma unroll
                    for (int u = 0; u < CH; ++u) bn[q][u] = wp[((int64_t)q * GALL + (ch + 1) * CH + u) * 64 * gstep];
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const float4 a = At4[at4_slot(ks * GS + ch * CH + u, hi, r31)];
#pragma unroll
                    for (int q = 0; q < CBW; ++q) {
                        c[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bc[q][u].x, c[q], 0, 0, 0);
                        c[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bc[q][u].y, c[q], 0, 0, 0);
                        c[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bc[q][u].z, c[q], 0, 0, 0);
                        c[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bc[q][u].w, c[q], 0, 0, 0);
                    }
                }
                if (ch + 2 < NCH) {
#pragma unroll
                    for (int q = 0; q < CBW; ++q)
#pragma unroll
                        for (int u = 0; u < CH; ++u) bc[q][u] = wp[((int64_t)q * GALL + (ch + 2) * CH + u) * 64 * gstep];
                }
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const float4 a = At4[at4_slot(ks * GS + (ch + 1) * CH + u, hi, r31)];
#pragma unroll
                    for (int q = 0; q < CBW; ++q) {
                        c[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bn[q][u].x, c[q], 0, 0, 0);
                        c[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bn[q][u].y, c[q], 0, 0, 0);
                        c[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bn[q][u].z, c[q], 0, 0, 0);
                        c[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bn[q][u].w, c[q], 0, 0, 0);
                    }
                }
            }
            if (NCH & 1) {
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const float4 a = At4[at4_slot(ks * GS + (NCH - 1) * CH + u, hi, r31)];
#pragma unroll
                    for (int q = 0; q < CBW; ++q) {
                        c[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bc[q][u].x, c[q], 0, 0, 0);
                        c[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bc[q][u].y, c[q], 0, 0, 0);
                        c[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bc[q][u].z, c[q], 0, 0, 0);
                        c[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bc[q][u].w, c[q], 0, 0, 0);
                    }
                }
            }
            if (!ROOT_LDS) {
                // ---- root block (the node's own row) takes block 0's slots once everyone is done with the tile
                float4 rb[CBW][GRS > 0 ? GRS : 1];
                const float4* wr = reinterpret_cast<const float4*>(Wp) + ((int64_t)(p * NCB + cb0) * GALL + GMAIN + ks * GRS) * 64 + lane;
#pragma unroll
                for (int q = 0; q < CBW; ++q)
#pragma unroll
                    for (int u = 0; u < GRS; ++u) rb[q][u] = wr[((int64_t)q * GALL + u) * 64];
                lds_barrier();
                frag_store<CW, CPL>(At, 0, sub, il, root);
                __syncthreads();
#pragma unroll
                for (int u = 0; u < GRS; ++u) {
                    const float4 a = At4[at4_slot(ks * GRS + u, hi, r31)];
#pragma unroll
                    for (int q = 0; q < CBW; ++q) {
                        c[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, rb[q][u].x, c[q], 0, 0, 0);
                        c[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, rb[q][u].y, c[q], 0, 0, 0);
                        c[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, rb[q][u].z, c[q], 0, 0, 0);
                        c[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, rb[q][u].w, c[q], 0, 0, 0);
                    }
                }
            }
            lds_barrier();                 // the tile is rewritten by the next pass / tile (or becomes `red`)
        }
        // ---- epilogue: bias, leaky_relu + dropout, 128-B row segments
        if constexpr (KS == 1) {
#pragma unroll
            for (int q = 0; q < CBW; ++q) {
                const int colg = (cb0 + q) * 32 + r31;
                const float bz = bias ? bias[colg] : 0.f;
                if (act.enabled && act.thr16) {
                    // Dropout draws: one 64-bit hash serves the four columns of a float4 group, i.e. the four lanes of a quad
                    // for the same row.  Lane j of a quad hashes rows j, j + 4, j + 8, j + 12 of the accumulator and the quad
                    // shares them through quad_perm broadcasts: 4 hashes + 32 DPP moves per lane and column block instead of
                    // 16 hashes (as the H = 64 kernel; the draws are those of qot_act_fwd / act_apply1).
                    const uint64_t dstep = (uint64_t)act.step[0];
                    const int jq = r31 & 3;
                    uint32_t mylo[4], myhi[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int rr = jq + 4 * t;
                        const int row = (rr & 3) + 8 * (rr >> 2) + 4 * hi;
                        const uint64_t flat = (uint64_t)((tile0 + row) * H + colg);
                        const uint64_t z = act_hash64(act.seed, dstep, flat >> 2);
                        mylo[t] = (uint32_t)z; myhi[t] = (uint32_t)(z >> 32);
                    }
                    uint32_t zlo[16], zhi[16];
#define QOT_GEN_ZSHARE(S)                                                                                \
                    _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                      \
                        zlo[(S) + 4 * t] = __builtin_amdgcn_update_dpp(0, mylo[t], (S) * 0x55, 0xF, 0xF, true); \
                        zhi[(S) + 4 * t] = __builtin_amdgcn_update_dpp(0, myhi[t], (S) * 0x55, 0xF, 0xF, true); \
                    }
                    QOT_GEN_ZSHARE(0) QOT_GEN_ZSHARE(1) QOT_GEN_ZSHARE(2) QOT_GEN_ZSHARE(3)
#undef QOT_GEN_ZSHARE
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
                        const int64_t io = tile0 + row;
                        const float v = c[q][r] + bz;
                        float y = v > 0.f ? v : act.slope * v;
                        const uint32_t half = (jq & 2) ? zhi[r] : zlo[r];
                        const bool keep = ((half >> (16 * (jq & 1))) & 0xFFFFu) >= act.thr16;
                        y = keep ? y * act.keep_scale : 0.f;
                        if (io < N) out[io * H + colg] = y;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
                        const int64_t io = tile0 + row;
                        if (io < N) out[io * H + colg] = act_apply1(c[q][r] + bz, act, (uint64_t)(io * H + colg));
                    }
                }
            }
        } else {
            float* red = At;               // [wave][16][64]
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = c[0][r];
            lds_barrier();
            constexpr int RPW = 16 / KS;   // accumulator registers finished by each wave of a column block
            const int colg = cb0 * 32 + r31;
            const float bz = (bias && colg < H) ? bias[colg] : 0.f;
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr) {
                const int r = ks * RPW + rr;
                float v = bz;
#pragma unroll
                for (int k2 = 0; k2 < KS; ++k2) v += red[((cb0 + NCB * k2) * 16 + r) * 64 + lane];
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int64_t io = tile0 + row;
                if (io < N && colg < H) out[io * H + colg] = act_apply1(v, act, (uint64_t)(io * H + colg));
            }
            lds_barrier();
        }
    }
}

// ---------------------------------------------------------------------------------------------- weight gradient
template <int H>
struct DwW {
    static constexpr int AC = H >= 32 ? 32 : 16;                    // input channels of a slice
    static constexpr int NAC = H / AC;
    static constexpr int OC = H >= 128 ? 128 : (H < 32 ? 32 : H);   // output columns of a slice (H = 16: padded)
    static constexpr int NOC = (H + OC - 1) / OC;
    static constexpr int OCB = OC / 32;
    static constexpr int NSLICE = NAC * NOC;
};

// Two roles per workgroup (512 threads): waves 0-3 GATHER tile t+1 (operand slice + g rows) into one half of a
// double-buffered LDS image while waves 4-7 MULTIPLY tile t out of the other half; one barrier per tile.  A gather
// round costs about as much as the 320 MFMAs it feeds (index chain, edge MLP, broadcasts: a cost per edge, whatever
// the channel count -- measured 677 us gather-only vs 793 us MFMA-only at H = 128), and with every wave doing both
// in turn the two added up (1208 us); side by side on the same SIMDs the vector and the matrix pipe run together.
template <int H, int D>
__global__ __launch_bounds__(512, 2) void nnconv_dw_gen_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ g, int ldg, const float* __restrict__ ea,
    const float* __restrict__ w1, const float* __restrict__ b1, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ col, const int32_t* __restrict__ eidx, const float* __restrict__ invdeg,
    float* __restrict__ slabs, int64_t N, int nsplit, int variant) {
    using W = DwW<H>;
    constexpr int K = 2 * D;
    constexpr int AC = W::AC, NAC = W::NAC, OC = W::OC, OCB = W::OCB, NSLICE = W::NSLICE;
    constexpr int ROWS = (K + 2) * AC;              // rows of the slice of dWcat: (kk, a - a0)
    constexpr int RB = ROWS / 32;
    static_assert(ROWS % 32 == 0, "K + 2 even");
    constexpr int NT = RB * OCB;                    // 32x32 accumulator tiles of the slice
    constexpr int TPW = (NT + 3) / 4;               // per consumer wave (tile t belongs to consumer wave t % 4)
    __shared__ __attribute__((aligned(16))) float Atile[2][32 * ROWS];
    __shared__ __attribute__((aligned(16))) float Gt[2][32 * OC];
    __shared__ float4 xchg[4][XchW<D>::NCH * 8 * 8];       // per producer wave: the gather's edge exchange slots
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool producer = wave < 4;
    const int cw = wave & 3;                        // consumer wave index
    const int r31 = lane & 31, hi = lane >> 5;
    const int sub = threadIdx.x & 7, il = (threadIdx.x & 255) >> 3;
    const int64_t ntiles = (N + 31) / 32;
    // Walk of the node tiles.  Plain: slice = blockIdx % NSLICE, tiles split, split + nsplit, ...  XCD-aware (workgroups
    // are dealt round-robin over the 8 XCDs, blockIdx % 8, each with a private L2): an XCD owns a contiguous eighth of the
    // tiles, all NSLICE slices of a split sit on that XCD, and its nsplit / 8 splits walk neighbouring tiles at the same
    // time -- the g rows one slice fetches the other slices find in that L2, and so do neighbouring tiles (the same
    // graph) the source rows they share.  Measured (r03, N = 256 000): HBM fetch 954 -> 289 MB at H = 128 (3.5x -> 1.15x
    // the algorithmic bytes) for 863 -> 877 us; at H = 256 (16 slices) 3367 -> 3506 us, so only widths with <= 4 slices
    // take it.  (Every split a contiguous run of tiles, or the slices' starts rotated: slower, tools/experiments.)
    int slice = blockIdx.x % NSLICE, split = blockIdx.x / NSLICE;
    int64_t t_first = split, t_step = nsplit, t_end = ntiles;
    if (nsplit % 8 == 0 && NSLICE <= 4) {
        const int xcd = blockIdx.x % 8, q = blockIdx.x / 8;
        const int sl = q / NSLICE;
        slice = q % NSLICE;
        split = sl * 8 + xcd;
        const int64_t chunk = (ntiles + 7) / 8;
        t_first = xcd * chunk + sl;
        t_step = nsplit / 8;
        t_end = (xcd + 1) * chunk < ntiles ? (xcd + 1) * chunk : ntiles;
    }
    const int n_mine = t_first < t_end ? (int)((t_end - t_first + t_step - 1) / t_step) : 0;
    auto tile_of = [&](int k) -> int64_t { return t_first + (int64_t)k * t_step; };
    const int a0 = (slice % NAC) * AC, o0 = (slice / NAC) * OC;

    // The two roles are two separate loops with the same barrier count, so that the consumers' 80 accumulator registers
    // are not live in the producers' code (one loop with a role branch inside spilled 192 B/lane at 128 registers).
    if (producer) {
#ifdef QOT_DIAG
        if (variant == 6) __builtin_amdgcn_s_setprio(2);
#endif
        auto fill = [&](int64_t tile, int buf) {
            const int64_t tile0 = tile * 32;
            const int64_t i = tile0 + il;
            {
                const bool ok = i < N;
#pragma unroll
                for (int q = 0; q < OC / 8; q += 4) {
                    const int cc = o0 + (OC / 8) * sub + q;
                    const float4 v = (ok && cc < H) ? ld4(g + i * ldg + cc) : f4zero();
                    *reinterpret_cast<float4*>(&Gt[buf][il * OC + (OC / 8) * sub + q]) = v;
                }
            }
            constexpr int CPL = AC / 8;
            float acc[K + 1][CPL], root[CPL];
#ifdef QOT_DIAG
            if (variant == 1) {
#pragma unroll
                for (int kk = 0; kk <= K; ++kk)
#pragma unroll
                    for (int c_ = 0; c_ < CPL; ++c_) acc[kk][c_] = 1.f + kk;
#pragma unroll
                for (int c_ = 0; c_ < CPL; ++c_) root[c_] = 1.f;
            } else
#endif
            gen_gather_xch<D, CPL, false, 8>(x, ldx, a0 + CPL * sub, ea, w1, b1, rowptr, col, eidx, invdeg, i, N, xchg[wave], acc, root);
#pragma unroll
            for (int kk = 0; kk <= K; ++kk)
#pragma unroll
                for (int c_ = 0; c_ < CPL; ++c_) Atile[buf][il * ROWS + kk * AC + CPL * sub + c_] = acc[kk][c_];
#pragma unroll
            for (int c_ = 0; c_ < CPL; ++c_) Atile[buf][il * ROWS + (K + 1) * AC + CPL * sub + c_] = root[c_];
        };
        if (n_mine > 0) fill(tile_of(0), 0);
        __syncthreads();
        int buf = 0;
#pragma unroll 1
        for (int k = 0; k < n_mine; ++k, buf ^= 1) {
            if (k + 1 < n_mine) fill(tile_of(k + 1), buf ^ 1);
            __syncthreads();
        }
        return;
    }
#ifdef QOT_DIAG
    if (variant == 5) __builtin_amdgcn_s_setprio(2);
#endif
    f32x16 dw[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dw[t][r] = 0.f;
    // tile tt = cw + 4t has column block tt % OCB = cw % OCB for every t (OCB divides 4): one B operand per step
    static_assert(4 % OCB == 0, "column block per wave");
    int aoff[TPW];
    const int boff = hi * OC + (cw % OCB) * 32 + r31;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tt = cw + 4 * t;
        const int rb = (tt < NT) ? tt / OCB : 0;
        aoff[t] = hi * ROWS + rb * 32 + r31;
    }
    __syncthreads();
    int buf = 0;
#pragma unroll 1
    for (int k = 0; k < n_mine; ++k, buf ^= 1) {
#ifdef QOT_DIAG
        if (variant != 3)
#endif
        {
            // dW[(kk,a), o] += sum over the tile's nodes: A operand = the tile read transposed (node = k index)
            const float* At = Atile[buf];
            const float* Gb = Gt[buf];
#pragma unroll 4
            for (int s = 0; s < 16; ++s) {
                float av[TPW];
                const float bv = Gb[boff + 2 * s * OC];
#pragma unroll
                for (int t = 0; t < TPW; ++t) av[t] = At[aoff[t] + 2 * s * ROWS];
#pragma unroll
                for (int t = 0; t < TPW; ++t)
                    if (cw + 4 * t < NT) dw[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv, dw[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // slab[blockIdx][row (kk, a - a0)][o - o0]
    float* slab = slabs + (int64_t)(split * NSLICE + slice) * ROWS * OC;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tt = cw + 4 * t;
        if (tt < NT) {
            const int rb = tt / OCB, cb = tt % OCB;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
                slab[(rb * 32 + row) * OC + cb * 32 + r31] = dw[t][r];
            }
        }
    }
}

// sum over the node splits (fixed order), written straight into the parameters' layouts:
//   dst = [ g(nn.2.weight)[a*H+o, k] (H*H*K) | g(nn.2.bias)[a*H+o] (H*H) | g(lin.weight)[o, a] (H*H) ]
// LANES lanes per output element stride over the splits and meet in a fixed butterfly (a serial loop over hundreds of
// slabs per element is a chain of dependent-latency loads: 320 us at the reference's own H = 16 scale).
template <int H, int LANES>
__global__ void nnconv_dw_final_kernel(const float* __restrict__ slabs, int nsplit, int K, float* __restrict__ dst) {
    using W = DwW<H>;
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t e = t / LANES;
    const int sub = (int)(t % LANES);
    const int64_t hh = (int64_t)H * H;
    const bool live = e < hh * (K + 2);
    float s = 0.f;
    if (live) {
        int kk, a, o;
        if (e < hh * K) { kk = (int)(e % K); const int64_t ao = e / K; a = (int)(ao / H); o = (int)(ao % H); }
        else if (e < hh * (K + 1)) { kk = K; const int64_t ao = e - hh * K; a = (int)(ao / H); o = (int)(ao % H); }
        else { kk = K + 1; const int64_t oa = e - hh * (K + 1); o = (int)(oa / H); a = (int)(oa % H); }
        const int slice = (a / W::AC) + W::NAC * (o / W::OC);
        const int rows = (K + 2) * W::AC;
        const int64_t off = ((int64_t)slice * rows + kk * W::AC + (a % W::AC)) * W::OC + (o % W::OC);
        const int64_t stride = (int64_t)W::NSLICE * rows * W::OC;
        for (int sp = sub; sp < nsplit; sp += LANES) s += slabs[sp * stride + off];
    }
    if constexpr (LANES > 1) {
#pragma unroll
        for (int o2 = LANES / 2; o2 > 0; o2 >>= 1) s += __shfl_xor(s, o2);
    }
    if (live && sub == 0) dst[e] = s;
}

// ---------------------------------------------------------------------------------------------- grad of nn.0.*
// Bp layout (functional.nnconv_gen_gradh_perm_index): pass p (32 input channels, or H when H < 32), 32-column block
// nb of GA_p (column n = k*CWG + al), group gq of 8 output channels o, lane l, r:
//   Bp[(((p*NBG + nb)*GH + gq)*64 + l)*4 + r] = W2[(p*CWG + al)*H + o, k],  o = 8 gq + 2r + (l>>5),  n = nb*32 + (l&31)
constexpr int kEdgeCap = 512;      // edges of a tile staged in LDS (the rest is read directly)

template <int H, int D>
__global__ __launch_bounds__(256, 2) void nnconv_gradh_gen_kernel(
    const float* __restrict__ g, int ldg, const float* __restrict__ x, int ldx, const float* __restrict__ ea,
    const float* __restrict__ w1, const float* __restrict__ b1, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ col, const int32_t* __restrict__ eidx, const float* __restrict__ invdeg,
    const float* __restrict__ Bp, float* __restrict__ partials, int64_t N) {
    constexpr int K = 2 * D;
    static_assert(K <= 8, "one k per lane of the 8-lane group");
    constexpr int CWG = H < 32 ? H : 32;
    constexpr int NPG = H / CWG;
    constexpr int CPL = CWG / 8;                  // 4 (or 2 at H = 16)
    constexpr int NBG = K * CWG / 32;             // 32-column blocks of GA_p
    constexpr int GH = H / 8;                     // groups of 8 output channels
    constexpr int LDGA = K * CWG + 4;
    constexpr int EPF = 8;                        // edge slots of a lane group whose source rows are requested ahead
    __shared__ __attribute__((aligned(16))) float Gt[GH * 2 * 32 * 4];
    __shared__ __attribute__((aligned(16))) float GAt[32 * LDGA > 2 * (D + 1) * 256 ? 32 * LDGA : 2 * (D + 1) * 256];
    // the tile's first kEdgeCap edges, staged once per tile: source row, local destination row, 1/deg of the
    // destination, edge features
    __shared__ int sj[kEdgeCap], sr[kEdgeCap];
    __shared__ float ssc[kEdgeCap];
    __shared__ __attribute__((aligned(16))) float sea[kEdgeCap * D];
    __shared__ int rp_l[36];
    float4* Gt4 = reinterpret_cast<float4*>(Gt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r31 = lane & 31, hi = lane >> 5;
    const int sub = threadIdx.x & 7, il = threadIdx.x >> 3;
    const int64_t ntiles = (N + 31) / 32;

    float wrow[D], brow, aw[D], ab = 0.f;         // lane-owned row k = sub of the first edge-MLP layer
    brow = (sub < K) ? b1[sub] : 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) { wrow[d] = (sub < K) ? w1[sub * D + d] : 0.f; aw[d] = 0.f; }

    // One edge of the pass: dots of the destination's GA rows (LDS) with this pass's channels of the source row, the 8
    // values transposed over the 8 lanes (lane k ends with the total of k), and -- the gradient being LINEAR in the
    // dots -- the pass's share goes straight into the lane's sums: grad_h[e, k] = [pre_k > 0] / deg * sum over passes,
    // so no per-edge partial has to wait for the last pass (r02 kept them in LDS, which bounded the edges of a tile
    // that one MFMA phase could serve: a 600-edge tile of a power-law graph repeated its MFMA phases three times).
    auto edge = [&](int r, const float (&row)[CPL], const float (&ee)[D], float sc) {
        float pd[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float a = 0.f;
            if (k < K) {
                float ga[CPL];
                ldv<CPL>(&GAt[r * LDGA + k * CWG + CPL * sub], ga);
#pragma unroll
                for (int c_ = 0; c_ < CPL; ++c_) a = fmaf(ga[c_], row[c_], a);
            }
            pd[k] = a;
        }
        float t4[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float keep = (sub & 4) ? pd[m + 4] : pd[m];
            const float send = (sub & 4) ? pd[m] : pd[m + 4];
            t4[m] = keep + dpp_move<0x141>(send);
        }
        float t2[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const float keep = (sub & 2) ? t4[m + 2] : t4[m];
            const float send = (sub & 2) ? t4[m] : t4[m + 2];
            t2[m] = keep + dpp_move<0x4E>(send);
        }
        const float keep = (sub & 1) ? t2[1] : t2[0];
        const float send = (sub & 1) ? t2[0] : t2[1];
        const float tot = keep + dpp_move<0xB1>(send);          // k = sub
        float pre = brow;
#pragma unroll
        for (int d = 0; d < D; ++d) pre = fmaf(wrow[d], ee[d], pre);
        const float gh = (pre > 0.f && sub < K) ? tot * sc : 0.f;
        ab += gh;
#pragma unroll
        for (int d = 0; d < D; ++d) aw[d] = fmaf(gh, ee[d], aw[d]);
    };
    auto staged_ee = [&](int s, float (&ee)[D]) {
#pragma unroll
        for (int d = 0; d < D; ++d) ee[d] = sea[s * D + d];
    };

#pragma unroll 1
    for (int64_t it = 0;; ++it) {
        const int64_t tile = xcd_tile(it, ntiles);
        if (tile < 0) break;
        const int64_t tile0 = tile * 32;
        const int64_t i = tile0 + il;
        const int64_t tend = (tile0 + 32 < N) ? tile0 + 32 : N;
        const int e_t0 = rowptr[tile0], e_t1 = rowptr[tend];
        if (e_t0 == e_t1) continue;               // no in-edges in this tile (uniform)
        const int nt = e_t1 - e_t0;
        const int ncap = nt < kEdgeCap ? nt : kEdgeCap;
        // g tile -> LDS, fragment-grouped over the output channels o
        for (int gq = sub; gq < GH; gq += 8) {
            float4 g0 = f4zero(), g1 = f4zero();
            if (i < N) { g0 = ld4(g + i * ldg + 8 * gq); g1 = ld4(g + i * ldg + 8 * gq + 4); }
            Gt4[at4_slot(gq, 0, il)] = make_float4(g0.x, g0.z, g1.x, g1.z);
            Gt4[at4_slot(gq, 1, il)] = make_float4(g0.y, g0.w, g1.y, g1.w);
        }
        // ---- stage the tile's edges (r03): every thread one or two edges (coalesced col / edge id, then the features),
        // every destination marks its slots.  The per-edge work is then dealt EVENLY over the 32 lane groups (slot s ->
        // group s % 32) whatever the degrees: its results are sums over all lanes (aw, ab), so no lane group has to own a
        // destination -- a degree-64 row costs what 64 edges cost, not 8 serial batches of dependent loads in every
        // channel pass, and no group idles behind the longest row of its wave.
        {
            int beg = 0, end = 0;
            float sc_i = 0.f;
            if (i < N) { beg = rowptr[i]; end = rowptr[i + 1]; sc_i = invdeg[i]; }
            int ej[kEdgeCap / 256];
            int64_t ee_[kEdgeCap / 256];
#pragma unroll
            for (int q = 0; q < kEdgeCap / 256; ++q) {
                const int s = threadIdx.x + 256 * q;
                ej[q] = 0; ee_[q] = 0;
                if (s < ncap) { ej[q] = col[e_t0 + s]; ee_[q] = eidx[e_t0 + s]; }
            }
            const int lim = e_t0 + ncap;
            for (int p = beg + sub; p < end && p < lim; p += 8) { sr[p - e_t0] = il; ssc[p - e_t0] = sc_i; }
            if (threadIdx.x < 33) rp_l[threadIdx.x] = rowptr[tile0 + threadIdx.x < N ? tile0 + threadIdx.x : N];
#pragma unroll
            for (int q = 0; q < kEdgeCap / 256; ++q) {
                const int s = threadIdx.x + 256 * q;
                if (s < ncap) {
                    sj[s] = ej[q];
#pragma unroll
                    for (int d = 0; d < D; ++d) sea[s * D + d] = ea[ee_[q] * D + d];
                }
            }
        }
        const int mine = (ncap - il + 31) / 32;           // staged slots il, il + 32, ... of this lane group
#pragma unroll 1
        for (int p = 0; p < NPG; ++p) {
            lds_barrier();            // Gt / the staged edges written; GAt free
            // the group's first source rows for this pass are requested before the MFMA phase and used after it
            float xr[EPF][CPL];
            const int cbase = p * CWG + CPL * sub;
#pragma unroll
            for (int t = 0; t < EPF; ++t)
                if (t < mine) ldv<CPL>(x + (int64_t)sj[il + 32 * t] * ldx + cbase, xr[t]);
            // GA_p tile on the matrix cores
            for (int nb = wave; nb < NBG; nb += 4) {
                const float4* bp = reinterpret_cast<const float4*>(Bp) + ((int64_t)(p * NBG + nb) * GH) * 64 + lane;
                f32x16 c;
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = 0.f;
                // weight fragments of chunk ch+1 are requested before the MFMAs of chunk ch issue (loaded right in
                // front of their use, every group of 4 MFMAs waited for an L2 round trip)
                constexpr int GC = (GH % 4 == 0) ? 4 : 2;
                constexpr int NGC = GH / GC;
                float4 b0[GC], b1[GC];
#pragma unroll
                for (int v = 0; v < GC; ++v) b0[v] = bp[v * 64];
#pragma unroll 1
                for (int ch = 0; ch < NGC; ch += 2) {
                    if (ch + 1 < NGC) {
#pragma unroll
                        for (int v = 0; v < GC; ++v) b1[v] = bp[((ch + 1) * GC + v) * 64];
                    }
#pragma unroll
                    for (int v = 0; v < GC; ++v) c = mfma_group(Gt4, ch * GC + v, hi, r31, b0[v], c);
                    if (ch + 1 < NGC) {
                        if (ch + 2 < NGC) {
#pragma unroll
                            for (int v = 0; v < GC; ++v) b0[v] = bp[((ch + 2) * GC + v) * 64];
                        }
#pragma unroll
                        for (int v = 0; v < GC; ++v) c = mfma_group(Gt4, (ch + 1) * GC + v, hi, r31, b1[v], c);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
                    GAt[row * LDGA + nb * 32 + r31] = c[r];
                }
            }
            lds_barrier();
            // per-edge dots over this pass's channels: the staged slots of this group ...
#pragma unroll
            for (int t = 0; t < EPF; ++t) {
                if (t < mine) {
                    const int s = il + 32 * t;
                    float ee[D];
                    staged_ee(s, ee);
                    edge(sr[s], xr[t], ee, ssc[s]);
                }
            }
            for (int t = EPF; t < mine; ++t) {            // (more than 256 staged edges in the tile)
                const int s = il + 32 * t;
                float row[CPL], ee[D];
                ldv<CPL>(x + (int64_t)sj[s] * ldx + cbase, row);
                staged_ee(s, ee);
                edge(sr[s], row, ee, ssc[s]);
            }
            // ... and what the tile has beyond the staged ones (a hub's tile): straight from memory, destination by
            // bisection of the tile's row pointers
            for (int s = kEdgeCap + il; s < nt; s += 32) {
                const int pp = e_t0 + s;
                int lo = 0;
#pragma unroll
                for (int st = 16; st > 0; st >>= 1)
                    if (rp_l[lo + st] <= pp) lo += st;
                const int64_t e = eidx[pp];
                float row[CPL], ee[D];
                ldv<CPL>(x + (int64_t)col[pp] * ldx + cbase, row);
#pragma unroll
                for (int d = 0; d < D; ++d) ee[d] = ea[e * D + d];
                edge(lo, row, ee, invdeg[tile0 + lo]);
            }
        }
        lds_barrier();                // GAt / the staged edges are rewritten by the next tile
    }
    // block partial: sum the 32 lane groups (fixed order) -> partials[blk][K*(D+1)]
    lds_barrier();
    float* red = GAt;
#pragma unroll
    for (int d = 0; d < D; ++d) red[(d * 32 + il) * 8 + sub] = aw[d];
    red[(D * 32 + il) * 8 + sub] = ab;
    lds_barrier();
    if (threadIdx.x < (D + 1) * 8) {
        const int s8 = threadIdx.x & 7, d = threadIdx.x >> 3;
        float sum = 0.f;
        for (int g32 = 0; g32 < 32; ++g32) sum += red[(d * 32 + g32) * 8 + s8];
        if (s8 < K) partials[(int64_t)blockIdx.x * (K * (D + 1)) + (d < D ? s8 * D + d : K * D + s8)] = sum;
    }
}

__global__ void gen_partial_sum_kernel(const float* __restrict__ partials, int nblk, int n, int KD,
                                       float* __restrict__ gw1, float* __restrict__ gb1) {
    const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);      // one wave per output
    if (t >= n) return;
    const float s = wave_sum_partials(partials, nblk, n, t);
    if ((threadIdx.x & 63) == 0) { if (t < KD) gw1[t] = s; else gb1[t - KD] = s; }
}

}  // namespace qot

using namespace qot;

#define QOT_DISPATCH_GEN_H(H, ...)                               \
    switch (H) {                                                 \
        case 16:  { constexpr int kH = 16;  __VA_ARGS__; } break;  \
        case 32:  { constexpr int kH = 32;  __VA_ARGS__; } break;  \
        case 64:  { constexpr int kH = 64;  __VA_ARGS__; } break;  \
        case 128: { constexpr int kH = 128; __VA_ARGS__; } break;  \
        case 256: { constexpr int kH = 256; __VA_ARGS__; } break;  \
        default: return QOT_ERR_UNSUPPORTED;                     \
    }
#define QOT_DISPATCH_D4(D, ...)                                  \
    switch (D) {                                                 \
        case 1: { constexpr int kD = 1; __VA_ARGS__; } break;    \
        case 2: { constexpr int kD = 2; __VA_ARGS__; } break;    \
        case 3: { constexpr int kD = 3; __VA_ARGS__; } break;    \
        case 4: { constexpr int kD = 4; __VA_ARGS__; } break;    \
        default: return QOT_ERR_UNSUPPORTED;                     \
    }

#ifdef QOT_DIAG
static int g_gen_variant = 0;
extern "C" void qot_debug_gen_variant(int v) { g_gen_variant = v; }
#endif

// Width-generic form of qot_nnconv_fused (same arguments; w_perm in the per-pass layout documented above).
int qot_nnconv_gen_launch(const float* x, int ld_x, const float* edge_attr, const float* w1, const float* b1,
                          const int32_t* rowptr, const int32_t* col, const int32_t* edge_ids, const float* invdeg,
                          int transpose, const float* w_perm, const float* bias, float* out, int64_t N, int H, int D,
                          const ActParams& ap, hipStream_t stream) {
#ifdef QOT_DIAG
    const int variant = g_gen_variant;
#else
    const int variant = 0;
#endif
    int grid = grid_for(N, 32);
    if (grid > 2 * num_cus()) grid = 2 * num_cus();
    QOT_DISPATCH_GEN_H(H, QOT_DISPATCH_D4(D, {
        if (transpose)
            nnconv_gen_kernel<kH, kD, true><<<grid, 256, 0, stream>>>(x, ld_x, edge_attr, w1, b1, rowptr, col, edge_ids,
                                                                       invdeg, w_perm, bias, out, N, ap, variant);
        else
            nnconv_gen_kernel<kH, kD, false><<<grid, 256, 0, stream>>>(x, ld_x, edge_attr, w1, b1, rowptr, col, edge_ids,
                                                                        invdeg, w_perm, bias, out, N, ap, variant);
    }));
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

static int dw_splits(int64_t N, int nslice) {
    const int64_t ntiles = (N + 31) / 32;
    int64_t s = ((int64_t)num_cus() + nslice - 1) / nslice;          // one 8-wave workgroup per CU in all
    const int64_t by_work = (ntiles + 7) / 8;                         // a workgroup's slab should stand for >= 8 tiles
    if (s > by_work) s = by_work;
    if (s < 1) s = 1;
    return (int)s;
}

static int dw_nslice(int H) {
    switch (H) {
        case 16: return DwW<16>::NSLICE;
        case 32: return DwW<32>::NSLICE;
        case 64: return DwW<64>::NSLICE;
        case 128: return DwW<128>::NSLICE;
        case 256: return DwW<256>::NSLICE;
    }
    return 0;
}
static int dw_oc(int H) { return H >= 128 ? 128 : (H < 32 ? 32 : H); }

extern "C" size_t qot_nnconv_dw_workspace_floats(int64_t N, int H, int D) {
    const int ns = dw_nslice(H);
    if (!ns || N <= 0) return 16;
    return (size_t)dw_splits(N, ns) * ns * (size_t)((2 * D + 2) * (H >= 32 ? 32 : 16)) * dw_oc(H);
}

// Weight gradient of NNConv for any supported width: grad_params = [g(nn.2.weight) | g(nn.2.bias) | g(lin.weight)]
// ((2D+2)*H*H floats, the parameters' own layouts) from x, grad_out and the CSR (by destination).
extern "C" int qot_nnconv_dw(const float* x, int ld_x, const float* grad_out, int ld_g, const float* edge_attr,
                             const float* w1, const float* b1, const int32_t* rowptr, const int32_t* col,
                             const int32_t* eid, const float* invdeg, float* grad_params, float* workspace, int64_t N,
                             int H, int D, qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N <= 0 || !rowptr) return QOT_ERR_BADARG;
    if (!x || !grad_out || !w1 || !b1 || !invdeg || !grad_params || !workspace || (ld_x & 3) || (ld_g & 3))
        return QOT_ERR_BADARG;
    const int ns = dw_nslice(H);
    if (!ns) return QOT_ERR_UNSUPPORTED;
    const int nsplit = dw_splits(N, ns);
#ifdef QOT_DIAG
    const int variant = g_gen_variant;
#else
    const int variant = 0;
#endif
    QOT_DISPATCH_GEN_H(H, QOT_DISPATCH_D4(D, {
        nnconv_dw_gen_kernel<kH, kD><<<nsplit * ns, 512, 0, stream>>>(x, ld_x, grad_out, ld_g, edge_attr, w1, b1, rowptr,
                                                                      col, eid, invdeg, workspace, N, nsplit, variant);
        QOT_LAUNCH_CHECK();
        const int64_t elems = (int64_t)(2 * kD + 2) * kH * kH;
        if (nsplit >= 64)
            nnconv_dw_final_kernel<kH, 16><<<grid_for(elems * 16, 256), 256, 0, stream>>>(workspace, nsplit, 2 * kD, grad_params);
        else if (nsplit >= 8)
            nnconv_dw_final_kernel<kH, 4><<<grid_for(elems * 4, 256), 256, 0, stream>>>(workspace, nsplit, 2 * kD, grad_params);
        else
            nnconv_dw_final_kernel<kH, 1><<<grid_for(elems, 256), 256, 0, stream>>>(workspace, nsplit, 2 * kD, grad_params);
    }));
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

int qot_nnconv_gradh_gen_launch(const float* grad_out, int ld_g, const float* x, int ld_x, const float* edge_attr,
                                const float* w1, const float* b1, const int32_t* rowptr, const int32_t* col,
                                const int32_t* eid, const float* invdeg, const float* b_perm, float* gw1, float* gb1,
                                float* workspace, int64_t N, int H, int D, hipStream_t stream) {
    int grid = grid_for(N > 0 ? N : 1, 32);
    if (grid > 2 * num_cus()) grid = 2 * num_cus();
    const int K = 2 * D;
    QOT_DISPATCH_GEN_H(H, QOT_DISPATCH_D4(D, {
        nnconv_gradh_gen_kernel<kH, kD><<<grid, 256, 0, stream>>>(grad_out, ld_g, x, ld_x, edge_attr, w1, b1, rowptr, col,
                                                                  eid, invdeg, b_perm, workspace, N);
    }));
    QOT_LAUNCH_CHECK();
    const int n = K * (D + 1);
    gen_partial_sum_kernel<<<grid_for(n, 4), 256, 0, stream>>>(workspace, grid, n, K * D, gw1, gb1);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}
