// grad of the edge MLP's first layer at H = 64 (nnconv_gradh64_kernel), split off nnconv_mfma.hip so that THIS kernel can be
// compiled with -mllvm -amdgpu-sched-strategy=max-ilp (Makefile): measured 87 -> 82 us at cfg2, while the same strategy costs
// the adjoint kernel 2.6 us and spills the forward one (r03; tools/experiments/r03_sched_bench.sh).
#include "common.hpp"
#include "mfma_tile.hpp"

namespace qot {

// ---------------------------------------------------------------------------------------------
// grad of the edge MLP's first layer (nn.0.weight / nn.0.bias), fused.
//   dL/dh_e[k] = invdeg_i * < GA_i[k,:], x_j > ,  GA_i[k,a] = sum_o g_i[o] * W2[a*64+o, k]
// The unfused path materialises GA ([N, 512] fp32, a 6.7 GFLOP library GEMM measured at 291 us)
// and re-reads it per destination.  Here, per tile of 32 destinations:
//   1. g rows -> LDS in MFMA fragment order (8 KB)
//   2. GA tile [32 x 512] = g_tile @ Bm on the matrix cores (4 waves x 4 column blocks,
//      128 v_mfma_f32_32x32x2_f32 per wave), accumulators -> LDS row-major (66 KB, padded rows)
//   3. the tile's edges, staged in LDS at the top of the tile, dealt evenly over the 32 lane groups (r03; r02
//      gave every destination a lane group: its GA rows stayed in registers, but a group walked its edges in
//      batches of dependent loads and idled behind the longest row of its wave): per edge the destination's GA
//      rows come from LDS, 8 partial dots, transpose-reduced over the 8 lanes (7 DPP moves) so lane s ends with
//      k = s; relu mask recomputed from the edge features; the source rows are requested before phase 2.
// Per-lane accumulators of (gw1[k,:], gb1[k]) live across the persistent loop; block partials
// are summed in a fixed order afterwards (bitwise reproducible, no float atomics).
constexpr int kGaLd = 516;    // padded GA row (floats): rows 4 banks apart

constexpr int kGhCap = 256;   // edges of a tile staged in LDS by the grad-h kernel (the rest is read directly)

// VARIANT (diagnostic build only): 0 production; 2 no dot phase; 3 no GA MFMA phase; 5 GA phase with the weight fragments
// of block 0 reused (no L2 stream)
template <int D, int VARIANT = 0>
__global__ __launch_bounds__(256, 2) void nnconv_gradh64_kernel(
    const float* __restrict__ g, int ldg, const float* __restrict__ x, int ldx, const float* __restrict__ ea,
    const float* __restrict__ w1, const float* __restrict__ b1, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ col, const int32_t* __restrict__ eidx, const float* __restrict__ invdeg,
    const float* __restrict__ Bp, float* __restrict__ partials, int64_t N) {
    constexpr int K = 2 * D;
    static_assert(K <= 8, "one k per lane of the 8-lane group");
    constexpr int NB = K * 2;               // 32-column blocks of GA (K*64/32)
    constexpr int NBW = NB / 4;             // per wave
    constexpr int EPG = kGhCap / 32;        // staged edge slots per lane group
    constexpr int EPF = 5;                  // ... of which the source rows are requested in front of the MFMA phase
    __shared__ __attribute__((aligned(16))) float Gt[8 * 2 * 32 * 4];     // g tile (x 1/deg), fragment-grouped
    __shared__ __attribute__((aligned(16))) float GAt[32 * (K * 64 + 4)];  // GA tile, row-major padded
    // the tile's first kGhCap edges, staged once per tile: source row, local destination row, edge features
    __shared__ int sj[kGhCap], sr[kGhCap];
    __shared__ __attribute__((aligned(16))) float sea[kGhCap * D];
    __shared__ int rp_l[36];
    constexpr int LDGA = K * 64 + 4;
    float4* Gt4 = reinterpret_cast<float4*>(Gt);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r31 = lane & 31, hi = lane >> 5;
    const int sub = threadIdx.x & 7, il = threadIdx.x >> 3;
    const int c0 = 8 * sub;
    const int64_t ntiles = (N + 31) / 32;

    // lane-owned slice of the first edge-MLP layer: row k = sub
    float wrow[D], brow, aw[D], ab = 0.f;
    brow = (sub < K) ? b1[sub] : 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) { wrow[d] = (sub < K) ? w1[sub * D + d] : 0.f; aw[d] = 0.f; }

    // One edge: dots of the destination's GA rows (LDS; 1/deg is in the g tile) with the source row's 8 channels of this
    // lane, transposed over the 8 lanes of the group (lane k ends with the total of k), relu mask from the edge features.
    auto edge = [&](int r, const float4& x0, const float4& x1, const float (&ee)[D]) {
        float pd[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float a = 0.f;
            if (k < K) {
                const float4 u0 = *reinterpret_cast<const float4*>(&GAt[r * LDGA + k * 64 + c0]);
                const float4 u1 = *reinterpret_cast<const float4*>(&GAt[r * LDGA + k * 64 + c0 + 4]);
                float a0 = u0.x * x0.x, a1 = u0.y * x0.y;        // two interleaved partial dots
                a0 = fmaf(u0.z, x0.z, a0); a1 = fmaf(u0.w, x0.w, a1);
                a0 = fmaf(u1.x, x1.x, a0); a1 = fmaf(u1.y, x1.y, a1);
                a0 = fmaf(u1.z, x1.z, a0); a1 = fmaf(u1.w, x1.w, a1);
                a = a0 + a1;
            }
            pd[k] = a;
        }
        float t4[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float keep = (sub & 4) ? pd[m + 4] : pd[m];
            const float send = (sub & 4) ? pd[m] : pd[m + 4];
            t4[m] = keep + dpp_move<0x141>(send);      // partner 7 - sub (bit 2 differs)
        }
        float t2[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const float keep = (sub & 2) ? t4[m + 2] : t4[m];
            const float send = (sub & 2) ? t4[m] : t4[m + 2];
            t2[m] = keep + dpp_move<0x4E>(send);       // partner sub ^ 2
        }
        const float keep = (sub & 1) ? t2[1] : t2[0];
        const float send = (sub & 1) ? t2[0] : t2[1];
        const float tot = keep + dpp_move<0xB1>(send);   // partner sub ^ 1; k = sub
        float pre = brow;
#pragma unroll
        for (int d = 0; d < D; ++d) pre = fmaf(wrow[d], ee[d], pre);
        const float gh = (pre > 0.f && sub < K) ? tot : 0.f;
        ab += gh;
#pragma unroll
        for (int d = 0; d < D; ++d) aw[d] = fmaf(gh, ee[d], aw[d]);
    };

#pragma unroll 1
    for (int64_t it = 0;; ++it) {
        const int64_t tile = xcd_tile(it, ntiles);
        if (tile < 0) break;
        const int64_t tile0 = tile * 32;
        const int64_t i = tile0 + il;
        const int64_t tend = (tile0 + 32 < N) ? tile0 + 32 : N;
        // Weight fragments of this wave's first GA block are requested before anything else of the tile, those of block
        // t + 1 before the MFMAs of block t (two buffers).
        float4 bf[2][8];
        {
            const float4* bp = reinterpret_cast<const float4*>(Bp) + (int64_t)(wave * NBW) * 8 * 64 + lane;
#pragma unroll
            for (int gq = 0; gq < 8; ++gq) bf[0][gq] = bp[gq * 64];
        }
        // 1. the tile's edges, staged (r03): every thread one edge (coalesced col / edge id, then its features), every
        // destination marks its slots; the per-edge work is dealt EVENLY over the 32 lane groups afterwards (slot s ->
        // group s % 32) whatever the degrees -- its results are sums over all lanes, no group has to own a destination.
        const int e_t0 = rowptr[tile0], e_t1 = rowptr[tend];
        const int nt = e_t1 - e_t0;
        const int ncap = nt < kGhCap ? nt : kGhCap;
        int64_t my_e = 0;
        if ((int)threadIdx.x < ncap) {
            sj[threadIdx.x] = col[e_t0 + threadIdx.x];
            my_e = eidx[e_t0 + threadIdx.x];
        }
        if (threadIdx.x < 33) rp_l[threadIdx.x] = rowptr[tile0 + threadIdx.x < N ? tile0 + threadIdx.x : N];
        // g tile x 1/deg -> LDS (fragment-grouped, group index = sub)
        {
            float4 g0 = f4zero(), g1 = f4zero();
            if (i < N) {
                const float sc = invdeg[i];
                const int beg = rowptr[i], end = rowptr[i + 1];
                g0 = scale4(sc, ld4(g + i * ldg + c0)); g1 = scale4(sc, ld4(g + i * ldg + c0 + 4));
                const int lim = e_t0 + ncap;
                for (int p = beg + sub; p < end && p < lim; p += 8) sr[p - e_t0] = il;
            }
            Gt4[at4_slot(sub, 0, il)] = make_float4(g0.x, g0.z, g1.x, g1.z);
            Gt4[at4_slot(sub, 1, il)] = make_float4(g0.y, g0.w, g1.y, g1.w);
        }
        float my_ea[D];
#pragma unroll
        for (int d = 0; d < D; ++d) my_ea[d] = ((int)threadIdx.x < ncap) ? ea[my_e * D + d] : 0.f;
        lds_barrier();
        // the group's first source rows are requested before the MFMA phase and used after it
        const int mine = (ncap - il + 31) / 32;           // staged slots il, il + 32, ... of this lane group
        float4 xr0[EPF], xr1[EPF];
#pragma unroll
        for (int t = 0; t < EPF; ++t) {
            if (t < mine) {
                const float* xp = x + (int64_t)sj[il + 32 * t] * ldx + c0;
                xr0[t] = ld4(xp); xr1[t] = ld4(xp + 4);
            }
        }
        // 2. GA tile on the matrix cores
        if (VARIANT != 3) {
            float4 af[8];
#pragma unroll
            for (int gq = 0; gq < 8; ++gq) af[gq] = Gt4[at4_slot(gq, hi, r31)];
#pragma unroll
            for (int t = 0; t < NBW; ++t) {
                const int nb = wave * NBW + t;
                if (t + 1 < NBW) {
                    const float4* bp = reinterpret_cast<const float4*>(Bp) +
                                       (int64_t)((VARIANT == 5) ? wave * NBW : nb + 1) * 8 * 64 + lane;
#pragma unroll
                    for (int gq = 0; gq < 8; ++gq) bf[(t + 1) & 1][gq] = bp[gq * 64];
                }
                f32x16 c;
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = 0.f;
#pragma unroll
                for (int gq = 0; gq < 8; ++gq) {
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[gq].x, bf[t & 1][gq].x, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[gq].y, bf[t & 1][gq].y, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[gq].z, bf[t & 1][gq].z, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[gq].w, bf[t & 1][gq].w, c, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
                    GAt[row * LDGA + nb * 32 + r31] = c[r];
                }
            }
        }
        if ((int)threadIdx.x < ncap) {
#pragma unroll
            for (int d = 0; d < D; ++d) sea[threadIdx.x * D + d] = my_ea[d];
        }
        lds_barrier();
        // 3. per-edge dots: the staged slots of this group ...
        if (VARIANT != 2) {
#pragma unroll
            for (int t = 0; t < EPF; ++t) {
                if (t < mine) {
                    const int s = il + 32 * t;
                    float ee[D];
#pragma unroll
                    for (int d = 0; d < D; ++d) ee[d] = sea[s * D + d];
                    edge(sr[s], xr0[t], xr1[t], ee);
                }
            }
            for (int t = EPF; t < mine; ++t) {
                const int s = il + 32 * t;
                const float* xp = x + (int64_t)sj[s] * ldx + c0;
                const float4 y0 = ld4(xp), y1 = ld4(xp + 4);
                float ee[D];
#pragma unroll
                for (int d = 0; d < D; ++d) ee[d] = sea[s * D + d];
                edge(sr[s], y0, y1, ee);
            }
            // ... and what the tile has beyond the staged ones (a hub's tile): straight from memory, destination by
            // bisection of the tile's row pointers
            for (int s = kGhCap + il; s < nt; s += 32) {
                const int pp = e_t0 + s;
                int lo = 0;
#pragma unroll
                for (int st = 16; st > 0; st >>= 1)
                    if (rp_l[lo + st] <= pp) lo += st;
                const int64_t e = eidx[pp];
                const float* xp = x + (int64_t)col[pp] * ldx + c0;
                const float4 y0 = ld4(xp), y1 = ld4(xp + 4);
                float ee[D];
#pragma unroll
                for (int d = 0; d < D; ++d) ee[d] = ea[e * D + d];
                edge(lo, y0, y1, ee);
            }
        }
        lds_barrier();        // GAt / Gt / the staged edges are rewritten by the next tile
    }
    // block partial: sum the 32 lane groups (fixed order) -> partials[blk][K*(D+1)]
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

}  // namespace qot

using namespace qot;

// launch of nnconv_gradh64_kernel (qot_nnconv_gradh_fused, nnconv_mfma.hip); variant: diagnostic builds only
int qot_nnconv_gradh64_launch(const float* grad_out, int ld_g, const float* x, int ld_x, const float* edge_attr, const float* w1,
                              const float* b1, const int32_t* rowptr, const int32_t* col, const int32_t* eid, const float* invdeg,
                              const float* b_perm, float* workspace, int64_t N, int D, int grid, int variant, hipStream_t stream) {
#ifdef QOT_DIAG
    if (variant >= 1 && variant <= 5 && D == 4) {
#define QOT_GH_V(V) nnconv_gradh64_kernel<4, V><<<grid, 256, 0, stream>>>(grad_out, ld_g, x, ld_x, edge_attr, w1, b1, rowptr, col, \
                                                                        eid, invdeg, b_perm, workspace, N)
        switch (variant) {
            case 1: QOT_GH_V(1); break;
            case 2: QOT_GH_V(2); break;
            case 3: QOT_GH_V(3); break;
            case 4: QOT_GH_V(4); break;
            default: QOT_GH_V(5); break;
        }
#undef QOT_GH_V
        QOT_LAUNCH_CHECK();
        return QOT_OK;
    }
#endif
    (void)variant;
    QOT_DISPATCH_D(D, {
        if (kD <= 4)
            nnconv_gradh64_kernel<(kD <= 4 ? kD : 4)><<<grid, 256, 0, stream>>>(
                grad_out, ld_g, x, ld_x, edge_attr, w1, b1, rowptr, col, eid, invdeg, b_perm, workspace, N);
    });
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}
