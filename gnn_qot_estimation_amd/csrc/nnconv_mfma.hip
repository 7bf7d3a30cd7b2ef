// NNConv fused gather -> LDS operand tile -> fp32 MFMA  (H = 64).
//
// out[i,:] = bias + A_i @ Wcat,  A_i = [inv_i * sum_e h_e[k] x_j (k<K) | inv_i * sum_e x_j | x_i]
// (see nnconv.hip for the algebra; reference site topological_training/models.py:57).
// The unfused path writes A ([N, 640] fp32 = 262 MB at B=1024) to HBM and reads it back in
// a library GEMM; here a workgroup builds the 32 x 640 operand tile of 32 consecutive
// destinations directly in LDS and multiplies it on the matrix cores, so HBM sees only
// x (read), the edge features / CSR, and out (written).
//
//   phase 1  8 lanes per destination (2 x float4 = 64 channels), all 32 destinations at once;
//            blocks 0..K land in LDS in fragment-grouped order (at4_slot): two conflict-free
//            ds_write_b128 per lane per block.
//   phase 2  4 waves = 2 column halves x 2 K halves; each wave chains
//            v_mfma_f32_32x32x2_f32 on one 32x32 accumulator.  A fragments: ONE conflict-free
//            ds_read_b128 per 4 MFMAs (measured: a ds_read_b32 + swizzle per MFMA capped the
//            loop at ~105 of the ~135 TFLOP/s the register-fed loop reaches).
//            B fragment: Wcat pre-permuted on the host side into fragment order so a lane
//            reads 16 B = 4 consecutive k-steps, a wave 1 KiB contiguous (L2-resident, 160 KB).
//   root     the destination's own row (block K+1) is kept in registers during phase 2 and
//            then overwrites block 0's slots: 72 KB of LDS instead of 80 KB, which is what
//            lets TWO workgroups share a CU (2 x 80 KB needs every byte of the 160 KiB and
//            was measured to run one workgroup per CU: gather and MFMA phases serialised).
//   phase 3  K halves summed through LDS, bias added, 128-B row segments stored.
// One workgroup gathers while its CU-mate owns the MFMA pipes.  TRANSPOSE runs the adjoint (grad_x) over the CSC with the mean scale taken at
// the gathered end.
#include "common.hpp"
#include "mfma_tile.hpp"
#include "nnconv_finalize_dev.hpp"

namespace qot {

// Phase 1.  8 lanes per destination (2 x float4 = 64 channels each), so the 32 destinations of
// a tile are gathered in ONE pass by the 256 threads.  Lane `sub` of a destination's group
// prefetches the (source, edge id, edge features) of in-edge `sub` and evaluates that edge's
// h = relu(W1 ea + b1) once; the per-edge loop then only broadcasts (8-lane shuffles) and
// streams the 256-B source rows, four edges in flight per group.
// Returns the destination's own row (root block) in registers; blocks 0..K go to LDS.
// GV (diagnostic build only): 6 every gathered row read from one of two hot addresses; 7 arithmetic ids instead of the
// index chain; 8 one of the K weighted sums only
template <int D, bool TRANSPOSE, int GV = 0>
__device__ __forceinline__ void nnconv_gather_tile(
    float* __restrict__ At, const float* __restrict__ x, int ldx, const float* __restrict__ ea,
    const float* __restrict__ w1, const float* __restrict__ b1, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ col, const int32_t* __restrict__ eidx, const float* __restrict__ invdeg,
    int64_t tile0, int64_t N, float4& root0, float4& root1) {
    constexpr int K = 2 * D;
    const int sub = threadIdx.x & 7, il = threadIdx.x >> 3;
    const int c0 = 8 * sub;
    const int64_t i = tile0 + il;
    float4 acc0[K + 1], acc1[K + 1];        // written (not accumulated) by the first edge slot: no zero fill per tile
    root0 = f4zero(); root1 = f4zero();
    int beg = 0, end = 0;
    // the row's own loads (mean scale, root row) are issued BEFORE the edge loop: behind it they were one more
    // exposed memory round trip per tile
    float srow = 0.f;
    if (i < N) {
        beg = rowptr[i]; end = rowptr[i + 1];
        if (!TRANSPOSE) srow = invdeg[i];
        root0 = ld4(x + i * ldx + c0);
        root1 = ld4(x + i * ldx + c0 + 4);
    }
    const int own = (i < N) ? (int)i : 0;   // dead edge slots read the destination's own row (times zero)
    // One batch = up to 8 in-edges of every destination, one edge per lane: indices, edge features, edge-MLP hidden vector
    // (the mean scale 1/deg of the forward form is folded into it), then two blocks of four slots.  Lanes 0..7 of a group
    // hold edges base..base+7; broadcasts with compile-time source lanes run on the DPP path (as __shfl with a runtime lane
    // each was a ds_bpermute_b32 round trip: ten per edge).  Dead slots (lane p >= end) carry h = 0 and scale = 0: their
    // terms vanish without per-use selects.  FIRST: the batch that every destination runs (also one without in-edges); its
    // first slot WRITES the accumulators.
#define QOT_ACC(FIRSTSLOT, H, X, A) ((FIRSTSLOT) ? scale4((H), (X)) : fma4((H), (X), (A)))
#define QOT_EDGE4(U0, FIRST)                                                                            \
        {                                                                                               \
            float4 xa[4], xb[4];                                                                        \
            float sc[4];                                                                                \
            int jj[4];                                                                                  \
            jj[0] = group8_bcast<U0 + 0>(myj); jj[1] = group8_bcast<U0 + 1>(myj);                       \
            jj[2] = group8_bcast<U0 + 2>(myj); jj[3] = group8_bcast<U0 + 3>(myj);                       \
            sc[0] = group8_bcast<U0 + 0>(mysc); sc[1] = group8_bcast<U0 + 1>(mysc);                     \
            sc[2] = group8_bcast<U0 + 2>(mysc); sc[3] = group8_bcast<U0 + 3>(mysc);                     \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                             \
                const float* xr = x + (int64_t)(GV == 6 ? (jj[u] & 1) : jj[u]) * ldx + c0;              \
                xa[u] = ld4(xr);                                                                        \
                xb[u] = ld4(xr + 4);                                                                    \
            }                                                                                           \
            _Pragma("unroll") for (int kk = 0; kk < (GV == 8 ? 1 : K); ++kk) {                          \
                const float h0 = group8_bcast<U0 + 0>(myh[kk]), h1 = group8_bcast<U0 + 1>(myh[kk]);     \
                acc0[kk] = QOT_ACC(FIRST, h0, xa[0], acc0[kk]); acc1[kk] = QOT_ACC(FIRST, h0, xb[0], acc1[kk]); \
                acc0[kk] = fma4(h1, xa[1], acc0[kk]); acc1[kk] = fma4(h1, xb[1], acc1[kk]);             \
            }                                                                                           \
            if (GV == 8 && (FIRST)) {                                                                   \
                _Pragma("unroll") for (int kk = 1; kk < K; ++kk) { acc0[kk] = f4zero(); acc1[kk] = f4zero(); } \
            }                                                                                           \
            acc0[K] = QOT_ACC(FIRST, sc[0], xa[0], acc0[K]); acc1[K] = QOT_ACC(FIRST, sc[0], xb[0], acc1[K]); \
            acc0[K] = fma4(sc[1], xa[1], acc0[K]); acc1[K] = fma4(sc[1], xb[1], acc1[K]);               \
            /* four rows stay in flight, the vector work is skipped in pairs (a wave runs as many slots */ \
            /* as its highest-degree row needs, and VALU time is MFMA time on this part)                */ \
            if (cnt > U0 + 2) {                                                                         \
                _Pragma("unroll") for (int kk = 0; kk < (GV == 8 ? 1 : K); ++kk) {                      \
                    const float h2 = group8_bcast<U0 + 2>(myh[kk]), h3 = group8_bcast<U0 + 3>(myh[kk]); \
                    acc0[kk] = fma4(h2, xa[2], acc0[kk]); acc1[kk] = fma4(h2, xb[2], acc1[kk]);         \
                    acc0[kk] = fma4(h3, xa[3], acc0[kk]); acc1[kk] = fma4(h3, xb[3], acc1[kk]);         \
                }                                                                                       \
                _Pragma("unroll") for (int u = 2; u < 4; ++u) {                                         \
                    acc0[K] = fma4(sc[u], xa[u], acc0[K]);                                              \
                    acc1[K] = fma4(sc[u], xb[u], acc1[K]);                                              \
                }                                                                                       \
            }                                                                                           \
        }
#define QOT_FWD_BATCH(FIRST)                                                                            \
    {                                                                                                   \
        const int p = base + sub;                                                                       \
        int myj = own;                                                                                  \
        float myh[K], mysc = 0.f;                                                                       \
        _Pragma("unroll") for (int kk = 0; kk < K; ++kk) myh[kk] = 0.f;                                 \
        if (p < end) {                                                                                  \
            myj = (GV == 7) ? (int)(i ^ (p & 31)) : col[p];                                             \
            const int64_t e = (GV == 7) ? (int64_t)p : (int64_t)eidx[p];                                \
            float ee[D];                                                                                \
            _Pragma("unroll") for (int d = 0; d < D; ++d) ee[d] = ea[e * D + d];                        \
            mysc = TRANSPOSE ? invdeg[myj] : srow;                                                      \
            _Pragma("unroll") for (int kk = 0; kk < K; ++kk) {                                          \
                float h = b1[kk];                                                                       \
                _Pragma("unroll") for (int d = 0; d < D; ++d) h = fmaf(w1[kk * D + d], ee[d], h);       \
                myh[kk] = fmaxf(h, 0.f) * mysc;                                                         \
            }                                                                                           \
        }                                                                                               \
        const int cnt = (end - base < 8) ? end - base : 8;                                              \
        QOT_EDGE4(0, FIRST)                                                                             \
        if (cnt > 4) QOT_EDGE4(4, false)                                                                \
    }
    int base = beg;
    QOT_FWD_BATCH(true)
    for (base = beg + 8; base < end; base += 8) QOT_FWD_BATCH(false)
#undef QOT_FWD_BATCH
#undef QOT_EDGE4
#undef QOT_ACC
    float4* At4 = reinterpret_cast<float4*>(At);
#pragma unroll
    for (int kk = 0; kk <= K; ++kk) {       // channels c0+{0,2,4,6} -> hi 0 ; c0+{1,3,5,7} -> hi 1
        const int g = kk * 8 + sub;
        At4[at4_slot(g, 0, il)] = make_float4(acc0[kk].x, acc0[kk].z, acc1[kk].x, acc1[kk].z);
        At4[at4_slot(g, 1, il)] = make_float4(acc0[kk].y, acc0[kk].w, acc1[kk].y, acc1[kk].w);
    }
}

#ifdef QOT_DIAG
#define QOT_DIAG_SECTION 1
#include "diag/nnconv_mfma_diag.inc"
#undef QOT_DIAG_SECTION
#endif

// in-kernel cycle stamps: diagnostic build only (make DIAG=1); nothing in the release build
#ifdef QOT_DIAG
#define QOT_DIAG_SECTION 2
#include "diag/nnconv_mfma_diag.inc"
#undef QOT_DIAG_SECTION
#else
#define QOT_STAMP(slot)
#endif

template <int D, bool TRANSPOSE, int VARIANT = 0>
__global__ __launch_bounds__(256, 2) void nnconv_mfma64_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ ea, const float* __restrict__ w1,
    const float* __restrict__ b1, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const int32_t* __restrict__ eidx, const float* __restrict__ invdeg, const float* __restrict__ Wp,
    const float* __restrict__ bias, float* __restrict__ out, int64_t N, ActParams act) {
    constexpr int K = 2 * D;
    constexpr int KM = (K + 1) * 64;        // inner dimension held in LDS (blocks 0..K)
    constexpr int GM = KM / 16;             // float4 B groups per wave for the main part
    constexpr int CH = 4;                       // prefetch chunk (GM = 8D+4 is divisible by 4); 6 spilled inside the MFMA loop
    __shared__ __attribute__((aligned(16))) float At[KM * 32];
    const int64_t ntiles = (N + 31) / 32;
    // read once per kernel: inside the tile loop each was a load consumed on the spot (an L2 round trip per tile)
    const float bz = bias ? bias[((threadIdx.x >> 6) & 1) * 32 + (threadIdx.x & 31)] : 0.f;
    const uint64_t drop_step = (act.enabled && act.thr16) ? (uint64_t)act.step[0] : 0;
    // persistent: 2 workgroups per CU walk the tiles; the two CU-mates drift out of phase so
    // one gathers while the other owns the MFMA pipes
#pragma unroll 1
    for (int64_t it = 0;; ++it) {
    const int64_t tile = xcd_tile(it, ntiles);
    if (tile < 0) break;
    const int64_t tile0 = tile * 32;
    float4 root0, root1;
    unsigned long long t_prev = 0;
    if (VARIANT == 3 || VARIANT == 9) t_prev = __builtin_amdgcn_s_memtime();

#ifdef QOT_DIAG
    if (VARIANT == 11) {
        nnconv_gather_tile_rank1_shape<D>(At, x, ldx, rowptr, col, tile0, N, root0, root1);
    } else
#endif
    if (VARIANT != 1 && VARIANT != 4 && VARIANT != 5 && VARIANT != 9) {
        nnconv_gather_tile<D, TRANSPOSE, (VARIANT >= 6 && VARIANT <= 8) ? VARIANT : 0>(At, x, ldx, ea, w1, b1, rowptr, col, eidx, invdeg, tile0, N, root0, root1);
    } else {
        for (int t = threadIdx.x; t < KM * 32; t += 256) At[t] = 1.0f + (float)(t & 7);
        root0 = root1 = make_float4(1.f, 1.f, 1.f, 1.f);
    }
    const float4* At4 = reinterpret_cast<const float4*>(At);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nh = wave & 1, kh = wave >> 1;
    const int r31 = lane & 31, hi = lane >> 5;
    // Wp: [nh][group g of 4 k-steps][lane][4]; wave (nh, kh) owns main groups kh*GM.. and
    // root groups (2*GM + kh*4)..
    const float4* wpn = reinterpret_cast<const float4*>(Wp) + (int64_t)nh * (2 * GM + 8) * 64 + lane;
    const float4* wp = wpn + (int64_t)kh * GM * 64;
    // one base pointer + immediate offsets (as separate expressions the four addresses were materialised
    // outside the tile loop, spilled, and reloaded from scratch -- with a full wait each -- every tile)
    const float4* rbp = wpn + (int64_t)(2 * GM + kh * 4) * 64;
    asm volatile("" : "+v"(rbp));          // opaque: one live base per tile, the four offsets stay immediates
    // The first weight fragments of the tile (root block + three chunks) are requested BEFORE the barrier that
    // publishes the operand tile, and that barrier orders LDS traffic only: requested behind a __syncthreads() their
    // L2 round trip was exposed once per tile, now it runs while the wave waits for the slowest gatherer.
    float4 rb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) rb[u] = rbp[u * 64];
    // B fragments are requested NB - 1 chunks (48 MFMAs, ~3000 cycles) ahead, NB buffers rotating with
    // compile-time roles: an L2 round trip is then covered by this wave's own MFMAs, so a workgroup keeps
    // the pipe fed while its CU-mate is gathering (with one chunk of lookahead the MFMA phase only ran at
    // full rate when both workgroups were in it).
    constexpr int NCH = GM / CH;
    constexpr int NB = 4;                      // buffers: fragments are requested NB - 1 chunks ahead
    float4 bb[NB][CH];
#pragma unroll
    for (int q = 0; q < NB - 1; ++q)
#pragma unroll
        for (int u = 0; u < CH; ++u) bb[q][u] = wp[(q * CH + u) * 64];
    QOT_STAMP(0)
    lds_barrier();
    QOT_STAMP(1)
    if (VARIANT == 2) {   // ablation: gather only
        if (threadIdx.x < 32 && tile0 + threadIdx.x < N) out[(tile0 + threadIdx.x) * 64] = At[threadIdx.x * 33] + root0.x + rb[0].x + bb[0][0].x;
        __syncthreads();
        continue;
    }

    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    int ch = 0;
    // (r04, measured: the A fragment of group L + 1 read from LDS before the four MFMAs of group L instead of right in front
    //  of its use -- `ds_read_b128; s_waitcnt lgkmcnt(0)` before every group as it stands -- 96.0 -> 96.9 us: no gain, not kept)
#pragma unroll 1
    for (; ch + NB <= NCH; ch += NB) {
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            if (ch + q + NB - 1 < NCH && VARIANT != 4 && VARIANT != 5 && VARIANT != 9) {       // 4 / 5: weight fragments not streamed
#pragma unroll
                for (int u = 0; u < CH; ++u) bb[(q + NB - 1) % NB][u] = wp[((ch + q + NB - 1) * CH + u) * 64];
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                if (VARIANT == 5 || VARIANT == 9) {                            // 5 / 9: operand tile not read either
                    const float4 a = rb[u & 3], b = bb[q][u];
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, c, 0, 0, 0);
                } else {
                    c = mfma_group(At4, kh * GM + (ch + q) * CH + u, hi, r31, bb[q][u], c);
                }
            }
        }
    }
    // tail: chunks ch .. NCH-1 are already in buffers 0 .. (their loads were issued above)
#pragma unroll
    for (int q = 0; q < NB - 1; ++q) {
        if (ch + q < NCH) {
#pragma unroll
            for (int u = 0; u < CH; ++u) c = mfma_group(At4, kh * GM + (ch + q) * CH + u, hi, r31, bb[q][u], c);
        }
    }
    QOT_STAMP(2)
    lds_barrier();                         // everyone is done with blocks 0..K
    QOT_STAMP(3)
    {   // root block (x_i itself) reuses block 0's slots
        float4* At4 = reinterpret_cast<float4*>(At);
        const int sub = threadIdx.x & 7, il = threadIdx.x >> 3;
        At4[at4_slot(sub, 0, il)] = make_float4(root0.x, root0.z, root1.x, root1.z);
        At4[at4_slot(sub, 1, il)] = make_float4(root0.y, root0.w, root1.y, root1.w);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) c = mfma_group(At4, kh * 4 + u, hi, r31, rb[u], c);
    QOT_STAMP(4)
    // K halves meet through LDS; every wave finishes 8 of the 16 accumulator registers of its
    // (column half), so the stores are spread over all four waves
    float* red = At + 64 * 32;             // disjoint from block 0, which other waves may still read
    // (compile-time register indices in both branches: a runtime index into the accumulator
    //  vector would be lowered through scratch)
    if (kh) {
#pragma unroll
        for (int r = 0; r < 8; ++r) red[((2 + nh) * 8 + r) * 64 + lane] = c[r];
    } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) red[(nh * 8 + r) * 64 + lane] = c[8 + r];
    }
    lds_barrier();
    {
        const int colg = nh * 32 + r31;
        float v[8];
        if (kh) {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = c[8 + r] + red[(nh * 8 + r) * 64 + lane] + bz;
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = c[r] + red[((2 + nh) * 8 + r) * 64 + lane] + bz;
        }
        // Dropout draws: one 64-bit hash serves the four columns of a float4 group, i.e. the four lanes of
        // a quad for the same row.  Each lane hashes two of its eight rows and the quad shares them through
        // quad_perm broadcasts (2 hashes + 12 DPP moves per lane instead of 8 hashes: the 64-bit
        // multiplies of the hash were ~5 % of this kernel's VALU time).
        uint32_t zlo[8], zhi[8];
        const bool drop = act.enabled && act.thr16;
        if (drop) {
            const int jq = r31 & 3;
            uint32_t mylo[2], myhi[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int rr = kh * 8 + 2 * jq + t;
                const int row = (rr & 3) + 8 * (rr >> 2) + 4 * hi;
                const uint64_t flat = (uint64_t)((tile0 + row) * 64 + colg);
                const uint64_t z = act_hash64(act.seed, drop_step, flat >> 2);
                mylo[t] = (uint32_t)z; myhi[t] = (uint32_t)(z >> 32);
            }
#define QOT_ZSHARE(S)                                                                                   \
            zlo[2 * (S)] = __builtin_amdgcn_update_dpp(0, mylo[0], (S) * 0x55, 0xF, 0xF, true);         \
            zhi[2 * (S)] = __builtin_amdgcn_update_dpp(0, myhi[0], (S) * 0x55, 0xF, 0xF, true);         \
            zlo[2 * (S) + 1] = __builtin_amdgcn_update_dpp(0, mylo[1], (S) * 0x55, 0xF, 0xF, true);     \
            zhi[2 * (S) + 1] = __builtin_amdgcn_update_dpp(0, myhi[1], (S) * 0x55, 0xF, 0xF, true);
            QOT_ZSHARE(0) QOT_ZSHARE(1) QOT_ZSHARE(2) QOT_ZSHARE(3)
#undef QOT_ZSHARE
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int rr = kh * 8 + r;
            const int row = (rr & 3) + 8 * (rr >> 2) + 4 * hi;
            const int64_t i = tile0 + row;
            float y = v[r];
            if (act.enabled) {
                y = y > 0.f ? y : act.slope * y;
                if (drop) {
                    const uint64_t z = ((uint64_t)zhi[r] << 32) | zlo[r];
                    const bool keep = ((uint32_t)(z >> (16 * (colg & 3))) & 0xFFFFu) >= act.thr16;
                    y = keep ? y * act.keep_scale : 0.f;
                }
            }
            if (i < N) out[i * 64 + colg] = y;
        }
    }
    QOT_STAMP(5)
    lds_barrier();                         // `red` is consumed before the next tile's gather overwrites it
    QOT_STAMP(6)
    }
}

// ---------------------------------------------------------------------------------------------
// Backward of NNConv, data AND weight gradients from ONE gather  (H = 64).
//   U_j = [ sum_{e: j->i} h_e[k] invdeg_i g_i (k<K) | sum_e invdeg_i g_i | g_j ]     (adjoint tile)
//   grad_x_j  = U_j @ WcatT                         (as nnconv_mfma64<TRANSPOSE>)
//   gWcat     = sum_j x_j (x) U_j  ("X^T U")        -- grad of nn.2.weight / nn.2.bias / lin.weight
// The separate path re-gathered the forward operand A ([N,640] = 262 MB written, then read by a
// split-K GEMM).  Here the U tile that the adjoint needs anyway is multiplied twice: rows as the
// A operand (grad_x) and, read transposed from the same LDS image, against the tile's own x rows
// (weight gradient).  512 threads = 8 waves, one workgroup per CU: every wave keeps 5 of the 40
// 32x32 accumulator tiles of gWcat^T across the persistent loop (80 VGPRs), partials go to one slab
// per workgroup and are summed in a fixed order afterwards (bitwise reproducible).

// VARIANT (diagnostic build only): 0 production; 1 every gathered row read from ONE hot address (no row-load latency);
// 2 no grad_x loop; 3 no weight-gradient loop; 4 no gather; 5 neither MFMA loop (gather + exchange only)
template <int D, int VARIANT = 0>
__global__ __launch_bounds__(512, 2) void nnconv_adjoint_dw64_kernel(
    const float* __restrict__ g, int ldg, const float* __restrict__ xf, int ldx, const float* __restrict__ ea,
    const float* __restrict__ w1, const float* __restrict__ b1, const int32_t* __restrict__ rowptr_t,
    const int32_t* __restrict__ col_t, const int32_t* __restrict__ eid_t, const float* __restrict__ invdeg,
    const float* __restrict__ Wp, float* __restrict__ gx, float* __restrict__ slabs, int64_t N) {
    constexpr int K = 2 * D;
    constexpr int KT = (K + 2) * 64;        // all K+2 blocks live in LDS (one workgroup per CU: room)
    constexpr int GQ = KT / 8 / 4;          // float4 groups per K quarter (20 at D=4)
    constexpr int MB = KT / 32;             // 32-row blocks of gWcat^T (20)
    constexpr int TPW = MB * 2 / 8;         // dW tiles per wave (5)
    static_assert(GQ % 4 == 0 || GQ % 5 == 0, "chunking");
    constexpr int CH = (GQ % 4 == 0) ? 4 : 5;
    __shared__ __attribute__((aligned(16))) float Ut[KT * 32];
    __shared__ __attribute__((aligned(16))) float red[8 * 3 * 4 * 64];
    __shared__ __attribute__((aligned(16))) float xs[32 * 64];      // the tile's own x rows (B operand of X^T U)
    float4* Ut4 = reinterpret_cast<float4*>(Ut);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r31 = lane & 31, hi = lane >> 5;
    const int nh = wave & 1, kq = wave >> 1;                 // grad_x: column half x K quarter
    const int ah = wave & 1, mb0 = wave >> 1;                // dW: x-channel half, row blocks mb0 + 4t
    const int64_t ntiles = (N + 31) / 32;

    f32x16 dw[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dw[t][r] = 0.f;
    // Transposed-read addressing of the U tile for this lane's dW rows (kcol = 32*(mb0 + 4t) + r31).
    // Float index of U[row jj][kcol] = base(t) + ((jj ^ m) << 2) with base(t) = base(0) + 4096 t and the
    // XOR mask m = (4 mb0 + (r31 >> 3)) & 7 the same for every t.  With jj = 8a + 2b + hi the XOR only
    // touches b and hi: (jj ^ m) << 2 = 32 a + ((((b ^ (m >> 1)) << 1) | (hi ^ (m & 1))) << 2), so four
    // per-lane offsets (b = 0..3) plus compile-time immediates (a, t) address all 80 reads of a tile.
    int dwoff[4];
    {
        const int kcol = mb0 * 32 + r31;
        const int gg = kcol >> 3, within = kcol & 7;
        const int base0 = ((2 * gg + (within & 1)) * 32) * 4 + (within >> 1);
        const int m = gg & 7;
#pragma unroll
        for (int b = 0; b < 4; ++b) dwoff[b] = base0 + ((((b ^ (m >> 1)) << 1) | (hi ^ (m & 1))) << 2);
    }
    const float4* wp = reinterpret_cast<const float4*>(Wp) + ((int64_t)nh * (KT / 8) + kq * GQ) * 64 + lane;
    // (r04, measured and not kept, tools/ab_adjoint.py, interleaved rounds in one process: static `s_setprio 1` for waves 4-7
    // -- 172.7 us either way; the half index folded into the tile's XOR swizzle, i ^ ((g + 4 hi) & 7), which makes the
    // transposed reads of the X^T U loop conflict-free -- 171.9 us either way: that loop does not wait for LDS bandwidth)
#ifdef QOT_ADJ_SETPRIO
    // static priority for the second-dispatched half of the workgroup (waves 4-7 share their SIMDs with waves 0-3 and lose
    // the age-based arbitration on every phase): MI355X_MICROARCH.md, "Two waves per SIMD", item 4
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif

#pragma unroll 1
    for (int64_t it = 0;; ++it) {
        const int64_t tile = xcd_tile(it, ntiles);
        if (tile < 0) break;
        const int64_t tile0 = tile * 32;
        // the tile's own x rows are requested BEFORE the gather and written to LDS behind it (loaded behind the
        // gather, as first written, the load was one exposed memory round trip per tile)
        float4 xown;
        {
            const int64_t j = tile0 + (threadIdx.x >> 4);
            xown = (j < N) ? ld4(xf + j * ldx + 4 * (threadIdx.x & 15)) : f4zero();
        }
        // ---- gather: 16 lanes per source node j (float4 = 64 channels), out-edges over the CSC
        if (VARIANT != 4) {
            const int sub = threadIdx.x & 15, il = threadIdx.x >> 4;
            const int c0 = 4 * sub;
            const int64_t j = tile0 + il;
            float4 acc[K + 2];          // blocks 0..K are WRITTEN by the first edge slot: no zero fill per tile
            acc[K + 1] = f4zero();
            int beg = 0, end = 0;
            if (j < N) { beg = rowptr_t[j]; end = rowptr_t[j + 1]; acc[K + 1] = ld4(g + j * ldg + c0); }
            const int own = (j < N) ? (int)j : 0;   // dead slots read the node's own g row (times zero), never another node's
            // One batch = up to 16 out-edges of every node, one edge per lane of the row (zeros past the end); broadcasts with
            // compile-time source lanes are one DPP move each (row_newbcast) -- with a runtime lane (__shfl) every one of the
            // ten per edge was a ds_bpermute_b32 + s_waitcnt round trip.  FIRST: the batch every node runs (also one without
            // out-edges); its first slot writes the accumulators.
#define QOT_ADJ_EDGE(U)                                                                                  \
                {                                                                                        \
                    const int64_t i = (VARIANT == 1) ? (int64_t)(row16_bcast<U>(myi) & 1) : (int64_t)row16_bcast<U>(myi); \
                    gr[(U) & 3] = ld4(g + i * ldg + c0);                                                 \
                    sc[(U) & 3] = row16_bcast<U>(mysc);                                                  \
                }
#define QOT_ADJ_FMA(U, FIRSTSLOT)                                                                        \
                {                                                                                        \
                    _Pragma("unroll") for (int kk = 0; kk < (VARIANT == 6 ? 1 : K); ++kk)                \
                        acc[kk] = (FIRSTSLOT) ? scale4(row16_bcast<U>(myh[kk]), gr[(U) & 3])             \
                                              : fma4(row16_bcast<U>(myh[kk]), gr[(U) & 3], acc[kk]);     \
                    if (VARIANT == 6 && (FIRSTSLOT)) {                                                   \
                        _Pragma("unroll") for (int kk = 1; kk < K; ++kk) acc[kk] = f4zero();             \
                    }                                                                                    \
                    acc[K] = (FIRSTSLOT) ? scale4(sc[(U) & 3], gr[(U) & 3]) : fma4(sc[(U) & 3], gr[(U) & 3], acc[K]); \
                }
#define QOT_ADJ_EDGE4(U0, FIRST)                                                                         \
                {                                                                                        \
                    float4 gr[4];                                                                        \
                    float sc[4];                                                                         \
                    QOT_ADJ_EDGE(U0) QOT_ADJ_EDGE(U0 + 1) QOT_ADJ_EDGE(U0 + 2) QOT_ADJ_EDGE(U0 + 3)      \
                    QOT_ADJ_FMA(U0, FIRST) QOT_ADJ_FMA(U0 + 1, false)                                    \
                    /* four rows stay in flight, but the vector work is skipped in pairs: a wave runs as */ \
                    /* many slots as its highest-degree row needs, and VALU time is MFMA time here      */ \
                    if (cnt > U0 + 2) { QOT_ADJ_FMA(U0 + 2, false) QOT_ADJ_FMA(U0 + 3, false) }          \
                }
#define QOT_ADJ_BATCH(FIRST)                                                                             \
            {                                                                                            \
                const int p = base + sub;                                                                \
                int myi = own;                                                                           \
                float myh[K], mysc = 0.f;                                                                \
                _Pragma("unroll") for (int kk = 0; kk < K; ++kk) myh[kk] = 0.f;                          \
                if (p < end) {                                                                           \
                    myi = (VARIANT == 7) ? (int)(j ^ (p & 31)) : col_t[p];      /* 7: arithmetic ids */  \
                    const int64_t e = (VARIANT == 7) ? (int64_t)p : (int64_t)eid_t[p];                   \
                    float ee[D];                                                                         \
                    _Pragma("unroll") for (int d = 0; d < D; ++d) ee[d] = ea[e * D + d];                 \
                    mysc = (VARIANT == 7) ? 0.25f : invdeg[myi];                                         \
                    _Pragma("unroll") for (int kk = 0; kk < K; ++kk) {                                   \
                        float h = b1[kk];                                                                \
                        _Pragma("unroll") for (int d = 0; d < D; ++d) h = fmaf(w1[kk * D + d], ee[d], h); \
                        myh[kk] = fmaxf(h, 0.f) * mysc;                                                  \
                    }                                                                                    \
                }                                                                                        \
                const int cnt = (end - base < 16) ? end - base : 16;                                     \
                QOT_ADJ_EDGE4(0, FIRST)                                                                  \
                if (cnt > 4) QOT_ADJ_EDGE4(4, false)                                                     \
                if (cnt > 8) QOT_ADJ_EDGE4(8, false)                                                     \
                if (cnt > 12) QOT_ADJ_EDGE4(12, false)                                                   \
            }
            int base = beg;
            QOT_ADJ_BATCH(true)
            for (base = beg + 16; base < end; base += 16) QOT_ADJ_BATCH(false)
#undef QOT_ADJ_BATCH
#undef QOT_ADJ_EDGE4
#undef QOT_ADJ_FMA
#undef QOT_ADJ_EDGE
            // channels c0..c0+3 of block kk: k = 64 kk + c0 + t -> group 8 kk + sub/2,
            // (r, hi) = (2 (sub&1) + t/2, t&1): two 8-byte stores per block, conflict-free
#pragma unroll
            for (int kk = 0; kk < K + 2; ++kk) {
                const int gg = kk * 8 + (sub >> 1);
                float2* s0 = reinterpret_cast<float2*>(&Ut4[at4_slot(gg, 0, il)]) + (sub & 1);
                float2* s1 = reinterpret_cast<float2*>(&Ut4[at4_slot(gg, 1, il)]) + (sub & 1);
                *s0 = make_float2(acc[kk].x, acc[kk].z);
                *s1 = make_float2(acc[kk].y, acc[kk].w);
            }
        }
        // x rows of the tile's own nodes (B operand of the weight-gradient product) go to LDS next to the U
        // tile (16 lanes per row, one float4 each): holding them in 16 VGPRs through both MFMA loops was
        // part of what pushed this kernel into scratch.
        {
            const int sub = threadIdx.x & 15, il = threadIdx.x >> 4;
            *reinterpret_cast<float4*>(&xs[il * 64 + 4 * sub]) = xown;
        }
        // WcatT fragments: two buffers used alternately, two chunks per trip of a rolled loop.  Written as
        // `bc = bn` copies the compiler put `s_waitcnt vmcnt(0)` at the top of every chunk, i.e. waited for
        // the prefetch it had just issued (fully unrolled it spilled instead).
        constexpr int NCH = GQ / CH;
        float4 bc[CH], bn[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) bc[u] = wp[u * 64];
        lds_barrier();
        // ---- grad_x: this wave's K quarter of U @ WcatT
        f32x16 c;
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
        if (VARIANT != 2 && VARIANT != 5) {
            int ch = 0;
#pragma unroll 1
            for (; ch + 1 < NCH; ch += 2) {
#pragma unroll
                for (int u = 0; u < CH; ++u) bn[u] = wp[((ch + 1) * CH + u) * 64];
#pragma unroll
                for (int u = 0; u < CH; ++u) c = mfma_group(Ut4, kq * GQ + ch * CH + u, hi, r31, bc[u], c);
                if (ch + 2 < NCH) {
#pragma unroll
                    for (int u = 0; u < CH; ++u) bc[u] = wp[((ch + 2) * CH + u) * 64];
                }
#pragma unroll
                for (int u = 0; u < CH; ++u) c = mfma_group(Ut4, kq * GQ + (ch + 1) * CH + u, hi, r31, bn[u], c);
            }
            if (NCH & 1) {
#pragma unroll
                for (int u = 0; u < CH; ++u) c = mfma_group(Ut4, kq * GQ + (NCH - 1) * CH + u, hi, r31, bc[u], c);
            }
        }
        // (published BEFORE the weight-gradient loop so that the 16 accumulator registers are dead there)
        // ---- K quarters of grad_x meet through LDS: quarter q finishes accumulator registers
        // 4q..4q+3 of its column half (hands the other 12 over), so all 8 waves share the stores.
        // red[((owner*2 + nh)*3 + slot)*4 + rr][lane], slot = which of the 3 other quarters wrote it
#define QOT_GIVE(OWNER, SLOT)                                                                        \
        _Pragma("unroll") for (int rr = 0; rr < 4; ++rr)                                             \
            red[((((OWNER) * 2 + nh) * 3 + (SLOT)) * 4 + rr) * 64 + lane] = c[4 * (OWNER) + rr];
        if (kq == 0) { QOT_GIVE(1, 0) QOT_GIVE(2, 0) QOT_GIVE(3, 0) }
        else if (kq == 1) { QOT_GIVE(0, 0) QOT_GIVE(2, 1) QOT_GIVE(3, 1) }
        else if (kq == 2) { QOT_GIVE(0, 1) QOT_GIVE(1, 1) QOT_GIVE(3, 2) }
        else { QOT_GIVE(0, 2) QOT_GIVE(1, 2) QOT_GIVE(2, 2) }
#undef QOT_GIVE
        // ---- weight gradient: gWcat^T[(k,o)][a] += sum_j U[j][(k,o)] x[j][a]
        // Step s = 4a + b multiplies tile rows 2s, 2s+1.  Software-pipelined by hand: the TPW transposed
        // reads of step s+1 are issued before the TPW MFMAs of step s (left to itself the compiler
        // reused ONE temporary and put `ds_read_b32; s_waitcnt lgkmcnt(0)` in front of every MFMA --
        // half of all MFMAs of this kernel waited for their own LDS read).
        if (VARIANT != 3 && VARIANT != 5) {
            float abuf[2][TPW], xbuf[2];
            const float* xsl = xs + hi * 64 + ah * 32 + r31;          // row 2s + hi -> + 128 s
#pragma unroll
            for (int t = 0; t < TPW; ++t) abuf[0][t] = Ut[dwoff[0] + t * 4096];
            xbuf[0] = xsl[0];
#pragma unroll
            for (int sidx = 0; sidx < 16; ++sidx) {
                const int cur = sidx & 1;
                if (sidx + 1 < 16) {
                    const int sn = sidx + 1;
#pragma unroll
                    for (int t = 0; t < TPW; ++t) abuf[cur ^ 1][t] = Ut[dwoff[sn & 3] + (sn >> 2) * 32 + t * 4096];
                    xbuf[cur ^ 1] = xsl[sn * 128];
                }
                __builtin_amdgcn_sched_barrier(0);      // the reads of step s+1 stay in front of the MFMAs of step s
#pragma unroll
                for (int t = 0; t < TPW; ++t)
                    dw[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(abuf[cur][t], xbuf[cur], dw[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        lds_barrier();
        {
            float v[4];
            if (kq == 0) { v[0] = c[0]; v[1] = c[1]; v[2] = c[2]; v[3] = c[3]; }
            else if (kq == 1) { v[0] = c[4]; v[1] = c[5]; v[2] = c[6]; v[3] = c[7]; }
            else if (kq == 2) { v[0] = c[8]; v[1] = c[9]; v[2] = c[10]; v[3] = c[11]; }
            else { v[0] = c[12]; v[1] = c[13]; v[2] = c[14]; v[3] = c[15]; }
            const int colg = nh * 32 + r31;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
                for (int slot = 0; slot < 3; ++slot) v[rr] += red[(((kq * 2 + nh) * 3 + slot) * 4 + rr) * 64 + lane];
                const int reg = 4 * kq + rr;
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * hi;
                const int64_t j = tile0 + row;
                if (j < N) gx[j * 64 + colg] = v[rr];
            }
        }
        lds_barrier();            // Ut / red are rewritten by the next tile
    }
    // slab[blk][(k,o) row][a]
    float* slab = slabs + (int64_t)blockIdx.x * KT * 64;
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
            slab[((mb0 + 4 * t) * 32 + row) * 64 + ah * 32 + r31] = dw[t][r];
        }
}

// level-1/level-2 slab sums (same scheme as gemm_tn.hip; duplicated signature, internal linkage)
__global__ void adj_slab_reduce_kernel(const float* __restrict__ slabs, int nslabs, int per_group, int64_t elems,
                                       int64_t slab_stride, float* __restrict__ dst, int64_t dst_stride) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t * 4 >= elems) return;
    const int s0 = blockIdx.y * per_group;
    int s1 = s0 + per_group;
    if (s1 > nslabs) s1 = nslabs;
    float4 acc = f4zero();
    int sidx = s0;
    for (; sidx + 8 <= s1; sidx += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ld4(slabs + (int64_t)(sidx + u) * slab_stride + 4 * t);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = add4(acc, v[u]);
    }
    for (; sidx < s1; ++sidx) acc = add4(acc, ld4(slabs + (int64_t)sidx * slab_stride + 4 * t));
    st4(dst + blockIdx.y * dst_stride + 4 * t, acc);
}

// final slab sum written straight into the parameters' own layouts (no permute/copy kernels after):
//   dst = [ g(nn.2.weight)[a*64+o, k] (64*64*K) | g(nn.2.bias)[a*64+o] (64*64) | g(lin.weight)[o, a] (64*64) ]
// from gWcat^T[kb*64 + o][a]
__global__ void adj_slab_final_params_kernel(const float* __restrict__ slabs, int nslabs, int64_t elems,
                                             int64_t slab_stride, float* __restrict__ dst, int K) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t * 4 >= elems) return;
    float4 acc = f4zero();
    for (int sidx = 0; sidx < nslabs; ++sidx) acc = add4(acc, ld4(slabs + (int64_t)sidx * slab_stride + 4 * t));
    const int64_t e = 4 * t;
    const int a0 = (int)(e & 63), o = (int)((e >> 6) & 63), kb = (int)(e >> 12);
    const float v[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int a = a0 + c;
        int64_t idx;
        if (kb < K) idx = ((int64_t)a * 64 + o) * K + kb;
        else if (kb == K) idx = (int64_t)4096 * K + a * 64 + o;
        else idx = (int64_t)4096 * (K + 1) + o * 64 + a;
        dst[idx] = v[c];
    }
}

// ---------------------------------------------------------------------------------------------
// grad of the edge MLP's first layer: nnconv_gradh64_kernel lives in nnconv_gradh64.hip (a translation unit of its own:
// it is built with another machine-scheduler strategy, see the Makefile)
__global__ void gradh_partial_sum_kernel(const float* __restrict__ partials, int nblk, int n, int KD,
                                         float* __restrict__ gw1, float* __restrict__ gb1) {
    const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);      // one wave per output
    if (t >= n) return;
    const float s = wave_sum_partials(partials, nblk, n, t);
    if ((threadIdx.x & 63) == 0) { if (t < KD) gw1[t] = s; else gb1[t - KD] = s; }
}

}  // namespace qot

using namespace qot;

// width-generic kernels (nnconv_gen.hip)
int qot_nnconv_gen_launch(const float* x, int ld_x, const float* edge_attr, const float* w1, const float* b1,
                          const int32_t* rowptr, const int32_t* col, const int32_t* edge_ids, const float* invdeg,
                          int transpose, const float* w_perm, const float* bias, float* out, int64_t N, int H, int D,
                          const ActParams& ap, hipStream_t stream);
int qot_nnconv_gradh_gen_launch(const float* grad_out, int ld_g, const float* x, int ld_x, const float* edge_attr,
                                const float* w1, const float* b1, const int32_t* rowptr, const int32_t* col,
                                const int32_t* eid, const float* invdeg, const float* b_perm, float* gw1, float* gb1,
                                float* workspace, int64_t N, int H, int D, hipStream_t stream);

#ifdef QOT_DIAG
extern "C" int qot_nnconv_fused_ws(const float* x, int ld_x, const float* edge_attr, const float* w1,
                                   const float* b1, const int32_t* rowptr, const int32_t* col,
                                   const int32_t* edge_ids, const float* invdeg, int transpose,
                                   const float* w_perm, const float* bias, float* out, int64_t N, int H, int D,
                                   int act, float act_slope, float act_p, uint64_t act_seed,
                                   const int64_t* act_step, qot_stream_t stream);

static int g_variant = 0;   // ablation switch for tools/ablate_nnconv.py (0 = production)
extern "C" void qot_debug_set_variant(int v) { g_variant = v; }
extern "C" void qot_debug_stamps(unsigned long long* host8, int reset) {
    if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(qot::g_stamps), z, sizeof(z)); }
    else (void)hipMemcpyFromSymbol(host8, HIP_SYMBOL(qot::g_stamps), 8 * sizeof(unsigned long long));
}
#endif

// Wp layout (built by the caller, see functional.nnconv_perm_index): with GT = (K+2)*64/8 groups
// of 4 k-steps, for column half nh, group g, lane l, r in 0..3:
//   Wp[((nh*GT + g)*64 + l)*4 + r] = Wcat[8*g + 2*r + (l>>5)][nh*32 + (l&31)]
// edge_ids[p] = original edge id of slot p of the index being walked (CSR: eid; CSC: eid_t).
extern "C" int qot_nnconv_fused(const float* x, int ld_x, const float* edge_attr, const float* w1,
                                const float* b1, const int32_t* rowptr, const int32_t* col,
                                const int32_t* edge_ids, const float* invdeg, int transpose,
                                const float* w_perm, const float* bias, float* out, int64_t N, int H, int D,
                                int act, float act_slope, float act_p, uint64_t act_seed,
                                const int64_t* act_step, qot_stream_t stream) {
    if (N < 0 || !rowptr) return QOT_ERR_BADARG;
    if (H != 16 && H != 32 && H != 64 && H != 128 && H != 256) return QOT_ERR_UNSUPPORTED;
    if (N == 0) return QOT_OK;
    if (!x || !w1 || !b1 || !invdeg || !w_perm || !out || (ld_x & 3)) return QOT_ERR_BADARG;
    if (H != 64 || transpose == 2 || transpose == 3) {
        // other widths (and, for A/B measurements, H = 64 with transpose = 2 / 3): the per-pass tile kernel
        if (D > 4) return QOT_ERR_UNSUPPORTED;
        return qot_nnconv_gen_launch(x, ld_x, edge_attr, w1, b1, rowptr, col, edge_ids, invdeg, transpose & 1, w_perm,
                                     bias, out, N, H, D, make_act(act, act_slope, act_p, act_seed, act_step),
                                     (hipStream_t)stream);
    }
#ifdef QOT_DIAG
    static int use_ws = -1;
    if (use_ws < 0) { const char* e = getenv("QOT_NNCONV_WS"); use_ws = (e && e[0] == '1') ? 1 : 0; }
    if (use_ws && D <= 4 && !g_variant)
        return qot_nnconv_fused_ws(x, ld_x, edge_attr, w1, b1, rowptr, col, edge_ids, invdeg, transpose, w_perm, bias,
                                   out, N, H, D, act, act_slope, act_p, act_seed, act_step, stream);
#endif
    int grid = grid_for(N, 32);
    if (grid > 2 * num_cus()) grid = 2 * num_cus();
    const ActParams ap = make_act(act, act_slope, act_p, act_seed, act_step);
#ifdef QOT_DIAG
    if (g_variant >= 100 && g_variant < 110 && D == 4 && !transpose) {      // 100 + v: variant v with ONE workgroup per CU
        const int v = g_variant - 100, g1 = grid > num_cus() ? num_cus() : grid;
        if (v == 0)
            nnconv_mfma64_kernel<4, false, 0><<<g1, 256, 0, (hipStream_t)stream>>>(
                x, ld_x, edge_attr, w1, b1, rowptr, col, edge_ids, invdeg, w_perm, bias, out, N, ap);
        else if (v == 1)
            nnconv_mfma64_kernel<4, false, 1><<<g1, 256, 0, (hipStream_t)stream>>>(
                x, ld_x, edge_attr, w1, b1, rowptr, col, edge_ids, invdeg, w_perm, bias, out, N, ap);
        else if (v == 4)
            nnconv_mfma64_kernel<4, false, 4><<<g1, 256, 0, (hipStream_t)stream>>>(
                x, ld_x, edge_attr, w1, b1, rowptr, col, edge_ids, invdeg, w_perm, bias, out, N, ap);
        else if (v == 5)
            nnconv_mfma64_kernel<4, false, 5><<<g1, 256, 0, (hipStream_t)stream>>>(
                x, ld_x, edge_attr, w1, b1, rowptr, col, edge_ids, invdeg, w_perm, bias, out, N, ap);
        else if (v == 9)
            nnconv_mfma64_kernel<4, false, 9><<<g1, 256, 0, (hipStream_t)stream>>>(
                x, ld_x, edge_attr, w1, b1, rowptr, col, edge_ids, invdeg, w_perm, bias, out, N, ap);
        else
            nnconv_mfma64_kernel<4, false, 2><<<g1, 256, 0, (hipStream_t)stream>>>(
                x, ld_x, edge_attr, w1, b1, rowptr, col, edge_ids, invdeg, w_perm, bias, out, N, ap);
        QOT_LAUNCH_CHECK();
        return QOT_OK;
    }
    if (g_variant && g_variant <= 10 && D == 4 && !transpose) {
        if (g_variant == 3)
            nnconv_mfma64_kernel<4, false, 3><<<grid, 256, 0, (hipStream_t)stream>>>(
                x, ld_x, edge_attr, w1, b1, rowptr, col, edge_ids, invdeg, w_perm, bias, out, N, ap);
        else if (g_variant == 1)
            nnconv_mfma64_kernel<4, false, 1><<<grid, 256, 0, (hipStream_t)stream>>>(
                x, ld_x, edge_attr, w1, b1, rowptr, col, edge_ids, invdeg, w_perm, bias, out, N, ap);
        else if (g_variant == 4)
            nnconv_mfma64_kernel<4, false, 4><<<grid, 256, 0, (hipStream_t)stream>>>(
                x, ld_x, edge_attr, w1, b1, rowptr, col, edge_ids, invdeg, w_perm, bias, out, N, ap);
        else if (g_variant == 5)
            nnconv_mfma64_kernel<4, false, 5><<<grid, 256, 0, (hipStream_t)stream>>>(
                x, ld_x, edge_attr, w1, b1, rowptr, col, edge_ids, invdeg, w_perm, bias, out, N, ap);
        else if (g_variant == 6)
            nnconv_mfma64_kernel<4, false, 6><<<grid, 256, 0, (hipStream_t)stream>>>(
                x, ld_x, edge_attr, w1, b1, rowptr, col, edge_ids, invdeg, w_perm, bias, out, N, ap);
        else if (g_variant == 7)
            nnconv_mfma64_kernel<4, false, 7><<<grid, 256, 0, (hipStream_t)stream>>>(
                x, ld_x, edge_attr, w1, b1, rowptr, col, edge_ids, invdeg, w_perm, bias, out, N, ap);
        else if (g_variant == 8)
            nnconv_mfma64_kernel<4, false, 8><<<grid, 256, 0, (hipStream_t)stream>>>(
                x, ld_x, edge_attr, w1, b1, rowptr, col, edge_ids, invdeg, w_perm, bias, out, N, ap);
        else if (g_variant == 9)
            nnconv_mfma64_kernel<4, false, 9><<<grid, 256, 0, (hipStream_t)stream>>>(
                x, ld_x, edge_attr, w1, b1, rowptr, col, edge_ids, invdeg, w_perm, bias, out, N, ap);
        else if (g_variant == 10)
            nnconv_mfma64_kernel<4, false, 11><<<grid, 256, 0, (hipStream_t)stream>>>(
                x, ld_x, edge_attr, w1, b1, rowptr, col, edge_ids, invdeg, w_perm, bias, out, N, ap);
        else
            nnconv_mfma64_kernel<4, false, 2><<<grid, 256, 0, (hipStream_t)stream>>>(
                x, ld_x, edge_attr, w1, b1, rowptr, col, edge_ids, invdeg, w_perm, bias, out, N, ap);
        QOT_LAUNCH_CHECK();
        return QOT_OK;
    }
#endif
    QOT_DISPATCH_D(D, {
        if (transpose)
            nnconv_mfma64_kernel<kD, true><<<grid, 256, 0, (hipStream_t)stream>>>(
                x, ld_x, edge_attr, w1, b1, rowptr, col, edge_ids, invdeg, w_perm, bias, out, N, ap);
        else
            nnconv_mfma64_kernel<kD, false><<<grid, 256, 0, (hipStream_t)stream>>>(
                x, ld_x, edge_attr, w1, b1, rowptr, col, edge_ids, invdeg, w_perm, bias, out, N, ap);
    });
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

// Bp layout: for 32-column block nb of GA (column n = k*64 + a), group gq of 4 k-steps over o,
// lane l, r:  Bp[((nb*8 + gq)*64 + l)*4 + r] = W2[a*64 + o, k]  with o = 8*gq + 2*r + (l>>5),
// n = nb*32 + (l&31)  (built by functional.nnconv_gradh_perm_index).
int qot_nnconv_gradh64_launch(const float* grad_out, int ld_g, const float* x, int ld_x, const float* edge_attr, const float* w1,
                              const float* b1, const int32_t* rowptr, const int32_t* col, const int32_t* eid, const float* invdeg,
                              const float* b_perm, float* workspace, int64_t N, int D, int grid, int variant, hipStream_t stream);
extern "C" size_t qot_nnconv_gradh_workspace_floats(int D) {
    return (size_t)2 * 256 * 2 * (size_t)(2 * D * (D + 1));
}

extern "C" int qot_nnconv_gradh_fused(const float* grad_out, int ld_g, const float* x, int ld_x,
                                      const float* edge_attr, const float* w1, const float* b1,
                                      const int32_t* rowptr, const int32_t* col, const int32_t* eid,
                                      const float* invdeg, const float* b_perm, float* gw1, float* gb1,
                                      float* workspace, int64_t N, int H, int D, qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N < 0 || !rowptr) return QOT_ERR_BADARG;
    if ((H != 16 && H != 32 && H != 64 && H != 128 && H != 256) || D > 4) return QOT_ERR_UNSUPPORTED;
    if (!grad_out || !x || !w1 || !b1 || !invdeg || !b_perm || (!gw1 != !gb1) || !workspace || (ld_g & 3) || (ld_x & 3))
        return QOT_ERR_BADARG;
    if (!gw1 && H != 64) return QOT_ERR_UNSUPPORTED;
#ifdef QOT_DIAG
    if (H == 64 && g_variant == 9)       // A/B: the generic kernel at H = 64 (b_perm in ITS layout)
        return qot_nnconv_gradh_gen_launch(grad_out, ld_g, x, ld_x, edge_attr, w1, b1, rowptr, col, eid, invdeg, b_perm,
                                           gw1, gb1, workspace, N, H, D, stream);
#endif
    if (H != 64)     // other widths: GA built 32 input channels at a time (nnconv_gen.hip)
        return qot_nnconv_gradh_gen_launch(grad_out, ld_g, x, ld_x, edge_attr, w1, b1, rowptr, col, eid, invdeg, b_perm,
                                           gw1, gb1, workspace, N, H, D, stream);
    int grid = grid_for(N > 0 ? N : 1, 32);
    if (grid > 2 * num_cus()) grid = 2 * num_cus();
    const int K = 2 * D;
    int variant = 0;
#ifdef QOT_DIAG
    if (g_variant >= 31 && g_variant <= 35 && D == 4) variant = g_variant - 30;
#endif
    {
        const int rc = qot_nnconv_gradh64_launch(grad_out, ld_g, x, ld_x, edge_attr, w1, b1, rowptr, col, eid, invdeg, b_perm,
                                                 workspace, N, D, grid, variant, stream);
        if (rc != QOT_OK) return rc;
    }
    QOT_LAUNCH_CHECK();
    if (!gw1) return QOT_OK;          // partials left in the workspace for qot_nnconv_bwd_finalize
    const int n = K * (D + 1);
    gradh_partial_sum_kernel<<<grid_for(n, 4), 256, 0, stream>>>(workspace, grid, n, K * D, gw1, gb1);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

// Both second-stage sums of the H = 64 NNConv backward in one launch: after qot_nnconv_adjoint_dw(param_layout = 2:
// slabs left in adj_workspace) and qot_nnconv_gradh_fused(gw1 = gb1 = NULL: partials left in gradh_workspace).
extern "C" int qot_nnconv_bwd_finalize(const float* adj_workspace, const float* gradh_workspace, float* grad_params,
                                       float* gw1, float* gb1, int64_t N, int H, int D, qot_stream_t stream_) {
    if (N <= 0 || H != 64 || D < 1 || D > 4) return (H != 64 || D > 4) ? QOT_ERR_UNSUPPORTED : QOT_ERR_BADARG;
    qot_role_t r{};                      // nnconv_bwd_finalize64_body through the multi-role launch (roles.hip)
    r.kind = QOT_ROLE_NNCONV_FINALIZE64;
    r.p[0] = adj_workspace; r.p[1] = gradh_workspace; r.p[2] = grad_params; r.p[3] = gw1; r.p[4] = gb1;
    r.i[0] = N; r.i[1] = D;
    return qot_run_roles(&r, 1, stream_);
}

// grad_x + weight gradient of NNConv in one pass (H = 64, D <= 4).  w_perm: WcatT in fragment
// order (as for qot_nnconv_fused transpose=1).  gwcat_t[(K+2)*64, 64] receives gWcat^T per block:
// gwcat_t[k*64 + o][a] = d/dWcat[k*64 + a][o].  workspace: qot_nnconv_adjoint_dw_workspace_floats(D).
extern "C" size_t qot_nnconv_adjoint_dw_workspace_floats(int D) {
    return (size_t)(num_cus() > 256 ? num_cus() : 256) * kAdjBlocksPerCu * (size_t)((2 * D + 2) * 64) * 64;
}

extern "C" int qot_nnconv_adjoint_dw(const float* grad_out, int ld_g, const float* x, int ld_x,
                                     const float* edge_attr, const float* w1, const float* b1,
                                     const int32_t* rowptr_t, const int32_t* col_t, const int32_t* eid_t,
                                     const float* invdeg, const float* w_perm, float* grad_x, float* gwcat_t,
                                     int param_layout, float* workspace, int64_t N, int H, int D,
                                     qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N <= 0 || !rowptr_t) return QOT_ERR_BADARG;
    if (H != 64 || D > 4) return QOT_ERR_UNSUPPORTED;
    if (!grad_out || !x || !w1 || !b1 || !invdeg || !w_perm || !grad_x || !gwcat_t || !workspace || (ld_g & 3))
        return QOT_ERR_BADARG;
    int grid = grid_for(N, 32);
    const int cap = num_cus() * kAdjBlocksPerCu;
    if (grid > cap) grid = cap;
#ifdef QOT_DIAG
    if (g_variant >= 11 && g_variant <= 17 && D == 4) {
#define QOT_ADJ_V(V) nnconv_adjoint_dw64_kernel<4, V><<<grid, 512, 0, stream>>>(grad_out, ld_g, x, ld_x, edge_attr, w1, b1, \
            rowptr_t, col_t, eid_t, invdeg, w_perm, grad_x, workspace, N)
        switch (g_variant) {
            case 11: QOT_ADJ_V(1); break;
            case 12: QOT_ADJ_V(2); break;
            case 13: QOT_ADJ_V(3); break;
            case 14: QOT_ADJ_V(4); break;
            case 16: QOT_ADJ_V(6); break;
            case 17: QOT_ADJ_V(7); break;
            default: QOT_ADJ_V(5); break;
        }
#undef QOT_ADJ_V
        QOT_LAUNCH_CHECK();
        return QOT_OK;
    }
#endif
    QOT_DISPATCH_D(D, {
        if (kD <= 4)
            nnconv_adjoint_dw64_kernel<(kD <= 4 ? kD : 4)><<<grid, 512, 0, stream>>>(
                grad_out, ld_g, x, ld_x, edge_attr, w1, b1, rowptr_t, col_t, eid_t, invdeg, w_perm, grad_x,
                workspace, N);
    });
    QOT_LAUNCH_CHECK();
    if (param_layout == 2) return QOT_OK;      // profiling: main kernel only (slabs left unsummed)
    const int64_t elems = (int64_t)(2 * D + 2) * 64 * 64;
    const int per_group = 16;
    const int groups = (grid + per_group - 1) / per_group;
    adj_slab_reduce_kernel<<<dim3(grid_for(elems / 4, 256), groups), 256, 0, stream>>>(
        workspace, grid, per_group, elems, elems, workspace, (int64_t)per_group * elems);
    QOT_LAUNCH_CHECK();
    if (param_layout)
        adj_slab_final_params_kernel<<<grid_for(elems / 4, 256), 256, 0, stream>>>(
            workspace, groups, elems, (int64_t)per_group * elems, gwcat_t, 2 * D);
    else
        adj_slab_reduce_kernel<<<dim3(grid_for(elems / 4, 256), 1), 256, 0, stream>>>(
            workspace, groups, groups, elems, (int64_t)per_group * elems, gwcat_t, 0);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}
