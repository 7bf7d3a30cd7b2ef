// The large-M form of gemm.hip's NT product: 256 x 256 x 32 tiles, one 256-thread workgroup per CU, each of the four
// waves owning a 128 x 128 quarter (4 x 4 MFMA tiles of 32 x 32: 256 accumulator registers per lane, in AGPRs).
//
// Why a second tile size (measured, tools/bench_gemm_k.py on the diagnostic build): the 128 x 128 kernel's MFMA stream
// alone runs at 141-147 TFLOP/s, its fragment reads take that to 131-133, the stage's global loads / LDS stores to
// 124-127 -- not latency (operands from one L2-resident row: no change; barrier removed: no change) but the number of
// non-MFMA instructions issued per MFMA.  A wave that owns 128 x 128 instead of 64 x 64 issues HALF the ds_read_b128 per
// MFMA (8 per 64 instead of 4 per 16) and half the global loads / LDS stores (16 + 16 per 256 instead of 8 + 8 per 64);
// the library's own kernel for this shape is a 256 x 256 x 32 tile for the same reason.
//
// Measured negative (r04): the stages of all tiles as one branch-free stream (stage s + 1 -- also the next tile's first
// -- loaded and stored while stage s is multiplied; rolling fragments, 48 registers; 8-row epilogue bands behind the stage
// buffers, no barrier around the epilogue).  Left to the scheduler, the single big block had its global loads sunk next
// to the LDS stores that consume them: 118 / 118 TFLOP/s (plain / BatchNorm prologue) against 128 / 125 for this file,
// whose `if (!last)` branches keep the loads a stage ahead; with the interleave pinned by sched_group_barrier (2 MFMA :
// 1 load, 4 MFMA : 1 LDS store, fragment reads as registers come free) the loads still stayed late: 101 / 114.  Kept:
// the branchy loop.
//
// Reference sites as in gemm.hip: lightpath_training/models.py:13,30 (GATConv's projection) and its autograd under
// lightpath_training/train.py:128.
#include <cstdlib>
#include "common.hpp"
#include "mfma_tile.hpp"

namespace qot {

constexpr int kG2T = 256, kG2BK = 32;
constexpr int kG2Stage4 = 2048;                          // float4 per operand and stage (256 rows x 32 k)
constexpr size_t kG2StageBytes = (size_t)2 * 2 * kG2Stage4 * 16;    // two stages x two operands = 128 KB
constexpr int kG2AttFloats = 2 * 1024;                   // NT: attention vectors of up to 8 heads behind the stages

#ifdef QOT_DIAG
__device__ int g_gemm256_variant;  // ablation bits of the NT kernel (tools/bench_gemm_k.py): 1 no global loads in the loop,
#define G2_VAR(bit) (g2_var & (bit))   // 2 no LDS stores, 4 no barrier, 8 no C stores, 16 no fragment reads
#else
#define G2_VAR(bit) 0
#endif

struct G2Frag { float4 a[4], b[4]; };

// float4 slot of (k group g of 8, half hi, row): lane (hi, r) of an MFMA reads slot (2g + hi) * 256 + (row ^ 2g) --
// any constant XOR keeps the 16-lane groups of a ds_read_b128 on 16 distinct slots of one aligned block -- and the 8-lane
// groups of the ds_write_b128 (four k groups of two neighbouring rows) land on 8 distinct slots mod 8.
__device__ __forceinline__ int g2_slot(int g, int hi, int row) { return (2 * g + hi) * 256 + (row ^ (g << 1)); }

__device__ __forceinline__ void g2_read_frag(G2Frag& f, const float4* __restrict__ As, const float4* __restrict__ Bs, int g,
                                             int wm, int wn, int hi, int r31) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f.a[i] = As[g2_slot(g, hi, wm * 128 + i * 32 + r31)];
#pragma unroll
    for (int j = 0; j < 4; ++j) f.b[j] = Bs[g2_slot(g, hi, wn * 128 + j * 32 + r31)];
}

// 64 MFMAs of one k group: 16 independent accumulators per k pair
__device__ __forceinline__ void g2_group_mfma(const G2Frag& f, f32x16 (&c)[4][4]) {
#define QOT_STEP(COMP)                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                            \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                        \
            c[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i].COMP, f.b[j].COMP, c[i][j], 0, 0, 0);
    QOT_STEP(x) QOT_STEP(y) QOT_STEP(z) QOT_STEP(w)
#undef QOT_STEP
}

__device__ __forceinline__ void g2_zero(f32x16 (&c)[4][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) c[i][j][r] = 0.f;
}

// One 32-row band of a wave's quarter (MFMA row i) through the wave's OWN 16 KB of LDS -- no workgroup barrier: a wave's
// LDS operations complete in order -- so that the global stores are 16 B per lane, 512 contiguous bytes per row.
// emit(pm, q4, v): row pm of the band, float4 q4 of the wave's 128 columns.
template <class Emit>
__device__ __forceinline__ void g2_band_out(float* __restrict__ reg, const f32x16 (&c)[4][4], int i, int hi, int r31, int lane,
                                            Emit emit) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pm = (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int pn = j * 32 + r31;
            reg[pm * 128 + (pn ^ ((pm & 7) << 2))] = c[i][j][r];
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int idx = it * 64 + lane;
        const int pm = idx >> 5, q4 = idx & 31;
        emit(pm, q4, ld4(reg + pm * 128 + ((4 * q4) ^ ((pm & 7) << 2))));
    }
}

// ---- NT: C[M, N] = A'[M, K] . B[N, K]^T (+ bias), persistent over the output tiles ------------------------------
// Workgroup (xcd = id % 8, slot = id / 8) walks the tiles q = slot, slot + per_x, ... of ITS XCD: row block 8 (q / ntn)
// + xcd, column tile q % ntn -- the column tiles of a row block run side by side on one XCD (one HBM fetch of the A
// rows).  The first stage of the next tile is requested before the epilogue of the current one.
template <bool AFFINE, bool LOGITS>
__global__ __launch_bounds__(256) void gemm256_nt_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                         int64_t ldb, float* __restrict__ C, int64_t ldc, int64_t M, int N,
                                                         int K, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const float* __restrict__ bias,
                                                         const float* __restrict__ att_src, const float* __restrict__ att_dst,
                                                         float* __restrict__ a_src, float* __restrict__ a_dst) {
    extern __shared__ __attribute__((aligned(16))) float4 g2lds[];       // [stage][operand][slot], then the attention vectors
    mfma_acc_in_agprs();
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, hi = lane >> 5, r31 = lane & 31;
    const int ntn = (N + kG2T - 1) / kG2T;
    const int64_t ntm = (M + kG2T - 1) / kG2T;
    const int xcd = blockIdx.x % 8, per_x = gridDim.x / 8;
    int64_t q = blockIdx.x / 8;
    int64_t rb = (q / ntn) * 8 + xcd;
    int ct = (int)(q % ntn);
    if (rb >= ntm) return;
    float* att = reinterpret_cast<float*>(g2lds + 4 * kG2Stage4);
    if (LOGITS) {                                          // att_src | att_dst of all heads (N <= 1024: host side)
        for (int u = t; u < N; u += 256) {
            att[u] = att_src[u];
            att[1024 + u] = att_dst[u];
        }
    }
    const int gk = t & 3, row0 = t >> 2;                   // my k group; my rows: row0 + 64 j
    const float* ap[4];
    const float* bp[4];
    auto point = [&](int64_t rb_, int ct_) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t ar = rb_ * kG2T + row0 + 64 * j;
            const int br = ct_ * kG2T + row0 + 64 * j;
            ap[j] = A + (ar < M ? ar : M - 1) * lda + 8 * gk;          // clamped, not zeroed: see gemm_nt_kernel
            bp[j] = B + (int64_t)(br < N ? br : N - 1) * ldb + 8 * gk;
        }
    };
    // The A half of a stage is in flight under k groups 0-1 and stored behind them, the B half under groups 2-3: 32
    // prefetch registers instead of 64 (with both halves in flight at once the affine / logits forms spilled), and
    // half a stage -- 128 MFMAs, 3.4 us -- is still several memory latencies.
    float4 pa[4][2], pb[4][2], ps[2], pt[2];
    auto load_a = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { pa[j][0] = ld4(ap[j] + k0); pa[j][1] = ld4(ap[j] + k0 + 4); }
        if (AFFINE) {
            ps[0] = ld4(scale + k0 + 8 * gk); ps[1] = ld4(scale + k0 + 8 * gk + 4);
            pt[0] = ld4(shift + k0 + 8 * gk); pt[1] = ld4(shift + k0 + 8 * gk + 4);
        }
    };
    auto load_b = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { pb[j][0] = ld4(bp[j] + k0); pb[j][1] = ld4(bp[j] + k0 + 4); }
    };
    auto stash_a = [&](int s) {
        float4* As = g2lds + (2 * s) * kG2Stage4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = row0 + 64 * j;
            if (AFFINE) {
                pa[j][0] = affine_relu4<true>(pa[j][0], ps[0], pt[0]);
                pa[j][1] = affine_relu4<true>(pa[j][1], ps[1], pt[1]);
            }
            As[g2_slot(gk, 0, row)] = pa[j][0];
            As[g2_slot(gk, 1, row)] = pa[j][1];
        }
    };
    auto stash_b = [&](int s) {
        float4* Bs = g2lds + (2 * s + 1) * kG2Stage4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = row0 + 64 * j;
            Bs[g2_slot(gk, 0, row)] = pb[j][0];
            Bs[g2_slot(gk, 1, row)] = pb[j][1];
        }
    };
    const int nk = K / kG2BK;
#ifdef QOT_DIAG
    const int g2_var = g_gemm256_variant;
#endif
    f32x16 c[4][4];
    G2Frag f0, f1;
    point(rb, ct);
    load_a(0);
    load_b(0);
    stash_a(0);
    stash_b(0);
    __syncthreads();
    int cur = 0;
    for (;;) {
        g2_zero(c);
        const float4* As = g2lds + (2 * cur) * kG2Stage4;
        g2_read_frag(f0, As, As + kG2Stage4, 0, wm, wn, hi, r31);
        if (G2_VAR(16)) f1 = f0;
        int64_t nrb = 0;
        int nct = 0;
        bool have_next = false;
#pragma unroll 1
        for (int kt = 0; kt < nk; ++kt) {
            const bool last = kt + 1 == nk;
            const int kn = last ? 0 : (kt + 1) * kG2BK;
            if (last) {                                    // the next tile's first stage, under this stage and the epilogue
                q += per_x;
                nrb = (q / ntn) * 8 + xcd;
                nct = (int)(q % ntn);
                have_next = nrb < ntm;
                if (have_next) point(nrb, nct);
            }
            const bool more = !last || have_next;
            if (more && !G2_VAR(1)) load_a(kn);
            As = g2lds + (2 * cur) * kG2Stage4;
            if (!G2_VAR(16)) g2_read_frag(f1, As, As + kG2Stage4, 1, wm, wn, hi, r31);
            g2_group_mfma(f0, c);
            if (!G2_VAR(16)) g2_read_frag(f0, As, As + kG2Stage4, 2, wm, wn, hi, r31);
            g2_group_mfma(f1, c);
            if (!last && !G2_VAR(2)) stash_a(cur ^ 1);     // (the last stage keeps its halves in registers over the epilogue:
            if (more && !last && !G2_VAR(1)) load_b(kn);   //  only the A half fits there, the B half is requested after it)
            if (!G2_VAR(16)) g2_read_frag(f1, As, As + kG2Stage4, 3, wm, wn, hi, r31);
            g2_group_mfma(f0, c);
            if (!last) {
                if (!G2_VAR(2)) stash_b(cur ^ 1);
                if (!G2_VAR(4)) lds_barrier();
                cur ^= 1;
                As = g2lds + (2 * cur) * kG2Stage4;
                if (!G2_VAR(16)) g2_read_frag(f0, As, As + kG2Stage4, 0, wm, wn, hi, r31);
            }
            g2_group_mfma(f1, c);
        }
        // ---- epilogue: both stage buffers are free once every wave is past its last fragment read ----
        lds_barrier();
        float* reg = reinterpret_cast<float*>(g2lds) + wave * 4096;
        const int64_t m0 = rb * kG2T + wm * 128;
        const int n0 = ct * kG2T + wn * 128;
#pragma unroll                                   // c[i] must stay a register array: i is a compile-time constant per copy
        for (int i = 0; i < 4; ++i) {
            g2_band_out(reg, c, i, hi, r31, lane, [&](int pm, int q4, float4 v) {
                const int64_t row = m0 + i * 32 + pm;
                const int col = n0 + 4 * q4;
                if (row < M && col < N) {
                    if (bias) v = add4(v, ld4(bias + col));
                    if (!G2_VAR(8) || v.x == 12345.678f) st4(C + row * ldc + col, v);
                }
            });
            if (LOGITS) {
                // the wave's 128 columns are ONE attention head (see gemm_nt_kernel<., LOGITS>): lane = (row pm, half)
                const int pm = lane >> 1, half = lane & 1;
                const int64_t row = m0 + i * 32 + pm;
                float s_ = 0.f, d_ = 0.f;
                if (n0 < N) {
#pragma unroll 2                                 // fully unrolled, the 64 loads of a lane are hoisted together: spills
                    for (int u = 0; u < 16; ++u) {
                        const int cq = 64 * half + 4 * u;
                        float4 v = ld4(reg + pm * 128 + (cq ^ ((pm & 7) << 2)));
                        if (bias) v = add4(v, ld4(bias + n0 + cq));
                        s_ += dot4(v, ld4(att + n0 + cq));
                        d_ += dot4(v, ld4(att + 1024 + n0 + cq));
                    }
                }
                s_ += dpp_move<0xB1>(s_);              // quad_perm [1,0,3,2]: lane ^ 1
                d_ += dpp_move<0xB1>(d_);
                if (half == 0 && row < M && n0 < N) {
                    const int heads = N / 128;
                    a_src[row * heads + n0 / 128] = s_;
                    a_dst[row * heads + n0 / 128] = d_;
                }
            }
        }
        if (!have_next) break;
        lds_barrier();                                     // the bands are read: the stage buffers may be written again
        rb = nrb;
        ct = nct;
        cur = 0;
        load_b(0);                                         // W rows: L2 hits
        stash_a(0);
        stash_b(0);
        lds_barrier();
    }
}

}  // namespace qot

using namespace qot;

#ifdef QOT_DIAG
extern "C" int qot_debug_gemm256_variant(int v) {
    return hipMemcpyToSymbol(HIP_SYMBOL(qot::g_gemm256_variant), &v, sizeof(int)) == hipSuccess ? 0 : 1;
}
#endif

// 1 when the 256 x 256 forms take a product of this size (enough tiles for every CU; narrower N leaves half a tile idle)
extern "C" int qot_gemm256_takes(int64_t M, int N) {
    if (getenv("QOT_NO_GEMM256")) return 0;
    return N >= kG2T && ((M + kG2T - 1) / kG2T) * ((N + kG2T - 1) / kG2T) >= num_cus();
}

int gemm256_nt_launch(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K,
                      const float* scale, const float* shift, const float* bias, const float* att_src, const float* att_dst,
                      float* a_src, float* a_dst, hipStream_t stream) {
    static size_t allowed[4][kMaxDevices];
    const bool logits = att_src != nullptr;
    if (logits && N > 1024) return QOT_ERR_UNSUPPORTED;
    const size_t lds = kG2StageBytes + (logits ? kG2AttFloats * sizeof(float) : 0);
    const int64_t tiles = ((M + kG2T - 1) / kG2T) * ((N + kG2T - 1) / kG2T);
    int64_t grid = num_cus();
    if (grid > tiles) grid = tiles;
    grid = (grid + 7) / 8 * 8;                           // whole XCD rounds; workgroups without a tile return at once
#define QOT_G2_NT(AFF, LOG, IDX)                                                                                     \
    {                                                                                                                \
        const int rc = ensure_dyn_lds(reinterpret_cast<const void*>(gemm256_nt_kernel<AFF, LOG>), lds, allowed[IDX]); \
        if (rc != QOT_OK) return rc;                                                                                 \
        gemm256_nt_kernel<AFF, LOG><<<(int)grid, 256, lds, stream>>>(A, lda, B, ldb, C, ldc, M, N, K, scale, shift, bias, \
                                                                    att_src, att_dst, a_src, a_dst);                 \
    }
    if (scale && logits) QOT_G2_NT(true, true, 0)
    else if (scale) QOT_G2_NT(true, false, 1)
    else if (logits) QOT_G2_NT(false, true, 2)
    else QOT_G2_NT(false, false, 3)
#undef QOT_G2_NT
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}
