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

// Fragments of a wave: a[i] (its four 32-row bands), b[2][j] (its four 32-column bands, current and next k group).  A k
// group's 64 MFMAs run band by band (i-major), so a[i] is dead after its 16 MFMAs and takes the NEXT group's fragment at
// once: 48 fragment registers instead of 64 for two full sets.
struct G2Frag { float4 a[4], b[2][4]; };

// float4 slot of (k group g of 8, half hi, row): lane (hi, r) of an MFMA reads slot (2g + hi) * 256 + (row ^ 2g) --
// any constant XOR keeps the 16-lane groups of a ds_read_b128 on 16 distinct slots of one aligned block -- and the 8-lane
// groups of the ds_write_b128 (four k groups of two neighbouring rows) land on 8 distinct slots mod 8.
__device__ __forceinline__ int g2_slot(int g, int hi, int row) { return (2 * g + hi) * 256 + (row ^ (g << 1)); }

__device__ __forceinline__ float4 g2_read_a(const float4* __restrict__ As, int g, int i, int wm, int hi, int r31) {
    return As[g2_slot(g, hi, wm * 128 + i * 32 + r31)];
}
__device__ __forceinline__ void g2_read_b(float4 (&b)[4], const float4* __restrict__ Bs, int g, int wn, int hi, int r31) {
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = Bs[g2_slot(g, hi, wn * 128 + j * 32 + r31)];
}

// 16 MFMAs of band i of one k group: four accumulators take turns
__device__ __forceinline__ void g2_band_mfma(const float4& a, const float4 (&b)[4], f32x16 (&c)[4]) {
#define QOT_STEP(COMP)                                                                                       \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                            \
        c[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.COMP, b[j].COMP, c[j], 0, 0, 0);
    QOT_STEP(x) QOT_STEP(y) QOT_STEP(z) QOT_STEP(w)
#undef QOT_STEP
}

// k group G (compile-time 0..3) of the stage in (As, Bs): its 64 MFMAs; the fragments of the group after it -- group G + 1
// of the same stage, or group 0 of (An, Bn) behind group 3 -- are read as registers come free.  SKIP: diagnostic build.
template <int G>
__device__ __forceinline__ void g2_group(G2Frag& f, const float4* __restrict__ As, const float4* __restrict__ Bs,
                                         const float4* __restrict__ An, const float4* __restrict__ Bn, int wm, int wn, int hi,
                                         int r31, f32x16 (&c)[4][4], bool skip_reads) {
    constexpr int NG = (G + 1) & 3;
    const float4* Ar = G == 3 ? An : As;
    const float4* Br = G == 3 ? Bn : Bs;
    if (!skip_reads) g2_read_b(f.b[(G + 1) & 1], Br, NG, wn, hi, r31);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        g2_band_mfma(f.a[i], f.b[G & 1], c[i]);
        if (!skip_reads) f.a[i] = g2_read_a(Ar, NG, i, wm, hi, r31);
    }
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
// rows).
// The stages of all its tiles form ONE stream: while stage s is multiplied, stage s + 1 -- the next 32 k of the tile or
// the first 32 k of the next tile -- is loaded and stored to the other LDS buffer, so a tile boundary costs no pipeline
// refill.  The loop body has no branch: one scheduling region of 192 MFMAs, 24 fragment reads, 16 (20) global loads and
// 16 LDS stores whose interleave is pinned with sched_group_barrier: with ONE wave per SIMD nothing else fills the matrix
// pipe while a clump of loads / stores / reads issues (an LDS store takes 13-26 issue cycles; the branchy first version
// of this kernel had them in clumps of 8 between runs of ~60 MFMAs: 4-7 % on the diagnostic build's ablations, 128 / 125
// TFLOP/s plain / BatchNorm prologue).  The operands come through BUFFER loads (one descriptor per matrix, 32-bit byte
// offsets, the k offset in the scalar offset field): this compiler's sched_group_barrier VMEM masks do not match
// global_load (a FLAT instruction) -- with global loads the same pins left them next to the LDS stores that consume them
// (101 / 114 TFLOP/s), and without pins the scheduler sinks them there as well (118 / 118).  With buffer loads the pins
// hold: 133-136 / 128-131 TFLOP/s (the library's plain product: 135).  Matrices above 4 GB take the 128 x 128 kernel.
// The epilogue stages 8-row bands through 4 KB of LDS per wave BEHIND the stage buffers (they hold the next tile's first
// stage by then) and needs no workgroup barrier.
constexpr int kG2BandFloats = 8 * 128;                    // per wave
constexpr size_t kG2BandBytes = (size_t)4 * kG2BandFloats * sizeof(float);

template <bool AFFINE, bool LOGITS>
__global__ __launch_bounds__(256) void gemm256_nt_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                         int64_t ldb, float* __restrict__ C, int64_t ldc, int64_t M, int N,
                                                         int K, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const float* __restrict__ bias,
                                                         const float* __restrict__ att_src, const float* __restrict__ att_dst,
                                                         float* __restrict__ a_src, float* __restrict__ a_dst) {
    extern __shared__ __attribute__((aligned(16))) float4 g2lds[];       // [stage][operand][slot] | bands | attention vectors
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
    float* band = reinterpret_cast<float*>(g2lds + 4 * kG2Stage4) + wave * kG2BandFloats;
    float* att = reinterpret_cast<float*>(g2lds + 4 * kG2Stage4) + 4 * kG2BandFloats;
    if (LOGITS) {                                          // att_src | att_dst of all heads (N <= 1024: host side)
        for (int u = t; u < N; u += 256) {
            att[u] = att_src[u];
            att[1024 + u] = att_dst[u];
        }
    }
    const int gk = t & 3, row0 = t >> 2;                   // my k group; my rows: row0 + 64 j
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, 0xFFFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rbd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, 0xFFFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(AFFINE ? scale : A), 0, 0xFFFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(AFFINE ? shift : A), 0, 0xFFFFFFFF, 0x00020000);
    unsigned int ao[4], bo[4];                             // byte offsets of my rows (host side: M * lda * 4 < 2^32)
    auto point = [&](int64_t rb_, int ct_) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t ar = rb_ * kG2T + row0 + 64 * j;
            const int br = ct_ * kG2T + row0 + 64 * j;
            ao[j] = (unsigned int)(((ar < M ? ar : M - 1) * lda + 8 * gk) * 4);          // clamped, not zeroed
            bo[j] = (unsigned int)((((int64_t)(br < N ? br : N - 1)) * ldb + 8 * gk) * 4);
        }
    };
    auto bld = [&](const __amdgpu_buffer_rsrc_t& r, unsigned int voff, int soff) -> float4 {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    // The A half of the next stage is in flight under k group 0 and stored under group 1, the B half (W rows: L2 hits) in
    // flight under group 1 and stored under group 2: the two halves never hold registers at the same time.
    float4 pa[4][2], pb[4][2], ps[2], pt[2];
    auto load_a = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { pa[j][0] = bld(ra, ao[j], 4 * k0); pa[j][1] = bld(ra, ao[j] + 16, 4 * k0); }
        if (AFFINE) {
            ps[0] = bld(rsc, 32u * gk, 4 * k0); ps[1] = bld(rsc, 32u * gk + 16, 4 * k0);
            pt[0] = bld(rsh, 32u * gk, 4 * k0); pt[1] = bld(rsh, 32u * gk + 16, 4 * k0);
        }
    };
    auto load_b = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { pb[j][0] = bld(rbd, bo[j], 4 * k0); pb[j][1] = bld(rbd, bo[j] + 16, 4 * k0); }
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
    G2Frag f;
    point(rb, ct);
    load_a(0);
    load_b(0);
    stash_a(0);
    stash_b(0);
    __syncthreads();
    g2_read_b(f.b[0], g2lds + kG2Stage4, 0, wn, hi, r31);
#pragma unroll
    for (int i = 0; i < 4; ++i) f.a[i] = g2_read_a(g2lds, 0, i, wm, hi, r31);
    int cur = 0;
    int64_t nrb = rb;
    int nct = ct;
    bool have_next = true;
    // scheduling masks of sched_group_barrier
#define QOT_M(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0);
#define QOT_V(n) __builtin_amdgcn_sched_group_barrier(0x020, n, 0);
#define QOT_D(n) __builtin_amdgcn_sched_group_barrier(0x100, n, 0);
#define QOT_W(n) __builtin_amdgcn_sched_group_barrier(0x200, n, 0);
    for (;;) {                                             // tiles
        g2_zero(c);
#pragma unroll 1
        for (int kt = 0; kt < nk; ++kt) {                  // stages
        // the stage that is loaded while this one is multiplied
        int kn = kt + 1;
        if (kn == nk) {
            kn = 0;
            q += per_x;
            nrb = (q / ntn) * 8 + xcd;
            nct = (int)(q % ntn);
            have_next = nrb < ntm;
            if (have_next) point(nrb, nct);                // (none left: the loads below re-read this tile's first stage,
        }                                                  //  the copy is never multiplied -- no branch in the body)
        const float4* As = g2lds + (2 * cur) * kG2Stage4;
        const float4* Bs = As + kG2Stage4;
        const float4* An = g2lds + (2 * (cur ^ 1)) * kG2Stage4;
        const float4* Bn = An + kG2Stage4;
        const bool skip = G2_VAR(16);
        if (!G2_VAR(1)) load_a(kn * kG2BK);
        g2_group<0>(f, As, Bs, An, Bn, wm, wn, hi, r31, c, skip);
        if (!G2_VAR(2)) stash_a(cur ^ 1);
        if (!G2_VAR(1)) load_b(kn * kG2BK);
        g2_group<1>(f, As, Bs, An, Bn, wm, wn, hi, r31, c, skip);
        if (!G2_VAR(2)) stash_b(cur ^ 1);
        g2_group<2>(f, As, Bs, An, Bn, wm, wn, hi, r31, c, skip);
#ifndef QOT_G2_NO_PIN
        // group 0: 8 (12) global loads of the A half
        QOT_M(2) QOT_V(1) QOT_M(2) QOT_D(1) QOT_M(2) QOT_V(1) QOT_M(2) QOT_D(1)
        QOT_M(2) QOT_V(1) QOT_M(2) QOT_D(1) QOT_M(2) QOT_V(1) QOT_M(2) QOT_D(1) QOT_D(1)
        QOT_M(4) QOT_V(1) QOT_M(4) QOT_V(1) QOT_M(4) QOT_V(1) QOT_M(4) QOT_V(1) QOT_D(1)
        if (AFFINE) { QOT_M(4) QOT_V(1) QOT_M(4) QOT_V(1) QOT_M(4) QOT_V(1) QOT_M(4) QOT_V(1) } else { QOT_M(16) }
        QOT_D(1) QOT_M(16) QOT_D(1)
        // group 1: 8 LDS stores of the A half, then the 8 global loads of the B half
        QOT_M(2) QOT_W(1) QOT_M(2) QOT_D(1) QOT_M(2) QOT_W(1) QOT_M(2) QOT_D(1)
        QOT_M(2) QOT_W(1) QOT_M(2) QOT_D(1) QOT_M(2) QOT_W(1) QOT_M(2) QOT_D(1) QOT_D(1)
        QOT_M(4) QOT_W(1) QOT_M(4) QOT_W(1) QOT_M(4) QOT_W(1) QOT_M(4) QOT_W(1) QOT_D(1)
        QOT_M(2) QOT_V(1) QOT_M(2) QOT_V(1) QOT_M(2) QOT_V(1) QOT_M(2) QOT_V(1)
        QOT_M(2) QOT_V(1) QOT_M(2) QOT_V(1) QOT_M(2) QOT_V(1) QOT_M(2) QOT_V(1) QOT_D(1)
        QOT_M(16) QOT_D(1)
        // group 2: 8 LDS stores of the B half
        QOT_M(2) QOT_W(1) QOT_M(2) QOT_D(1) QOT_M(2) QOT_W(1) QOT_M(2) QOT_D(1)
        QOT_M(2) QOT_W(1) QOT_M(2) QOT_D(1) QOT_M(2) QOT_W(1) QOT_M(2) QOT_D(1) QOT_D(1)
        QOT_M(4) QOT_W(1) QOT_M(4) QOT_W(1) QOT_M(4) QOT_W(1) QOT_M(4) QOT_W(1) QOT_D(1)
        QOT_M(16) QOT_D(1) QOT_M(16) QOT_D(1)
#endif
        if (!G2_VAR(4)) lds_barrier();
        g2_group<3>(f, As, Bs, An, Bn, wm, wn, hi, r31, c, skip);
#ifndef QOT_G2_NO_PIN
        QOT_M(4) QOT_D(1) QOT_M(4) QOT_D(1) QOT_M(4) QOT_D(1) QOT_M(4) QOT_D(1) QOT_D(1)
        QOT_M(16) QOT_D(1) QOT_M(16) QOT_D(1) QOT_M(16) QOT_D(1)
#endif
        cur ^= 1;
        }
        // ---- the tile is complete: 16 bands of 8 rows per wave through the wave's own LDS ----
        const int64_t m0 = rb * kG2T + wm * 128;
        const int n0 = ct * kG2T + wn * 128;
#pragma unroll                                   // c[i][.][4 qd + rr] must stay register-indexed: constants per copy
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int pm = rr + 4 * hi;           // MFMA row (rr + 8 qd + 4 hi) of tile i = row pm of band (i, qd)
                        const int pn = j * 32 + r31;
                        band[pm * 128 + (pn ^ (pm << 2))] = c[i][j][4 * qd + rr];
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int64_t rbase = m0 + i * 32 + 8 * qd;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int idx = it * 64 + lane;
                    const int pm = idx >> 5, q4 = idx & 31;
                    float4 v = ld4(band + pm * 128 + ((4 * q4) ^ (pm << 2)));
                    const int64_t row = rbase + pm;
                    const int col = n0 + 4 * q4;
                    if (row < M && col < N) {
                        if (bias) v = add4(v, ld4(bias + col));
                        if (!G2_VAR(8) || v.x == 12345.678f) st4(C + row * ldc + col, v);
                    }
                }
                if (LOGITS) {
                    // the wave's 128 columns are ONE attention head (see gemm_nt_kernel<., LOGITS>): lane = (row pm, 16 columns)
                    const int pm = lane >> 3, part = lane & 7;
                    const int64_t row = rbase + pm;
                    float s_ = 0.f, d_ = 0.f;
                    if (n0 < N) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int cq = 16 * part + 4 * u;
                            float4 v = ld4(band + pm * 128 + (cq ^ (pm << 2)));
                            if (bias) v = add4(v, ld4(bias + n0 + cq));
                            s_ += dot4(v, ld4(att + n0 + cq));
                            d_ += dot4(v, ld4(att + 1024 + n0 + cq));
                        }
                    }
                    s_ = group_sum<8>(s_);
                    d_ = group_sum<8>(d_);
                    if (part == 0 && row < M && n0 < N) {
                        const int heads = N / 128;
                        a_src[row * heads + n0 / 128] = s_;
                        a_dst[row * heads + n0 / 128] = d_;
                    }
                }
                asm volatile("" ::: "memory");
            }
        if (!have_next) break;
        rb = nrb;
        ct = nct;
    }
#undef QOT_M
#undef QOT_V
#undef QOT_D
#undef QOT_W
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
    const size_t lds = kG2StageBytes + kG2BandBytes + (logits ? kG2AttFloats * sizeof(float) : 0);
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
