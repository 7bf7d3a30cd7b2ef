// Dense fp32 projections of LightpathGNN on the matrix cores (gfx950, v_mfma_f32_32x32x2_f32).
//
// Reference sites: lightpath_training/models.py:13,30 -- GATConv's shared projection z = x W^T ([N, 4C] x [4C, 4C] from
// the second layer on; SURVEY 8(d) cfg3: N ~ 720 k, 4C = 512) -- and its autograd under loss.backward()
// (lightpath_training/train.py:128): grad_x = g W, grad_W = g^T x.  Rounds 1-2 sent all three to the library (58.6 % of
// cfg3's kernel time, the weight-gradient shape at 91 TFLOP/s).  Two kernels here:
//
//   NT   C[M, N] = A'[M, K] . B[N, K]^T (+ bias)      both operands k-contiguous (row-major [rows, K])
//        A' = A, or relu(A * scale[k] + shift[k]) applied while the tile is loaded: BatchNorm(+ReLU) of the previous
//        layer (lightpath_training/models.py:31-32) folded into the projection's operand load, so the normalised
//        activations are never written to / re-read from HBM (one [N, 4C] pass each way per layer).
//   TN   C[M, N] = A[K, M]^T . B'[K, N]                both operands k-strided (row index = k): the weight gradient
//        g^T y with y = relu(x * scale[n] + shift[n]) recomputed from x on load.  K ~ 7e5: split over gridDim.z,
//        partial planes summed afterwards in a fixed order (bitwise reproducible).
//
// Tile 128 x 128 x 32 per 256-thread workgroup (2 x 2 waves of 64 x 64 = 2 x 2 MFMA tiles: one ds_read_b128 feeds four
// MFMAs, the ratio the fused NNConv kernels run at), operands staged through LDS in the fragment-grouped order of
// mfma_tile.hpp (k = 8g + 2r + hi  ->  float4 slot (2g + hi) * 128 + row', component r), double buffered (64 KB: two
// workgroups per CU), next stage prefetched into registers under the current stage's 64 MFMAs per wave.
#include <cstdlib>
#include "common.hpp"
#include "mfma_tile.hpp"

namespace qot {

constexpr int kGemmBM = 128, kGemmBN = 128, kGemmBK = 32;

#ifdef QOT_DIAG
__device__ int g_gemm_variant;     // ablation bits of the NT kernel (tools/bench_gemm_k.py): 1 no global loads in the loop,
#define GEMM_VAR(bit) (gemm_var & (bit))   // 2 no LDS stores, 4 no barrier, 8 no C stores, 16 no fragment reads
#else
#define GEMM_VAR(bit) 0
#endif

// float4 slot of (group g, k parity hi, row) inside one operand stage; the XOR spreads the four groups a quarter-wave
// writes (NT loader) over all banks, reads of 16 consecutive rows stay a permutation of 16 consecutive slots
__device__ __forceinline__ int gemm_slot(int g, int hi, int row) { return (2 * g + hi) * 128 + (row ^ ((g & 3) << 2)); }

struct GemmFrag { float4 a[2], b[2]; };

__device__ __forceinline__ void gemm_read_frag(GemmFrag& f, const float4* __restrict__ As, const float4* __restrict__ Bs, int g,
                                               int wm, int wn, int hi, int r31) {
#pragma unroll
    for (int i = 0; i < 2; ++i) f.a[i] = As[gemm_slot(g, hi, wm * 64 + i * 32 + r31)];
#pragma unroll
    for (int j = 0; j < 2; ++j) f.b[j] = Bs[gemm_slot(g, hi, wn * 64 + j * 32 + r31)];
}

// 16 MFMAs of one k group; the four accumulators take turns: consecutive MFMAs never wait for each other's result
__device__ __forceinline__ void gemm_group_mfma(const GemmFrag& f, f32x16 (&c)[2][2]) {
#define QOT_STEP(COMP)                                                                                       \
    c[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[0].COMP, f.b[0].COMP, c[0][0], 0, 0, 0);             \
    c[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[0].COMP, f.b[1].COMP, c[0][1], 0, 0, 0);             \
    c[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[1].COMP, f.b[0].COMP, c[1][0], 0, 0, 0);             \
    c[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[1].COMP, f.b[1].COMP, c[1][1], 0, 0, 0);
    QOT_STEP(x) QOT_STEP(y) QOT_STEP(z) QOT_STEP(w)
#undef QOT_STEP
}

// Main loop over nk stages of 32 k (4 groups of 16 MFMAs per wave).  Measured on the first version (one block of 64
// MFMAs, then LDS stores of the next stage, then the barrier): 96.8 TFLOP/s at cfg3's shape against 133 with the loop's
// global loads / LDS stores / barrier removed -- every stage paid the LDS write burst, the barrier skew and an
// exposed LDS read latency behind the barrier.  Now fragments are read one group ahead, the next stage's LDS stores are
// issued between the groups of the current stage, and the LAST group of a stage is multiplied behind the barrier, after
// the first fragments of the next stage have been requested: its 16 MFMAs (1024 cycles) cover that LDS round trip.
template <class Load, class Stash>
__device__ __forceinline__ void gemm_mainloop(float4 (&lds)[2][2][1024], int64_t nk, int wm, int wn, int hi, int r31,
                                              f32x16 (&c)[2][2], Load load, Stash stash, int gemm_var = 0) {
    GemmFrag f0, f1;
    if (nk > 0) gemm_read_frag(f0, lds[0][0], lds[0][1], 0, wm, wn, hi, r31);
    if (GEMM_VAR(16)) f1 = f0;
#pragma unroll 1
    for (int64_t kt = 0; kt < nk; ++kt) {
        const int cur = (int)(kt & 1);
        const bool more = kt + 1 < nk;
        if (more && !GEMM_VAR(1)) load(kt + 1);                              // global -> registers, under the MFMAs below
        if (!GEMM_VAR(16)) gemm_read_frag(f1, lds[cur][0], lds[cur][1], 1, wm, wn, hi, r31);
        gemm_group_mfma(f0, c);
        if (!GEMM_VAR(16)) gemm_read_frag(f0, lds[cur][0], lds[cur][1], 2, wm, wn, hi, r31);
        gemm_group_mfma(f1, c);
        if (!GEMM_VAR(16)) gemm_read_frag(f1, lds[cur][0], lds[cur][1], 3, wm, wn, hi, r31);
        gemm_group_mfma(f0, c);
        if (more && !GEMM_VAR(2)) stash(cur ^ 1);                            // registers -> the other LDS buffer
        if (!GEMM_VAR(4)) lds_barrier();
        if (more && !GEMM_VAR(16)) gemm_read_frag(f0, lds[cur ^ 1][0], lds[cur ^ 1][1], 0, wm, wn, hi, r31);
        gemm_group_mfma(f1, c);                                              // group 3 of this stage, from registers
    }
}

// C tile of a workgroup through LDS (free after the main loop: 128 x 128 floats = the whole 64 KB) so that the global
// stores are 16 B per lane, 512 contiguous bytes per row and wave instruction (as 4-B stores straight from the MFMA
// layout -- 128-B pieces, 64 instructions per wave -- the epilogue cost 15 % of the kernel).  rowmap(p): output row of
// tile row p (-1 = out of range), colbase: first output column, ncols valid columns (multiple of 4).
__device__ __forceinline__ void gemm_tile_to_lds(float* __restrict__ tile, const f32x16 (&c)[2][2], int wm, int wn, int hi,
                                                 int r31) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pm = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int pn = wn * 64 + j * 32 + r31;
                tile[pm * 128 + (pn ^ ((pm & 7) << 2))] = c[i][j][r];      // XOR on float4 granularity: conflict-free both ways
            }
}

// ---- NT: both operands row-major [rows, K] ---------------------------------------------------------------------
// thread t owns (row, group) pairs pi = t, t + 256 of each operand: row = pi / 4, g = pi % 4 -> 8 consecutive k (32 B);
// four neighbouring lanes read one 128-B row segment
// LOGITS (r03): the 128 columns of an output tile are ONE attention head of GATConv (C = 128 channels per head), so the
// head's two attention logits of the tile's rows, a_src[m, h] = <out[m, h, :], att_src[h, :]> and a_dst likewise
// (lightpath_training/models.py:13 -> PyG GATConv's alpha_src / alpha_dst), are complete inside the tile: they are formed
// from the LDS image of the epilogue and the separate pass over the [M, 4C] matrix (gat_logits_kernel) is not needed.
template <bool AFFINE, bool LOGITS = false>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                         int64_t ldb, float* __restrict__ C, int64_t ldc, int64_t M, int N,
                                                         int K, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const float* __restrict__ bias,
                                                         const float* __restrict__ att_src = nullptr,
                                                         const float* __restrict__ att_dst = nullptr,
                                                         float* __restrict__ a_src = nullptr, float* __restrict__ a_dst = nullptr,
                                                         int kslices = 1) {
    __shared__ __attribute__((aligned(16))) float4 lds[2][2][1024];       // [stage][operand][slot]
    mfma_acc_in_agprs();
    // kslices > 1 (qot_gemm_nt_planes: few output tiles, long K): workgroup id = slice * tiles + tile multiplies the K range
    // [slice * K / kslices, ...) into plane `slice` of C ([kslices][M][ldc]); the caller sums the planes in order
    const int64_t tiles_all = (int64_t)((M + kGemmBM - 1) / kGemmBM) * ((N + kGemmBN - 1) / kGemmBN);
    const int64_t bid = kslices > 1 ? (int64_t)blockIdx.x % tiles_all : (int64_t)blockIdx.x;
    if (kslices > 1) {
        const int64_t slice = (int64_t)blockIdx.x / tiles_all;
        K /= kslices;
        A += slice * K;
        B += slice * K;
        C += slice * M * ldc;
    }
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, hi = lane >> 5, r31 = lane & 31;
    // Workgroups are dealt round-robin over the 8 XCDs (id % 8), each with its own L2: the column tiles of ONE row block
    // go to ONE XCD, back to back, so that the A rows they share are fetched from HBM once (with the plain order --
    // column tile = id % ntn -- the four workgroups that share a row block sat on four XCDs: A was fetched four times).
    const int ntn = (N + kGemmBN - 1) / kGemmBN;
    const int64_t ntm = (M + kGemmBM - 1) / kGemmBM;
    int64_t rb;
    int ct;
    {
        const int64_t id = bid;
        const int64_t full = (ntm / 8) * 8;                  // row blocks that split evenly over the XCDs
        if (id < full * ntn) {
            const int64_t xcd = id % 8, j = id / 8;
            rb = xcd + 8 * (j / ntn);
            ct = (int)(j % ntn);
        } else {                                             // the last < 8 row blocks: plain order
            const int64_t j = id - full * ntn;
            rb = full + j / ntn;
            ct = (int)(j % ntn);
        }
    }
    const int64_t m0 = rb * kGemmBM;
    const int n0 = ct * kGemmBN;
    // rows past the end are clamped, not zeroed: row m of A only reaches row m of C (column n of B only column n), and
    // the epilogue stores neither
    const int gk = t & 3;                          // the k group (8 consecutive k) of both of this thread's rows
    const float* ap[2];
    const float* bp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int64_t ar = m0 + ((t + 256 * j) >> 2);
        const int br = n0 + ((t + 256 * j) >> 2);
        ap[j] = A + (ar < M ? ar : M - 1) * lda + 8 * gk;
        bp[j] = B + (int64_t)(br < N ? br : N - 1) * ldb + 8 * gk;
    }
    float4 pa[2][2], pb[2][2], ps[2], pt[2];
    auto load = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            pa[j][0] = ld4(ap[j] + k0); pa[j][1] = ld4(ap[j] + k0 + 4);
            pb[j][0] = ld4(bp[j] + k0); pb[j][1] = ld4(bp[j] + k0 + 4);
        }
        if (AFFINE) {                  // requested here, applied in stash(): the loads stay in flight under the MFMAs
            ps[0] = ld4(scale + k0 + 8 * gk); ps[1] = ld4(scale + k0 + 8 * gk + 4);
            pt[0] = ld4(shift + k0 + 8 * gk); pt[1] = ld4(shift + k0 + 8 * gk + 4);
        }
    };
    // Fragment order of this kernel: MFMA r of group g multiplies the k pair {8g + r, 8g + 4 + r} (the hi = 0 lanes
    // supply the first, the hi = 1 lanes the second; A and B agree, which is all the product needs), so the float4 a
    // lane feeds to four consecutive MFMAs is four CONSECUTIVE k of its row: the 16-byte pieces go from the global load
    // to LDS as they are (the interleaved order of mfma_tile.hpp cost 52 register moves per stage here, and on this part
    // every VALU instruction is an MFMA issue slot lost).
    auto stash = [&](int s) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (t + 256 * j) >> 2;
            if (AFFINE) {
                pa[j][0] = affine_relu4<true>(pa[j][0], ps[0], pt[0]);
                pa[j][1] = affine_relu4<true>(pa[j][1], ps[1], pt[1]);
            }
            lds[s][0][gemm_slot(gk, 0, row)] = pa[j][0];
            lds[s][0][gemm_slot(gk, 1, row)] = pa[j][1];
            lds[s][1][gemm_slot(gk, 0, row)] = pb[j][0];
            lds[s][1][gemm_slot(gk, 1, row)] = pb[j][1];
        }
    };
    f32x16 c[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) c[i][j][r] = 0.f;
    const int nk = K / kGemmBK;
    load(0);
    stash(0);
    __syncthreads();
#ifdef QOT_DIAG
    const int gemm_var = g_gemm_variant;
#else
    const int gemm_var = 0;
#endif
    gemm_mainloop(lds, nk, wm, wn, hi, r31, c, [&](int64_t kt) { load((int)kt * kGemmBK); }, stash, gemm_var);
    // epilogue through LDS: tile row p = output row m0 + p, tile column q = output column n0 + q
    float* tile = reinterpret_cast<float*>(&lds[0][0][0]);
    gemm_tile_to_lds(tile, c, wm, wn, hi, r31);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int idx = it * 256 + t;                 // float4 index: row = idx / 32, quad = idx % 32
        const int pm = idx >> 5, q4 = idx & 31;
        const int64_t row = m0 + pm;
        const int col = n0 + 4 * q4;
        if (row < M && col < N) {
            float4 v = ld4(tile + pm * 128 + ((4 * q4) ^ ((pm & 7) << 2)));
            if (bias) { const float4 bz = ld4(bias + col); v = add4(v, bz); }
            if (!GEMM_VAR(8) || v.x == 12345.678f) st4(C + row * ldc + col, v);
        }
    }
    if (LOGITS) {
        // thread t: tile row t / 2, columns 64 (t & 1) .. + 64; the two halves of a row meet in neighbouring lanes
        const int pm = t >> 1, half = t & 1;
        const int64_t row = m0 + pm;
        float ps = 0.f, pd = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int cq = 64 * half + 4 * q;
            float4 v = ld4(tile + pm * 128 + (cq ^ ((pm & 7) << 2)));
            if (bias) v = add4(v, ld4(bias + n0 + cq));
            ps += dot4(v, ld4(att_src + n0 + cq));
            pd += dot4(v, ld4(att_dst + n0 + cq));
        }
        ps += dpp_move<0xB1>(ps);              // quad_perm [1,0,3,2]: lane ^ 1
        pd += dpp_move<0xB1>(pd);
        if (half == 0 && row < M) {
            const int heads = N / kGemmBN;
            a_src[row * heads + ct] = ps;
            a_dst[row * heads + ct] = pd;
        }
    }
}

// ---- TN: both operands [K, rows] (row index = k), split-K over gridDim.z ---------------------------------------
// The operands arrive with the OUTPUT index contiguous, the opposite of what a float4 fragment wants.  Rounds 3's kernel
// transposed 8 x 4 blocks in registers on the way to LDS (32 moves + 32 selects per thread and stage, every one an MFMA
// issue slot lost).  Here the stage keeps the global order -- T[k][c], 32 k rows of 128 floats, 16-byte pieces stored as
// loaded -- and the MFMA operands are read one dword per MFMA, two at a time:
//   MFMA 2p / 2p + 1 of a stage multiply the k pairs {4p, 4p + 2} / {4p + 1, 4p + 3}: lane (hi, r) reads rows
//   4p + 2hi and 4p + 2hi + 1 of its column with ONE ds_read2_b32 (second address + 128 dwords);
//   column c of row k sits at dword c ^ (32 * ((k >> 1) & 1)), so the hi = 0 and hi = 1 halves of a wave fall on
//   disjoint bank halves: no conflicts on either side (a quarter-wave writes 256 contiguous bytes).
// Every thread stages four k rows of BOTH operands (thread t: quad q = t % 32 of the tile's 128 columns, rows 4 (t / 32)
// + s), so the BatchNorm/ReLU prologue of the B operand is spread over all four waves.
__device__ __forceinline__ int tn_dword(int k, int c) { return k * 128 + (c ^ (((k >> 1) & 1) << 5)); }

struct TnFrag { float a[2][2], b[2][2]; };       // [tile][MFMA of the pair]

__device__ __forceinline__ void tn_read_frag(TnFrag& f, const float* __restrict__ As, const float* __restrict__ Bs, int p,
                                             int wm, int wn, int hi, int r31) {
    const int k = 4 * p + 2 * hi;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float* s = As + tn_dword(k, wm * 64 + i * 32 + r31);
        f.a[i][0] = s[0];
        f.a[i][1] = s[128];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float* s = Bs + tn_dword(k, wn * 64 + j * 32 + r31);
        f.b[j][0] = s[0];
        f.b[j][1] = s[128];
    }
}

__device__ __forceinline__ void tn_pair_mfma(const TnFrag& f, f32x16 (&c)[2][2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        c[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[0][u], f.b[0][u], c[0][0], 0, 0, 0);
        c[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[0][u], f.b[1][u], c[0][1], 0, 0, 0);
        c[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[1][u], f.b[0][u], c[1][0], 0, 0, 0);
        c[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[1][u], f.b[1][u], c[1][1], 0, 0, 0);
    }
}

template <bool AFFINE>
__global__ __launch_bounds__(256, 2) void gemm_tn_split_kernel(const float* __restrict__ A, int64_t lda,
                                                               const float* __restrict__ B, int64_t ldb,
                                                               float* __restrict__ Cpart, int M, int N, int64_t K,
                                                               int64_t kchunk, int nsplit, const float* __restrict__ scale,
                                                               const float* __restrict__ shift) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][32 * 128];     // [stage][operand][k][column']
    mfma_acc_in_agprs();
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, hi = lane >> 5, r31 = lane & 31;
    // all output tiles of ONE split (they share its K chunk of both operands) on ONE XCD (id % 8), back to back
    const int ntn = (N + kGemmBN - 1) / kGemmBN, ntm = (M + kGemmBM - 1) / kGemmBM;
    const int tiles = ntn * ntm;
    int split, tile_id;
    {
        const int id = blockIdx.x;
        const int full = (nsplit / 8) * 8;
        if (id < full * tiles) {
            const int xcd = id % 8, j = id / 8;
            split = xcd + 8 * (j / tiles);
            tile_id = j % tiles;
        } else {
            const int j = id - full * tiles;
            split = full + j / tiles;
            tile_id = j % tiles;
        }
    }
    const int m0 = (tile_id / ntn) * kGemmBM, n0 = (tile_id % ntn) * kGemmBN;
    const int64_t kbeg = (int64_t)split * kchunk;
    const int64_t kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
    const int q = t & 31, kr = 4 * (t >> 5);             // column quad, first of my four k rows inside a stage
    // columns past the end are clamped, not zeroed (column m of A only reaches row m of C; the epilogue skips it);
    // M, N multiples of 4 (checked by the host side)
    const int ca = (m0 + 4 * q + 3 < M) ? m0 + 4 * q : M - 4;
    const int cb = (n0 + 4 * q + 3 < N) ? n0 + 4 * q : N - 4;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = f4zero();
    if (AFFINE) { sc = ld4(scale + cb); sh = ld4(shift + cb); }
    float4 pa[4], pb[4];
    uint32_t oa[4], ob[4];                               // (kr + s) * ld + column: < 2^32 (checked by the host side)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        oa[s] = (uint32_t)((kr + s) * lda + ca);
        ob[s] = (uint32_t)((kr + s) * ldb + cb);
    }
    int64_t pk0 = 0;
    auto load = [&](int64_t k0) {
        pk0 = k0;
        if (k0 + kGemmBK <= kend) {                      // uniform: all but the last stage of a ragged chunk
            // uniform base (scalar registers, advanced by the scalar unit) + per-thread 32-bit offsets that never
            // change: no vector address arithmetic in the loop
            const float* a = A + k0 * lda;
            const float* b = B + k0 * ldb;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                pa[s] = ld4(a + oa[s]);
                pb[s] = ld4(b + ob[s]);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int64_t k = (k0 + kr + s < kend) ? k0 + kr + s : kend - 1;
                pa[s] = ld4(A + k * lda + ca);
                pb[s] = ld4(B + k * ldb + cb);
            }
        }
    };
    auto stash = [&](int st) {
        // transform what was loaded (not in load(): a use right behind a load keeps only that load in flight)
        if (AFFINE) {
#pragma unroll
            for (int s = 0; s < 4; ++s) pb[s] = affine_relu4<true>(pb[s], sc, sh);
        }
        if (pk0 + kGemmBK > kend) {                      // rows past the chunk contribute nothing: zero BOTH operands' copies
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (pk0 + kr + s >= kend) { pa[s] = f4zero(); pb[s] = f4zero(); }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            st4(&lds[st][0][tn_dword(kr + s, 4 * q)], pa[s]);
            st4(&lds[st][1][tn_dword(kr + s, 4 * q)], pb[s]);
        }
    };
    f32x16 c[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) c[i][j][r] = 0.f;
    const int64_t nk = (kend - kbeg + kGemmBK - 1) / kGemmBK;
    if (nk > 0) {
        load(kbeg);
        stash(0);
    }
    __syncthreads();
    // stages of 8 k quads (8 MFMAs per wave each); fragments one quad ahead, the next stage's LDS stores between the
    // quads, the last quad multiplied behind the barrier (same schedule as gemm_mainloop)
    TnFrag f0, f1;
    if (nk > 0) tn_read_frag(f0, lds[0][0], lds[0][1], 0, wm, wn, hi, r31);
#pragma unroll 1
    for (int64_t kt = 0; kt < nk; ++kt) {
        const int cur = (int)(kt & 1);
        const bool more = kt + 1 < nk;
        if (more) load(kbeg + (kt + 1) * kGemmBK);
#pragma unroll
        for (int p = 0; p < 6; p += 2) {
            tn_read_frag(f1, lds[cur][0], lds[cur][1], p + 1, wm, wn, hi, r31);
            tn_pair_mfma(f0, c);
            tn_read_frag(f0, lds[cur][0], lds[cur][1], p + 2, wm, wn, hi, r31);
            tn_pair_mfma(f1, c);
        }
        tn_read_frag(f1, lds[cur][0], lds[cur][1], 7, wm, wn, hi, r31);
        tn_pair_mfma(f0, c);
        if (more) stash(cur ^ 1);
        lds_barrier();
        if (more) tn_read_frag(f0, lds[cur ^ 1][0], lds[cur ^ 1][1], 0, wm, wn, hi, r31);
        tn_pair_mfma(f1, c);
    }
    float* Cp = Cpart + (int64_t)split * M * N;
    // epilogue through LDS: tile row p = output row m0 + p, tile column q = output column n0 + q
    float* tile = &lds[0][0][0];
    gemm_tile_to_lds(tile, c, wm, wn, hi, r31);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int idx = it * 256 + t;                 // float4 index: row = idx / 32, quad = idx % 32
        const int pm = idx >> 5, q4 = idx & 31;
        const int row = m0 + pm, col = n0 + 4 * q4;
        if (row < M && col < N) st4(Cp + (int64_t)row * N + col, ld4(tile + pm * 128 + ((4 * q4) ^ ((pm & 7) << 2))));
    }
}

// The same product with the loop body free of branches (whole stages only; the stage after the last one re-reads it, the
// copy is never multiplied; a ragged end of K is one masked stage behind the loop), operands through buffer loads and the
// interleave pinned with sched_group_barrier as in gemm256.hip -- left to the scheduler the fragment reads sat one or two
// MFMAs in front of their use (`ds_read2; s_waitcnt lgkmcnt(0)` before every group of six).
template <bool AFFINE>
__global__ __launch_bounds__(256, 2) void gemm_tn_full_kernel(const float* __restrict__ A, int64_t lda,
                                                              const float* __restrict__ B, int64_t ldb,
                                                              float* __restrict__ Cpart, int M, int N, int64_t K,
                                                              int64_t kchunk, int nsplit, const float* __restrict__ scale,
                                                              const float* __restrict__ shift) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][32 * 128];     // [stage][operand][k][column']
    mfma_acc_in_agprs();
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, hi = lane >> 5, r31 = lane & 31;
    const int ntn = (N + kGemmBN - 1) / kGemmBN, ntm = (M + kGemmBM - 1) / kGemmBM;
    const int tiles = ntn * ntm;
    int split, tile_id;
    {
        const int id = blockIdx.x;
        const int full = (nsplit / 8) * 8;
        if (id < full * tiles) {
            const int xcd = id % 8, j = id / 8;
            split = xcd + 8 * (j / tiles);
            tile_id = j % tiles;
        } else {
            const int j = id - full * tiles;
            split = full + j / tiles;
            tile_id = j % tiles;
        }
    }
    const int m0 = (tile_id / ntn) * kGemmBM, n0 = (tile_id % ntn) * kGemmBN;
    const int64_t kbeg = (int64_t)split * kchunk;
    const int64_t kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
    const int q = t & 31, kr = 4 * (t >> 5);
    const int ca = (m0 + 4 * q + 3 < M) ? m0 + 4 * q : M - 4;
    const int cb = (n0 + 4 * q + 3 < N) ? n0 + 4 * q : N - 4;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = f4zero();
    if (AFFINE) { sc = ld4(scale + cb); sh = ld4(shift + cb); }
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, 0xFFFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, 0xFFFFFFFF, 0x00020000);
    unsigned int oa[4], ob[4];                           // byte offsets inside a stage (host side: K * ld * 4 < 2^32)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        oa[s] = (unsigned int)(((kr + s) * lda + ca) * 4);
        ob[s] = (unsigned int)(((kr + s) * ldb + cb) * 4);
    }
    auto bld = [&](const __amdgpu_buffer_rsrc_t& r, unsigned int voff, unsigned int soff) -> float4 {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    };
    float4 pa[4], pb[4];
    auto load = [&](int64_t k0) {
        const unsigned int sa = (unsigned int)(k0 * lda * 4), sb = (unsigned int)(k0 * ldb * 4);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            pa[s] = bld(ra, oa[s], sa);
            pb[s] = bld(rb, ob[s], sb);
        }
    };
    auto stash = [&](int st) {
        if (AFFINE) {
#pragma unroll
            for (int s = 0; s < 4; ++s) pb[s] = affine_relu4<true>(pb[s], sc, sh);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            st4(&lds[st][0][tn_dword(kr + s, 4 * q)], pa[s]);
            st4(&lds[st][1][tn_dword(kr + s, 4 * q)], pb[s]);
        }
    };
    f32x16 c[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) c[i][j][r] = 0.f;
    const int64_t nk = (kend - kbeg) / kGemmBK;           // whole stages; a ragged end of K (last split only) follows the loop
    const int tail = (int)((kend - kbeg) - nk * kGemmBK);
    if (nk > 0) {
        load(kbeg);
        stash(0);
    }
    __syncthreads();
    TnFrag f0, f1;
    if (nk > 0) tn_read_frag(f0, lds[0][0], lds[0][1], 0, wm, wn, hi, r31);
#define QOT_M(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0);
#define QOT_V(n) __builtin_amdgcn_sched_group_barrier(0x020, n, 0);
#define QOT_D(n) __builtin_amdgcn_sched_group_barrier(0x100, n, 0);
#define QOT_W(n) __builtin_amdgcn_sched_group_barrier(0x200, n, 0);
#pragma unroll 1
    for (int64_t kt = 0; kt < nk; ++kt) {
        const int cur = (int)(kt & 1);
        const int64_t kn = (kt + 1 < nk) ? kt + 1 : kt;
        load(kbeg + kn * kGemmBK);
        tn_read_frag(f1, lds[cur][0], lds[cur][1], 1, wm, wn, hi, r31);
        tn_pair_mfma(f0, c);
        tn_read_frag(f0, lds[cur][0], lds[cur][1], 2, wm, wn, hi, r31);
        tn_pair_mfma(f1, c);
        tn_read_frag(f1, lds[cur][0], lds[cur][1], 3, wm, wn, hi, r31);
        tn_pair_mfma(f0, c);
        tn_read_frag(f0, lds[cur][0], lds[cur][1], 4, wm, wn, hi, r31);
        tn_pair_mfma(f1, c);
        tn_read_frag(f1, lds[cur][0], lds[cur][1], 5, wm, wn, hi, r31);
        tn_pair_mfma(f0, c);
        stash(cur ^ 1);
        tn_read_frag(f0, lds[cur][0], lds[cur][1], 6, wm, wn, hi, r31);
        tn_pair_mfma(f1, c);
        tn_read_frag(f1, lds[cur][0], lds[cur][1], 7, wm, wn, hi, r31);
        tn_pair_mfma(f0, c);
        // quads 0-1: the eight loads of the next stage; quads 5-6: its eight LDS stores; a quad's four fragment reads
        // under the MFMAs of the quad before it
        QOT_M(2) QOT_V(1) QOT_D(1) QOT_M(2) QOT_V(1) QOT_D(1) QOT_M(2) QOT_V(1) QOT_D(1) QOT_M(2) QOT_V(1) QOT_D(1)
        QOT_M(2) QOT_V(1) QOT_D(1) QOT_M(2) QOT_V(1) QOT_D(1) QOT_M(2) QOT_V(1) QOT_D(1) QOT_M(2) QOT_V(1) QOT_D(1)
        QOT_M(2) QOT_D(1) QOT_M(2) QOT_D(1) QOT_M(2) QOT_D(1) QOT_M(2) QOT_D(1)
        QOT_M(2) QOT_D(1) QOT_M(2) QOT_D(1) QOT_M(2) QOT_D(1) QOT_M(2) QOT_D(1)
        QOT_M(2) QOT_D(1) QOT_M(2) QOT_D(1) QOT_M(2) QOT_D(1) QOT_M(2) QOT_D(1)
        QOT_M(2) QOT_W(1) QOT_D(1) QOT_M(2) QOT_W(1) QOT_D(1) QOT_M(2) QOT_W(1) QOT_D(1) QOT_M(2) QOT_W(1) QOT_D(1)
        QOT_M(2) QOT_W(1) QOT_D(1) QOT_M(2) QOT_W(1) QOT_D(1) QOT_M(2) QOT_W(1) QOT_D(1) QOT_M(2) QOT_W(1) QOT_D(1)
        lds_barrier();
        tn_read_frag(f0, lds[cur ^ 1][0], lds[cur ^ 1][1], 0, wm, wn, hi, r31);
        tn_pair_mfma(f1, c);
        QOT_M(2) QOT_D(1) QOT_M(2) QOT_D(1) QOT_M(2) QOT_D(1) QOT_M(2) QOT_D(1)
    }
#undef QOT_M
#undef QOT_V
#undef QOT_D
#undef QOT_W
    if (tail > 0) {
        // the last < 32 rows of K (K not a multiple of 32: last split only): one masked stage, not pipelined
        __syncthreads();
        const int64_t k0 = kbeg + nk * kGemmBK;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool ok = kr + s < tail;
            const int64_t k = ok ? k0 + kr + s : k0;
            pa[s] = ld4(A + k * lda + ca);
            pb[s] = ld4(B + k * ldb + cb);
            if (AFFINE) pb[s] = affine_relu4<true>(pb[s], sc, sh);
            if (!ok) { pa[s] = f4zero(); pb[s] = f4zero(); }
            st4(&lds[0][0][tn_dword(kr + s, 4 * q)], pa[s]);
            st4(&lds[0][1][tn_dword(kr + s, 4 * q)], pb[s]);
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            tn_read_frag(f0, lds[0][0], lds[0][1], p, wm, wn, hi, r31);
            tn_pair_mfma(f0, c);
        }
        __syncthreads();
    }
    float* Cp = Cpart + (int64_t)split * M * N;
    float* tile = &lds[0][0][0];
    gemm_tile_to_lds(tile, c, wm, wn, hi, r31);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int idx = it * 256 + t;
        const int pm = idx >> 5, q4 = idx & 31;
        const int row = m0 + pm, col = n0 + 4 * q4;
        if (row < M && col < N) st4(Cp + (int64_t)row * N + col, ld4(tile + pm * 128 + ((4 * q4) ^ ((pm & 7) << 2))));
    }
}

}  // namespace qot

using namespace qot;

// gemm256.hip: the 256 x 256 forms for large M
extern "C" int qot_gemm256_takes(int64_t M, int N);
int gemm256_nt_launch(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K,
                      const float* scale, const float* shift, const float* bias, const float* att_src, const float* att_dst,
                      float* a_src, float* a_dst, hipStream_t stream);

#ifdef QOT_DIAG
extern "C" int qot_debug_gemm_variant(int v) {
    return hipMemcpyToSymbol(HIP_SYMBOL(qot::g_gemm_variant), &v, sizeof(int)) == hipSuccess ? 0 : 1;
}
#endif

// C[M, N] = A'[M, K] . B[N, K]^T (+ bias[N]);  A' = relu(A * scale[k] + shift[k]) when scale != NULL.
// K multiple of 32, lda / ldb multiples of 4, 16-byte aligned operands.
extern "C" int qot_gemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
                           int N, int K, const float* scale, const float* shift, const float* bias, qot_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0) return QOT_ERR_BADARG;
    if (M == 0) return QOT_OK;
    if (!A || !B || !C || (scale && !shift)) return QOT_ERR_BADARG;
    if ((K % kGemmBK) || (N & 3) || (lda & 3) || (ldb & 3) || (ldc & 3) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) ||
        ((uintptr_t)C & 15) || ((uintptr_t)bias & 15))
        return QOT_ERR_UNSUPPORTED;
    // (the 256 x 256 kernel addresses its operands with 32-bit byte offsets)
    const bool fits32 = (uint64_t)M * (uint64_t)lda * 4 < (1ull << 32) && (uint64_t)N * (uint64_t)ldb * 4 < (1ull << 32);
    if (qot_gemm256_takes(M, N) && fits32)
        return gemm256_nt_launch(A, lda, B, ldb, C, ldc, M, N, K, scale, shift, bias, nullptr, nullptr, nullptr, nullptr,
                                 (hipStream_t)stream);
    const int64_t tiles = ((M + kGemmBM - 1) / kGemmBM) * ((N + kGemmBN - 1) / kGemmBN);
    if (tiles > 0x7fffffff) return QOT_ERR_UNSUPPORTED;
    if (scale)
        gemm_nt_kernel<true><<<(int)tiles, 256, 0, (hipStream_t)stream>>>(A, lda, B, ldb, C, ldc, M, N, K, scale, shift, bias);
    else
        gemm_nt_kernel<false><<<(int)tiles, 256, 0, (hipStream_t)stream>>>(A, lda, B, ldb, C, ldc, M, N, K, nullptr, nullptr, bias);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

// qot_gemm_nt for FEW output tiles and a long inner dimension: Cpart[kslices][M, N], plane s = A[:, Ks] . B[:, Ks]^T over its
// slice of K (K / kslices a multiple of 32); the caller sums the planes in order (QOT_ROLE_SUM_ROWS).
extern "C" int qot_gemm_nt_planes(const float* A, int64_t lda, const float* B, int64_t ldb, float* Cpart, int64_t M, int N,
                                  int K, int kslices, qot_stream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0 || kslices <= 0) return QOT_ERR_BADARG;
    if (!A || !B || !Cpart) return QOT_ERR_BADARG;
    if ((K % kslices) || ((K / kslices) % kGemmBK) || (N & 3) || (lda & 3) || (ldb & 3) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) ||
        ((uintptr_t)Cpart & 15))
        return QOT_ERR_UNSUPPORTED;
    const int64_t tiles = ((M + kGemmBM - 1) / kGemmBM) * ((N + kGemmBN - 1) / kGemmBN) * kslices;
    if (tiles > 0x7fffffff) return QOT_ERR_UNSUPPORTED;
    gemm_nt_kernel<false><<<(int)tiles, 256, 0, (hipStream_t)stream>>>(A, lda, B, ldb, Cpart, N, M, N, K, nullptr, nullptr, nullptr,
                                                                     nullptr, nullptr, nullptr, nullptr, kslices);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

// qot_gemm_nt with GATConv's attention logits from the epilogue: N = heads * 128 (one output tile per head), att_src /
// att_dst [heads * 128], a_src / a_dst [M, heads] (see gemm_nt_kernel<., LOGITS>)
extern "C" int qot_gemm_nt_logits(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
                                  int N, int K, const float* scale, const float* shift, const float* bias,
                                  const float* att_src, const float* att_dst, float* a_src, float* a_dst,
                                  qot_stream_t stream) {
    if (M < 0 || N <= 0 || K <= 0) return QOT_ERR_BADARG;
    if (M == 0) return QOT_OK;
    if (!A || !B || !C || (scale && !shift) || !att_src || !att_dst || !a_src || !a_dst) return QOT_ERR_BADARG;
    if ((K % kGemmBK) || (N % kGemmBN) || (lda & 3) || (ldb & 3) || (ldc & 3) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) ||
        ((uintptr_t)C & 15) || ((uintptr_t)bias & 15) || ((uintptr_t)att_src & 15) || ((uintptr_t)att_dst & 15))
        return QOT_ERR_UNSUPPORTED;
    if (qot_gemm256_takes(M, N) && N <= 1024 && (uint64_t)M * (uint64_t)lda * 4 < (1ull << 32) &&
        (uint64_t)N * (uint64_t)ldb * 4 < (1ull << 32))
        return gemm256_nt_launch(A, lda, B, ldb, C, ldc, M, N, K, scale, shift, bias, att_src, att_dst, a_src, a_dst,
                                 (hipStream_t)stream);
    const int64_t tiles = ((M + kGemmBM - 1) / kGemmBM) * (N / kGemmBN);
    if (tiles > 0x7fffffff) return QOT_ERR_UNSUPPORTED;
    if (scale)
        gemm_nt_kernel<true, true><<<(int)tiles, 256, 0, (hipStream_t)stream>>>(A, lda, B, ldb, C, ldc, M, N, K, scale, shift, bias,
                                                                              att_src, att_dst, a_src, a_dst);
    else
        gemm_nt_kernel<false, true><<<(int)tiles, 256, 0, (hipStream_t)stream>>>(A, lda, B, ldb, C, ldc, M, N, K, nullptr, nullptr,
                                                                               bias, att_src, att_dst, a_src, a_dst);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

// split count of qot_gemm_tn for a [K, M]^T [K, N] product (enough workgroups to fill the part, chunks of >= 128 rows)
extern "C" int qot_gemm_tn_splits(int M, int N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int64_t tiles = (int64_t)((M + kGemmBM - 1) / kGemmBM) * ((N + kGemmBN - 1) / kGemmBN);
    int64_t s = (4 * (int64_t)num_cus() + tiles - 1) / tiles;
    const int64_t maxs = (K + 127) / 128;                    // chunks of >= 128 rows (four stages)
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    if (s > 256) s = 256;
    return (int)s;
}

// Cpart[splits, M, N]: plane z = A[kz, M]^T . B'[kz, N] over its K chunk (B' = relu(B * scale[n] + shift[n]) when scale !=
// NULL); the caller sums the planes in order (QOT_ROLE_SUM_ROWS).  M, N multiples of 4; splits = qot_gemm_tn_splits.
extern "C" int qot_gemm_tn_planes(const float* A, int64_t lda, const float* B, int64_t ldb, float* Cpart, int M, int N,
                                  int64_t K, int splits, const float* scale, const float* shift, qot_stream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0 || splits <= 0) return QOT_ERR_BADARG;
    if (!A || !B || !Cpart || (scale && !shift)) return QOT_ERR_BADARG;
    if ((M & 3) || (N & 3) || (lda & 3) || (ldb & 3) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)Cpart & 15) ||
        lda < M || ldb < N || lda > (1 << 24) || ldb > (1 << 24))      // a stage's 32 rows are addressed with 32-bit offsets
        return QOT_ERR_UNSUPPORTED;
    int64_t kchunk = (K + splits - 1) / splits;
    kchunk = (kchunk + kGemmBK - 1) / kGemmBK * kGemmBK;
    const int64_t grid = (int64_t)((N + kGemmBN - 1) / kGemmBN) * ((M + kGemmBM - 1) / kGemmBM) * splits;
    if (grid > 0x7fffffff) return QOT_ERR_UNSUPPORTED;
    // every stage full and 32-bit byte offsets: the branch-free, pinned form
    const bool full = (uint64_t)K * (uint64_t)lda * 4 < (1ull << 32) &&
                      (uint64_t)K * (uint64_t)ldb * 4 < (1ull << 32) && !getenv("QOT_NO_GEMM_TN_FULL");
    if (full && scale)
        gemm_tn_full_kernel<true><<<(int)grid, 256, 0, (hipStream_t)stream>>>(A, lda, B, ldb, Cpart, M, N, K, kchunk, splits, scale, shift);
    else if (full)
        gemm_tn_full_kernel<false><<<(int)grid, 256, 0, (hipStream_t)stream>>>(A, lda, B, ldb, Cpart, M, N, K, kchunk, splits, nullptr, nullptr);
    else if (scale)
        gemm_tn_split_kernel<true><<<(int)grid, 256, 0, (hipStream_t)stream>>>(A, lda, B, ldb, Cpart, M, N, K, kchunk, splits, scale, shift);
    else
        gemm_tn_split_kernel<false><<<(int)grid, 256, 0, (hipStream_t)stream>>>(A, lda, B, ldb, Cpart, M, N, K, kchunk, splits, nullptr, nullptr);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}
