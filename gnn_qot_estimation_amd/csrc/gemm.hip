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
#include "common.hpp"
#include "mfma_tile.hpp"

namespace qot {

constexpr int kGemmBM = 128, kGemmBN = 128, kGemmBK = 32;

// float4 slot of (group g, k parity hi, row) inside one operand stage; the XOR spreads the four groups a quarter-wave
// writes (NT loader) over all banks, reads of 16 consecutive rows stay a permutation of 16 consecutive slots
__device__ __forceinline__ int gemm_slot(int g, int hi, int row) { return (2 * g + hi) * 128 + (row ^ ((g & 3) << 2)); }

template <bool AFFINE>
__device__ __forceinline__ float4 affine_relu4(float4 v, float4 s, float4 t) {
    if (!AFFINE) return v;
    return make_float4(fmaxf(fmaf(v.x, s.x, t.x), 0.f), fmaxf(fmaf(v.y, s.y, t.y), 0.f),
                       fmaxf(fmaf(v.z, s.z, t.z), 0.f), fmaxf(fmaf(v.w, s.w, t.w), 0.f));
}

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
                                              f32x16 (&c)[2][2], Load load, Stash stash) {
    GemmFrag f0, f1;
    if (nk > 0) gemm_read_frag(f0, lds[0][0], lds[0][1], 0, wm, wn, hi, r31);
#pragma unroll 1
    for (int64_t kt = 0; kt < nk; ++kt) {
        const int cur = (int)(kt & 1);
        const bool more = kt + 1 < nk;
        if (more) load(kt + 1);                                              // global -> registers, under the MFMAs below
        gemm_read_frag(f1, lds[cur][0], lds[cur][1], 1, wm, wn, hi, r31);
        gemm_group_mfma(f0, c);
        gemm_read_frag(f0, lds[cur][0], lds[cur][1], 2, wm, wn, hi, r31);
        gemm_group_mfma(f1, c);
        gemm_read_frag(f1, lds[cur][0], lds[cur][1], 3, wm, wn, hi, r31);
        gemm_group_mfma(f0, c);
        if (more) stash(cur ^ 1);                                            // registers -> the other LDS buffer
        lds_barrier();
        if (more) gemm_read_frag(f0, lds[cur ^ 1][0], lds[cur ^ 1][1], 0, wm, wn, hi, r31);
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
                                                         float* __restrict__ a_src = nullptr, float* __restrict__ a_dst = nullptr) {
    __shared__ __attribute__((aligned(16))) float4 lds[2][2][1024];       // [stage][operand][slot]
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
        const int64_t id = blockIdx.x;
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
    int64_t arow[2];
    int brow[2], gk[2];
    bool aok[2], bok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pi = t + 256 * j;
        gk[j] = pi & 3;
        arow[j] = m0 + (pi >> 2);
        brow[j] = n0 + (pi >> 2);
        aok[j] = arow[j] < M;
        bok[j] = brow[j] < N;
        if (!aok[j]) arow[j] = M - 1;
        if (!bok[j]) brow[j] = N - 1;
    }
    float4 pa[2][2], pb[2][2], ps[2][2], pt[2][2];
    auto load = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float* ap = A + arow[j] * lda + k0 + 8 * gk[j];
            const float* bp = B + (int64_t)brow[j] * ldb + k0 + 8 * gk[j];
            pa[j][0] = ld4(ap); pa[j][1] = ld4(ap + 4);
            pb[j][0] = ld4(bp); pb[j][1] = ld4(bp + 4);
            if (AFFINE) {              // requested here, applied in stash(): the loads stay in flight under the MFMAs
                ps[j][0] = ld4(scale + k0 + 8 * gk[j]); ps[j][1] = ld4(scale + k0 + 8 * gk[j] + 4);
                pt[j][0] = ld4(shift + k0 + 8 * gk[j]); pt[j][1] = ld4(shift + k0 + 8 * gk[j] + 4);
            }
        }
    };
    auto stash = [&](int s) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (t + 256 * j) >> 2, g = gk[j];
            if (AFFINE) {
                pa[j][0] = affine_relu4<true>(pa[j][0], ps[j][0], pt[j][0]);
                pa[j][1] = affine_relu4<true>(pa[j][1], ps[j][1], pt[j][1]);
            }
            if (!aok[j]) { pa[j][0] = f4zero(); pa[j][1] = f4zero(); }
            if (!bok[j]) { pb[j][0] = f4zero(); pb[j][1] = f4zero(); }
            lds[s][0][gemm_slot(g, 0, row)] = make_float4(pa[j][0].x, pa[j][0].z, pa[j][1].x, pa[j][1].z);
            lds[s][0][gemm_slot(g, 1, row)] = make_float4(pa[j][0].y, pa[j][0].w, pa[j][1].y, pa[j][1].w);
            lds[s][1][gemm_slot(g, 0, row)] = make_float4(pb[j][0].x, pb[j][0].z, pb[j][1].x, pb[j][1].z);
            lds[s][1][gemm_slot(g, 1, row)] = make_float4(pb[j][0].y, pb[j][0].w, pb[j][1].y, pb[j][1].w);
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
    gemm_mainloop(lds, nk, wm, wn, hi, r31, c, [&](int64_t kt) { load((int)kt * kGemmBK); }, stash);
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
            st4(C + row * ldc + col, v);
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
// threads 0..127 stage A, 128..255 stage B: thread u owns (quad q = u % 32, group g = u / 32): eight k rows x four
// consecutive output rows (one float4 per k row: a wave reads 512 contiguous bytes per k row), transposed in registers
// into the fragment order.  Output row 4q + c is kept at tile row c*32 + q (writes of a quarter-wave then fall on 16
// consecutive slots); the epilogue undoes the permutation.
template <bool AFFINE>
__global__ __launch_bounds__(256, 2) void gemm_tn_split_kernel(const float* __restrict__ A, int64_t lda,
                                                               const float* __restrict__ B, int64_t ldb,
                                                               float* __restrict__ Cpart, int M, int N, int64_t K,
                                                               int64_t kchunk, int nsplit, const float* __restrict__ scale,
                                                               const float* __restrict__ shift) {
    __shared__ __attribute__((aligned(16))) float4 lds[2][2][1024];
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
    const bool isb = t >= 128;
    const int u = t & 127, q = u & 31, g = u >> 5;
    const float* src = isb ? B : A;
    const int64_t ld = isb ? ldb : lda;
    const int c0 = (isb ? n0 : m0) + 4 * q;
    const int lim = isb ? N : M;
    const bool cok = c0 + 3 < lim;                 // M, N multiples of 4 (checked by the host side)
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = f4zero();
    if (AFFINE && isb && cok) { sc = ld4(scale + c0); sh = ld4(shift + c0); }
    float4 pv[8];
    int64_t pk0 = 0;
    auto load = [&](int64_t k0) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int64_t k = k0 + 8 * g + s;
            const bool ok = cok && k < kend;
            pv[s] = ld4(src + (ok ? k : kbeg) * ld + (cok ? c0 : 0));
        }
        pk0 = k0;
    };
    auto stash = [&](int st) {
        float4* dst = lds[st][isb ? 1 : 0];
        // transform / zero what was loaded (not in load(): a use right behind a load keeps only that load in flight)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (AFFINE && isb) pv[s] = affine_relu4<true>(pv[s], sc, sh);
            if (!(cok && pk0 + 8 * g + s < kend)) pv[s] = f4zero();
        }
#define QOT_T(COMP, CI)                                                                                      \
        dst[gemm_slot(g, 0, (CI) * 32 + q)] = make_float4(pv[0].COMP, pv[2].COMP, pv[4].COMP, pv[6].COMP);   \
        dst[gemm_slot(g, 1, (CI) * 32 + q)] = make_float4(pv[1].COMP, pv[3].COMP, pv[5].COMP, pv[7].COMP);
        QOT_T(x, 0) QOT_T(y, 1) QOT_T(z, 2) QOT_T(w, 3)
#undef QOT_T
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
    gemm_mainloop(lds, nk, wm, wn, hi, r31, c, [&](int64_t kt) { load(kbeg + kt * kGemmBK); }, stash);
    float* Cp = Cpart + (int64_t)split * M * N;
    // epilogue through LDS; tile row p = c*32 + q holds output row 4q + c (columns likewise)
    float* tile = reinterpret_cast<float*>(&lds[0][0][0]);
    gemm_tile_to_lds(tile, c, wm, wn, hi, r31);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int idx = it * 256 + t;                 // output-ordered float4: local row ro = idx / 32, quad qo = idx % 32
        const int ro = idx >> 5, qo = idx & 31;
        const int pm = (ro & 3) * 32 + (ro >> 2);     // tile row of output row ro
        const int row = m0 + ro, col = n0 + 4 * qo;
        if (row < M && col < N) {
            // output columns 4qo + cc live at tile columns cc*32 + qo
            float v[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int pn = cc * 32 + qo;
                v[cc] = tile[pm * 128 + (pn ^ ((pm & 7) << 2))];
            }
            st4(Cp + (int64_t)row * N + col, make_float4(v[0], v[1], v[2], v[3]));
        }
    }
}

}  // namespace qot

using namespace qot;

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
    const int64_t tiles = ((M + kGemmBM - 1) / kGemmBM) * ((N + kGemmBN - 1) / kGemmBN);
    if (tiles > 0x7fffffff) return QOT_ERR_UNSUPPORTED;
    if (scale)
        gemm_nt_kernel<true><<<(int)tiles, 256, 0, (hipStream_t)stream>>>(A, lda, B, ldb, C, ldc, M, N, K, scale, shift, bias);
    else
        gemm_nt_kernel<false><<<(int)tiles, 256, 0, (hipStream_t)stream>>>(A, lda, B, ldb, C, ldc, M, N, K, nullptr, nullptr, bias);
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

// split count of qot_gemm_tn for a [K, M]^T [K, N] product (enough workgroups to fill the part, chunks of >= 1024 rows)
extern "C" int qot_gemm_tn_splits(int M, int N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int64_t tiles = (int64_t)((M + kGemmBM - 1) / kGemmBM) * ((N + kGemmBN - 1) / kGemmBN);
    int64_t s = (4 * (int64_t)num_cus() + tiles - 1) / tiles;
    const int64_t maxs = (K + 1023) / 1024;
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
    if ((M & 3) || (N & 3) || (lda & 3) || (ldb & 3) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)Cpart & 15))
        return QOT_ERR_UNSUPPORTED;
    int64_t kchunk = (K + splits - 1) / splits;
    kchunk = (kchunk + kGemmBK - 1) / kGemmBK * kGemmBK;
    const int64_t grid = (int64_t)((N + kGemmBN - 1) / kGemmBN) * ((M + kGemmBM - 1) / kGemmBM) * splits;
    if (grid > 0x7fffffff) return QOT_ERR_UNSUPPORTED;
    if (scale)
        gemm_tn_split_kernel<true><<<(int)grid, 256, 0, (hipStream_t)stream>>>(A, lda, B, ldb, Cpart, M, N, K, kchunk, splits, scale, shift);
    else
        gemm_tn_split_kernel<false><<<(int)grid, 256, 0, (hipStream_t)stream>>>(A, lda, B, ldb, Cpart, M, N, K, kchunk, splits, nullptr, nullptr);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}
