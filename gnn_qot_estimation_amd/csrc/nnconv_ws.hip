// NNConv forward / adjoint, weight-stationary form  (H = 64, D <= 4).
//
// Same product as nnconv_mfma64_kernel (out = bias + A @ Wcat, reference site
// topological_training/models.py:57), different dataflow:
//   * Wcat never moves: each of the 8 waves of a workgroup keeps its K-eighth of Wcat
//     ((K+2)*64/8 rows x 64 columns = 80 VGPRs at D = 4) in registers for the whole persistent
//     loop.  The tile kernel re-streamed the 160 KB weight image from L2 for every 32-row tile
//     (the L2-fed MFMA loop tops out at ~110 TFLOP/s; register-fed ~134).
//   * the operand tile A ([32, 640]) is never materialised: the tile's gathered source rows are
//     copied raw into LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no VALU), the per-edge
//     weights h_e (edge MLP hidden vector, mean scale) sit beside them, and every lane forms its own
//     MFMA A operand  A[i, k*64 + c] = sum_e h_e[k] x_{j(e)}[c]  with D-ish FMAs in the shadow of the
//     MFMAs.  Gather and multiply are one phase instead of two serialised ones.
//   * lane (row r = lane & 31, half h = lane >> 5) of wave w covers k in [80w + 40h, 80w + 40h + 40)
//     in blocks of 8 (one slab, 8 consecutive channels = two ds_read_b128 of the source row per edge);
//     both 32-column halves are multiplied from the same A value (2 MFMAs per A value).
//   * LDS image of a source row: 16 x 16-B chunks, chunk q of slot s at position q ^ (s & 15)
//     (the swizzle is applied on the DMA's per-lane SOURCE address; destination is lane-linear).
//   * tiles whose in-edge count exceeds the staging capacity run extra rounds (A is linear in the
//     edges, partial operands accumulate in the same MFMA accumulators).
#include "common.hpp"

namespace qot {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

constexpr int kWsEC = 224;                 // edge slots per staging round
constexpr int kWsSlots = kWsEC + 32;       // + the 32 destinations' own rows (root slab)
constexpr int kWsHW = 12;                  // per-slot weight row: K+2 <= 10 floats, padded

__device__ __forceinline__ void ws_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ int64_t ws_xcd_tile(int64_t it, int64_t ntiles) {
    const int nx = 8;
    if ((int)gridDim.x % nx != 0) {
        const int64_t t = (int64_t)blockIdx.x + it * gridDim.x;
        return t < ntiles ? t : -1;
    }
    const int xcd = blockIdx.x % nx, slot = blockIdx.x / nx, per_x = gridDim.x / nx;
    const int64_t chunk = (ntiles + nx - 1) / nx;
    const int64_t local = slot + it * per_x;
    const int64_t t = xcd * chunk + local;
    return (local < chunk && t < ntiles) ? t : -1;
}

// diagnostic build only (STAMP): per-phase clock sums, one adder per wave
__device__ unsigned long long g_ws_stamps[8];
#define QOT_WS_STAMP(slot)                                                           \
    if (STAMP) {                                                                     \
        unsigned long long _t = __builtin_amdgcn_s_memtime();                        \
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_ws_stamps[slot], _t - t_prev);     \
        t_prev = _t;                                                                 \
    }

template <int D, bool TRANSPOSE, bool STAMP = false>
__global__ __launch_bounds__(512, 2) void nnconv_ws64_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ ea, const float* __restrict__ w1,
    const float* __restrict__ b1, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const int32_t* __restrict__ eidx, const float* __restrict__ invdeg, const float* __restrict__ Wp,
    const float* __restrict__ bias, float* __restrict__ out, int64_t N, ActParams act) {
    constexpr int K = 2 * D, S = K + 2, KT = S * 64;
    constexpr int NBLK = KT / 128;          // blocks of 8 k-steps per lane half
    constexpr int GT = KT / 8;              // Wp groups of 4 k-steps... (8 k values) per column half
    static_assert(S <= kWsHW, "weight row too narrow");
    __shared__ __attribute__((aligned(16))) float xbuf[kWsSlots * 64];
    __shared__ __attribute__((aligned(16))) float hbuf[kWsSlots * kWsHW];
    __shared__ int srcbuf[kWsSlots];
    __shared__ int rpbuf[36];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    const int base_kk = wave * (KT / 8) + h * (KT / 16);

    // resident weights: wreg[b][i][nh] = Wcat[base_kk + 8b + i][32 nh + r]   (Wp: see qot_nnconv_fused)
    float wreg[NBLK][8][2];
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) {
                const int g = (base_kk >> 3) + b;
                wreg[b][i][nh] = Wp[(((int64_t)nh * GT + g) * 64 + (i & 1) * 32 + r) * 4 + (i >> 1)];
            }

    const int64_t ntiles = (N + 31) / 32;
    const float4* xb4 = reinterpret_cast<const float4*>(xbuf);
#pragma unroll 1
    for (int64_t it = 0;; ++it) {
        const int64_t tile = ws_xcd_tile(it, ntiles);
        if (tile < 0) break;
        const int64_t tile0 = tile * 32;
        const int64_t tend = tile0 + 32 < N ? tile0 + 32 : N;
        const int e0 = rowptr[tile0];
        const int ne = rowptr[tend] - e0;
        f32x16 c0, c1;
#pragma unroll
        for (int j = 0; j < 16; ++j) { c0[j] = 0.f; c1[j] = 0.f; }
        const int64_t irow = tile0 + r;
        unsigned long long t_prev = 0;
        if (STAMP) t_prev = __builtin_amdgcn_s_memtime();
        const float inv = (!TRANSPOSE && irow < N) ? invdeg[irow] : 1.0f;

#pragma unroll 1
        for (int lo = 0; lo == 0 || lo < ne; lo += kWsEC) {
            const int cnt = ne - lo < kWsEC ? ne - lo : kWsEC;
            // ---- per-slot metadata: source row, edge-MLP hidden vector (x mean scale of the adjoint)
            if (t < 33) {
                const int64_t ii = tile0 + t < N ? tile0 + t : N;
                rpbuf[t] = rowptr[ii] - e0;
            }
            if (t < cnt) {
                const int p = e0 + lo + t;
                const int src = col[p];
                const int64_t e = eidx[p];
                float ee[D];
#pragma unroll
                for (int d = 0; d < D; ++d) ee[d] = ea[e * D + d];
                const float sc = TRANSPOSE ? invdeg[src] : 1.0f;
#pragma unroll
                for (int kk = 0; kk < K; ++kk) {
                    float hv = b1[kk];
#pragma unroll
                    for (int d = 0; d < D; ++d) hv = fmaf(w1[kk * D + d], ee[d], hv);
                    hbuf[t * kWsHW + kk] = fmaxf(hv, 0.f) * sc;
                }
                hbuf[t * kWsHW + K] = sc;
                hbuf[t * kWsHW + K + 1] = 0.f;
                srcbuf[t] = src;
            } else if (lo == 0 && t >= kWsEC && t < kWsSlots) {
                const int64_t ii = tile0 + (t - kWsEC);
#pragma unroll
                for (int kk = 0; kk <= K; ++kk) hbuf[t * kWsHW + kk] = 0.f;
                hbuf[t * kWsHW + K + 1] = ii < N ? 1.0f : 0.f;
                srcbuf[t] = (int)(ii < N ? ii : N - 1);
            }
            QOT_WS_STAMP(0)
            ws_lds_barrier();
            QOT_WS_STAMP(1)
            // ---- source rows -> LDS by DMA: 16 lanes per row, 4 rows per wave-instruction
#pragma unroll
            for (int q = 0; q < kWsSlots / 32; ++q) {
                const int sbase = 32 * q + 4 * wave;
                const int slot = sbase + (lane >> 4);
                const bool live = slot < cnt || (lo == 0 && slot >= kWsEC);
                if (live) {
                    const int src = srcbuf[slot];
                    const int chunk = (lane & 15) ^ (slot & 15);
                    __builtin_amdgcn_global_load_lds((glb_ptr_t)(x + (int64_t)src * ldx + chunk * 4),
                                                     (lds_ptr_t)(xbuf + sbase * 64), 16, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            QOT_WS_STAMP(2)
            ws_lds_barrier();
            QOT_WS_STAMP(3)

            // ---- operand formation + MFMA
            int beg = rpbuf[r] - lo, end = rpbuf[r + 1] - lo;
            beg = beg < 0 ? 0 : beg;
            end = end > kWsEC ? kWsEC : end;
            const int deg = end > beg ? end - beg : 0;
            const int n_it = deg + (lo == 0 ? 1 : 0);
            int maxit = n_it;
#pragma unroll
            for (int off = 32; off; off >>= 1) {
                const int o = __shfl_xor(maxit, off);
                maxit = o > maxit ? o : maxit;
            }
            maxit = __builtin_amdgcn_readfirstlane(maxit);
            const int ownslot = kWsEC + r;
#pragma unroll
            for (int b = 0; b < NBLK; ++b) {
                const int kk0 = base_kk + 8 * b;
                const int slab = kk0 >> 6, cq = (kk0 & 63) >> 2;
                float4 a0 = f4zero(), a1 = f4zero();
#pragma unroll 1
                for (int d0 = 0; d0 < maxit; d0 += 4) {      // 4 edges in flight: 4 weight reads + 8 row reads, then 32 FMAs
                    float wv[4];
                    float4 xa[4], xc[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int d = d0 + u;
                        const bool isedge = d < deg;
                        const int slot = isedge ? beg + d : ownslot;
                        const float wr = hbuf[slot * kWsHW + slab];
                        wv[u] = d < n_it ? (isedge ? wr * inv : wr) : 0.f;
                        const int pos = cq ^ (slot & 15);
                        xa[u] = xb4[slot * 16 + pos];
                        xc[u] = xb4[slot * 16 + (pos ^ 1)];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        a0 = fma4(wv[u], xa[u], a0);
                        a1 = fma4(wv[u], xc[u], a1);
                    }
                }
                const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], wreg[b][i][0], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], wreg[b][i][1], c1, 0, 0, 0);
                }
            }
            QOT_WS_STAMP(4)
            ws_lds_barrier();                      // every wave is done with this round's rows
            QOT_WS_STAMP(5)
        }

        // ---- K-eighths meet through LDS (aliases the row buffer), bias / activation, 256-B row stores
        float* red = xbuf;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int row = (j & 3) + 8 * (j >> 2) + 4 * h;
            red[(wave * 32 + row) * 64 + r] = c0[j];
            red[(wave * 32 + row) * 64 + 32 + r] = c1[j];
        }
        ws_lds_barrier();
        {
            const int row = t >> 4, c4 = t & 15;
            float4 v = bias ? ld4(bias + 4 * c4) : f4zero();
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const float4 p = xb4[(w * 32 + row) * 16 + c4];
                v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
            }
            const int64_t i = tile0 + row;
            if (i < N) {
                v = act_apply4(v, act, (uint64_t)(i * 16 + c4));
                *reinterpret_cast<float4*>(out + i * 64 + 4 * c4) = v;
            }
        }
        ws_lds_barrier();                          // `red` consumed before the next tile's DMA lands
        QOT_WS_STAMP(6)
    }
}

}  // namespace qot

using namespace qot;

static int g_ws_stamp = 0;   // tools/ablate_ws.py
extern "C" void qot_debug_ws_stamps(unsigned long long* host8, int mode) {
    if (mode == 1) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(qot::g_ws_stamps), z, sizeof(z)); g_ws_stamp = 1; }
    else if (mode == 2) g_ws_stamp = 0;
    else (void)hipMemcpyFromSymbol(host8, HIP_SYMBOL(qot::g_ws_stamps), 8 * sizeof(unsigned long long));
}

// Same contract as qot_nnconv_fused (nnconv_mfma.hip); D <= 4 only.
extern "C" int qot_nnconv_fused_ws(const float* x, int ld_x, const float* edge_attr, const float* w1,
                                   const float* b1, const int32_t* rowptr, const int32_t* col,
                                   const int32_t* edge_ids, const float* invdeg, int transpose,
                                   const float* w_perm, const float* bias, float* out, int64_t N, int H, int D,
                                   int act, float act_slope, float act_p, uint64_t act_seed,
                                   const int64_t* act_step, qot_stream_t stream) {
    if (N < 0 || !rowptr) return QOT_ERR_BADARG;
    if (H != 64 || D < 1 || D > 4) return QOT_ERR_UNSUPPORTED;
    if (N == 0) return QOT_OK;
    if (!x || !w1 || !b1 || !invdeg || !w_perm || !out || (ld_x & 3)) return QOT_ERR_BADARG;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    int grid = grid_for(N, 32);
    if (grid > ncu) grid = ncu;
    const ActParams ap = make_act(act, act_slope, act_p, act_seed, act_step);
    if (g_ws_stamp && D == 4 && !transpose) {
        nnconv_ws64_kernel<4, false, true><<<grid, 512, 0, (hipStream_t)stream>>>(
            x, ld_x, edge_attr, w1, b1, rowptr, col, edge_ids, invdeg, w_perm, bias, out, N, ap);
        QOT_LAUNCH_CHECK();
        return QOT_OK;
    }
    QOT_DISPATCH_D(D, {
        if (kD <= 4) {
            constexpr int kDD = kD <= 4 ? kD : 4;
            if (transpose)
                nnconv_ws64_kernel<kDD, true><<<grid, 512, 0, (hipStream_t)stream>>>(
                    x, ld_x, edge_attr, w1, b1, rowptr, col, edge_ids, invdeg, w_perm, bias, out, N, ap);
            else
                nnconv_ws64_kernel<kDD, false><<<grid, 512, 0, (hipStream_t)stream>>>(
                    x, ld_x, edge_attr, w1, b1, rowptr, col, edge_ids, invdeg, w_perm, bias, out, N, ap);
        }
    });
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}
