// Device bodies of the step's small dense pieces, shared by the multi-role launch (roles.hip).
#pragma once
#include "common.hpp"

namespace qot {

// ---- table projection ------------------------------------------------------------------------
struct Proj4 {
    const float* w[4];      // each [H, H] (out, in)
    const float* b[4];      // each [H]
};
// (selects instead of a runtime array index: that sends a struct built in registers to scratch)
__device__ __forceinline__ const float* proj_w(const Proj4& p, int s) { return s == 0 ? p.w[0] : (s == 1 ? p.w[1] : (s == 2 ? p.w[2] : p.w[3])); }
__device__ __forceinline__ const float* proj_b(const Proj4& p, int s) { return s == 0 ? p.b[0] : (s == 1 ? p.b[1] : (s == 2 ? p.b[2] : p.b[3])); }

// Rows / outputs per workgroup.  One row per workgroup (R = 1) is what a small table wants (V = 100 at cfg2: latency);
// a big one (V >= kTableBlockMinV: cfg4 / cfg5's 1000 nodes) is walked kTableBlock rows at a time so that a weight value
// read from L2 serves kTableBlock rows -- with one row per workgroup every workgroup re-read all 4 H^2 weights: 1 GB of L2
// traffic at V = 1000, H = 256 for a 0.5 GFLOP product (144 us forward, 238 us backward at cfg5).
constexpr int kTableBlock = 8;
constexpr int kTableBlockMinV = 512;
__host__ __device__ inline int table_rows_per_block(int64_t V) { return V >= kTableBlockMinV ? kTableBlock : 1; }

// out[v, s*H + o] = b_s[o] + sum_a table[v, a] * w_s[o, a];  one 256-thread workgroup per R table rows; `row`: R*H floats of LDS
template <int H, int R>
__device__ __forceinline__ void table_project_fwd_rows(const float* __restrict__ table, const Proj4& p,
                                                       float* __restrict__ out, int V, int v0, float* __restrict__ row) {
    for (int c = threadIdx.x; c < R * H; c += 256) {
        const int r = c / H, a = c % H;
        row[c] = v0 + r < V ? table[(int64_t)(v0 + r) * H + a] : 0.f;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 4 * H; c += 256) {
        // (H >= 64: the projection index is the same for a whole wave -- said so, the pointer selects below are scalar;
        // as a per-lane index the compiler turned them into an indexed read of a copy of the pointers in scratch)
        const int s = H >= 64 ? __builtin_amdgcn_readfirstlane(c / H) : c / H, o = c % H;
        const float* w = proj_w(p, s) + (int64_t)o * H;
        float acc[R];
        // (four loads and a value select: a select of the four bias POINTERS was compiled into an indexed read of a copy of
        // them in scratch)
        const float b0 = p.b[0][o], b1 = p.b[1][o], b2 = p.b[2][o], b3 = p.b[3][o];
        const float bz = s == 0 ? b0 : (s == 1 ? b1 : (s == 2 ? b2 : b3));
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = bz;
#pragma unroll 4
        for (int a = 0; a < H; a += 4) {
            const float4 ww = ld4(w + a);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float4 xv = *reinterpret_cast<const float4*>(row + r * H + a);
                acc[r] = fmaf(ww.x, xv.x, acc[r]); acc[r] = fmaf(ww.y, xv.y, acc[r]);
                acc[r] = fmaf(ww.z, xv.z, acc[r]); acc[r] = fmaf(ww.w, xv.w, acc[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (v0 + r < V) out[(int64_t)(v0 + r) * 4 * H + c] = acc[r];
    }
}

// R = 1 inside the multi-role kernel, R = kTableBlock in a kernel of its own (roles.hip): the blocked bodies' registers
// (150 + scratch against 74) must not set the occupancy of every other job of the multi-role launch.
template <int H, int R>
__device__ __forceinline__ void table_project_fwd_body(const float* __restrict__ table, const Proj4& p,
                                                       float* __restrict__ out, int64_t* __restrict__ counter,
                                                       int64_t* __restrict__ snapshot, int V, int vb, float* __restrict__ row) {
    if (counter && vb == 0 && threadIdx.x == 0) {   // as step_advance_kernel: this is the forward's first launch in table mode
        const int64_t cc = counter[0] + 1;
        counter[0] = cc;
        snapshot[0] = cc;
    }
    table_project_fwd_rows<H, R>(table, p, out, V, vb * R, row);
}

// Backward of the projection.  grads: gw [4H, H] | gb [4H] (packed q|k|v|skip order), gtable [V, H].
//   weight part, one workgroup per CB packed rows c:  gw[c, a] = sum_v gp[v, c] table[v, a],  gb[c] = sum_v gp[v, c]
//   table part, one workgroup per R table rows v:     gt[v, a] = sum_c gp[v, c] w_{s(c)}[o(c), a]
// 256 threads = H columns x PH phases of the reduction index (the loops are pure latency otherwise), phases meet in LDS
// in a fixed order.  Virtual blocks: [0, 4H / CB) weight part, then ceil(V / R) table part; CB = R = table_rows_per_block(V).
// `lds`: 512 * CB + R * 4H floats
template <int H, int CB>
__device__ __forceinline__ void table_project_bwd_w(const float* __restrict__ gp, const float* __restrict__ table,
                                                    float* __restrict__ gw, float* __restrict__ gb, int V, int c0,
                                                    float* __restrict__ lds) {
    constexpr int PH = (H >= 256) ? 1 : 256 / H;
    float* red = lds;                 // [CB][256]
    float* redb = lds + 256 * CB;     // [CB][256]
    const int a = threadIdx.x % H, ph = threadIdx.x / H;
    float acc[CB], sb[CB];
#pragma unroll
    for (int u = 0; u < CB; ++u) { acc[u] = 0.f; sb[u] = 0.f; }
    if constexpr (CB == 1) {
        const int c = c0;
        // (eight rows' operands requested before the first product: load -> use per trip serialises the round trips)
        int v = ph;
        for (; v + 7 * PH < V; v += 8 * PH) {
            float gv[8], tv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                gv[u] = gp[(int64_t)(v + u * PH) * 4 * H + c];
                tv[u] = table[(int64_t)(v + u * PH) * H + a];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc[0] = fmaf(gv[u], tv[u], acc[0]); sb[0] += gv[u]; }
        }
        for (; v < V; v += PH) {
            const float gv = gp[(int64_t)v * 4 * H + c];
            acc[0] = fmaf(gv, table[(int64_t)v * H + a], acc[0]);
            sb[0] += gv;
        }
    } else {
        static_assert(CB == 1 || CB == 8, "two float4 of gp per row");
        int v = ph;
        for (; v + 3 * PH < V; v += 4 * PH) {          // four rows in flight
            float4 g0[4], g1[4];
            float tv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* gr = gp + (int64_t)(v + u * PH) * 4 * H + c0;
                g0[u] = ld4(gr); g1[u] = ld4(gr + 4);
                tv[u] = table[(int64_t)(v + u * PH) * H + a];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float gv[8] = {g0[u].x, g0[u].y, g0[u].z, g0[u].w, g1[u].x, g1[u].y, g1[u].z, g1[u].w};
#pragma unroll
                for (int q = 0; q < 8; ++q) { acc[q] = fmaf(gv[q], tv[u], acc[q]); sb[q] += gv[q]; }
            }
        }
        for (; v < V; v += PH) {
            const float* gr = gp + (int64_t)v * 4 * H + c0;
            const float4 g0 = ld4(gr), g1 = ld4(gr + 4);
            const float tv = table[(int64_t)v * H + a];
            const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
            for (int q = 0; q < 8; ++q) { acc[q] = fmaf(gv[q], tv, acc[q]); sb[q] += gv[q]; }
        }
    }
#pragma unroll
    for (int u = 0; u < CB; ++u) { red[u * 256 + threadIdx.x] = acc[u]; redb[u * 256 + threadIdx.x] = sb[u]; }
    __syncthreads();
    if (ph == 0) {
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            float s = acc[u], t = sb[u];
            for (int k = 1; k < PH; ++k) { s += red[u * 256 + k * H + a]; t += redb[u * 256 + k * H + a]; }
            gw[(int64_t)(c0 + u) * H + a] = s;
            if (a == 0) gb[c0 + u] = t;
        }
    }
}

template <int H, int R>
__device__ __forceinline__ void table_project_bwd_t(const float* __restrict__ gp, const Proj4& p, float* __restrict__ gtable,
                                                    int V, int v0, float* __restrict__ lds) {
    constexpr int PH = (H >= 256) ? 1 : 256 / H;
    float* red = lds;                 // [R][256]
    float* g = lds + 512 * R;         // [R][4H]
    const int a = threadIdx.x % H, ph = threadIdx.x / H;
    for (int c = threadIdx.x; c < R * 4 * H; c += 256) {
        const int r = c / (4 * H);
        g[c] = v0 + r < V ? gp[(int64_t)(v0 + r) * 4 * H + c % (4 * H)] : 0.f;
    }
    __syncthreads();
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    for (int c0 = ph; c0 < 4 * H; c0 += 8 * PH) {              // 4H / PH trips: a multiple of 8 for every width
        float wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c0 + u * PH;
            const int cc = c < 4 * H ? c : ph;
            wv[u] = proj_w(p, cc / H)[(int64_t)(cc % H) * H + a];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c0 + u * PH;
            if (c < 4 * H) {
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = fmaf(g[r * 4 * H + c], wv[u], acc[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) red[r * 256 + threadIdx.x] = acc[r];
    __syncthreads();
    if (ph == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float s = acc[r];
            for (int k = 1; k < PH; ++k) s += red[r * 256 + k * H + a];
            if (v0 + r < V) gtable[(int64_t)(v0 + r) * H + a] = s;
        }
    }
}

template <int H, int R>
__device__ __forceinline__ void table_project_bwd_body(const float* __restrict__ gp, const float* __restrict__ table,
                                                       const Proj4& p, float* __restrict__ gtable,
                                                       float* __restrict__ gw, float* __restrict__ gb, int V, int vb,
                                                       float* __restrict__ lds) {
    constexpr int NW = 4 * H / R;
    if (vb < NW) table_project_bwd_w<H, R>(gp, table, gw, gb, V, vb * R, lds);
    else table_project_bwd_t<H, R>(gp, p, gtable, V, (vb - NW) * R, lds);
}

// ---- out[i] = concat(s0[0:n0], s1[0:n1], s2)[idx[i]] (0 where idx[i] < 0) ------------------------------------------
__device__ __forceinline__ void gather3_body(const float* __restrict__ s0, int n0, const float* __restrict__ s1, int n1,
                                             const float* __restrict__ s2, const int32_t* __restrict__ idx,
                                             float* __restrict__ out, int64_t n, int64_t vb) {
    const int64_t i = vb * 256 + threadIdx.x;
    if (i >= n) return;
    const int v = idx[i];          // v < 0: a padding slot (zero)
    out[i] = v < 0 ? 0.f : (v < n0 ? s0[v] : (v < n0 + n1 ? s1[v - n0] : s2[v - n0 - n1]));
}

}  // namespace qot
