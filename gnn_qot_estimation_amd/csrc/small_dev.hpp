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

// out[v, s*H + o] = b_s[o] + sum_a table[v, a] * w_s[o, a];  one 256-thread workgroup per table row v; `row`: H floats of LDS
template <int H>
__device__ __forceinline__ void table_project_fwd_body(const float* __restrict__ table, const Proj4& p,
                                                       float* __restrict__ out, int64_t* __restrict__ counter,
                                                       int64_t* __restrict__ snapshot, int v, float* __restrict__ row) {
    if (counter && v == 0 && threadIdx.x == 0) {   // as step_advance_kernel: this is the forward's first launch in table mode
        const int64_t cc = counter[0] + 1;
        counter[0] = cc;
        snapshot[0] = cc;
    }
    for (int c = threadIdx.x; c < H; c += 256) row[c] = table[(int64_t)v * H + c];
    __syncthreads();
    for (int c = threadIdx.x; c < 4 * H; c += 256) {
        const int s = c / H, o = c % H;
        const float* w = proj_w(p, s) + (int64_t)o * H;
        float acc = proj_b(p, s)[o];
#pragma unroll 8
        for (int a = 0; a < H; a += 4) {
            const float4 ww = ld4(w + a);
            acc = fmaf(ww.x, row[a], acc); acc = fmaf(ww.y, row[a + 1], acc);
            acc = fmaf(ww.z, row[a + 2], acc); acc = fmaf(ww.w, row[a + 3], acc);
        }
        out[(int64_t)v * 4 * H + c] = acc;
    }
}

// blocks [0, 4H): weight + bias gradient of packed row c (= s*H + o):  gw[c, a] = sum_v gp[v, c] table[v, a]
// blocks [4H, 4H + V): table gradient row v:                          gt[v, a] = sum_c gp[v, c] w_{s(c)}[o(c), a]
// grads: gw [4H, H] | gb [4H]   (packed q|k|v|skip order).  256 threads = H columns x PH phases of the
// reduction index (the loops are pure latency otherwise), phases meet in LDS in a fixed order.
// virtual block vb of 4H + V; `lds`: 512 + 4H floats
template <int H>
__device__ __forceinline__ void table_project_bwd_body(const float* __restrict__ gp, const float* __restrict__ table,
                                                       const Proj4& p, float* __restrict__ gtable,
                                                       float* __restrict__ gw, float* __restrict__ gb, int V, int vb,
                                                       float* __restrict__ lds) {
    constexpr int PH = (H >= 256) ? 1 : 256 / H;
    float* red = lds;
    float* redb = lds + 256;
    float* g = lds + 512;
    const int a = threadIdx.x % H, ph = threadIdx.x / H;
    float acc = 0.f, sb = 0.f;
    if (vb < 4 * H) {
        const int c = vb;
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
            for (int u = 0; u < 8; ++u) { acc = fmaf(gv[u], tv[u], acc); sb += gv[u]; }
        }
        for (; v < V; v += PH) {
            const float gv = gp[(int64_t)v * 4 * H + c];
            acc = fmaf(gv, table[(int64_t)v * H + a], acc);
            sb += gv;
        }
        red[threadIdx.x] = acc;
        redb[threadIdx.x] = sb;
        __syncthreads();
        if (ph == 0) {
            for (int k = 1; k < PH; ++k) { acc += red[k * H + a]; sb += redb[k * H + a]; }
            gw[(int64_t)c * H + a] = acc;
            if (a == 0) gb[c] = sb;
        }
    } else {
        const int v = vb - 4 * H;
        for (int c = threadIdx.x; c < 4 * H; c += 256) g[c] = gp[(int64_t)v * 4 * H + c];
        __syncthreads();
        for (int c0 = ph; c0 < 4 * H; c0 += 8 * PH) {              // 4H / PH = 16 H / 256 * ... trips: a multiple of 8 for every width
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
                if (c < 4 * H) acc = fmaf(g[c], wv[u], acc);
            }
        }
        red[threadIdx.x] = acc;
        __syncthreads();
        if (ph == 0) {
            for (int k = 1; k < PH; ++k) acc += red[k * H + a];
            gtable[(int64_t)v * H + a] = acc;
        }
    }
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
