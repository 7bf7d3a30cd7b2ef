// TransformerConv in table mode, GRAPH form: shared layouts and the two table-level device bodies that ride in the
// multi-role launches (roles.hip).
//
// The reference's only mode is x = node_embeddings(node_ids) with node_ids = arange(n) in every graph
// (topological_training/models.py:51-53, dataset.py:78).  Then every query / key / value / skip row is a row of the
// PROJECTED table T = emb W^T + b, and the dense part of an attention logit is an entry of the n x n score matrix
//     M = T_q T_k^T / sqrt(H),      <q_i, k_j + We ea_e> / sqrt(H) = M[r_i, r_j] + <P[r_i], ea_e>,   P = T_q We / sqrt(H)
// (r = node index inside its graph).  With n of the order of 100 both M (40 KB) and T_v (26 KB) live in LDS, a
// workgroup takes WHOLE graphs (a graph's index and edge-feature slices are contiguous), and per edge the forward reads
// one M entry and one value row from LDS -- no key gather, no dot, no lane reduction, no dependent global loads.
// Backward: the gradients of T_q / T_k return through grad M (one scalar per edge) and grad P; only grad T_v needs the
// per-edge H-vector.  Per workgroup ONE partial row
//     [ gTv n x H | gTs n x H | gM n x ldm | gP n x D | gWe H x D ]
// is left for the step's fixed-order row sum (QOT_ROLE_SUM_ROWS); the projection's backward then forms its q / k columns
//     grad T_q = (gM T_k + gP We^T) / sqrt(H),      grad T_k = gM^T T_q / sqrt(H)
// on the fly from the summed row (table_project_bwd_scores_body) -- the [n, 4H] table gradient is never materialised.
#pragma once
#include "common.hpp"
#include "small_dev.hpp"

namespace qot {

struct TgRow {          // float offsets inside a partial / summed row
    int ldm, off_gv, off_gs, off_gm, off_gp, off_gwe, len;
};
__host__ __device__ inline int pad4(int v) { return (v + 3) & ~3; }
__host__ __device__ inline TgRow tg_row(int n, int H, int D) {
    TgRow r;
    r.ldm = pad4(n);
    r.off_gv = 0;
    r.off_gs = n * H;
    r.off_gm = 2 * n * H;
    r.off_gp = r.off_gm + n * r.ldm;
    r.off_gwe = pad4(r.off_gp + n * D);
    r.len = pad4(r.off_gwe + H * D);
    return r;
}

constexpr int kTgMaxN = 128;             // nodes per graph the graph form takes (register-resident accumulators, LDS)
constexpr int kScoreRows = 4;            // query rows per workgroup of table_scores_body (latency-bound: many small workgroups)

// ---- M [n, ldm] and P [n, D] from the PARAMETERS (no dependence on the projected table: the job shares the forward
// prologue's launch with the projection itself).  With u_r = Wk^T T_q[r] and beta_r = <T_q[r], bk>:
//     <T_q[r], T_k[j]> = <emb[j], u_r> + beta_r
// so a workgroup of kScoreRows query rows needs those rows' T_q (R H^2 MACs), their u (R H^2) and R n dots of H -- no
// T_k.  256 threads; `lds`: kScoreRows * (3 H + 1) floats.
template <int H>
__device__ __forceinline__ void table_scores_body(const float* __restrict__ table, const float* __restrict__ wq,
                                                  const float* __restrict__ bq, const float* __restrict__ wk,
                                                  const float* __restrict__ bk, const float* __restrict__ we,
                                                  float* __restrict__ M, float* __restrict__ Pm, int n, int D, int vb,
                                                  float* __restrict__ lds) {
    constexpr int R = kScoreRows;
    float* er = lds;                 // [R][H] embedding rows
    float* tq = er + R * H;          // [R][H] projected query rows
    float* u = tq + R * H;           // [R][H]
    float* beta = u + R * H;         // [R]
    const int r0 = vb * R;
    const int ldm = pad4(n);
    const float rs = rsqrtf((float)H);
    for (int c = threadIdx.x; c < R * H; c += 256) {
        const int r = c / H;
        er[c] = r0 + r < n ? table[(int64_t)(r0 + r) * H + c % H] : 0.f;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < R * H; c += 256) {          // T_q rows, summed as qot_table_project_fwd sums them
        const int r = c / H, o = c % H;
        const float* w = wq + (int64_t)o * H;
        float acc = bq[o];
#pragma unroll 8
        for (int a = 0; a < H; a += 4) {          // (eight loads in flight: the job is a chain of L2 round trips otherwise)
            const float4 ww = ld4(w + a);
            const float4 xv = *reinterpret_cast<const float4*>(er + r * H + a);
            acc = fmaf(ww.x, xv.x, acc); acc = fmaf(ww.y, xv.y, acc);
            acc = fmaf(ww.z, xv.z, acc); acc = fmaf(ww.w, xv.w, acc);
        }
        tq[c] = acc;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < R * H; c += 256) {          // u[r][a] = sum_c T_q[r][c] wk[c][a]
        const int r = c / H, a = c % H;
        float acc = 0.f;
#pragma unroll 8
        for (int k = 0; k < H; ++k) acc = fmaf(tq[r * H + k], wk[(int64_t)k * H + a], acc);
        u[c] = acc;
    }
    if (threadIdx.x < R) {
        float acc = 0.f;
        for (int k = 0; k < H; ++k) acc = fmaf(tq[threadIdx.x * H + k], bk[k], acc);
        beta[threadIdx.x] = acc;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < R * ldm; c += 256) {        // M rows (pad columns: 0)
        const int r = c / ldm, j = c % ldm;
        if (r0 + r >= n) continue;
        float acc = 0.f;
        if (j < n) {
            const float* ej = table + (int64_t)j * H;
            acc = beta[r];
#pragma unroll 8
            for (int a = 0; a < H; a += 4) {
                const float4 ev = ld4(ej + a);
                const float4 uv = *reinterpret_cast<const float4*>(u + r * H + a);
                acc = fmaf(ev.x, uv.x, acc); acc = fmaf(ev.y, uv.y, acc);
                acc = fmaf(ev.z, uv.z, acc); acc = fmaf(ev.w, uv.w, acc);
            }
            acc *= rs;
        }
        M[(int64_t)(r0 + r) * ldm + j] = acc;
    }
    for (int c = threadIdx.x; c < R * D; c += 256) {          // P rows
        const int r = c / D, d = c % D;
        if (r0 + r >= n) continue;
        float acc = 0.f;
        for (int k = 0; k < H; ++k) acc = fmaf(tq[r * H + k], we[k * D + d], acc);
        Pm[(int64_t)(r0 + r) * D + d] = acc * rs;
    }
}

// ---- backward of the projection fed by the summed partial row S (see the header): as table_project_bwd_body (R = 1), the
// q / k columns of the table gradient formed where they are needed.  t4 = the projected table [V, 4H] (T_q | T_k | ...).
//   weight part, one workgroup per packed column c:  gcol[v] (v < n, 0 above) -> gw[c, :] = sum_v gcol[v] table[v, :], gb[c]
//   table part, one workgroup per table row v:       grow[4H] -> gt[v, :] = sum_c grow[c] w_{s(c)}[o(c), :]
//   one more workgroup: grad w_edge [H, D] = S's value-path share + T_q^T gP / sqrt(H)
// The jobs are chains of L2 round trips (a trip costs ~0.7 us next to the other jobs of the launch), so every loop requests a
// batch of independent loads before its first product, and the inner sums are split over thread halves that meet in LDS in a
// fixed order.  n <= kTgMaxN (= 128: two threads per row in the gradient-slice sums).
// `lds`: tg_bwd_scores_lds_floats(n, H, D) floats.
__host__ __device__ inline int tg_bwd_scores_lds_floats(int n, int H, int D) {
    const int a = 512 + (4 * H > kTgMaxN ? 4 * H : kTgMaxN) + 2 * kTgMaxN;     // red | redb | gradient slice (a column of up to
                                                                              // kTgMaxN rows, or a row of 4H) | halves
    const int b = n * H + pad4(n * D);                       // the w_edge job: T_q, gP
    return a > b ? a : b;
}

// sum_{k in [k0, k1)} x[k * xs] * y[k * ys]: sixteen terms' operands requested before the first product (a batch is one
// L2 round trip; the jobs of this launch are chains of them)
__device__ __forceinline__ float tg_strided_dot(const float* __restrict__ x, int64_t xs, const float* __restrict__ y, int64_t ys,
                                                int k0, int k1) {
    constexpr int U = 16;
    float acc = 0.f;
    for (int k = k0; k < k1; k += U) {
        float a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = k + u < k1 ? k + u : k0;
            a[u] = x[(int64_t)kk * xs]; b[u] = y[(int64_t)kk * ys];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = fmaf(k + u < k1 ? a[u] : 0.f, b[u], acc);
    }
    return acc;
}

template <int H>
__device__ __forceinline__ void table_project_bwd_scores_body(const float* __restrict__ S, const float* __restrict__ t4,
                                                              const float* __restrict__ we, const float* __restrict__ table,
                                                              const Proj4& p, float* __restrict__ gtable,
                                                              float* __restrict__ gw, float* __restrict__ gb,
                                                              float* __restrict__ gwe, int V, int n, int D, int vb,
                                                              float* __restrict__ lds) {
    constexpr int PH = (H >= 256) ? 1 : 256 / H;
    const TgRow L = tg_row(n, H, D);
    const float rs = rsqrtf((float)H);
    float* red = lds;                  // [256]
    float* redb = lds + 256;           // [256]
    float* gs = lds + 512;             // the gradient slice: column (n) or row (4H)
    float* half = gs + (4 * H > kTgMaxN ? 4 * H : kTgMaxN);      // [2][kTgMaxN] halves of a column sum
    const int a = threadIdx.x % H, ph = threadIdx.x / H;
    const float* gM = S + L.off_gm;
    if (vb < 4 * H) {
        // ---- weight part, packed column c
        const int c = vb, s = c / H, o = c % H;
        const int v = threadIdx.x % kTgMaxN, hf = threadIdx.x / kTgMaxN;          // two threads per row v
        const int mid = pad4((n + 1) / 2);
        const int k0 = hf == 0 ? 0 : (mid < n ? mid : n), k1 = hf == 0 ? (mid < n ? mid : n) : n;
        float part = 0.f;
        if (v < n) {
            if (s == 0) part = tg_strided_dot(gM + (int64_t)v * L.ldm, 1, t4 + H + o, 4 * H, k0, k1);        // sum_j gM[v][j] T_k[j][o]
            else if (s == 1) part = tg_strided_dot(gM + v, L.ldm, t4 + o, 4 * H, k0, k1);                    // sum_r gM[r][v] T_q[r][o]
            else if (hf == 0) part = S[(s == 2 ? L.off_gv : L.off_gs) + v * H + o];
        }
        half[hf * kTgMaxN + v] = part;
        __syncthreads();
        if (threadIdx.x < kTgMaxN) {
            float g = 0.f;
            if (v < n) {
                g = half[v] + half[kTgMaxN + v];
                if (s == 0) {
                    for (int d = 0; d < D; ++d) g = fmaf(S[L.off_gp + v * D + d], we[o * D + d], g);
                }
                if (s < 2) g *= rs;
            }
            gs[v] = g;
        }
        __syncthreads();
        float acc = 0.f, sb = 0.f;
        int vv = ph;
        for (; vv < n; vv += 16 * PH) {
            float tv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int v2 = vv + u * PH;
                tv[u] = table[(int64_t)(v2 < n ? v2 : vv) * H + a];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int v2 = vv + u * PH;
                const float gv = v2 < n ? gs[v2] : 0.f;
                acc = fmaf(gv, tv[u], acc);
                sb += gv;
            }
        }
        red[threadIdx.x] = acc;
        redb[threadIdx.x] = sb;
        __syncthreads();
        if (ph == 0) {
            float sa = acc, t = sb;
            for (int k = 1; k < PH; ++k) { sa += red[k * H + a]; t += redb[k * H + a]; }
            gw[(int64_t)c * H + a] = sa;
            if (a == 0) gb[c] = t;
        }
    } else if (vb < 4 * H + V) {
        // ---- table part, row v
        const int v = vb - 4 * H;
        if (v >= n) {                  // rows no node refers to: zero gradient
            if (ph == 0) gtable[(int64_t)v * H + a] = 0.f;
            return;
        }
        // the row of the gradient: thread (o, ph) sums a share of the inner index for the q and the k column o
        {
            const int per = (n + PH - 1) / PH;
            const int k0 = ph * per < n ? ph * per : n, k1 = (ph + 1) * per < n ? (ph + 1) * per : n;
            const float pq = tg_strided_dot(gM + (int64_t)v * L.ldm, 1, t4 + H + a, 4 * H, k0, k1);
            const float pk = tg_strided_dot(gM + v, L.ldm, t4 + a, 4 * H, k0, k1);
            red[threadIdx.x] = pq;
            redb[threadIdx.x] = pk;
        }
        __syncthreads();
        if (ph == 0) {
            float q = red[a], k = redb[a];
            for (int u = 1; u < PH; ++u) { q += red[u * H + a]; k += redb[u * H + a]; }
            for (int d = 0; d < D; ++d) q = fmaf(S[L.off_gp + v * D + d], we[a * D + d], q);
            gs[a] = q * rs;
            gs[H + a] = k * rs;
            gs[2 * H + a] = S[L.off_gv + v * H + a];
            gs[3 * H + a] = S[L.off_gs + v * H + a];
        }
        __syncthreads();
        float acc = 0.f;
        for (int c0 = ph; c0 < 4 * H; c0 += 16 * PH) {             // 4H / PH trips: a multiple of 16 for every width
            float wv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int c = c0 + u * PH;
                const int cc = c < 4 * H ? c : ph;
                wv[u] = proj_w(p, cc / H)[(int64_t)(cc % H) * H + a];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int c = c0 + u * PH;
                if (c < 4 * H) acc = fmaf(gs[c], wv[u], acc);
            }
        }
        __syncthreads();
        red[threadIdx.x] = acc;
        __syncthreads();
        if (ph == 0) {
            float sa = acc;
            for (int k = 1; k < PH; ++k) sa += red[k * H + a];
            gtable[(int64_t)v * H + a] = sa;
        }
    } else {
        // ---- grad w_edge: value path (already summed in S) + T_q^T gP / sqrt(H); T_q and gP staged in LDS
        float* tq = lds;                       // [n][H]
        float* gp = lds + n * H;               // [n][D]
        for (int i0 = threadIdx.x; i0 < n * H; i0 += 8 * 256) {
            float q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256;
                const int ii = i < n * H ? i : 0;
                q[u] = t4[(int64_t)(ii / H) * 4 * H + ii % H];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256;
                if (i < n * H) tq[i] = q[u];
            }
        }
        for (int i = threadIdx.x; i < n * D; i += 256) gp[i] = S[L.off_gp + i];
        __syncthreads();
        for (int o = threadIdx.x; o < H * D; o += 256) {
            const int c = o / D, d = o % D;
            float acc = 0.f;
            for (int r = 0; r < n; ++r) acc = fmaf(tq[r * H + c], gp[r * D + d], acc);
            gwe[o] = fmaf(rs, acc, S[L.off_gwe + o]);
        }
    }
}

}  // namespace qot
