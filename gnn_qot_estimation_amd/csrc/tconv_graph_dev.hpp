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
constexpr int kScoreRows = 8;            // query rows per workgroup of table_scores_body

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
#pragma unroll 4
        for (int a = 0; a < H; a += 4) {
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
#pragma unroll 4
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
// `lds`: 512 + max(4H, n) floats.
template <int H>
__device__ __forceinline__ float tg_gp_entry(const float* __restrict__ S, const TgRow& L, const float* __restrict__ t4,
                                             const float* __restrict__ we, int n, int D, int v, int c, float rs) {
    if (v >= n) return 0.f;
    const int s = c / H, o = c % H;
    if (s == 2) return S[L.off_gv + v * H + o];
    if (s == 3) return S[L.off_gs + v * H + o];
    float acc = 0.f;
    // (eight terms' operands requested before the first product: a load -> use chain per trip serialises the round trips)
    if (s == 0) {            // rs (sum_j gM[v][j] T_k[j][o] + sum_d gP[v][d] We[o][d])
        const float* gm = S + L.off_gm + (int64_t)v * L.ldm;
        const float* tk = t4 + H + o;
        int j = 0;
        for (; j + 8 <= n; j += 8) {
            const float4 ga = ld4(gm + j), gb4 = ld4(gm + j + 4);
            float k[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) k[u] = tk[(int64_t)(j + u) * 4 * H];
            acc = fmaf(ga.x, k[0], acc); acc = fmaf(ga.y, k[1], acc); acc = fmaf(ga.z, k[2], acc); acc = fmaf(ga.w, k[3], acc);
            acc = fmaf(gb4.x, k[4], acc); acc = fmaf(gb4.y, k[5], acc); acc = fmaf(gb4.z, k[6], acc); acc = fmaf(gb4.w, k[7], acc);
        }
        for (; j < n; ++j) acc = fmaf(gm[j], tk[(int64_t)j * 4 * H], acc);
        for (int d = 0; d < D; ++d) acc = fmaf(S[L.off_gp + v * D + d], we[o * D + d], acc);
    } else {                 // rs sum_r gM[r][v] T_q[r][o]
        const float* gm = S + L.off_gm + v;
        const float* tq = t4 + o;
        int r = 0;
        for (; r + 8 <= n; r += 8) {
            float g[8], q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                g[u] = gm[(int64_t)(r + u) * L.ldm];
                q[u] = tq[(int64_t)(r + u) * 4 * H];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fmaf(g[u], q[u], acc);
        }
        for (; r < n; ++r) acc = fmaf(gm[(int64_t)r * L.ldm], tq[(int64_t)r * 4 * H], acc);
    }
    return acc * rs;
}

template <int H>
__device__ __forceinline__ void table_project_bwd_scores_body(const float* __restrict__ S, const float* __restrict__ t4,
                                                              const float* __restrict__ we, const float* __restrict__ table,
                                                              const Proj4& p, float* __restrict__ gtable,
                                                              float* __restrict__ gw, float* __restrict__ gb, int V, int n,
                                                              int D, int vb, float* __restrict__ lds) {
    constexpr int PH = (H >= 256) ? 1 : 256 / H;
    const TgRow L = tg_row(n, H, D);
    const float rs = rsqrtf((float)H);
    float* red = lds;                  // [256]
    float* redb = lds + 256;           // [256]
    float* gs = lds + 512;             // the gradient slice: column (n) or row (4H)
    const int a = threadIdx.x % H, ph = threadIdx.x / H;
    if (vb < 4 * H) {
        const int c = vb;
        for (int v = threadIdx.x; v < n; v += 256) gs[v] = tg_gp_entry<H>(S, L, t4, we, n, D, v, c, rs);
        __syncthreads();
        float acc = 0.f, sb = 0.f;
        int v = ph;
        for (; v + 7 * PH < n; v += 8 * PH) {
            float tv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) tv[u] = table[(int64_t)(v + u * PH) * H + a];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const float gv = gs[v + u * PH]; acc = fmaf(gv, tv[u], acc); sb += gv; }
        }
        for (; v < n; v += PH) {
            const float gv = gs[v];
            acc = fmaf(gv, table[(int64_t)v * H + a], acc);
            sb += gv;
        }
        red[threadIdx.x] = acc;
        redb[threadIdx.x] = sb;
        __syncthreads();
        if (ph == 0) {
            float s = acc, t = sb;
            for (int k = 1; k < PH; ++k) { s += red[k * H + a]; t += redb[k * H + a]; }
            gw[(int64_t)c * H + a] = s;
            if (a == 0) gb[c] = t;
        }
    } else {
        const int v = vb - 4 * H;
        for (int c = threadIdx.x; c < 4 * H; c += 256) gs[c] = tg_gp_entry<H>(S, L, t4, we, n, D, v, c, rs);
        __syncthreads();
        float acc = 0.f;
        if (v < n) {                   // rows no node refers to: zero gradient
            for (int c0 = ph; c0 < 4 * H; c0 += 8 * PH) {
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
                    if (c < 4 * H) acc = fmaf(gs[c], wv[u], acc);
                }
            }
        }
        red[threadIdx.x] = acc;
        __syncthreads();
        if (ph == 0) {
            float s = acc;
            for (int k = 1; k < PH; ++k) s += red[k * H + a];
            if (v < V) gtable[(int64_t)v * H + a] = s;
        }
    }
}

}  // namespace qot
