// TransformerConv backward in table mode for LARGE tables (V ~ 1000: cfg4 / cfg5) -- the ROW form.
//
// With x = emb[node_ids] (topological_training/models.py:51-53) and node_ids == arange(n) in every graph
// (topological_training/dataset.py:78) node r of every graph shares table row r, and the logits' dense part is an entry of
// M = T_q T_k^T (qot_tconv_fwd_scores).  The autograd of TransformerConv.propagate (reached from loss.backward(),
// topological_training/train.py:115) then needs no per-node grad_q / grad_k at all:
//   grad M[r_i, r_j] += ds_e,   grad T_q = (grad M T_k + grad P W_e^T) / sqrt(H),   grad T_k = grad M^T T_q / sqrt(H)
// with ds_e = a_e (da_e - delta_i) and grad P[r] = sum over the destinations of row r of sum_e ds_e ea_e.
// The kernels of tconv.hip (destination pass: node r of 8 graphs per workgroup, key AND value row gathered per edge, two
// H-wide weighted key sums per destination; source pass: query AND gradient row per out-edge; per-workgroup partial
// table rows [B / 8, n, 4H] summed afterwards) become:
//   destination pass: a workgroup owns table row r for a slice of the graphs -- row r of M staged in LDS once, the value row
//     the only row gathered per edge, ds_e added into ONE LDS image of row r of grad M as 64-bit fixed point (2^-36: the
//     lane groups of a workgroup work on different graphs and meet on the same entries in no fixed order; integer sums do
//     not depend on it, so the result stays bit-reproducible, and the image costs 8 KB instead of a private fp32 row per
//     lane group -- 36 KB at H = 128, which left two workgroups per CU and made the pass slower than tconv.hip's) -- and
//     leaves ONE partial row [grad T_skip (H) | grad M (n) | grad P (D)] per (slice, r);
//   source pass: same ownership by source row: grad T_v[r] accumulated in registers over the slice.
// P slices of the graphs per row (load balance; hubs of power-law graphs sit at the same rows of every graph: rows are
// dealt slowest so that the heavy rows start first); the P partial rows are summed in order (QOT_ROLE_SUM_ROWS / the
// caller).  No atomics on global memory, fixed summation order.
#include "common.hpp"

namespace qot {

__host__ __device__ constexpr int trows_cpl() { return 4; }

template <int NV>
__device__ __forceinline__ float trows_dot(const float4 (&a)[NV], const float4 (&b)[NV]) {
    float s = dot4(a[0], b[0]);
#pragma unroll
    for (int v = 1; v < NV; ++v) s += dot4(a[v], b[v]);
    return s;
}
__device__ __forceinline__ float trows_comp(const float4& a, int c) { return c == 0 ? a.x : (c == 1 ? a.y : (c == 2 ? a.z : a.w)); }

// partial row layout of the destination pass: [grad T_skip H | grad M npad | grad P D | pad to 4]
__host__ __device__ inline int trows_npad(int n) { return (n + 31) / 32 * 32; }
__host__ __device__ inline int trows_ldrow(int n, int H, int D) { return (H + trows_npad(n) + D + 3) / 4 * 4; }

// grid = n * parts workgroups of 256 threads; workgroup vb: r = vb / parts, part = vb % parts (heavy rows first).
// LDS (dynamic): sGM[npad] (64-bit) | sM[npad]
template <int H, int D>
__global__ __launch_bounds__(256, (D <= 4 && H <= 128) ? 4 : 3) void tconv_bwd_dst_rows_kernel(
    const float* __restrict__ g, const float* __restrict__ q, const float* __restrict__ v_, int ld,
    const float* __restrict__ ea, const float* __restrict__ we, const float* __restrict__ stats,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colf, const int32_t* __restrict__ eid,
    const float* __restrict__ Mtab, int ldm, float* __restrict__ gskip, float* __restrict__ escr, float* __restrict__ delta,
    const float* __restrict__ y_act, ActParams act, int tile_n, int64_t tile_B, int parts, float* __restrict__ part_rows,
    float* __restrict__ wedge_partials) {
    constexpr int CPL = trows_cpl(), NV = CPL / 4;
    constexpr int TPR = H / CPL;
    constexpr int RPB = 256 / TPR;
    constexpr int DCH = D > 4 ? 4 : D;
    extern __shared__ float sh[];
    __shared__ float4 tred[NV][256];
    __shared__ float pred[RPB][D];
    __shared__ float wred[RPB * H * DCH];
    const int npad = trows_npad(tile_n);
    const int ldrow = trows_ldrow(tile_n, H, D);
    unsigned long long* sGM = reinterpret_cast<unsigned long long*>(sh);
    float* sM = sh + 2 * npad;
    constexpr float kFix = 68719476736.0f;            // 2^36
    const int sub = threadIdx.x % TPR, rloc = threadIdx.x / TPR;
    const int vb = blockIdx.x;
    const int r = vb / parts, part = vb % parts;
    const int64_t per = (tile_B + parts - 1) / parts;
    const int64_t b0 = (int64_t)part * per;
    const int64_t b1 = (b0 + per < tile_B) ? b0 + per : tile_B;
    const float rs = rsqrtf((float)H);
    const int c0 = CPL * sub;
    for (int t = threadIdx.x; t < npad; t += 256) sM[t] = t < tile_n ? rs * Mtab[(int64_t)r * ldm + t] : 0.f;
    for (int t = threadIdx.x; t < npad; t += 256) sGM[t] = 0ull;
    float4 qi[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) qi[v] = scale4(rs, ld4(q + (int64_t)r * ld + c0 + 4 * v));
    float wl[CPL][D];
#pragma unroll
    for (int c = 0; c < CPL; ++c)
#pragma unroll
        for (int d = 0; d < D; ++d) wl[c][d] = we[(c0 + c) * D + d];
    float qe[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float tq = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) tq = fmaf(trows_comp(qi[c >> 2], c & 3), wl[c][d], tq);
        qe[d] = group_sum<TPR>(tq);
    }
    __syncthreads();
    float4 rg[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) rg[v] = f4zero();
    float pdacc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) pdacc[d] = 0.f;
    float wc[CPL][D];
#pragma unroll
    for (int c = 0; c < CPL; ++c)
#pragma unroll
        for (int d = 0; d < D; ++d) wc[c][d] = 0.f;

    constexpr int BT = (TPR < 16) ? TPR : 16;                 // edges prefetched per batch
    // The per-destination chain is four dependent global round trips (row / statistics / edge range -> edge list -> edge
    // features -> value rows); the first level of the NEXT graph of this lane group is requested before the current one
    // is worked on (raw values: no arithmetic on them before the wait).
    float4 ngi[NV], nyy[NV];
    float nm = 0.f, nden = 1.f;
    int nbeg = 0, nend = 0;
    auto fetch = [&](int64_t b) {
        const int64_t i = b * tile_n + r;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            ngi[v] = ld4(g + i * H + c0 + 4 * v);
            if (y_act) nyy[v] = ld4(y_act + i * H + c0 + 4 * v);
        }
        nm = stats[2 * i];
        nden = stats[2 * i + 1];
        nbeg = rowptr[i];
        nend = rowptr[i + 1];
    };
    if (b0 + rloc < b1) fetch(b0 + rloc);
    for (int64_t b = b0 + rloc; b < b1; b += RPB) {
        const int64_t i = b * tile_n + r;
        float4 gi[NV], yyv[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) { gi[v] = ngi[v]; yyv[v] = nyy[v]; }
        const float m = nm;
        const float inv = 1.0f / nden;
        const int beg = nbeg, end = nend;
        if (b + RPB < b1) fetch(b + RPB);
        if (y_act) {      // grad_out arrives for y = dropout(leaky_relu(conv)): go back through it here (tconv_bwd_dst_kernel)
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int64_t flat = i * H + c0 + 4 * v;
                const float4 yy = yyv[v];
                uint64_t z = 0;
                if (act.thr16) z = act_hash64(act.seed, (uint64_t)act.step[0], (uint64_t)flat >> 2);
                float vi[4] = {gi[v].x, gi[v].y, gi[v].z, gi[v].w};
                const float vr[4] = {yy.x, yy.y, yy.z, yy.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool keep = act.thr16 ? (((uint32_t)(z >> (16 * c)) & 0xFFFFu) >= act.thr16) : true;
                    vi[c] = vi[c] * (keep ? act.keep_scale : 0.f) * (vr[c] > 0.f ? 1.0f : act.slope);
                }
                gi[v] = make_float4(vi[0], vi[1], vi[2], vi[3]);
            }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            st4(gskip + i * H + c0 + 4 * v, gi[v]);            // gradient wrt the conv output: the source pass reads it
            rg[v] = add4(rg[v], gi[v]);
        }
        float ge[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            float tg = 0.f;
#pragma unroll
            for (int c = 0; c < CPL; ++c) tg = fmaf(trows_comp(gi[c >> 2], c & 3), wl[c][d], tg);
            ge[d] = group_sum<TPR>(tg);
        }
        // pass 1: a_e and da_e of every in-edge (the lane that prefetched an edge owns its scalars), delta_i
        float sada = 0.f, p1[D], p2[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { p1[d] = 0.f; p2[d] = 0.f; }
        for (int base = beg; base < end; base += BT) {
            const int pme = base + sub;
            int myj = 0;
            float mye[D], mya = 0.f, myda = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) mye[d] = 0.f;
            const bool mine = sub < BT && pme < end;
            if (mine) {
                myj = colf[pme];
                const int64_t e = eid[pme];
#pragma unroll
                for (int d = 0; d < D; ++d) mye[d] = ea[e * D + d];
                float s = sM[myj];
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    s = fmaf(qe[d], mye[d], s);
                    myda = fmaf(ge[d], mye[d], myda);
                }
                mya = __expf(s - m) * inv;
            }
            const int cnt = (end - base < BT) ? end - base : BT;
            constexpr int UF = 4;            // value rows in flight per group
            for (int u0 = 0; u0 < cnt; u0 += UF) {
                float4 vr[UF][NV];
#pragma unroll
                for (int u = 0; u < UF; ++u) {
                    const int64_t j = __shfl(myj, u0 + u, TPR);
                    const int64_t jr = (u0 + u < cnt) ? j : 0;
#pragma unroll
                    for (int v = 0; v < NV; ++v) vr[u][v] = ld4(v_ + jr * ld + c0 + 4 * v);
                }
#pragma unroll
                for (int u = 0; u < UF; ++u) {
                    const float dv = group_sum<TPR>(trows_dot<NV>(gi, vr[u]));
                    if (sub == u0 + u) myda += dv;               // (u0 + u >= cnt: no lane owns it)
                }
            }
            if (mine) {
                const float ada = mya * myda;
                sada += ada;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    p1[d] = fmaf(ada, mye[d], p1[d]);
                    p2[d] = fmaf(mya, mye[d], p2[d]);
                }
                escr[2 * (int64_t)pme] = mya;
                escr[2 * (int64_t)pme + 1] = myda;
            }
        }
        sada = group_sum<TPR>(sada);
        float pd[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            p1[d] = group_sum<TPR>(p1[d]);
            p2[d] = group_sum<TPR>(p2[d]);
            pd[d] = p1[d] - sada * p2[d];
            pdacc[d] += pd[d];
        }
        if (sub == 0) delta[i] = sada;
        // pass 2: ds_e into this lane group's image of row r of grad M (its own escr entries: same lane wrote them)
        for (int base = beg; base < end; base += BT) {
            const int pme = base + sub;
            if (sub < BT && pme < end) {
                const int j = colf[pme];
                const float a = escr[2 * (int64_t)pme], da = escr[2 * (int64_t)pme + 1];
                const long long fx = __float2ll_rn(a * (da - sada) * kFix);
                atomicAdd(&sGM[j], (unsigned long long)fx);   // LDS, integer: order-independent
            }
        }
        // grad of lin_edge.weight, the part that is not a function of the table row: g_i[c] * p2_i[d]
#pragma unroll
        for (int c = 0; c < CPL; ++c)
#pragma unroll
            for (int d = 0; d < D; ++d) wc[c][d] = fmaf(trows_comp(gi[c >> 2], c & 3), p2[d], wc[c][d]);
    }
    // ---- the RPB lane groups meet in LDS, fixed order ----
#pragma unroll
    for (int v = 0; v < NV; ++v) tred[v][threadIdx.x] = rg[v];
    if (sub == 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) pred[rloc][d] = pdacc[d];
    }
    __syncthreads();
    float* out = part_rows + ((int64_t)part * tile_n + r) * ldrow;
    if (rloc == 0) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float4 a = tred[v][sub];
            for (int r2 = 1; r2 < RPB; ++r2) a = add4(a, tred[v][r2 * TPR + sub]);
            st4(out + c0 + 4 * v, a);
        }
    }
    for (int t = threadIdx.x; t < npad; t += 256) out[H + t] = (float)(long long)sGM[t] * (1.0f / kFix);
    if (threadIdx.x < D) {
        float s = pred[0][threadIdx.x];
        for (int r2 = 1; r2 < RPB; ++r2) s += pred[r2][threadIdx.x];
        out[H + npad + threadIdx.x] = s;
    } else if ((int)threadIdx.x < ldrow - H - npad) {
        out[H + npad + threadIdx.x] = 0.f;                 // the row's padding: the caller's product reads across it
    }
#pragma unroll
    for (int d0 = 0; d0 < D; d0 += DCH) {
        const int dc = (D - d0 < DCH) ? D - d0 : DCH;
        if (d0) __syncthreads();
#pragma unroll
        for (int c = 0; c < CPL; ++c)
#pragma unroll
            for (int dd = 0; dd < DCH; ++dd)
                if (d0 + dd < D) wred[rloc * H * DCH + dd * H + c0 + c] = wc[c][d0 + dd];
        __syncthreads();
        for (int o = threadIdx.x; o < H * dc; o += 256) {
            const int dd = o / H, ch = o % H;
            float sacc = 0.f;
            for (int r2 = 0; r2 < RPB; ++r2) sacc += wred[r2 * H * DCH + dd * H + ch];
            wedge_partials[(int64_t)blockIdx.x * H * D + ch * D + d0 + dd] = sacc;
        }
    }
}

// Source pass: grad T_v[r] = sum over the graphs of the slice, over the out-edges of node r, of a_e g_i  ->
// part_rows[part][r][H].  Same grid as the destination pass.
template <int H>
__global__ __launch_bounds__(256) void tconv_bwd_src_rows_kernel(
    const float* __restrict__ g, const float* __restrict__ escr, const int32_t* __restrict__ rowptr_t,
    const int32_t* __restrict__ col_t, const int32_t* __restrict__ pos_t, int tile_n, int64_t tile_B, int parts,
    float* __restrict__ part_rows) {
    constexpr int CPL = trows_cpl(), NV = CPL / 4;
    constexpr int TPR = H / CPL;
    constexpr int RPB = 256 / TPR;
    __shared__ float4 tred[NV][256];
    const int sub = threadIdx.x % TPR, rloc = threadIdx.x / TPR;
    const int vb = blockIdx.x;
    const int r = vb / parts, part = vb % parts;
    const int64_t per = (tile_B + parts - 1) / parts;
    const int64_t b0 = (int64_t)part * per;
    const int64_t b1 = (b0 + per < tile_B) ? b0 + per : tile_B;
    const int c0 = CPL * sub;
    float4 av[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) av[v] = f4zero();
    constexpr int BT = (TPR < 16) ? TPR : 16;
    // the next graph's out-edge range is requested while this one is worked on
    int nbeg = 0, nend = 0;
    if (b0 + rloc < b1) {
        const int64_t j0 = (b0 + rloc) * tile_n + r;
        nbeg = rowptr_t[j0];
        nend = rowptr_t[j0 + 1];
    }
    for (int64_t b = b0 + rloc; b < b1; b += RPB) {
        const int beg = nbeg, end = nend;
        if (b + RPB < b1) {
            const int64_t jn = (b + RPB) * tile_n + r;
            nbeg = rowptr_t[jn];
            nend = rowptr_t[jn + 1];
        }
        for (int base = beg; base < end; base += BT) {
            const int tme = base + sub;
            int myi = 0;
            float mya = 0.f;
            if (sub < BT && tme < end) {
                myi = col_t[tme];
                mya = escr[2 * (int64_t)pos_t[tme]];
            }
            const int cnt = (end - base < BT) ? end - base : BT;
            constexpr int UF = 4;
            for (int u0 = 0; u0 < cnt; u0 += UF) {
                float4 gr[UF][NV];
#pragma unroll
                for (int u = 0; u < UF; ++u) {
                    const int64_t i = __shfl(myi, u0 + u, TPR);
                    const bool live = u0 + u < cnt;
#pragma unroll
                    for (int v = 0; v < NV; ++v) gr[u][v] = ld4(g + (live ? i : 0) * H + c0 + 4 * v);
                }
#pragma unroll
                for (int u = 0; u < UF; ++u) {
                    const float a = (u0 + u < cnt) ? __shfl(mya, u0 + u, TPR) : 0.f;
#pragma unroll
                    for (int v = 0; v < NV; ++v) av[v] = fma4(a, gr[u][v], av[v]);
                }
            }
        }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) tred[v][threadIdx.x] = av[v];
    __syncthreads();
    if (rloc == 0) {
        float* out = part_rows + ((int64_t)part * tile_n + r) * H;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float4 a = tred[v][sub];
            for (int r2 = 1; r2 < RPB; ++r2) a = add4(a, tred[v][r2 * TPR + sub]);
            st4(out + c0 + 4 * v, a);
        }
    }
}

// Forward in the row form: the workgroup that owns table row r for a slice of the graphs stages row r of M once and holds
// q_r W_e and the skip row in registers; per destination only the edge list, the edge features and the value rows are read
// (tconv_fwd_kernel<., ., MT> reads the query and skip rows and looks M up in L2 for every destination), and the next
// graph's edge range is requested one graph ahead.  Same arithmetic, same order: bit-equal outputs.
template <int H, int D>
__global__ __launch_bounds__(256) void tconv_fwd_rows_kernel(
    const float* __restrict__ q, const float* __restrict__ v_, const float* __restrict__ skip, int ld,
    const float* __restrict__ ea, const float* __restrict__ we, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ colf, const int32_t* __restrict__ eid, const float* __restrict__ Mtab, int ldm,
    float* __restrict__ out, float* __restrict__ stats, int tile_n, int64_t tile_B, int parts, ActParams act) {
    constexpr int CPL = trows_cpl(), NV = CPL / 4;
    constexpr int TPR = H / CPL;
    constexpr int RPB = 256 / TPR;
    extern __shared__ float sh[];
    float* sM = sh;
    const int npad = trows_npad(tile_n);
    const int sub = threadIdx.x % TPR, rloc = threadIdx.x / TPR;
    const int vb = blockIdx.x;
    const int r = vb / parts, part = vb % parts;
    const int64_t per = (tile_B + parts - 1) / parts;
    const int64_t b0 = (int64_t)part * per;
    const int64_t b1 = (b0 + per < tile_B) ? b0 + per : tile_B;
    const float rs = rsqrtf((float)H);
    const int c0 = CPL * sub;
    for (int t = threadIdx.x; t < npad; t += 256) sM[t] = t < tile_n ? rs * Mtab[(int64_t)r * ldm + t] : 0.f;
    float4 qi[NV], sk[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        qi[v] = scale4(rs, ld4(q + (int64_t)r * ld + c0 + 4 * v));
        sk[v] = ld4(skip + (int64_t)r * ld + c0 + 4 * v);
    }
    float wl[CPL][D];
#pragma unroll
    for (int c = 0; c < CPL; ++c)
#pragma unroll
        for (int d = 0; d < D; ++d) wl[c][d] = we[(c0 + c) * D + d];
    float qe[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float t = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) t = fmaf(trows_comp(qi[c >> 2], c & 3), wl[c][d], t);
        qe[d] = group_sum<TPR>(t);
    }
    __syncthreads();
    constexpr int BT = (TPR < 16) ? TPR : 16;
    int nbeg = 0, nend = 0;
    if (b0 + rloc < b1) {
        const int64_t i0 = (b0 + rloc) * tile_n + r;
        nbeg = rowptr[i0];
        nend = rowptr[i0 + 1];
    }
    for (int64_t b = b0 + rloc; b < b1; b += RPB) {
        const int64_t i = b * tile_n + r;
        const int beg = nbeg, end = nend;
        if (b + RPB < b1) {
            const int64_t in_ = (b + RPB) * tile_n + r;
            nbeg = rowptr[in_];
            nend = rowptr[in_ + 1];
        }
        float m = -INFINITY, l = 0.f;
        float4 acc[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = f4zero();
        float aacc[D];
#pragma unroll
        for (int d = 0; d < D; ++d) aacc[d] = 0.f;
        for (int base = beg; base < end; base += BT) {
            const int pme = base + sub;
            int myj = 0;
            float mye[D], mym = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) mye[d] = 0.f;
            if (sub < BT && pme < end) {
                myj = colf[pme];
                const int64_t e = eid[pme];
#pragma unroll
                for (int d = 0; d < D; ++d) mye[d] = ea[e * D + d];
                mym = sM[myj];
            }
            const int cnt = (end - base < BT) ? end - base : BT;
            constexpr int UF = 4;
            for (int u0 = 0; u0 < cnt; u0 += UF) {
                float4 vr[UF][NV];
#pragma unroll
                for (int u = 0; u < UF; ++u) {
                    const int64_t j = __shfl(myj, u0 + u, TPR);
                    const int64_t jr = (u0 + u < cnt) ? j : 0;
#pragma unroll
                    for (int v = 0; v < NV; ++v) vr[u][v] = ld4(v_ + jr * ld + c0 + 4 * v);
                }
#pragma unroll
                for (int u = 0; u < UF; ++u) {
                    if (u0 + u < cnt) {                           // group-uniform
                        float ee[D];
#pragma unroll
                        for (int d = 0; d < D; ++d) ee[d] = __shfl(mye[d], u0 + u, TPR);
                        float s = __shfl(mym, u0 + u, TPR);
#pragma unroll
                        for (int d = 0; d < D; ++d) s = fmaf(qe[d], ee[d], s);
                        const float mn = fmaxf(m, s);
                        const float sc = __expf(m - mn);
                        const float pe = __expf(s - mn);
                        l = fmaf(l, sc, pe);
#pragma unroll
                        for (int v = 0; v < NV; ++v) acc[v] = fma4(pe, vr[u][v], scale4(sc, acc[v]));
#pragma unroll
                        for (int d = 0; d < D; ++d) aacc[d] = fmaf(pe, ee[d], aacc[d] * sc);
                        m = mn;
                    }
                }
            }
        }
        const float denom = l + 1e-16f;
        const float inv = 1.0f / denom;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4 o = scale4(inv, acc[v]);
            float oc[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int d = 0; d < D; ++d) oc[c] = fmaf(wl[4 * v + c][d], aacc[d] * inv, oc[c]);
            st4(out + i * H + c0 + 4 * v, act_apply4(make_float4(oc[0] + sk[v].x, oc[1] + sk[v].y, oc[2] + sk[v].z, oc[3] + sk[v].w),
                                                     act, (uint64_t)(i * H + c0 + 4 * v) >> 2));
        }
        if (sub == 0) {
            stats[2 * i] = (beg < end) ? m : 0.f;
            stats[2 * i + 1] = denom;
        }
    }
}

}  // namespace qot

using namespace qot;

// Row form of TransformerConv's backward (see the top of this file).  Layout helpers: the destination pass leaves
// parts * n rows of qot_tconv_rows_ld(n, H, D) floats, [grad T_skip (H) | grad M (qot_tconv_rows_npad(n)) | grad P (D)],
// row index part * n + r, and parts * n rows of H * D floats of lin_edge partials (workgroup order r * parts + part).
extern "C" int qot_tconv_rows_npad(int n) { return n > 0 ? trows_npad(n) : 0; }
extern "C" int qot_tconv_rows_ld(int n, int H, int D) { return n > 0 ? trows_ldrow(n, H, D) : 0; }

static size_t trows_lds_bytes(int n, int H) {
    (void)H;
    return (size_t)3 * trows_npad(n) * sizeof(float);        // 64-bit grad M image + the row of M
}

extern "C" int qot_tconv_rows_supported(int n, int H, int D) {
    if (n <= 0 || n > 8192 || D < 1 || D > 8) return 0;
    if (H != 64 && H != 128 && H != 256) return 0;
    return trows_lds_bytes(n, H) <= 96 * 1024;
}

// grad_out [N, H] (N = n * B, node_ids == arange(n) in every graph); q / v: columns of the projected table (row stride ld);
// scores [n, ld_scores] = T_q T_k^T (unscaled); colf = table row of every in-edge's source; grad_skip [N, H] out (gradient
// wrt the conv output behind the fused activation), escr [slots, 2], delta [N] out (for the source pass); part_rows
// [parts * n, qot_tconv_rows_ld], wedge_partials [parts * n, H * D] out.
extern "C" int qot_tconv_bwd_dst_rows(const float* grad_out, const float* q, const float* v, int ld, const float* edge_attr,
                                      const float* w_edge, const float* stats, const int32_t* rowptr, const int32_t* colf,
                                      const int32_t* eid, const float* scores, int ld_scores, float* grad_skip, float* escr,
                                      float* delta, const float* y_act, float act_slope, float act_p, uint64_t act_seed,
                                      const int64_t* act_step, int n, int64_t B, int parts, float* part_rows,
                                      float* wedge_partials, int H, int D, qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n <= 0 || B <= 0 || parts <= 0 || parts > B) return QOT_ERR_BADARG;
    if (!grad_out || !q || !v || !edge_attr || !w_edge || !stats || !rowptr || !colf || !eid || !scores || !grad_skip || !escr ||
        !delta || !part_rows || !wedge_partials || (ld & 3) || ld_scores < n)
        return QOT_ERR_BADARG;
    if (!qot_tconv_rows_supported(n, H, D) || (int64_t)n * parts > 0x7fffffff) return QOT_ERR_UNSUPPORTED;
    const ActParams ap = make_act(y_act ? 1 : 0, act_slope, act_p, act_seed, act_step);
    const size_t lds = trows_lds_bytes(n, H);
    static size_t allowed[3][8][kMaxDevices];
    int rc = QOT_ERR_UNSUPPORTED;
#define QOT_TROWS(HH, HI)                                                                                           \
    if (H == HH) {                                                                                                   \
        QOT_DISPATCH_D(D, {                                                                                          \
            rc = ensure_dyn_lds(reinterpret_cast<const void*>(tconv_bwd_dst_rows_kernel<HH, kD>), lds, allowed[HI][kD - 1]); \
            if (rc == QOT_OK)                                                                                        \
                tconv_bwd_dst_rows_kernel<HH, kD><<<n * parts, 256, lds, stream>>>(                                  \
                    grad_out, q, v, ld, edge_attr, w_edge, stats, rowptr, colf, eid, scores, ld_scores, grad_skip, escr, delta, \
                    y_act, ap, n, B, parts, part_rows, wedge_partials);                                              \
        });                                                                                                          \
    }
    QOT_TROWS(64, 0) QOT_TROWS(128, 1) QOT_TROWS(256, 2)
#undef QOT_TROWS
    if (rc != QOT_OK) return rc;
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

// grad T_v partials [parts * n, H] from grad_skip [N, H] and the destination pass's escr.
extern "C" int qot_tconv_bwd_src_rows(const float* grad_skip, const float* escr, const int32_t* rowptr_t, const int32_t* col_t,
                                      const int32_t* pos_t, int n, int64_t B, int parts, float* part_rows, int H,
                                      qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n <= 0 || B <= 0 || parts <= 0 || parts > B) return QOT_ERR_BADARG;
    if (!grad_skip || !escr || !rowptr_t || !col_t || !pos_t || !part_rows) return QOT_ERR_BADARG;
    if ((H != 64 && H != 128 && H != 256) || (int64_t)n * parts > 0x7fffffff) return QOT_ERR_UNSUPPORTED;
    if (H == 64) tconv_bwd_src_rows_kernel<64><<<n * parts, 256, 0, stream>>>(grad_skip, escr, rowptr_t, col_t, pos_t, n, B, parts, part_rows);
    else if (H == 128) tconv_bwd_src_rows_kernel<128><<<n * parts, 256, 0, stream>>>(grad_skip, escr, rowptr_t, col_t, pos_t, n, B, parts, part_rows);
    else tconv_bwd_src_rows_kernel<256><<<n * parts, 256, 0, stream>>>(grad_skip, escr, rowptr_t, col_t, pos_t, n, B, parts, part_rows);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

// Forward of the row form: out / stats as qot_tconv_fwd_scores (bit-equal), node_ids == arange(n) in each of the B graphs.
extern "C" int qot_tconv_fwd_rows(const float* q, const float* v, const float* skip, int ld, const float* scores, int ld_scores,
                                  const float* edge_attr, const float* w_edge, const int32_t* rowptr, const int32_t* colf,
                                  const int32_t* eid, float* out, float* stats, int n, int64_t B, int parts, int H, int D, int act,
                                  float act_slope, float act_p, uint64_t act_seed, const int64_t* act_step, qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n <= 0 || B <= 0 || parts <= 0 || parts > B) return QOT_ERR_BADARG;
    if (!q || !v || !skip || !scores || !edge_attr || !w_edge || !rowptr || !colf || !eid || !out || !stats || (ld & 3) ||
        ld_scores < n)
        return QOT_ERR_BADARG;
    if (!qot_tconv_rows_supported(n, H, D) || (int64_t)n * parts > 0x7fffffff) return QOT_ERR_UNSUPPORTED;
    const ActParams ap = make_act(act, act_slope, act_p, act_seed, act_step);
    const size_t lds = (size_t)trows_npad(n) * sizeof(float);
    if (lds > 64 * 1024) return QOT_ERR_UNSUPPORTED;
#define QOT_TROWS_F(HH)                                                                                              \
    if (H == HH) {                                                                                                   \
        QOT_DISPATCH_D(D, {                                                                                          \
            tconv_fwd_rows_kernel<HH, kD><<<n * parts, 256, lds, stream>>>(q, v, skip, ld, edge_attr, w_edge, rowptr, colf, eid, \
                                                                           scores, ld_scores, out, stats, n, B, parts, ap);      \
        });                                                                                                          \
    }
    QOT_TROWS_F(64) QOT_TROWS_F(128) QOT_TROWS_F(256)
#undef QOT_TROWS_F
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}
