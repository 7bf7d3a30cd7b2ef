// TransformerConv, table mode, GRAPH form (see tconv_graph_dev.hpp for the algebra): whole graphs per workgroup, the
// score matrix M = T_q T_k^T / sqrt(H) and the value table T_v resident in LDS.  Replaces [PyG-ext]
// TransformerConv.propagate + its autograd as reached from topological_training/models.py:53 when x =
// node_embeddings(arange(n)) in every graph (models.py:51-52, dataset.py:78) and n <= 128.
//
// Forward (qot_tconv_fwd_graph): NS graph slots of 256 threads per workgroup (NS = 4 when LDS allows).  Per graph:
//   A  the graph's CSR slice (contiguous), its edge features (through eid) -> LDS                 [all loads independent]
//   B  two lanes per destination: logits M[r][j] + <P[r], ea>, max, exp, sum, alpha, sum alpha ea  [LDS only]
//   C  H/4 lanes per destination: sum alpha T_v[j] (+ We (sum alpha ea) + T_skip[r]), activation, 16-B stores
// and the attention weights alpha[E] (CSR order) are kept for the backward: it needs no logits, no M, no softmax.
//
// Backward (qot_tconv_bwd_graph): one graph at a time per 1024-thread workgroup (every row of the LDS-resident
// accumulators has ONE owner, so all sums have a fixed order: bitwise reproducible).  Per graph:
//   0  index slices, alpha, edge features, g = grad_out through the fused leaky_relu + dropout (mask regenerated) -> LDS
//   1  destination pass (H/4 lanes per destination): da_e = <g_i, T_v[j]> + <We^T g_i, ea_e>, delta_i = sum alpha da,
//      ds_e = alpha_e (da_e - delta_i) -> gM[r][j] += ds_e (LDS), gP[r] += sum ds_e ea_e, gWe += g_i (x) sum alpha ea,
//      gTs[r] += g_i (registers)
//   2  source pass over the CSC: gTv[j] += sum_{e: j -> i} alpha_e g_i (registers; g rows from LDS)
// and at the end ONE partial row per workgroup (layout tg_row) for the step's fixed-order row sum.
#include "tconv_graph_dev.hpp"
#include "mfma_tile.hpp"      // num_cus()

namespace qot {

// in-kernel cycle stamps: diagnostic build only (make DIAG=1; tools/bench_tconv_graph.py); nothing in the release build
#ifdef QOT_DIAG
__device__ int g_tg_variant;           // ablation bits (tools/bench_tconv_graph.py): 1 no edge dots / ge, 2 no source pass, 4 no 1c,
#define TG_VAR(bit) (tg_var & (bit))   //   8 no activation backward in the commit
__device__ unsigned long long g_tg_stamps[16];
#else
#define TG_VAR(bit) 0
#endif
#if defined(QOT_DIAG) && defined(QOT_TG_STAMPS)      // (the stamps cost a third of the kernels' time: a build of their own)
// (sums kept in registers of thread 0, one atomic per slot at the end)
#define TG_STAMP_DECL                                                                        \
    unsigned long long tg_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                     \
    unsigned long long tg_t0 = __builtin_amdgcn_s_memtime();
#define TG_STAMP(slot)                                                                      \
    {                                                                                       \
        const unsigned long long tg_t1 = __builtin_amdgcn_s_memtime();                      \
        tg_acc[slot] += tg_t1 - tg_t0;                                                      \
        tg_t0 = tg_t1;                                                                      \
    }
#define TG_STAMP_FLUSH                                                                      \
    if (threadIdx.x == 0) {                                                                 \
        _Pragma("unroll") for (int q = 0; q < 12; ++q)                                      \
            if (tg_acc[q]) atomicAdd(&g_tg_stamps[q], tg_acc[q]);                           \
    }
#else
#define TG_STAMP_DECL
#define TG_STAMP(slot)
#define TG_STAMP_FLUSH
#endif

// ------------------------------------------------------------------------------------------------ forward
// Every phase that touches global memory is ONE batch of independent loads: under the burst of 256 workgroups starting
// together a dependent global round trip costs 0.7-1 us, so the count of SERIALISED trips -- not bytes -- sets the time
// (first version: ~10 trips, 19.5 us; measured with the diagnostic build's stamps, tools/bench_tconv_graph.py).
struct TgFwdLds {
    int tv, ts, slot0, slot_floats, rp, col, al, ea, aa;   // float offsets (rp .. aa: inside a slot)
    size_t bytes(int ns) const { return (size_t)(slot0 + ns * slot_floats) * 4; }
};
__host__ __device__ inline TgFwdLds tg_fwd_lds(int n, int H, int D, int max_e) {
    TgFwdLds L;
    const int me = pad4(max_e > 0 ? max_e : 1);
    L.tv = 0;                                    // (the score matrix M and P stay in L2: one lookup per edge, in the same
    L.ts = L.tv + n * H;                         //  round trip as the edge's features; 40 KB less to stage per workgroup)
    L.slot0 = L.ts + n * H;
    L.rp = 0;
    L.col = L.rp + pad4(n + 1);
    L.al = L.col + me;
    L.ea = L.al + me;
    L.aa = L.ea + pad4(me * D);
    L.slot_floats = L.aa + pad4(n * D);
    return L;
}

template <int H, int D>
__global__ __launch_bounds__(1024) void tconv_fwd_graph_kernel(
    const float* __restrict__ tv, const float* __restrict__ tskip, int ld, const float* __restrict__ M,
    const float* __restrict__ Pm, const float* __restrict__ we, const float* __restrict__ ea,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colf, const int32_t* __restrict__ eid,
    const int32_t* __restrict__ row, float* __restrict__ out, float* __restrict__ alpha, float* __restrict__ ea_csr,
    float* __restrict__ aa_out, int n, int64_t B, int max_e, ActParams act) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CPL = 4;                     // channels per lane in stage C (8 from H = 32 on: 128 VGPRs, 20.6 -> 23.4 us at cfg2)
    constexpr int NV = CPL / 4;
    constexpr int TPR = H / CPL;               // lanes per destination
    constexpr int G = 256 / TPR;               // destinations per slot and round
    constexpr int EB = 4;                      // edges a thread stages per trip
    const TgFwdLds L = tg_fwd_lds(n, H, D, max_e);
    const int ldm = pad4(n);
    const int NS = (int)blockDim.x >> 8;
    const int slot = (int)threadIdx.x >> 8, t = (int)threadIdx.x & 255;
    float* sTv = lds + L.tv;
    float* sTs = lds + L.ts;
    float* sb = lds + L.slot0 + slot * L.slot_floats;
    int* sRp = reinterpret_cast<int*>(sb + L.rp);      // RAW index slots (global), normalised by e0 where read
    int* sCol = reinterpret_cast<int*>(sb + L.col);
    float* sAl = sb + L.al;
    float* sEa = sb + L.ea;
    float* sAa = sb + L.aa;
#ifdef QOT_DIAG
    const int tg_var = g_tg_variant;       // forward ablation bits: 16 no stage C, 32 no stage B, 64 no edge staging (A),
#endif                                     // 128 no output stores, 256 no activation, 512 no table staging

    TG_STAMP_DECL
    const int sub = t % TPR, grp = t / TPR, c0 = CPL * sub;
    // ---- trip 1: the first graph's row pointers, the tables (one batch of loads), the per-thread constants
    const int64_t bfirst = (int64_t)blockIdx.x * NS + slot;
    int rp_raw = 0;
    if (bfirst < B && t <= n) rp_raw = rowptr[bfirst * n + t];
    {
        const int nT4 = n * H / 4, total = 2 * nT4;
        constexpr int TB = 4;
        for (int base = threadIdx.x; base < (TG_VAR(512) ? 0 : total); base += TB * (int)blockDim.x) {
            float4 v[TB];
#pragma unroll
            for (int u = 0; u < TB; ++u) {
                const int i = base + u * (int)blockDim.x;
                if (i < total) {
                    const int kk = i < nT4 ? i : i - nT4;
                    const int r = (4 * kk) / H, c = (4 * kk) % H;
                    v[u] = ld4((i < nT4 ? tv : tskip) + (int64_t)r * ld + c);
                }
            }
#pragma unroll
            for (int u = 0; u < TB; ++u) {
                const int i = base + u * (int)blockDim.x;
                if (i < total) st4(sTv + 4 * i, v[u]);                 // sTs follows sTv
            }
        }
    }
    const uint64_t stepv = act.thr16 ? (uint64_t)act.step[0] : 0;        // read once (see act_apply4s)
    float wl[CPL][D];
#pragma unroll
    for (int c = 0; c < CPL; ++c)
#pragma unroll
        for (int d = 0; d < D; ++d) wl[c][d] = we[(c0 + c) * D + d];
    if (bfirst < B && t <= n) sRp[t] = rp_raw;

    for (int64_t b0 = (int64_t)blockIdx.x * NS; b0 < B; b0 += (int64_t)gridDim.x * NS) {      // workgroup-uniform trips
        const int64_t b = b0 + slot;
        const bool live = b < B;
        const int64_t node0 = b * n;
        if (b0 != (int64_t)blockIdx.x * NS) {           // later graphs: their row pointers are a trip of their own
            __syncthreads();                            // the previous graph's stage C is done with the slot
            if (live && t <= n) sRp[t] = rowptr[node0 + t];
        }
        __syncthreads();
        TG_STAMP(0)
        int e0 = 0, eb = 0;
        if (live) {
            e0 = sRp[0];
            eb = sRp[n] - e0;
            eb = eb < 0 ? 0 : (eb > max_e ? max_e : eb);       // (the index build has checked the slices; memory safety)
        }
        // ---- A: trip 2 = index slice, trip 3 = edge features; the logits are formed here (one thread per edge)
        for (int pc = 0; pc < (TG_VAR(64) ? 0 : eb); pc += 256 * EB) {
            int cj[EB], rw[EB];
            int64_t ei[EB];
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                const int p = pc + t + 256 * u;
                const int pp = p < eb ? p : 0;
                cj[u] = colf[e0 + pp];
                rw[u] = row[e0 + pp];
                ei[u] = eid[e0 + pp];
            }
            float ev[EB][D], pr[EB][D], mv[EB];
            int jj[EB], rr[EB];
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                jj[u] = cj[u] < 0 ? 0 : (cj[u] >= n ? n - 1 : cj[u]);
                const int r = rw[u] - (int)node0;
                rr[u] = r < 0 ? 0 : (r >= n ? n - 1 : r);
                mv[u] = M[rr[u] * ldm + jj[u]];                          // L2-resident: the same round trip as the features
                if constexpr (D == 4) {
                    const float4 q = ld4(ea + ei[u] * 4);
                    ev[u][0] = q.x; ev[u][1] = q.y; ev[u][2] = q.z; ev[u][3] = q.w;
                    const float4 pq = ld4(Pm + rr[u] * 4);
                    pr[u][0] = pq.x; pr[u][1] = pq.y; pr[u][2] = pq.z; pr[u][3] = pq.w;
                } else {
#pragma unroll
                    for (int d = 0; d < D; ++d) { ev[u][d] = ea[ei[u] * D + d]; pr[u][d] = Pm[rr[u] * D + d]; }
                }
            }
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                const int p = pc + t + 256 * u;
                if (p < eb) {
                    float s = mv[u];
#pragma unroll
                    for (int d = 0; d < D; ++d) s = fmaf(pr[u][d], ev[u][d], s);
                    sCol[p] = jj[u];
                    sAl[p] = s;
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        sEa[p * D + d] = ev[u][d];
                        ea_csr[(int64_t)(e0 + p) * D + d] = ev[u][d];
                    }
                }
            }
        }
        __syncthreads();
        TG_STAMP(1)
        // ---- B: edge softmax, two lanes per destination (a quad per destination -- two rounds of 64 -- measured slower)
        if (live && !TG_VAR(32)) {
            const int l = t & 1;
            for (int r = t >> 1; r < n; r += 128) {
                int beg = sRp[r] - e0, end = sRp[r + 1] - e0;
                beg = beg < 0 ? 0 : (beg > eb ? eb : beg);
                end = end < beg ? beg : (end > eb ? eb : end);
                float m = -INFINITY;
                for (int p = beg + l; p < end; p += 2) m = fmaxf(m, sAl[p]);
                m = fmaxf(m, dpp_move<0xB1>(m));
                float lsum = 0.f, aa[D];
#pragma unroll
                for (int d = 0; d < D; ++d) aa[d] = 0.f;
                for (int p = beg + l; p < end; p += 2) {
                    const float ex = __expf(sAl[p] - m);
                    sAl[p] = ex;
                    lsum += ex;
#pragma unroll
                    for (int d = 0; d < D; ++d) aa[d] = fmaf(ex, sEa[p * D + d], aa[d]);
                }
                lsum += dpp_move<0xB1>(lsum);
                const float inv = 1.0f / (lsum + 1e-16f);
                for (int p = beg + l; p < end; p += 2) sAl[p] *= inv;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const float a2 = aa[d] + dpp_move<0xB1>(aa[d]);
                    if (l == 0) sAa[r * D + d] = a2 * inv;
                }
            }
        }
        __syncthreads();
        TG_STAMP(2)
        // ---- C: aggregate, root term, activation
        if (live && !TG_VAR(16)) {
            for (int r = grp; r < n; r += G) {
                int beg = sRp[r] - e0, end = sRp[r + 1] - e0;
                beg = beg < 0 ? 0 : (beg > eb ? eb : beg);
                end = end < beg ? beg : (end > eb ? eb : end);
                const int64_t i = node0 + r;
                float4 acc[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[v] = f4zero();
                int p = beg;
                for (; p + 4 <= end; p += 4) {
                    float a[4];
                    float4 tvv[4][NV];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        a[u] = sAl[p + u];
                        const float* rowp = sTv + sCol[p + u] * H + c0;
#pragma unroll
                        for (int v = 0; v < NV; ++v) tvv[u][v] = *reinterpret_cast<const float4*>(rowp + 4 * v);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int v = 0; v < NV; ++v) acc[v] = fma4(a[u], tvv[u][v], acc[v]);
                }
                for (; p < end; ++p) {
                    const float a1 = sAl[p];
                    const float* rowp = sTv + sCol[p] * H + c0;
#pragma unroll
                    for (int v = 0; v < NV; ++v) acc[v] = fma4(a1, *reinterpret_cast<const float4*>(rowp + 4 * v), acc[v]);
                }
                float ad[D];
#pragma unroll
                for (int d = 0; d < D; ++d) ad[d] = sAa[r * D + d];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const float4 sk = *reinterpret_cast<const float4*>(sTs + r * H + c0 + 4 * v);
                    float oc[4] = {acc[v].x, acc[v].y, acc[v].z, acc[v].w};
#pragma unroll
                    for (int d = 0; d < D; ++d)
#pragma unroll
                        for (int c = 0; c < 4; ++c) oc[c] = fmaf(wl[4 * v + c][d], ad[d], oc[c]);
                    const float4 o4 = TG_VAR(256) ? make_float4(oc[0] + sk.x, oc[1] + sk.y, oc[2] + sk.z, oc[3] + sk.w)
                                                  : act_apply4s(make_float4(oc[0] + sk.x, oc[1] + sk.y, oc[2] + sk.z, oc[3] + sk.w),
                                                                act, stepv, (uint64_t)(i * H + c0 + 4 * v) >> 2);
                    if (!TG_VAR(128) || o4.x == 12345.678f) st4(out + i * H + c0 + 4 * v, o4);
                }
            }
            for (int p = t; p < eb; p += 256) alpha[e0 + p] = sAl[p];
            for (int k2 = t; k2 < n * D; k2 += 256) aa_out[node0 * D + k2] = sAa[k2];
        }
        TG_STAMP(3)
    }
    TG_STAMP_FLUSH
}

// ------------------------------------------------------------------------------------------------ backward
// One graph at a time per 1024-thread workgroup; the NEXT graph's rows are requested (into registers) before the current
// graph is processed, so a graph costs no exposed global round trip.  Per graph:
//   commit  the prefetched registers -> LDS: g = grad_out through the fused activation (also: gTs += g), the index
//           slices, alpha, edge features (CSR order, left behind by the forward), sum alpha ea per node
//   1a      one THREAD per edge: da_e = <g_i, T_v[j]> (64 FMAs, no lane reduction; rows padded by 4 floats against bank
//           conflicts); one thread per (node, d): ge = We^T g_i
//   2       source pass over the CSC, H/4 lanes per source: gTv[j] += sum alpha_e g_i (registers)
//           gWe += sum_i g_i (x) (sum alpha ea)_i (registers, one output per thread)
//   1c      one thread per destination: delta_i, ds_e = alpha_e (da_e - delta_i) -> gM[r][j] (LDS, one owner per row: fixed
//           order), gP[r]
struct TgBwdLds {
    int tv, g, gm, gp, wet, rp, rpt, col, row, colt, post, al, ea, aa, da, ge, total;     // float offsets
    size_t bytes() const { return (size_t)total * 4; }
};
__host__ __device__ inline TgBwdLds tg_bwd_lds(int n, int H, int D, int max_e) {
    TgBwdLds L;
    const int me = pad4(max_e > 0 ? max_e : 1);
    const int gred = 16 * H * D;                 // the final cross-wave sum of gWe reuses the g tile
    L.tv = 0;
    L.g = L.tv + n * (H + 4);
    L.gm = L.g + (n * (H + 4) > gred ? n * (H + 4) : gred);
    L.gp = L.gm + n * pad4(n);
    L.wet = L.gp + pad4(n * D);
    L.rp = L.wet + H * D;
    L.rpt = L.rp + pad4(n + 1);
    L.col = L.rpt + pad4(n + 1);
    L.row = L.col + me;
    L.colt = L.row + me;
    L.post = L.colt + me;
    L.al = L.post + me;
    L.ea = L.al + me;
    L.aa = L.ea + pad4(me * D);
    L.da = L.aa + pad4(n * D);
    L.ge = L.da + me;
    L.total = L.ge + pad4(n * D);
    return L;
}

template <int H, int D, int NT>
__global__ __launch_bounds__(NT) void tconv_bwd_graph_kernel(
    const float* __restrict__ gout, const float* __restrict__ y_act, ActParams act, const float* __restrict__ tv, int ld,
    const float* __restrict__ we, const float* __restrict__ ea_csr, const float* __restrict__ alpha,
    const float* __restrict__ aa_in, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colf,
    const int32_t* __restrict__ row, const int32_t* __restrict__ rowptr_t, const int32_t* __restrict__ col_t,
    const int32_t* __restrict__ pos_t, float* __restrict__ partials, int n, int64_t B, int max_e) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TPR = H / 4;
    constexpr int NG = NT / TPR;                              // lane groups = source rows per round
    constexpr int MAXR = (kTgMaxN + NG - 1) / NG;               // rounds: n <= kTgMaxN
    constexpr int HP = H + 4;                                   // padded row of the g / T_v tiles
    constexpr int GK = (kTgMaxN * H / 4 + NT - 1) / NT;         // float4s of the g tile a thread owns
    constexpr int HD = H * D;
    constexpr int NWV = NT / 64;                                // waves
    const TgBwdLds L = tg_bwd_lds(n, H, D, max_e);
    const TgRow R = tg_row(n, H, D);
    const int ldm = R.ldm;
    float* sTv = lds + L.tv;
    float* sG = lds + L.g;
    float* sGM = lds + L.gm;
    float* sGP = lds + L.gp;
    float* sWeT = lds + L.wet;                                  // [D][H]
    int* sRp = reinterpret_cast<int*>(lds + L.rp);
    int* sRpT = reinterpret_cast<int*>(lds + L.rpt);
    int* sCol = reinterpret_cast<int*>(lds + L.col);
    int* sRow = reinterpret_cast<int*>(lds + L.row);
    int* sColT = reinterpret_cast<int*>(lds + L.colt);
    int* sPosT = reinterpret_cast<int*>(lds + L.post);
    float* sAl = lds + L.al;
    float* sEa = lds + L.ea;
    float* sAa = lds + L.aa;
    float* sDa = lds + L.da;
    float* sGe = lds + L.ge;
    const int tid = threadIdx.x;
    const int sub = tid % TPR, grp = tid / TPR, c0 = 4 * sub;
    const int nG4 = n * H / 4;
#ifdef QOT_DIAG
    const int tg_var = g_tg_variant;
#endif

    TG_STAMP_DECL
    // ---- the prefetch registers of ONE graph
    int pf_col = 0, pf_row = 0, pf_colt = 0, pf_post = 0, pf_rp = 0, pf_rpt = 0;
    float pf_al = 0.f, pf_aa = 0.f, pf_ea[D];
    float4 pf_g[GK], pf_y[GK];
    int pf_e0 = 0, pf_eb = 0;                   // the prefetched graph's slot range
    int64_t pf_node0 = 0;
    auto prefetch = [&](int64_t b, int e0, int eb) {
        pf_e0 = e0; pf_eb = eb; pf_node0 = b * n;
        const int p = tid < eb ? tid : 0;
        if (eb > 0) {
            pf_col = colf[e0 + p]; pf_row = row[e0 + p]; pf_colt = col_t[e0 + p]; pf_post = pos_t[e0 + p];
            pf_al = alpha[e0 + p];
            if constexpr (D == 4) {
                const float4 q = ld4(ea_csr + (int64_t)(e0 + p) * 4);
                pf_ea[0] = q.x; pf_ea[1] = q.y; pf_ea[2] = q.z; pf_ea[3] = q.w;
            } else {
#pragma unroll
                for (int d = 0; d < D; ++d) pf_ea[d] = ea_csr[(int64_t)(e0 + p) * D + d];
            }
        }
        const int r = tid <= n ? tid : 0;
        pf_rp = rowptr[pf_node0 + r];
        pf_rpt = rowptr_t[pf_node0 + r];
        pf_aa = aa_in[pf_node0 * D + (tid < n * D ? tid : 0)];
#pragma unroll
        for (int k = 0; k < GK; ++k) {
            const int f = tid + NT * k;
            const int64_t flat = pf_node0 * H + 4 * (int64_t)(f < nG4 ? f : 0);
            pf_g[k] = ld4(gout + flat);
            if (y_act) pf_y[k] = ld4(y_act + flat);
        }
    };

    // ---- trip 1: slot ranges of the first two graphs; the tables
    // (the ranges are uniform values; read through VECTOR loads -- an address the compiler cannot prove uniform -- because a
    // scalar load shares lgkmcnt with LDS: the next LDS wait would then be a wait for an L2 round trip)
    int vz = 0;
    asm volatile("" : "+v"(vz));
    int64_t b = blockIdx.x;
    int e0n = 0, e1n = 0;                        // slot range of the graph after the prefetched one, RAW (arithmetic on a
    {                                            // just-loaded value is scheduled behind the load, with its wait)
        const int a0 = rowptr[b * n + vz], a1 = rowptr[(b + 1) * n + vz];
        const int64_t b2 = b + gridDim.x;
        if (b2 < B) { e0n = rowptr[b2 * n + vz]; e1n = rowptr[(b2 + 1) * n + vz]; }
        prefetch(b, a0, a1 - a0);
    }
    for (int i = tid; i < nG4; i += NT) {
        const int r = (4 * i) / H, c = (4 * i) % H;
        st4(sTv + r * HP + c, ld4(tv + (int64_t)r * ld + c));
    }
    for (int i = tid; i < n * ldm; i += NT) sGM[i] = 0.f;
    for (int i = tid; i < n * D; i += NT) sGP[i] = 0.f;
    for (int i = tid; i < HD; i += NT) sWeT[(i % D) * H + i / D] = we[i];
    float4 gTs[GK], gTv[MAXR];
#pragma unroll
    for (int k = 0; k < GK; ++k) gTs[k] = f4zero();
#pragma unroll
    for (int k = 0; k < MAXR; ++k) gTv[k] = f4zero();
    float wc[4][D];                                   // gWe: this lane's four channels, summed over the rows of its lane group
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < D; ++d) wc[c][d] = 0.f;
    const uint64_t stepv = (y_act && act.thr16) ? (uint64_t)act.step[0] : 0;      // read once

    for (; b < B; b += gridDim.x) {
        const int e0 = pf_e0;
        int eb = pf_eb;
        eb = eb < 0 ? 0 : (eb > max_e ? max_e : eb);
        const int64_t node0 = pf_node0;
        __syncthreads();                  // tables / the previous graph's pass 1c is done with the tiles
        TG_STAMP(4)
        // ---- commit the prefetched graph to LDS
        if (tid < eb) {
            const int j = pf_col, it = pf_colt - (int)node0, rr = pf_row - (int)node0, ps = pf_post - e0;
            sCol[tid] = j < 0 ? 0 : (j >= n ? n - 1 : j);
            sRow[tid] = rr < 0 ? 0 : (rr >= n ? n - 1 : rr);
            sColT[tid] = it < 0 ? 0 : (it >= n ? n - 1 : it);
            sPosT[tid] = ps < 0 ? 0 : (ps >= eb ? eb - 1 : ps);
            sAl[tid] = pf_al;
#pragma unroll
            for (int d = 0; d < D; ++d) sEa[tid * D + d] = pf_ea[d];
        }
        for (int p = tid + NT; p < eb; p += NT) {            // graphs of more than 1024 edges: not prefetched
            const int j = colf[e0 + p], it = col_t[e0 + p] - (int)node0, rr = row[e0 + p] - (int)node0, ps = pos_t[e0 + p] - e0;
            sCol[p] = j < 0 ? 0 : (j >= n ? n - 1 : j);
            sRow[p] = rr < 0 ? 0 : (rr >= n ? n - 1 : rr);
            sColT[p] = it < 0 ? 0 : (it >= n ? n - 1 : it);
            sPosT[p] = ps < 0 ? 0 : (ps >= eb ? eb - 1 : ps);
            sAl[p] = alpha[e0 + p];
#pragma unroll
            for (int d = 0; d < D; ++d) sEa[p * D + d] = ea_csr[(int64_t)(e0 + p) * D + d];
        }
        for (int r2 = tid + NT; r2 <= n; r2 += NT) {               // (NT <= n: not prefetched)
            const int v = rowptr[node0 + r2] - e0, vt = rowptr_t[node0 + r2] - e0;
            sRp[r2] = v < 0 ? 0 : (v > eb ? eb : v);
            sRpT[r2] = vt < 0 ? 0 : (vt > eb ? eb : vt);
        }
        if (tid <= n) {
            const int v = pf_rp - e0, vt = pf_rpt - e0;
            sRp[tid] = v < 0 ? 0 : (v > eb ? eb : v);
            sRpT[tid] = vt < 0 ? 0 : (vt > eb ? eb : vt);
        }
        if (tid < n * D) sAa[tid] = pf_aa;
        for (int q = tid + NT; q < n * D; q += NT) sAa[q] = aa_in[node0 * D + q];
#pragma unroll
        for (int k = 0; k < GK; ++k) {
            const int f = tid + NT * k;
            if (f < nG4) {
                float4 gi = pf_g[k];
                if (y_act && !TG_VAR(8)) {
                    const int64_t flat = node0 * H + 4 * (int64_t)f;
                    const float4 yy = pf_y[k];
                    uint64_t z = 0;
                    if (act.thr16) z = act_hash64(act.seed, stepv, (uint64_t)flat >> 2);
                    float vi[4] = {gi.x, gi.y, gi.z, gi.w};
                    const float vr[4] = {yy.x, yy.y, yy.z, yy.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const bool keep = act.thr16 ? (((uint32_t)(z >> (16 * c)) & 0xFFFFu) >= act.thr16) : true;
                        vi[c] = vi[c] * (keep ? act.keep_scale : 0.f) * (vr[c] > 0.f ? 1.0f : act.slope);
                    }
                    gi = make_float4(vi[0], vi[1], vi[2], vi[3]);
                }
                gTs[k] = add4(gTs[k], gi);
                const int r = (4 * f) / H, c = (4 * f) % H;
                st4(sG + r * HP + c, gi);
            }
        }
        __syncthreads();
        TG_STAMP(5)
        // ---- request the next graph (and the range of the one after it): in flight under this graph's passes
        {
            const int64_t bn = b + gridDim.x;
            if (bn < B) {
                const int e0x = e0n, ebx = e1n - e0n;
                const int64_t b2 = bn + gridDim.x;
                if (b2 < B) { e0n = rowptr[b2 * n + vz]; e1n = rowptr[(b2 + 1) * n + vz]; }
                prefetch(bn, e0x, ebx);
            }
        }
        // ---- 1a: da_e = <g_i, T_v[j]>, one thread per edge
        for (int p = tid; p < (TG_VAR(1) ? 0 : eb); p += NT) {
            const float* gr = sG + sRow[p] * HP;
            const float* vr = sTv + sCol[p] * HP;
            float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll(NT >= 1024 ? 2 : 8)
            for (int c = 0; c < H; c += 4) {
                const float4 a = *reinterpret_cast<const float4*>(gr + c);
                const float4 v = *reinterpret_cast<const float4*>(vr + c);
                d0 = fmaf(a.x, v.x, d0); d1 = fmaf(a.y, v.y, d1); d2 = fmaf(a.z, v.z, d2); d3 = fmaf(a.w, v.w, d3);
            }
            sDa[p] = (d0 + d1) + (d2 + d3);
        }
        // ge[i][d] = <g_i, We[:, d]>, one thread per (node, d)
        for (int q = tid; q < (TG_VAR(1) ? 0 : n * D); q += NT) {
            const int i = q / D, d = q % D;
            const float* gr = sG + i * HP;
            const float* wr = sWeT + d * H;
            float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll(NT >= 1024 ? 2 : 8)
            for (int c = 0; c < H; c += 4) {
                const float4 a = *reinterpret_cast<const float4*>(gr + c);
                const float4 v = *reinterpret_cast<const float4*>(wr + c);
                d0 = fmaf(a.x, v.x, d0); d1 = fmaf(a.y, v.y, d1); d2 = fmaf(a.z, v.z, d2); d3 = fmaf(a.w, v.w, d3);
            }
            sGe[q] = (d0 + d1) + (d2 + d3);
        }
        TG_STAMP(6)
        // ---- 2: source pass; gWe (value path) += g_j (x) (sum alpha ea)_j for the same rows (registers, four channels per lane)
#pragma unroll
        for (int k = 0; k < MAXR; ++k) {
            const int j = grp + NG * k;
            if (j < n && !TG_VAR(2)) {
                const int beg = sRpT[j], end = sRpT[j + 1];
                const float4 gj = *reinterpret_cast<const float4*>(sG + j * HP + c0);
                float ad[D];
#pragma unroll
                for (int d = 0; d < D; ++d) ad[d] = sAa[j * D + d];
                float4 acc = gTv[k];
                int s = beg;
                for (; s + 2 <= end; s += 2) {
                    const float a0 = sAl[sPosT[s]], a1 = sAl[sPosT[s + 1]];
                    const float4 g0 = *reinterpret_cast<const float4*>(sG + sColT[s] * HP + c0);
                    const float4 g1 = *reinterpret_cast<const float4*>(sG + sColT[s + 1] * HP + c0);
                    acc = fma4(a0, g0, acc);
                    acc = fma4(a1, g1, acc);
                }
                if (s < end) acc = fma4(sAl[sPosT[s]], *reinterpret_cast<const float4*>(sG + sColT[s] * HP + c0), acc);
                gTv[k] = acc;
                const float gc[4] = {gj.x, gj.y, gj.z, gj.w};
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int d = 0; d < D; ++d) wc[c][d] = fmaf(gc[c], ad[d], wc[c][d]);
            }
        }
        __syncthreads();
        TG_STAMP(7)
        // ---- 1c: per destination, a QUAD of lanes (edges dealt over the four lanes, sums met by DPP): delta, ds -> gM, gP.
        // gM[r][j] += ds_e by LDS float atomics: row r belongs to this quad alone, a quad's adds of one instruction
        // to one address (duplicate edges) are applied by the LDS in lane order, its instructions in program order --
        // the sum has a fixed order, nothing waits for a read-modify-write round trip
        for (int r = tid >> 2; r < (TG_VAR(4) ? 0 : n); r += NT / 4) {
            const int l = tid & 3;
            const int beg = sRp[r], end = sRp[r + 1];
            float ge[D], p1[D];
#pragma unroll
            for (int d = 0; d < D; ++d) { ge[d] = sGe[r * D + d]; p1[d] = 0.f; }
            float sada = 0.f;
            for (int p = beg + l; p < end; p += 4) {
                float da = sDa[p];
#pragma unroll
                for (int d = 0; d < D; ++d) da = fmaf(ge[d], sEa[p * D + d], da);
                sDa[p] = da;                                           // the complete da_e (this lane reads it back below)
                const float ada = sAl[p] * da;
                sada += ada;
#pragma unroll
                for (int d = 0; d < D; ++d) p1[d] = fmaf(ada, sEa[p * D + d], p1[d]);
            }
            sada += dpp_move<0xB1>(sada);
            sada += dpp_move<0x4E>(sada);
            for (int p = beg + l; p < end; p += 4)
                atomicAdd(&sGM[r * ldm + sCol[p]], sAl[p] * (sDa[p] - sada));       // ds = alpha (da - delta)
#pragma unroll
            for (int d = 0; d < D; ++d) {
                float q = p1[d];
                q += dpp_move<0xB1>(q);
                q += dpp_move<0x4E>(q);
                if (l == 0) sGP[r * D + d] += q - sada * sAa[r * D + d];
            }
        }
        TG_STAMP(8)
    }
    __syncthreads();
    // ---- the workgroup's partial row
    float* prow = partials + (int64_t)blockIdx.x * R.len;
#pragma unroll
    for (int k = 0; k < GK; ++k) {
        const int f = tid + NT * k;
        if (f < nG4) st4(prow + R.off_gs + 4 * f, gTs[k]);
    }
#pragma unroll
    for (int k = 0; k < MAXR; ++k) {
        const int r = grp + NG * k;
        if (r < n) st4(prow + R.off_gv + r * H + c0, gTv[k]);
    }
    for (int i = tid; i < n * ldm; i += NT) prow[R.off_gm + i] = sGM[i];
    for (int i = tid; i < n * D; i += NT) prow[R.off_gp + i] = sGP[i];
    // gWe: the lane groups of a wave meet through shuffles, the waves through LDS (the g tile is free now), fixed order.  The
    // P path's share, T_q^T gP / sqrt(H), is added by the projection backward from the SUMMED row (qot_table_project_bwd_scores)
    {
        float* red = sG;                                            // [NWV][HD]
        constexpr int GPW = 64 / TPR > 0 ? 64 / TPR : 1;            // lane groups per wave
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int d = 0; d < D; ++d) {
                float v = wc[c][d];
#pragma unroll
                for (int o = TPR; o < 64 && GPW > 1; o <<= 1) v += __shfl_xor(v, o);
                if ((tid & 63) < TPR) red[(tid >> 6) * HD + (c0 + c) * D + d] = v;
            }
        __syncthreads();
        for (int o = tid; o < HD; o += NT) {
            float s2 = red[o];
#pragma unroll
            for (int w = 1; w < NWV; ++w) s2 += red[w * HD + o];
            prow[R.off_gwe + o] = s2;
        }
    }
    TG_STAMP(9)
    TG_STAMP_FLUSH
}

static size_t kLdsMax = 160 * 1024;
#ifndef QOT_TG_BWD_THREADS
#define QOT_TG_BWD_THREADS 512
#endif
constexpr int kTgBwdThreads = QOT_TG_BWD_THREADS;    // 512: two waves per SIMD with the whole register file for the prefetch

}  // namespace qot

using namespace qot;

#ifdef QOT_DIAG
extern "C" void qot_debug_tg_variant(int v) { (void)hipMemcpyToSymbol(HIP_SYMBOL(qot::g_tg_variant), &v, sizeof(int)); }
extern "C" void qot_debug_tg_stamps(unsigned long long* host16, int reset) {
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(qot::g_tg_stamps), z, sizeof(z)); }
    else (void)hipMemcpyFromSymbol(host16, HIP_SYMBOL(qot::g_tg_stamps), 16 * sizeof(unsigned long long));
}
#endif

static bool tg_width_ok(int H) { return H == 16 || H == 32 || H == 64 || H == 128 || H == 256; }

// floats of one partial / summed row of qot_tconv_bwd_graph: [gTv n*H | gTs n*H | gM n*ldm | gP n*D | gWe H*D] (padded)
extern "C" size_t qot_tconv_graph_row_floats(int n, int H, int D) {
    if (n <= 0 || !tg_width_ok(H) || D <= 0 || D > 8) return 0;
    return (size_t)tg_row(n, H, D).len;
}
// leading dimension of the score matrix M [n, ldm]
extern "C" int qot_tconv_graph_ldm(int n) { return n > 0 ? pad4(n) : 0; }
// workgroups (= partial rows) of qot_tconv_bwd_graph for B graphs
extern "C" int qot_tconv_bwd_graph_blocks(int64_t B) {
    if (B <= 0) return 0;
    const int cus = num_cus();
    return B < cus ? (int)B : cus;
}
// 1 when the graph form takes (n nodes per graph, at most max_e edges per graph, width H, edge_dim D), else 0
extern "C" int qot_tconv_graph_supported(int n, int max_e, int H, int D) {
    if (n <= 0 || n > kTgMaxN || max_e < 0 || !tg_width_ok(H) || D <= 0 || D > 8) return 0;
    if (tg_fwd_lds(n, H, D, max_e).bytes(1) > kLdsMax) return 0;
    if (tg_bwd_lds(n, H, D, max_e).bytes() > kLdsMax) return 0;
    return 1;
}

extern "C" int qot_tconv_fwd_graph(const float* t4, int ld, const float* M, const float* P, const float* w_edge,
                                   const float* edge_attr, const int32_t* rowptr, const int32_t* colf, const int32_t* eid,
                                   const int32_t* row, float* out, float* alpha, float* ea_csr, float* aa, int n, int64_t B,
                                   int max_e, int H, int D, int act, float act_slope, float act_p, uint64_t act_seed,
                                   const int64_t* act_step, qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n <= 0 || B < 0 || max_e < 0 || ld < 4 * H || (ld & 3)) return QOT_ERR_BADARG;
    if (B == 0) return QOT_OK;
    if (!t4 || !M || !P || !w_edge || !rowptr || !out || !aa) return QOT_ERR_BADARG;
    if (max_e > 0 && (!edge_attr || !colf || !eid || !row || !alpha || !ea_csr)) return QOT_ERR_BADARG;
    if (!qot_tconv_graph_supported(n, max_e, H, D)) return QOT_ERR_UNSUPPORTED;
    if ((int64_t)n * B >= (int64_t(1) << 31) / 2) return QOT_ERR_UNSUPPORTED;
    const TgFwdLds L = tg_fwd_lds(n, H, D, max_e);
    int ns = 4;
    while (ns > 1 && (L.bytes(ns) > kLdsMax || (B + ns - 1) / ns < (num_cus() + 1) / 2)) ns >>= 1;   // keep the chip covered
    const size_t lds = L.bytes(ns);
    int64_t grid = (B + ns - 1) / ns;
    if (grid > num_cus()) grid = num_cus();
    const ActParams ap = make_act(act, act_slope, act_p, act_seed, act_step);
    QOT_DISPATCH_H(H, QOT_DISPATCH_D(D, {
        static size_t allowed[kMaxDevices];
        const int lrc = ensure_dyn_lds(reinterpret_cast<const void*>(tconv_fwd_graph_kernel<kH, kD>), lds, allowed);
        if (lrc != QOT_OK) return lrc;
        tconv_fwd_graph_kernel<kH, kD><<<(int)grid, 256 * ns, lds, stream>>>(
            t4 + 2 * H, t4 + 3 * H, ld, M, P, w_edge, edge_attr, rowptr, colf, eid, row, out, alpha, ea_csr, aa, n, B, max_e,
            ap);
    }));
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

// partials: [qot_tconv_bwd_graph_blocks(B)][qot_tconv_graph_row_floats(n, H, D)]; y_act != NULL: grad_out is the gradient
// wrt y = dropout(leaky_relu(conv)) and y_act is that output (as qot_tconv_bwd_dst); alpha / ea_csr / aa: as the forward
// left them
extern "C" int qot_tconv_bwd_graph(const float* grad_out, const float* y_act, float act_slope, float act_p,
                                   uint64_t act_seed, const int64_t* act_step, const float* t4, int ld,
                                   const float* w_edge, const float* ea_csr, const float* alpha, const float* aa,
                                   const int32_t* rowptr, const int32_t* colf, const int32_t* row,
                                   const int32_t* rowptr_t, const int32_t* col_t, const int32_t* pos_t, float* partials,
                                   int n, int64_t B, int max_e, int H, int D, qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n <= 0 || B <= 0 || max_e < 0 || ld < 4 * H || (ld & 3)) return QOT_ERR_BADARG;
    if (!grad_out || !t4 || !w_edge || !aa || !rowptr || !rowptr_t || !partials) return QOT_ERR_BADARG;
    if (max_e > 0 && (!ea_csr || !alpha || !colf || !row || !col_t || !pos_t)) return QOT_ERR_BADARG;
    if (!qot_tconv_graph_supported(n, max_e, H, D)) return QOT_ERR_UNSUPPORTED;
    if ((int64_t)n * B >= (int64_t(1) << 31) / 2) return QOT_ERR_UNSUPPORTED;
    const size_t lds = tg_bwd_lds(n, H, D, max_e).bytes();
    const int grid = qot_tconv_bwd_graph_blocks(B);
    const ActParams ap = make_act(y_act ? 1 : 0, act_slope, act_p, act_seed, act_step);
    QOT_DISPATCH_H(H, QOT_DISPATCH_D(D, {
        static size_t allowed[kMaxDevices];
        constexpr int NT = kTgBwdThreads;
        const int lrc = ensure_dyn_lds(reinterpret_cast<const void*>(tconv_bwd_graph_kernel<kH, kD, NT>), lds, allowed);
        if (lrc != QOT_OK) return lrc;
        tconv_bwd_graph_kernel<kH, kD, NT><<<grid, NT, lds, stream>>>(
            grad_out, y_act, ap, t4 + 2 * H, ld, w_edge, ea_csr, alpha, aa, rowptr, colf, row, rowptr_t, col_t, pos_t,
            partials, n, B, max_e);
    }));
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

// M [n, qot_tconv_graph_ldm(n)] and P [n, D] from the parameters (one-role call of the multi-role launch)
extern "C" int qot_table_scores(const float* table, const float* wq, const float* bq, const float* wk, const float* bk,
                                const float* w_edge, float* M, float* P, int n, int H, int D, qot_stream_t stream) {
    qot_role_t r{};
    r.kind = QOT_ROLE_TABLE_SCORES;
    const void* ptrs[8] = {table, wq, bq, wk, bk, w_edge, M, P};
    for (int k = 0; k < 8; ++k) r.p[k] = ptrs[k];
    r.i[0] = n; r.i[1] = H; r.i[2] = D;
    return qot_run_roles(&r, 1, stream);
}

// backward of the table projection from the summed partial row S of qot_tconv_bwd_graph (one-role call); also completes
// grad w_edge [H, D] = S's value-path share + T_q^T gP / sqrt(H)
extern "C" int qot_table_project_bwd_scores(const float* S, const float* t4, const float* w_edge, const float* table,
                                            const float* wq, const float* wk, const float* wv, const float* ws,
                                            float* grad_table, float* grad_w, float* grad_b, float* grad_w_edge, int V,
                                            int n, int H, int D, qot_stream_t stream) {
    qot_role_t r{};
    r.kind = QOT_ROLE_TABLE_PROJECT_BWD_SCORES;
    const void* ptrs[12] = {S, t4, w_edge, table, wq, wk, wv, ws, grad_table, grad_w, grad_b, grad_w_edge};
    for (int k = 0; k < 12; ++k) r.p[k] = ptrs[k];
    r.i[0] = V; r.i[1] = n; r.i[2] = H; r.i[3] = D;
    return qot_run_roles(&r, 1, stream);
}
