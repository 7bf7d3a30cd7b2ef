// TransformerConv (heads = 1, edge features, root weight) -- fused edge-softmax-aggregate.
// Replaces [PyG-ext] TransformerConv.propagate reached from
// topological_training/models.py:53: 3 gathers [E,H], lin_edge
// materialisation [E,H], a 3-pass scatter softmax and a scatter_add become ONE pass over
// the destination-sorted CSR.
//
// Mapping: a group of TPR = H/CPL consecutive lanes owns one destination node; every lane holds CPL = 4 channels
// (one 16-B load per row per lane -> a row read is one contiguous H*4-byte burst; tconv_cpl below).  64/TPR
// destinations per wave.  The edge embedding is never formed:
//   <q_i, k_j + We ea> = <q_i, k_j> + <We^T q_i, ea>        (D-vector per destination)
//   sum_e a_e (v_j + We ea_e) = sum_e a_e v_j + We (sum_e a_e ea_e)
// so per edge only the 4 raw edge features are read.  Online softmax (running max / sum).
#include "common.hpp"

namespace qot {

// channels per lane.  8 from H = 64 on (half the lanes per destination, half the redundant softmax scalars per edge) was
// measured and is SLOWER at cfg2 (fwd 32 -> 36-40 us, destination pass 44 -> 51-54 us, source pass 24 -> 24.5-26 us with
// two or four rows in flight): fewer, longer waves at 3-4 waves per SIMD hide less of the dependent-load latency than
// the instruction count saves.  The kernels stay written for either value.
__host__ __device__ constexpr int tconv_cpl(int /*H*/) { return 4; }
__host__ __device__ constexpr int tconv_rpb(int H) { return 256 / (H / tconv_cpl(H)); }  // destinations per workgroup

template <int NV>
__device__ __forceinline__ float dotv(const float4 (&a)[NV], const float4 (&b)[NV]) {
    float s = dot4(a[0], b[0]);
#pragma unroll
    for (int v = 1; v < NV; ++v) s += dot4(a[v], b[v]);
    return s;
}
__device__ __forceinline__ float comp4(const float4& a, int c) { return c == 0 ? a.x : (c == 1 ? a.y : (c == 2 ? a.z : a.w)); }

// MT (table mode, rowmap != NULL): the dense part of the logit, <q_i, k_j>, is entry (row_i, row_j) of the V x V matrix
// Mtab = T_q T_k^T of the projected table (one small MFMA product per step, 4 MB at V = 1000: L2-resident) -- the lane that
// prefetches an edge's source row looks it up, and the per-edge key-row gather, the H-term dot and its lane-group reduction
// go; the value row is the only row gathered per edge.  (The tile form above computes rows of that matrix inside the
// workgroup, which only pays for small V.)
template <int H, int D, bool MT = false>
__global__ __launch_bounds__(256) void tconv_fwd_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v_,
    const float* __restrict__ skip, int ld, const float* __restrict__ ea,
    const float* __restrict__ we, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ col, const int32_t* __restrict__ eid, const int32_t* __restrict__ rowmap,
    float* __restrict__ out, float* __restrict__ stats, int64_t N, ActParams act,
    const float* __restrict__ Mtab = nullptr, int ldm = 0) {
    constexpr int CPL = tconv_cpl(H), NV = CPL / 4;
    constexpr int TPR = H / CPL;
    constexpr int RPB = 256 / TPR;
    const int sub = threadIdx.x % TPR;
    const int64_t i = (int64_t)xcd_block(blockIdx.x, gridDim.x) * RPB + threadIdx.x / TPR;
    if (i >= N) return;
    const float rs = rsqrtf((float)H);
    const int c0 = CPL * sub;
    // table mode: q/k/v/skip are rows of a projected embedding table; rowmap = node -> table row
    // for this node, and `col` already holds the table row of every in-edge's source
    const int64_t ri = rowmap ? (int64_t)rowmap[i] : i;

    float4 qi[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) qi[v] = scale4(rs, ld4(q + ri * ld + c0 + 4 * v));
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
        for (int c = 0; c < CPL; ++c) t = fmaf(comp4(qi[c >> 2], c & 3), wl[c][d], t);
        qe[d] = group_sum<TPR>(t);
    }

    float m = -INFINITY, l = 0.f;
    float4 acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = f4zero();
    float aacc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) aacc[d] = 0.f;

    // In-edges in batches of BT: lane `sub` of the group prefetches edge `sub`'s (source row,
    // edge features) so the per-edge dependent chain (col -> rows) is paid once per batch, and
    // four source rows are in flight per group.
    const int beg = rowptr[i], end = rowptr[i + 1];
    constexpr int BT = (TPR < 16) ? TPR : 16;                 // edges prefetched per batch
    for (int base = beg; base < end; base += BT) {
        const int pme = base + sub;
        int myj = 0;
        float mye[D], mym = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) mye[d] = 0.f;
        if (sub < BT && pme < end) {
            myj = col[pme];
            const int64_t e = eid[pme];
#pragma unroll
            for (int d = 0; d < D; ++d) mye[d] = ea[e * D + d];
            if (MT) mym = rs * Mtab[ri * ldm + myj];
        }
        const int cnt = (end - base < BT) ? end - base : BT;
        constexpr int UF = 4;            // source rows in flight per group (2 x UF x CPL registers)
        for (int u0 = 0; u0 < cnt; u0 += UF) {
            float4 kr[UF][NV], vr[UF][NV];
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                const int64_t j = __shfl(myj, u0 + u, TPR);
                const int64_t jr = (u0 + u < cnt) ? j : 0;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    if (!MT) kr[u][v] = ld4(k + jr * ld + c0 + 4 * v);
                    vr[u][v] = ld4(v_ + jr * ld + c0 + 4 * v);
                }
            }
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                if (u0 + u < cnt) {                           // group-uniform
                    float ee[D];
#pragma unroll
                    for (int d = 0; d < D; ++d) ee[d] = __shfl(mye[d], u0 + u, TPR);
                    float s = MT ? __shfl(mym, u0 + u, TPR) : group_sum<TPR>(dotv<NV>(qi, kr[u]));
#pragma unroll
                    for (int d = 0; d < D; ++d) s = fmaf(qe[d], ee[d], s);
                    float mn = fmaxf(m, s);
                    float sc = __expf(m - mn);
                    float pe = __expf(s - mn);
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
        const float4 sk = ld4(skip + ri * ld + c0 + 4 * v);
        st4(out + i * H + c0 + 4 * v, act_apply4(make_float4(oc[0] + sk.x, oc[1] + sk.y, oc[2] + sk.z, oc[3] + sk.w), act,
                                                 (uint64_t)(i * H + c0 + 4 * v) >> 2));
    }
    if (sub == 0) {
        stats[2 * i] = (beg < end) ? m : 0.f;
        stats[2 * i + 1] = denom;
    }
}

// Forward, TILE form of table mode (node_ids == arange(tile_n) in each of the tile_B graphs: what the reference's dataset
// emits, topological_training/dataset.py:78).  A workgroup takes node r of RPB consecutive graphs (the mapping the
// destination pass of the backward already uses), so all its destinations share ONE query row q_r, and the logits'
// dense part  <q_i, k_j> / sqrt(H)  is an entry of row r of the n x n matrix  T_q T_k^T  of the projected table: the
// workgroup computes that row ONCE (n dots of H, spread over its lane groups) into LDS, and an in-edge then costs one LDS
// lookup instead of a 4H-byte key-row gather, a dot and a lane-group reduction -- the value row is the only row gathered
// per edge.  Per-edge features stay with the lane that prefetched them (it forms the whole logit; the weighted feature
// sum is kept per owner lane and reduced once per destination): two broadcasts per edge instead of five.
// Same arithmetic as tconv_fwd_kernel up to the order of the H-term dot (row-of-S first, then + <We^T q, ea>).
template <int H, int D>
__global__ __launch_bounds__(256) void tconv_fwd_tile_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v_,
    const float* __restrict__ skip, int ld, const float* __restrict__ ea, const float* __restrict__ we,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colf, const int32_t* __restrict__ eid,
    const int32_t* __restrict__ rowmap, float* __restrict__ out, float* __restrict__ stats, int64_t N, int tile_n,
    int64_t tile_B, ActParams act) {
    constexpr int CPL = tconv_cpl(H), NV = CPL / 4;
    constexpr int TPR = H / CPL;
    constexpr int RPB = 256 / TPR;
    extern __shared__ float srow[];                       // [tile_n]: <q_r, k_j> / sqrt(H)
    const int sub = threadIdx.x % TPR, rloc = threadIdx.x / TPR;
    const int vb = xcd_block(blockIdx.x, gridDim.x);
    const int gb = vb / tile_n, r = vb % tile_n;
    const int64_t gph = (int64_t)gb * RPB + rloc;
    const bool live = gph < tile_B;
    const int64_t i = live ? gph * tile_n + r : (int64_t)gb * RPB * tile_n + r;     // dead groups: the first graph's node (reads only)
    const float rs = rsqrtf((float)H);
    const int c0 = CPL * sub;
    const int64_t ri = (int64_t)rowmap[i];
    float4 qi[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) qi[v] = scale4(rs, ld4(q + ri * ld + c0 + 4 * v));
    // in-edge range and the first batch's indices / features are requested before the row of S is formed
    const int beg = live ? rowptr[i] : 0, end = live ? rowptr[i + 1] : 0;
    // row r of T_q T_k^T: lane group rloc takes j = rloc, rloc + RPB, ...
    for (int j0 = rloc; j0 < tile_n; j0 += 4 * RPB) {
        float4 kr[4][NV];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = (j0 + u * RPB < tile_n) ? j0 + u * RPB : j0;
#pragma unroll
            for (int v = 0; v < NV; ++v) kr[u][v] = ld4(k + (int64_t)j * ld + c0 + 4 * v);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float s = group_sum<TPR>(dotv<NV>(qi, kr[u]));
            if (sub == 0 && j0 + u * RPB < tile_n) srow[j0 + u * RPB] = s;
        }
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
        for (int c = 0; c < CPL; ++c) t = fmaf(comp4(qi[c >> 2], c & 3), wl[c][d], t);
        qe[d] = group_sum<TPR>(t);
    }
    __syncthreads();
    if (!live) return;

    float m = -INFINITY, l = 0.f;
    float4 acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = f4zero();
    float aloc[D];                                        // sum_e a_e ea_e over the edges THIS lane prefetched
#pragma unroll
    for (int d = 0; d < D; ++d) aloc[d] = 0.f;
    constexpr int BT = (TPR < 16) ? TPR : 16;
    for (int base = beg; base < end; base += BT) {
        const int pme = base + sub;
        int myj = 0;
        float mys = 0.f, mye[D];
#pragma unroll
        for (int d = 0; d < D; ++d) mye[d] = 0.f;
        if (sub < BT && pme < end) {
            myj = colf[pme];
            const int64_t e = eid[pme];
#pragma unroll
            for (int d = 0; d < D; ++d) mye[d] = ea[e * D + d];
            mys = srow[myj];
#pragma unroll
            for (int d = 0; d < D; ++d) mys = fmaf(qe[d], mye[d], mys);          // the whole logit of my edge
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
                    const float s = __shfl(mys, u0 + u, TPR);
                    const float mn = fmaxf(m, s);
                    const float sc = __expf(m - mn);
                    const float pe = __expf(s - mn);
                    l = fmaf(l, sc, pe);
#pragma unroll
                    for (int v = 0; v < NV; ++v) acc[v] = fma4(pe, vr[u][v], scale4(sc, acc[v]));
                    const float mine = (sub == u0 + u) ? pe : 0.f;
#pragma unroll
                    for (int d = 0; d < D; ++d) aloc[d] = fmaf(mine, mye[d], aloc[d] * sc);
                    m = mn;
                }
            }
        }
    }
    float aacc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) aacc[d] = group_sum<TPR>(aloc[d]);
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
        const float4 sk = ld4(skip + ri * ld + c0 + 4 * v);
        st4(out + i * H + c0 + 4 * v, act_apply4(make_float4(oc[0] + sk.x, oc[1] + sk.y, oc[2] + sk.z, oc[3] + sk.w), act,
                                                 (uint64_t)(i * H + c0 + 4 * v) >> 2));
    }
    if (sub == 0) {
        stats[2 * i] = (beg < end) ? m : 0.f;
        stats[2 * i + 1] = denom;
    }
}

// Backward, destination pass.  With a_e the attention weight and da_e = <g_i, v_j + We ea_e>:
//   delta_i = sum_e a_e da_e ;  ds_e = a_e (da_e - delta_i)
//   grad_q_i = (sum_e ds_e k_j + We sum_e ds_e ea_e)/sqrt(H)
//            = ((A1 - delta A2) + We (P1 - delta P2))/sqrt(H)
// with A1 = sum a da k_j, A2 = sum a k_j, P1 = sum a da ea, P2 = sum a ea -- a single
// sweep over the in-edges, no second gather of k.
template <int H, int D>
__global__ __launch_bounds__(256) void tconv_bwd_dst_kernel(
    const float* __restrict__ g, const float* __restrict__ q, const float* __restrict__ k,
    const float* __restrict__ v_, int ld, const float* __restrict__ ea,
    const float* __restrict__ we, const float* __restrict__ stats,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const int32_t* __restrict__ eid, const int32_t* __restrict__ rowmap, float* __restrict__ gq,
    float* __restrict__ gskip, int ld_g, float* __restrict__ escr, float* __restrict__ delta, float* __restrict__ pds, float* __restrict__ pal,
    int64_t N, const float* __restrict__ y_act, ActParams act, float* __restrict__ wedge_partials,
    int tile_n, int64_t tile_B, float* __restrict__ gpart) {
    constexpr int CPL = tconv_cpl(H), NV = CPL / 4;
    constexpr int TPR = H / CPL;
    constexpr int RPB = 256 / TPR;
    constexpr int DCH = D > 4 ? 4 : D;             // lin_edge partials meet in LDS in chunks of <= 4 edge features
    __shared__ float wred[RPB * H * DCH];
    __shared__ float4 tred[2][NV][256];
    const int sub = threadIdx.x % TPR;
    const int rloc = threadIdx.x / TPR;
    // Row of this lane group.  Tile mode (table mode with node_ids == arange(n) per graph, gpart !=
    // NULL): the workgroup takes node r of RPB consecutive graphs, so that the table gradient -- the
    // sum over graphs -- is pre-reduced over those RPB rows in LDS instead of being written per node
    // and re-read by a row-sum (4H floats per node each way).
    int64_t i, prow = 0;
    {
        const int vb = xcd_block(blockIdx.x, gridDim.x);
        if (gpart) {
            const int gb = vb / tile_n, r = vb % tile_n;
            const int64_t gph = (int64_t)gb * RPB + rloc;
            i = gph < tile_B ? gph * tile_n + r : N;
            prow = (int64_t)gb * tile_n + r;
        } else {
            i = (int64_t)vb * RPB + rloc;
        }
    }
    float4 rq[NV], rg[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) { rq[v] = f4zero(); rg[v] = f4zero(); }
    const int c0 = CPL * sub;
    // rows past N (last block) contribute zeros to the block reduction below; no thread leaves
    // before the barrier (a wave can hold live and dead rows)
    float wc[CPL][D];
#pragma unroll
    for (int c = 0; c < CPL; ++c)
#pragma unroll
        for (int d = 0; d < D; ++d) wc[c][d] = 0.f;
    if (i < N) {
    const float rs = rsqrtf((float)H);
    const int64_t ri = rowmap ? (int64_t)rowmap[i] : i;

    float4 qi[NV], gi[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        qi[v] = scale4(rs, ld4(q + ri * ld + c0 + 4 * v));
        gi[v] = ld4(g + i * H + c0 + 4 * v);
    }
    if (y_act) {      // grad_out arrives for y = dropout(leaky_relu(conv)): go back through it here
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int64_t flat = i * H + c0 + 4 * v;
            const float4 yy = ld4(y_act + flat);
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
    float wl[CPL][D];
#pragma unroll
    for (int c = 0; c < CPL; ++c)
#pragma unroll
        for (int d = 0; d < D; ++d) wl[c][d] = we[(c0 + c) * D + d];
    float qe[D], ge[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float tq = 0.f, tg = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            tq = fmaf(comp4(qi[c >> 2], c & 3), wl[c][d], tq);
            tg = fmaf(comp4(gi[c >> 2], c & 3), wl[c][d], tg);
        }
        qe[d] = group_sum<TPR>(tq);
        ge[d] = group_sum<TPR>(tg);
    }
    const float m = stats[2 * i];
    const float inv = 1.0f / stats[2 * i + 1];

    float4 a1[NV], a2[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) { a1[v] = f4zero(); a2[v] = f4zero(); }
    float p1[D], p2[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { p1[d] = 0.f; p2[d] = 0.f; }
    float sada = 0.f;

    const int beg = rowptr[i], end = rowptr[i + 1];
    constexpr int BT = (TPR < 16) ? TPR : 16;
    for (int base = beg; base < end; base += BT) {
        const int pme = base + sub;
        int myj = 0;
        float mye[D];
#pragma unroll
        for (int d = 0; d < D; ++d) mye[d] = 0.f;
        if (sub < BT && pme < end) {
            myj = col[pme];
            const int64_t e = eid[pme];
#pragma unroll
            for (int d = 0; d < D; ++d) mye[d] = ea[e * D + d];
        }
        const int cnt = (end - base < BT) ? end - base : BT;
        constexpr int UF = 4;            // source rows in flight per group (2 x UF x CPL registers)
        for (int u0 = 0; u0 < cnt; u0 += UF) {
            float4 kr[UF][NV], vr[UF][NV];
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                const int64_t j = __shfl(myj, u0 + u, TPR);
                const int64_t jr = (u0 + u < cnt) ? j : 0;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    kr[u][v] = ld4(k + jr * ld + c0 + 4 * v);
                    vr[u][v] = ld4(v_ + jr * ld + c0 + 4 * v);
                }
            }
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                if (u0 + u < cnt) {
                    const int p = base + u0 + u;
                    float ee[D];
#pragma unroll
                    for (int d = 0; d < D; ++d) ee[d] = __shfl(mye[d], u0 + u, TPR);
                    float s = group_sum<TPR>(dotv<NV>(qi, kr[u])), da = group_sum<TPR>(dotv<NV>(gi, vr[u]));
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        s = fmaf(qe[d], ee[d], s);
                        da = fmaf(ge[d], ee[d], da);
                    }
                    const float a = __expf(s - m) * inv;
                    const float ada = a * da;
                    sada += ada;
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        a1[v] = fma4(ada, kr[u][v], a1[v]);
                        a2[v] = fma4(a, kr[u][v], a2[v]);
                    }
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        p1[d] = fmaf(ada, ee[d], p1[d]);
                        p2[d] = fmaf(a, ee[d], p2[d]);
                    }
                    if (sub == 0) {
                        escr[2 * (int64_t)p] = a;
                        escr[2 * (int64_t)p + 1] = da;
                    }
                }
            }
        }
    }
    float pd[D];
#pragma unroll
    for (int d = 0; d < D; ++d) pd[d] = p1[d] - sada * p2[d];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const float4 r = sub4(a1[v], scale4(sada, a2[v]));
        float rc[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int d = 0; d < D; ++d) rc[c] = fmaf(wl[4 * v + c][d], pd[d], rc[c]);
            rc[c] *= rs;
        }
        rq[v] = make_float4(rc[0], rc[1], rc[2], rc[3]);
        rg[v] = gi[v];
        if (!gpart) st4(gq + i * ld_g + c0 + 4 * v, rq[v]);
        if (gskip) st4(gskip + i * ld_g + c0 + 4 * v, gi[v]);     // grad of the skip projection is grad_out itself
    }
    if (sub == 0) {
        delta[i] = sada;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            pds[i * D + d] = pd[d];
            pal[i * D + d] = p2[d];
        }
    }
    // grad of lin_edge.weight: gWe[c,d] += q_i[c]/sqrt(H) * pd_i[d] + g_i[c] * p2_i[d]
#pragma unroll
    for (int c = 0; c < CPL; ++c)
#pragma unroll
        for (int d = 0; d < D; ++d) wc[c][d] = fmaf(comp4(qi[c >> 2], c & 3), pd[d], comp4(gi[c >> 2], c & 3) * p2[d]);
    }   // i < N
    if (wedge_partials) {
        // rows of the block meet in LDS, blocks in qot_tconv_bwd_dst's final fixed-order sum
#pragma unroll
        for (int d0 = 0; d0 < D; d0 += DCH) {
            const int dc = (D - d0 < DCH) ? D - d0 : DCH;
            if (d0) __syncthreads();
#pragma unroll
            for (int c = 0; c < CPL; ++c)
#pragma unroll
                for (int dd = 0; dd < DCH; ++dd)
                    if (d0 + dd < D) wred[rloc * H * DCH + dd * H + c0 + c] = wc[c][d0 + dd];   // [row][feature][channel]: lanes
            __syncthreads();                                                                  // write consecutive 16-B pieces
            for (int o = threadIdx.x; o < H * dc; o += 256) {
                const int dd = o / H, ch = o % H;
                float sacc = 0.f;
                for (int r = 0; r < RPB; ++r) sacc += wred[r * H * DCH + dd * H + ch];
                wedge_partials[(int64_t)blockIdx.x * H * D + ch * D + d0 + dd] = sacc;
            }
        }
    }
    if (gpart) {       // rows of the RPB graphs meet in LDS (fixed order): one partial table row per workgroup
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            tred[0][v][threadIdx.x] = rq[v];
            tred[1][v][threadIdx.x] = rg[v];
        }
        __syncthreads();
        if (rloc == 0) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                float4 a = tred[0][v][sub], bsum = tred[1][v][sub];
                for (int r2 = 1; r2 < RPB; ++r2) {
                    a = add4(a, tred[0][v][r2 * TPR + sub]);
                    bsum = add4(bsum, tred[1][v][r2 * TPR + sub]);
                }
                st4(gpart + prow * 4 * H + c0 + 4 * v, a);
                st4(gpart + prow * 4 * H + 3 * H + c0 + 4 * v, bsum);
            }
        }
    }
}

// Backward, source pass over the CSC: grad_v_j = sum_{e: j->i} a_e g_i,
// grad_k_j = sum_e ds_e q_i / sqrt(H).
template <int H>
__global__ __launch_bounds__(256) void tconv_bwd_src_kernel(
    const float* __restrict__ g, int ld_go, const float* __restrict__ q, int ld,
    const float* __restrict__ escr, const float* __restrict__ delta,
    const int32_t* __restrict__ rowptr_t, const int32_t* __restrict__ col_t,
    const int32_t* __restrict__ pos_t, const int32_t* __restrict__ qmap_t, float* __restrict__ gk,
    float* __restrict__ gv, int ld_g, int64_t N, int tile_n, int64_t tile_B, float* __restrict__ gpart) {
    constexpr int CPL = tconv_cpl(H), NV = CPL / 4;
    constexpr int TPR = H / CPL;
    constexpr int RPB = 256 / TPR;
    __shared__ float4 tred[2][NV][256];
    const int sub = threadIdx.x % TPR;
    const int rloc = threadIdx.x / TPR;
    int64_t j, prow = 0;
    {
        const int vb = xcd_block(blockIdx.x, gridDim.x);
        if (gpart) {                                   // tile mode: see tconv_bwd_dst_kernel
            const int gb = vb / tile_n, r = vb % tile_n;
            const int64_t gph = (int64_t)gb * RPB + rloc;
            j = gph < tile_B ? gph * tile_n + r : N;
            prow = (int64_t)gb * tile_n + r;
        } else {
            j = (int64_t)vb * RPB + rloc;
        }
    }
    if (j >= N && !gpart) return;
    const float rs = rsqrtf((float)H);
    const int c0 = CPL * sub;
    float4 ak[NV], av[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) { ak[v] = f4zero(); av[v] = f4zero(); }
    int beg = 0, end = 0;
    if (j < N) { beg = rowptr_t[j]; end = rowptr_t[j + 1]; }
    constexpr int BT = (TPR < 16) ? TPR : 16;
    for (int base = beg; base < end; base += BT) {
        const int tme = base + sub;
        int myi = 0, myq = 0;
        float mya = 0.f, myds = 0.f;
        if (sub < BT && tme < end) {
            myi = col_t[tme];
            myq = qmap_t ? qmap_t[tme] : myi;
            const int64_t p = pos_t[tme];
            mya = escr[2 * p];
            myds = mya * (escr[2 * p + 1] - delta[myi]) * rs;
        }
        const int cnt = (end - base < BT) ? end - base : BT;
        constexpr int UF = 4;
        for (int u0 = 0; u0 < cnt; u0 += UF) {
            float4 gr[UF][NV], qr[UF][NV];
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                const int64_t i = __shfl(myi, u0 + u, TPR);
                const int64_t iq = __shfl(myq, u0 + u, TPR);
                const bool live = u0 + u < cnt;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    gr[u][v] = ld4(g + (live ? i : 0) * ld_go + c0 + 4 * v);
                    qr[u][v] = ld4(q + (live ? iq : 0) * ld + c0 + 4 * v);
                }
            }
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                const bool live = u0 + u < cnt;
                const float a = live ? __shfl(mya, u0 + u, TPR) : 0.f;
                const float ds = live ? __shfl(myds, u0 + u, TPR) : 0.f;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    av[v] = fma4(a, gr[u][v], av[v]);
                    ak[v] = fma4(ds, qr[u][v], ak[v]);
                }
            }
        }
    }
    if (!gpart) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            st4(gk + j * ld_g + c0 + 4 * v, ak[v]);
            st4(gv + j * ld_g + c0 + 4 * v, av[v]);
        }
        return;
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        tred[0][v][threadIdx.x] = ak[v];
        tred[1][v][threadIdx.x] = av[v];
    }
    __syncthreads();
    if (rloc == 0) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float4 a = tred[0][v][sub], bsum = tred[1][v][sub];
            for (int r2 = 1; r2 < RPB; ++r2) {
                a = add4(a, tred[0][v][r2 * TPR + sub]);
                bsum = add4(bsum, tred[1][v][r2 * TPR + sub]);
            }
            st4(gpart + prow * 4 * H + H + c0 + 4 * v, a);
            st4(gpart + prow * 4 * H + 2 * H + c0 + 4 * v, bsum);
        }
    }
}

__global__ void partial_sum_kernel(const float* __restrict__ partials, int nblk, int n, float* __restrict__ out) {
    const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);      // one wave per output
    if (t >= n) return;
    const float s = wave_sum_partials(partials, nblk, n, t);
    if ((threadIdx.x & 63) == 0) out[t] = s;
}

// out[g*n + t] = sum of partials[b*n + t] over the blocks b of group g (one wave per output)
__global__ void partial_sum_groups_kernel(const float* __restrict__ partials, int nblk, int n, int per_group,
                                          float* __restrict__ out) {
    const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= n) return;
    const int b0 = blockIdx.y * per_group;
    const int cnt = (b0 + per_group <= nblk) ? per_group : nblk - b0;
    const float s = wave_sum_partials(partials + (int64_t)b0 * n, cnt, n, t);
    if ((threadIdx.x & 63) == 0) out[(int64_t)blockIdx.y * n + t] = s;
}

// Coalesced form of the group sum: 1024 threads = 64 consecutive outputs x 16 part sub-groups, every load a
// 256-B row segment, sub-groups meet in LDS (the one-wave-per-output form above reads each partial with a
// stride of n floats; 10.9 us for the 6.5 MB of cfg2's lin_edge partials).
__global__ __launch_bounds__(1024) void partial_sum_groups_rows_kernel(const float* __restrict__ partials, int nblk,
                                                                       int n, int per_group, float* __restrict__ out) {
    __shared__ float red[16][64];
    const int o = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + o;
    const int b0 = blockIdx.y * per_group;
    const int cnt = (b0 + per_group <= nblk) ? per_group : nblk - b0;
    float acc = 0.f;
    if (t < n) {
        const int per = (cnt + 15) / 16;
        const int p0 = sg * per, p1 = (p0 + per < cnt) ? p0 + per : cnt;
        int p = p0;
        for (; p + 8 <= p1; p += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partials[(int64_t)(b0 + p + u) * n + t];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; p < p1; ++p) acc += partials[(int64_t)(b0 + p) * n + t];
    }
    red[sg][o] = acc;
    __syncthreads();
    if (sg == 0 && t < n) {
        float s = red[0][o];
#pragma unroll
        for (int q = 1; q < 16; ++q) s += red[q][o];
        out[(int64_t)blockIdx.y * n + t] = s;
    }
}

constexpr int kWedgeBlocks = 512;
constexpr int kWedgeGroup = 128;

}  // namespace qot

using namespace qot;

extern "C" int qot_tconv_fwd(const float* q, const float* k, const float* v, const float* skip, int ld,
                             const float* edge_attr, const float* w_edge, const int32_t* rowptr,
                             const int32_t* col, const int32_t* eid, const int32_t* rowmap, float* out,
                             float* stats, int64_t N, int H, int D, int act, float act_slope, float act_p,
                             uint64_t act_seed, const int64_t* act_step, qot_stream_t stream) {
    if (N < 0 || !rowptr) return QOT_ERR_BADARG;
    if (N == 0) return QOT_OK;
    if (!q || !k || !v || !skip || !out || !stats || !w_edge || (ld & 3)) return QOT_ERR_BADARG;
    QOT_DISPATCH_H(H, QOT_DISPATCH_D(D, {
        constexpr int RPB = tconv_rpb(kH);
        tconv_fwd_kernel<kH, kD><<<grid_for(N, RPB), 256, 0, (hipStream_t)stream>>>(
            q, k, v, skip, ld, edge_attr, w_edge, rowptr, col, eid, rowmap, out, stats, N,
            make_act(act, act_slope, act_p, act_seed, act_step));
    }));
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

// Table mode with the score matrix (tconv_fwd_kernel<., ., MT>): scores[V, ld_scores] = T_q T_k^T (unscaled), rowmap = table
// row of every node, col = table row of every in-edge's source.  k is not read.
extern "C" int qot_tconv_fwd_scores(const float* q, const float* v, const float* skip, int ld, const float* scores,
                                    int ld_scores, const float* edge_attr, const float* w_edge, const int32_t* rowptr,
                                    const int32_t* col, const int32_t* eid, const int32_t* rowmap, float* out, float* stats,
                                    int64_t N, int H, int D, int act, float act_slope, float act_p, uint64_t act_seed,
                                    const int64_t* act_step, qot_stream_t stream) {
    if (N < 0 || !rowptr) return QOT_ERR_BADARG;
    if (N == 0) return QOT_OK;
    if (!q || !v || !skip || !out || !stats || !w_edge || !scores || !rowmap || !col || (ld & 3) || ld_scores <= 0)
        return QOT_ERR_BADARG;
    QOT_DISPATCH_H(H, QOT_DISPATCH_D(D, {
        constexpr int RPB = tconv_rpb(kH);
        tconv_fwd_kernel<kH, kD, true><<<grid_for(N, RPB), 256, 0, (hipStream_t)stream>>>(
            q, nullptr, v, skip, ld, edge_attr, w_edge, rowptr, col, eid, rowmap, out, stats, N,
            make_act(act, act_slope, act_p, act_seed, act_step), scores, ld_scores);
    }));
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

// Tile form of table mode (see tconv_fwd_tile_kernel): col = table row of every in-edge's source, rowmap = table row of every
// node, node_ids == arange(tile_n) in each of the tile_B graphs (N = tile_n * tile_B).
extern "C" int qot_tconv_fwd_tile(const float* q, const float* k, const float* v, const float* skip, int ld,
                                  const float* edge_attr, const float* w_edge, const int32_t* rowptr,
                                  const int32_t* col, const int32_t* eid, const int32_t* rowmap, float* out,
                                  float* stats, int64_t N, int H, int D, int tile_n, int64_t tile_B, int act,
                                  float act_slope, float act_p, uint64_t act_seed, const int64_t* act_step,
                                  qot_stream_t stream) {
    if (N <= 0 || !rowptr || tile_n <= 0 || tile_B <= 0 || (int64_t)tile_n * tile_B != N) return QOT_ERR_BADARG;
    if (!q || !k || !v || !skip || !out || !stats || !w_edge || !rowmap || !col || (ld & 3)) return QOT_ERR_BADARG;
    if (tile_n > 12288) return QOT_ERR_UNSUPPORTED;                 // the row of T_q T_k^T lives in LDS
    QOT_DISPATCH_H(H, QOT_DISPATCH_D(D, {
        constexpr int RPB = tconv_rpb(kH);
        const int64_t blocks = (int64_t)tile_n * grid_for(tile_B, RPB);
        if (blocks > 0x7fffffff) return QOT_ERR_UNSUPPORTED;
        tconv_fwd_tile_kernel<kH, kD><<<(int)blocks, 256, (size_t)tile_n * sizeof(float), (hipStream_t)stream>>>(
            q, k, v, skip, ld, edge_attr, w_edge, rowptr, col, eid, rowmap, out, stats, N, tile_n, tile_B,
            make_act(act, act_slope, act_p, act_seed, act_step));
    }));
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_tconv_bwd_dst(const float* grad_out, const float* q, const float* k, const float* v,
                                 int ld, const float* edge_attr, const float* w_edge,
                                 const float* stats, const int32_t* rowptr, const int32_t* col,
                                 const int32_t* eid, const int32_t* rowmap, float* grad_q, float* grad_skip,
                                 int ld_g, float* escr, float* delta, float* pds, float* pal,
                                 const float* y_act, float act_slope, float act_p, uint64_t act_seed,
                                 const int64_t* act_step, float* grad_w_edge, float* workspace,
                                 int tile_n, int64_t tile_B, float* grad_part,
                                 int64_t N, int H, int D, qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N < 0 || !rowptr) return QOT_ERR_BADARG;
    if (N == 0) return QOT_OK;
    if (!grad_out || !q || !k || !v || !stats || (!grad_q && !grad_part) || !delta || !pds || !pal || (ld & 3) || (ld_g & 3))
        return QOT_ERR_BADARG;
    if (grad_part && (tile_n <= 0 || tile_B <= 0 || (int64_t)tile_n * tile_B != N || !rowmap)) return QOT_ERR_BADARG;
    if (grad_w_edge && !workspace) return QOT_ERR_BADARG;
    const ActParams ap = make_act(y_act ? 1 : 0, act_slope, act_p, act_seed, act_step);
    int blocks = 0;
    QOT_DISPATCH_H(H, QOT_DISPATCH_D(D, {
        constexpr int RPB = tconv_rpb(kH);
        blocks = grad_part ? tile_n * grid_for(tile_B, RPB) : grid_for(N, RPB);
        tconv_bwd_dst_kernel<kH, kD><<<blocks, 256, 0, stream>>>(
            grad_out, q, k, v, ld, edge_attr, w_edge, stats, rowptr, col, eid, rowmap, grad_q, grad_skip, ld_g, escr,
            delta, pds, pal, N, y_act, ap, workspace, tile_n, tile_B, grad_part);
    }));
    QOT_LAUNCH_CHECK();
    // grad_w_edge == NULL with a workspace: the block partials [blocks, H*D] stay in the workspace and the caller sums
    // them (QOT_ROLE_SUM_ROWS; qot_tconv_bwd_dst_blocks gives the row count)
    if (grad_w_edge) {
        const int n = H * D;
        if (blocks > 2 * kWedgeGroup) {       // two levels: groups of kWedgeGroup blocks, then the groups
            const int groups = grid_for(blocks, kWedgeGroup);
            float* level1 = workspace + (size_t)blocks * n;
            partial_sum_groups_rows_kernel<<<dim3(grid_for(n, 64), groups), 1024, 0, stream>>>(workspace, blocks, n,
                                                                                            kWedgeGroup, level1);
            QOT_LAUNCH_CHECK();
            partial_sum_kernel<<<grid_for(n, 4), 256, 0, stream>>>(level1, groups, n, grad_w_edge);
        } else {
            partial_sum_kernel<<<grid_for(n, 4), 256, 0, stream>>>(workspace, blocks, n, grad_w_edge);
        }
        QOT_LAUNCH_CHECK();
    }
    return QOT_OK;
}

// workgroups (= rows of lin_edge partials in the workspace) of qot_tconv_bwd_dst
extern "C" int64_t qot_tconv_bwd_dst_blocks(int64_t N, int H, int tile_n, int64_t tile_B) {
    if (N <= 0 || (H != 16 && H != 32 && H != 64 && H != 128 && H != 256)) return 0;
    const int rpb = tconv_rpb(H);
    return tile_n > 0 ? (int64_t)tile_n * ((tile_B + rpb - 1) / rpb) : (N + rpb - 1) / rpb;
}

// destinations per workgroup of the TransformerConv kernels at width H (table mode pre-reduces the table gradient over
// that many graphs: grad_part is [ceil(tile_B / RPB), tile_n, 4H])
extern "C" int qot_tconv_rows_per_block(int H) {
    if (H != 16 && H != 32 && H != 64 && H != 128 && H != 256) return 0;
    return tconv_rpb(H);
}

// floats of workspace qot_tconv_bwd_dst needs when grad_w_edge is requested (tile mode: pass
// N = tile_n * ceil(tile_B / RPB) * RPB, RPB = qot_tconv_rows_per_block(H))
extern "C" size_t qot_tconv_bwd_dst_workspace_floats(int64_t N, int H, int D) {
    if (N <= 0 || H < 4 || D <= 0) return 0;
    const int64_t rpb = tconv_rpb(H);
    const int64_t blocks = (N + rpb - 1) / rpb;
    const int64_t groups = (blocks + kWedgeGroup - 1) / kWedgeGroup;
    return (size_t)(blocks + groups) * (size_t)H * (size_t)D;
}

extern "C" int qot_tconv_bwd_src(const float* grad_out, int ld_go, const float* q, int ld, const float* escr,
                                 const float* delta, const int32_t* rowptr_t, const int32_t* col_t,
                                 const int32_t* pos_t, const int32_t* qmap_t, float* grad_k, float* grad_v,
                                 int ld_g, int tile_n, int64_t tile_B, float* grad_part, int64_t N, int H,
                                 qot_stream_t stream) {
    if (N < 0 || !rowptr_t) return QOT_ERR_BADARG;
    if (N == 0) return QOT_OK;
    if (!grad_out || !q || !delta || ((!grad_k || !grad_v) && !grad_part) || (ld & 3) || (ld_g & 3) || (ld_go & 3))
        return QOT_ERR_BADARG;
    if (grad_part && (tile_n <= 0 || tile_B <= 0 || (int64_t)tile_n * tile_B != N)) return QOT_ERR_BADARG;
    QOT_DISPATCH_H(H, {
        constexpr int RPB = tconv_rpb(kH);
        const int blocks = grad_part ? tile_n * grid_for(tile_B, RPB) : grid_for(N, RPB);
        tconv_bwd_src_kernel<kH><<<blocks, 256, 0, (hipStream_t)stream>>>(
            grad_out, ld_go, q, ld, escr, delta, rowptr_t, col_t, pos_t, qmap_t, grad_k, grad_v, ld_g, N, tile_n,
            tile_B, grad_part);
    });
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}
