// GATConv (concat heads, self loops, LeakyReLU(0.2) logits) -- fused edge-softmax-aggregate.
// Replaces [PyG-ext] GATConv.edge_updater/propagate reached from
// lightpath_training/models.py:30.  The self-loop rewrite of the edge
// list is done once by qot_csr_build(gat_self_loops=1); logits are scalars per (edge, head)
// (a_src[j,h] + a_dst[i,h]) so the softmax needs no cross-lane traffic at all: every lane
// tracks the running (max, sum) of the head its channels belong to.
//
// Mapping: TPR = min(64, HC/4) lanes per destination, NV = HC/(4*TPR) float4 per lane;
// float4 number v of lane `sub` covers channels 4*(sub + TPR*v) .. +3, all in one head.
#include <cstdint>
#include "common.hpp"
#include "mfma_tile.hpp"      // num_cus()

namespace qot {

template <int HEADS, int C>
struct GatCfg {
    static constexpr int HC = HEADS * C;
    static constexpr int TPR = (HC / 4 < 64) ? HC / 4 : 64;
    static constexpr int NV = HC / (4 * TPR);
    static constexpr int RPB = 256 / TPR;
    static constexpr int LPH = (C / 4 < TPR) ? C / 4 : TPR;  // lanes that share one head
};

// Attention logits from the projected rows (App. B.3): a_src[n,h] = <z[n,h,:], att_src[h,:]>, a_dst likewise.
// One pass over z; a head's channels sit on LPH consecutive lanes of one float4 slot, so the reduction is a
// DPP / shuffle butterfly inside that lane group.
template <int HEADS, int C>
__global__ __launch_bounds__(256) void gat_logits_kernel(const float* __restrict__ z, const float* __restrict__ att_src,
                                                         const float* __restrict__ att_dst, float* __restrict__ a_src,
                                                         float* __restrict__ a_dst, int64_t N) {
    using G = GatCfg<HEADS, C>;
    const int sub = threadIdx.x % G::TPR;
    const int64_t i = (int64_t)blockIdx.x * G::RPB + threadIdx.x / G::TPR;
    if (i >= N) return;
#pragma unroll
    for (int v = 0; v < G::NV; ++v) {
        const int c = 4 * (sub + G::TPR * v);
        const float4 zv = ld4(z + i * G::HC + c);
        const float ps = group_sum<G::LPH>(dot4(zv, ld4(att_src + c)));
        const float pd = group_sum<G::LPH>(dot4(zv, ld4(att_dst + c)));
        if (c % C == 0) { a_src[i * HEADS + c / C] = ps; a_dst[i * HEADS + c / C] = pd; }
    }
}

// THIN instantiations (r04): the projected rows z_j = W x_j of a layer with a handful of input features (LightpathGNN's first:
// K = 5) are FORMED where the dense kernels read them -- K scalars of x_j instead of a 4C-float row, 4 K FMAs per float4 with W
// (transposed, [K][4C]) in LDS -- so that z is never written or read (1.47 GB each way per pass at cfg3).  Same walk, same cache.
#define QOT_GAT_THIN_ROWS                                                                                             \
    extern __shared__ float4 thin_w[];                           /* [K][HC / 4], THIN only */                         \
    float pfx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     /* the prefetched node's features */                 \
    if constexpr (THIN) {                                                                                             \
        for (int e_ = threadIdx.x; e_ < K * G::HC; e_ += 256)                                                         \
            reinterpret_cast<float*>(thin_w)[e_] = wlin[(int64_t)(e_ % G::HC) * K + e_ / G::HC];                      \
        __syncthreads();                                                                                              \
    }                                                                                                                 \
    auto thin_read = [&](int64_t j_, float(&xk_)[8]) {                                                                \
        _Pragma("unroll") for (int k_ = 0; k_ < 8; ++k_) xk_[k_] = k_ < K ? xin[j_ * K + k_] : 0.f;                   \
    };                                                                                                                \
    auto thin_form = [&](const float(&xk_)[8], float4(&r_)[G::NV]) {                                                  \
        const int sub_ = threadIdx.x % G::TPR;                                                                        \
        _Pragma("unroll") for (int v_ = 0; v_ < G::NV; ++v_) r_[v_] = f4zero();                                       \
        _Pragma("unroll") for (int k_ = 0; k_ < 8; ++k_)                                                              \
            if (k_ < K) {                                                                                             \
                _Pragma("unroll") for (int v_ = 0; v_ < G::NV; ++v_)                                                  \
                    r_[v_] = fma4(xk_[k_], thin_w[k_ * (G::HC / 4) + sub_ + G::TPR * v_], r_[v_]);                    \
            }                                                                                                         \
    };

// Forward.  Every workgroup walks a contiguous chunk of destinations (slot s takes rows r0+s, r0+s+RPB, ...), two
// in-edges in flight per lane group.  bn_partials != NULL: per-workgroup column (mean, M2 = sum of squared deviations) of
// out - bias land in partials[blk][2][HC] -- the BatchNorm that follows (lightpath_training/models.py:31) gets its batch
// statistics without another pass over the [N, 4C] matrix.  Every lane accumulates relative to the FIRST value it sees
// (a data value: sums of x - x0 and (x - x0)^2 do not cancel however far the column's mean is from zero or from the
// bias), lanes and workgroups are merged with Chan's pairwise formula in a fixed order (qot_bn_stats_from_partials).
// (r02 kept sums of (out - bias) and its square: for a channel with |mean - bias| >> std -- behind a ReLU, or large
// activations -- E[d^2] - E[d]^2 lost the variance's leading bits.)
template <int HEADS, int C, bool THIN = false>
__global__ __launch_bounds__(256, (HEADS * C >= 1024) ? 2 : 4) void gat_fwd_kernel(
    const float* __restrict__ z, const float* __restrict__ a_src, const float* __restrict__ a_dst,
    const float* __restrict__ bias, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ col, float* __restrict__ out, float* __restrict__ stats, int64_t N,
    float ns, float* __restrict__ bn_partials, const float* __restrict__ xin = nullptr,
    const float* __restrict__ wlin = nullptr, int K = 0) {
    using G = GatCfg<HEADS, C>;
    QOT_GAT_THIN_ROWS
    // r04 walk: every lane group (slot) takes a CONTIGUOUS run of its workgroup's chunk and keeps the source rows of the
    // destination it just finished in a slot-private LDS image (no barrier: only this lane group reads or writes it).  The
    // batches' graphs are numbered along their structure -- LightpathGNN's are chains (lightpath_training/dataset.py): the
    // sources of destination i are i - 1, i, i + 1 -- so of the next destination's rows all but one are already there; that
    // one is requested a destination ahead (pf), and the index reads two ahead.  Every z row then leaves L2 once (the
    // interleaved walk of r03 re-read 46 % of z: 2.15 GB for 1.47 at cfg3) and a destination waits for no round trip that
    // was not started a whole destination earlier.  Any other numbering stays correct: a source that is neither cached nor
    // prefetched is a plain load, EB of them in flight as before.
    constexpr int EB = G::NV >= 4 ? 2 : 3;
    constexpr int HC4 = G::HC / 4;
    __shared__ float4 cache[G::RPB * EB * HC4];
    static_assert(sizeof(float4) * G::RPB * EB * HC4 >= 2 * 256 * sizeof(float4) + 256 * sizeof(int), "reduction scratch aliases the cache");
    __shared__ float lcache[EB * G::NV * 256];          // the cached rows' source logits, [row][v][thread]
    float4* red1 = cache;
    float4* red2 = cache + 256;
    int* rcnt = reinterpret_cast<int*>(cache + 512);
    const int sub = threadIdx.x % G::TPR;
    // a full wave per destination (HC >= 256): the slot is wave-uniform -- said so, the walk's index state (row pointers,
    // sources, cache tags, the decisions made on them) lives in scalar registers and the row addresses are a scalar base +
    // one lane offset
    const int slot = G::TPR == 64 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (int)(threadIdx.x / G::TPR);
    float4* mine = cache + (size_t)slot * EB * HC4;
    const int blk = xcd_block(blockIdx.x, gridDim.x);
    const int64_t chunk = ((N + gridDim.x - 1) / gridDim.x + G::RPB - 1) / G::RPB * G::RPB;
    const int64_t r0 = (int64_t)blk * chunk;
    const int64_t r1 = (r0 + chunk < N) ? r0 + chunk : N;
    const int64_t run = chunk / G::RPB;
    const int64_t i0 = r0 + (int64_t)slot * run;
    const int64_t i1 = (i0 + run < r1) ? i0 + run : r1;
    int hh[G::NV];
    float4 s1[G::NV], s2[G::NV], d0[G::NV];
#pragma unroll
    for (int v = 0; v < G::NV; ++v) {
        hh[v] = (4 * (sub + G::TPR * v)) / C;
        s1[v] = f4zero(); s2[v] = f4zero(); d0[v] = f4zero();
    }
    // index reads, branch-free: rows past the matrix are clamped to its last row and slots past a destination's in-edges to
    // its last one (what they fetch is never used)
    const int elast = rowptr[N] - 1;
    auto ptrs = [&](int64_t r, int& b, int& e_) {
        const int64_t rc = r < N ? r : N - 1;
        b = rowptr[rc];
        e_ = rowptr[rc + 1];
    };
    auto srcs = [&](int b, int e_, int (&j)[EB]) {
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            int q = (b + e < e_) ? b + e : e_ - 1;
            q = q < elast ? q : elast;
            j[e] = elast >= 0 ? col[q > 0 ? q : 0] : 0;
        }
    };
    int beg, end, nbeg, nend, b2, e2;
    int jc[EB], jn[EB], cid[EB];
    int pfid = -1;
    float4 pf[G::NV];
    float pfa[G::NV], adn[G::NV];
#pragma unroll
    for (int v = 0; v < G::NV; ++v) {
        pf[v] = f4zero();
        pfa[v] = 0.f;
        adn[v] = a_dst[(i0 < N ? i0 : N - 1) * HEADS + hh[v]];
    }
#pragma unroll
    for (int e = 0; e < EB; ++e) cid[e] = -1;
    ptrs(i0, beg, end);
    ptrs(i0 + 1, nbeg, nend);
    ptrs(i0 + 2, b2, e2);
    srcs(beg, end, jc);
    srcs(nbeg, nend, jn);
    int nrows = 0;
    for (int64_t i = i0; i < i1; ++i, ++nrows) {
        int b3, e3, j2[EB];
        ptrs(i + 3, b3, e3);
        srcs(b2, e2, j2);
        // the row of destination i + 1 that this destination does not bring along: requested now
        int pid = -1;
        if (i + 1 < i1) {
#pragma unroll
            for (int e = EB - 1; e >= 0; --e) {
                bool have = nbeg + e >= nend;
#pragma unroll
                for (int k = 0; k < EB; ++k) have = have || (beg + k < end && jn[e] == jc[k]);
                if (!have) pid = jn[e];
            }
        }
        const bool any = beg < end;
        float ad[G::NV], m[G::NV], l[G::NV];
        float4 acc[G::NV];
#pragma unroll
        for (int v = 0; v < G::NV; ++v) {
            ad[v] = adn[v];
            adn[v] = a_dst[(i + 1 < N ? i + 1 : i) * HEADS + hh[v]];
            m[v] = -INFINITY;
            l[v] = 0.f;
            acc[v] = f4zero();
        }
        for (int p = beg; p < end; p += EB) {
            float4 zz[EB][G::NV];
            float as_[EB][G::NV];
            const bool first = p == beg;
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                const int64_t j = jc[e];
                int where = -1;                              // -1: load, EB: the prefetched row, k: cache row k
                if (first) {
#pragma unroll
                    for (int k = 0; k < EB; ++k) where = (jc[e] == cid[k]) ? k : where;
                    where = (jc[e] == pfid) ? EB : where;
                }
                if (where < 0) {
                    if constexpr (THIN) {
                        float xk[8];
                        thin_read(j, xk);
                        thin_form(xk, zz[e]);
                    }
#pragma unroll
                    for (int v = 0; v < G::NV; ++v) {
                        if constexpr (!THIN) zz[e][v] = ld4(z + j * G::HC + 4 * (sub + G::TPR * v));
                        as_[e][v] = a_src[j * HEADS + hh[v]];
                    }
                } else if (where == EB) {
                    if constexpr (THIN) thin_form(pfx, zz[e]);
#pragma unroll
                    for (int v = 0; v < G::NV; ++v) {
                        if constexpr (!THIN) zz[e][v] = pf[v];
                        as_[e][v] = pfa[v];
                    }
                } else {
#pragma unroll
                    for (int v = 0; v < G::NV; ++v) {
                        zz[e][v] = mine[where * HC4 + sub + G::TPR * v];
                        as_[e][v] = lcache[(where * G::NV + v) * 256 + threadIdx.x];
                    }
                }
            }
            if (first && pid >= 0) {             // (pf has been handed to zz: its registers take the next destination's row)
#pragma unroll
                for (int v = 0; v < G::NV; ++v) {
                    if constexpr (!THIN) pf[v] = ld4(z + (int64_t)pid * G::HC + 4 * (sub + G::TPR * v));
                    pfa[v] = a_src[(int64_t)pid * HEADS + hh[v]];
                }
                if constexpr (THIN) thin_read(pid, pfx);
            }
            if (first) {          // this destination's rows are the next one's cache (slots past its in-edges: empty)
#pragma unroll
                for (int e = 0; e < EB; ++e) {
                    cid[e] = (p + e < end) ? jc[e] : -1;
#pragma unroll
                    for (int v = 0; v < G::NV; ++v) {
                        mine[e * HC4 + sub + G::TPR * v] = zz[e][v];
                        lcache[(e * G::NV + v) * 256 + threadIdx.x] = as_[e][v];
                    }
                }
            }
            const int pnext = p + EB;
            if (pnext < end) {                 // a destination with more than EB in-edges: the next batch of sources
#pragma unroll
                for (int e = 0; e < EB; ++e) jc[e] = col[(pnext + e < end) ? pnext + e : pnext];
            }
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                if (e > 0 && p + e >= end) break;
#pragma unroll
                for (int v = 0; v < G::NV; ++v) {
                    const float raw = as_[e][v] + ad[v];
                    const float s = raw > 0.f ? raw : ns * raw;
                    const float mn = fmaxf(m[v], s);
                    const float sc = __expf(m[v] - mn);
                    const float pe = __expf(s - mn);
                    l[v] = fmaf(l[v], sc, pe);
                    acc[v] = fma4(pe, zz[e][v], scale4(sc, acc[v]));
                    m[v] = mn;
                }
            }
        }
        if (beg >= end) {
#pragma unroll
            for (int e = 0; e < EB; ++e) cid[e] = -1;
        }
#pragma unroll
        for (int v = 0; v < G::NV; ++v) {
            const int c = 4 * (sub + G::TPR * v);
            const float denom = l[v] + 1e-16f;
            const float4 d = scale4(1.0f / denom, acc[v]);            // out - bias
            st4(out + i * G::HC + c, add4(d, ld4(bias + c)));
            if (nrows == 0) d0[v] = d;
            const float4 e = sub4(d, d0[v]);
            s1[v] = add4(s1[v], e);
            s2[v] = make_float4(fmaf(e.x, e.x, s2[v].x), fmaf(e.y, e.y, s2[v].y), fmaf(e.z, e.z, s2[v].z), fmaf(e.w, e.w, s2[v].w));
            if (c % C == 0)
                *reinterpret_cast<float2*>(stats + (i * HEADS + hh[v]) * 2) = make_float2((beg < end) ? m[v] : 0.f, denom);
        }
        beg = nbeg; end = nend; nbeg = b2; nend = e2; b2 = b3; e2 = e3;
        pfid = any ? pid : -1;
#pragma unroll
        for (int e = 0; e < EB; ++e) { jc[e] = jn[e]; jn[e] = j2[e]; }
    }
    if (bn_partials) {          // fixed order: slots 0..RPB-1 of this workgroup (Chan merges), then the workgroups
        // this slot's (count, mean, M2) per column from its shifted sums
        const float fn = (float)nrows, rn = nrows > 0 ? 1.0f / fn : 0.f;
#pragma unroll
        for (int v = 0; v < G::NV; ++v) {
            const float4 mu = make_float4(fmaf(s1[v].x, rn, d0[v].x), fmaf(s1[v].y, rn, d0[v].y), fmaf(s1[v].z, rn, d0[v].z),
                                          fmaf(s1[v].w, rn, d0[v].w));
            const float4 m2 = make_float4(fmaxf(s2[v].x - s1[v].x * s1[v].x * rn, 0.f), fmaxf(s2[v].y - s1[v].y * s1[v].y * rn, 0.f),
                                          fmaxf(s2[v].z - s1[v].z * s1[v].z * rn, 0.f), fmaxf(s2[v].w - s1[v].w * s1[v].w * rn, 0.f));
            __syncthreads();
            red1[threadIdx.x] = mu;
            red2[threadIdx.x] = m2;
            rcnt[threadIdx.x] = nrows;
            __syncthreads();
            if (slot == 0) {
                float4 ma = red1[sub], qa = red2[sub];
                float na = (float)rcnt[sub];
                for (int k = 1; k < G::RPB; ++k) {
                    const float nb = (float)rcnt[k * G::TPR + sub];
                    if (nb == 0.f) continue;
                    const float4 mb = red1[k * G::TPR + sub], qb = red2[k * G::TPR + sub];
                    const float nn = na + nb, fb = nb / nn, w = na * fb;       // w = na nb / (na + nb)
                    const float4 dl = sub4(mb, ma);
                    ma = make_float4(fmaf(dl.x, fb, ma.x), fmaf(dl.y, fb, ma.y), fmaf(dl.z, fb, ma.z), fmaf(dl.w, fb, ma.w));
                    qa = make_float4(qa.x + qb.x + dl.x * dl.x * w, qa.y + qb.y + dl.y * dl.y * w, qa.z + qb.z + dl.z * dl.z * w,
                                     qa.w + qb.w + dl.w * dl.w * w);
                    na = nn;
                }
                const int c = 4 * (sub + G::TPR * v);
                st4(bn_partials + ((int64_t)blk * 2) * G::HC + c, ma);
                st4(bn_partials + ((int64_t)blk * 2 + 1) * G::HC + c, qa);
            }
        }
    }
}

// destination pass of the backward: per (edge, head) alpha and dalpha = <g_i,h, z_j,h>,
// delta_{i,h} = sum_e alpha dalpha, grad_a_dst[i,h] = sum_e alpha (dalpha - delta) lrelu'(raw).
// Same walk as the forward (r04): a contiguous run of destinations per lane group, the previous destination's z rows and
// source logits in a slot-private LDS image, the one row the next destination adds and its own g row, logit and softmax
// statistics requested a destination ahead, the index reads two ahead.  (r03: one workgroup per RPB destinations, one
// in-edge at a time: seven dependent round trips per chain destination, 733 us at cfg3 for 3.0 GB.)
template <int HEADS, int C, bool THIN = false>
__global__ __launch_bounds__(256, (HEADS * C >= 1024) ? 2 : 4) void gat_bwd_dst_kernel(
    const float* __restrict__ g, const float* __restrict__ z, const float* __restrict__ a_src,
    const float* __restrict__ a_dst, const float* __restrict__ stats,
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, float* __restrict__ gad,
    float* __restrict__ escr, float* __restrict__ delta, int64_t N, float ns, float* __restrict__ bias_partials,
    const float* __restrict__ xin = nullptr, const float* __restrict__ wlin = nullptr, int K = 0) {
    // bias_partials != NULL: the column sums of g (GATConv's bias gradient) of this workgroup's rows land in
    // bias_partials[blk][HC] -- every g row passes through here once anyway (r03: a pass of its own over [N, 4C], 178 us
    // per layer at cfg3); gat_bias_final_kernel adds the workgroups' rows in a fixed order.
    using G = GatCfg<HEADS, C>;
    QOT_GAT_THIN_ROWS
    constexpr int EB = G::NV >= 4 ? 2 : 3;
    constexpr int HC4 = G::HC / 4;
    __shared__ float4 cache[G::RPB * EB * HC4];
    __shared__ float lcache[EB * G::NV * 256];
    static_assert(sizeof(float4) * G::RPB * EB * HC4 >= 256 * sizeof(float4), "reduction scratch aliases the cache");
    float4 bsum[G::NV];
#pragma unroll
    for (int v = 0; v < G::NV; ++v) bsum[v] = f4zero();
    const int sub = threadIdx.x % G::TPR;
    const int slot = G::TPR == 64 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (int)(threadIdx.x / G::TPR);
    float4* mine = cache + (size_t)slot * EB * HC4;
    const int blk = xcd_block(blockIdx.x, gridDim.x);
    const int64_t chunk = ((N + gridDim.x - 1) / gridDim.x + G::RPB - 1) / G::RPB * G::RPB;
    const int64_t r0 = (int64_t)blk * chunk;
    const int64_t r1 = (r0 + chunk < N) ? r0 + chunk : N;
    const int64_t run = chunk / G::RPB;
    const int64_t i0 = r0 + (int64_t)slot * run;
    const int64_t i1 = (i0 + run < r1) ? i0 + run : r1;
    int hh[G::NV];
#pragma unroll
    for (int v = 0; v < G::NV; ++v) hh[v] = (4 * (sub + G::TPR * v)) / C;
    const int elast = rowptr[N] - 1;
    auto ptrs = [&](int64_t r, int& b_, int& e_) {
        const int64_t rc = r < N ? r : N - 1;
        b_ = rowptr[rc];
        e_ = rowptr[rc + 1];
    };
    auto srcs = [&](int b_, int e_, int (&j)[EB]) {
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            int q = (b_ + e < e_) ? b_ + e : e_ - 1;
            q = q < elast ? q : elast;
            j[e] = elast >= 0 ? col[q > 0 ? q : 0] : 0;
        }
    };
    // the destination's own operands, one destination ahead
    float adn[G::NV], mn_[G::NV], dn_[G::NV];
    float4 gn[G::NV];
    auto own = [&](int64_t r) {
        const int64_t rc = r < N ? r : N - 1;
#pragma unroll
        for (int v = 0; v < G::NV; ++v) {
            adn[v] = a_dst[rc * HEADS + hh[v]];
            const float2 st = *reinterpret_cast<const float2*>(stats + (rc * HEADS + hh[v]) * 2);
            mn_[v] = st.x;
            dn_[v] = st.y;
            gn[v] = ld4(g + rc * G::HC + 4 * (sub + G::TPR * v));
        }
    };
    int beg, end, nbeg, nend, b2, e2;
    int jc[EB], jn[EB], cid[EB];
    int pfid = -1;
    float4 pf[G::NV];
    float pfa[G::NV];
#pragma unroll
    for (int v = 0; v < G::NV; ++v) { pf[v] = f4zero(); pfa[v] = 0.f; }
#pragma unroll
    for (int e = 0; e < EB; ++e) cid[e] = -1;
    ptrs(i0, beg, end);
    ptrs(i0 + 1, nbeg, nend);
    ptrs(i0 + 2, b2, e2);
    srcs(beg, end, jc);
    srcs(nbeg, nend, jn);
    own(i0);
    for (int64_t i = i0; i < i1; ++i) {
        int b3, e3, j2[EB];
        ptrs(i + 3, b3, e3);
        srcs(b2, e2, j2);
        int pid = -1;
        if (i + 1 < i1) {
#pragma unroll
            for (int e = EB - 1; e >= 0; --e) {
                bool have = nbeg + e >= nend;
#pragma unroll
                for (int k = 0; k < EB; ++k) have = have || (beg + k < end && jn[e] == jc[k]);
                if (!have) pid = jn[e];
            }
        }
        const bool any = beg < end;
        float ad[G::NV], m[G::NV], inv[G::NV], sada[G::NV], sal[G::NV], salk[G::NV];
        float4 gi[G::NV];
#pragma unroll
        for (int v = 0; v < G::NV; ++v) {
            ad[v] = adn[v];
            m[v] = mn_[v];
            inv[v] = 1.0f / dn_[v];
            gi[v] = gn[v];
            bsum[v] = add4(bsum[v], gi[v]);
            sada[v] = sal[v] = salk[v] = 0.f;
        }
        own(i + 1);
        for (int p = beg; p < end; p += EB) {
            float4 zz[EB][G::NV];
            float as_[EB][G::NV];
            const bool first = p == beg;
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                const int64_t j = jc[e];
                int where = -1;                              // -1: load, EB: the prefetched row, k: cache row k
                if (first) {
#pragma unroll
                    for (int k = 0; k < EB; ++k) where = (jc[e] == cid[k]) ? k : where;
                    where = (jc[e] == pfid) ? EB : where;
                }
                if (where < 0) {
                    if constexpr (THIN) {
                        float xk[8];
                        thin_read(j, xk);
                        thin_form(xk, zz[e]);
                    }
#pragma unroll
                    for (int v = 0; v < G::NV; ++v) {
                        if constexpr (!THIN) zz[e][v] = ld4(z + j * G::HC + 4 * (sub + G::TPR * v));
                        as_[e][v] = a_src[j * HEADS + hh[v]];
                    }
                } else if (where == EB) {
                    if constexpr (THIN) thin_form(pfx, zz[e]);
#pragma unroll
                    for (int v = 0; v < G::NV; ++v) {
                        if constexpr (!THIN) zz[e][v] = pf[v];
                        as_[e][v] = pfa[v];
                    }
                } else {
#pragma unroll
                    for (int v = 0; v < G::NV; ++v) {
                        zz[e][v] = mine[where * HC4 + sub + G::TPR * v];
                        as_[e][v] = lcache[(where * G::NV + v) * 256 + threadIdx.x];
                    }
                }
            }
            if (first && pid >= 0) {
#pragma unroll
                for (int v = 0; v < G::NV; ++v) {
                    if constexpr (!THIN) pf[v] = ld4(z + (int64_t)pid * G::HC + 4 * (sub + G::TPR * v));
                    pfa[v] = a_src[(int64_t)pid * HEADS + hh[v]];
                }
                if constexpr (THIN) thin_read(pid, pfx);
            }
            if (first) {
#pragma unroll
                for (int e = 0; e < EB; ++e) {
                    cid[e] = (p + e < end) ? jc[e] : -1;
#pragma unroll
                    for (int v = 0; v < G::NV; ++v) {
                        mine[e * HC4 + sub + G::TPR * v] = zz[e][v];
                        lcache[(e * G::NV + v) * 256 + threadIdx.x] = as_[e][v];
                    }
                }
            }
            const int pnext = p + EB;
            if (pnext < end) {
#pragma unroll
                for (int e = 0; e < EB; ++e) jc[e] = col[(pnext + e < end) ? pnext + e : pnext];
            }
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                if (e > 0 && p + e >= end) break;
#pragma unroll
                for (int v = 0; v < G::NV; ++v) {
                    const int c = 4 * (sub + G::TPR * v);
                    const float da = group_sum<G::LPH>(dot4(gi[v], zz[e][v]));
                    const float raw = as_[e][v] + ad[v];
                    const float s_ = raw > 0.f ? raw : ns * raw;
                    const float lk = raw > 0.f ? 1.0f : ns;
                    const float a = __expf(s_ - m[v]) * inv[v];
                    sada[v] = fmaf(a, da, sada[v]);
                    sal[v] = fmaf(a * lk, da, sal[v]);
                    salk[v] = fmaf(a, lk, salk[v]);
                    if (c % C == 0)
                        *reinterpret_cast<float2*>(escr + ((int64_t)(p + e) * HEADS + hh[v]) * 2) = make_float2(a, da);
                }
            }
        }
        if (!any) {
#pragma unroll
            for (int e = 0; e < EB; ++e) cid[e] = -1;
        }
#pragma unroll
        for (int v = 0; v < G::NV; ++v) {
            const int c = 4 * (sub + G::TPR * v);
            if (c % C == 0) {
                delta[i * HEADS + hh[v]] = sada[v];
                gad[i * HEADS + hh[v]] = sal[v] - sada[v] * salk[v];
            }
        }
        beg = nbeg; end = nend; nbeg = b2; nend = e2; b2 = b3; e2 = e3;
        pfid = any ? pid : -1;
#pragma unroll
        for (int e = 0; e < EB; ++e) { jc[e] = jn[e]; jn[e] = j2[e]; }
    }
    if (bias_partials) {                       // slots 0..RPB-1 in order
#pragma unroll
        for (int v = 0; v < G::NV; ++v) {
            __syncthreads();
            cache[threadIdx.x] = bsum[v];
            __syncthreads();
            if (threadIdx.x < G::TPR) {
                float4 a = cache[sub];
                for (int k = 1; k < G::RPB; ++k) a = add4(a, cache[k * G::TPR + sub]);
                st4(bias_partials + (int64_t)blk * G::HC + 4 * (sub + G::TPR * v), a);
            }
        }
    }
}

// source pass over the CSC: grad_z_j = sum_{e: j->i} alpha_e g_i ; grad_a_src[j,h] = sum_e ds_e.
// The forward's walk (r04) over the transposed index: a contiguous run of sources per lane group, the g rows (and
// a_dst / delta) of the previous source's destinations in a slot-private LDS image, the one row the next source adds
// requested a source ahead, the next source's per-edge (alpha, dalpha) pairs requested as soon as this one's are used, the
// index reads (destinations and CSR positions) two ahead.  (r03: one workgroup per RPB sources, one out-edge at a time:
// 823 us at cfg3 for 3.0 GB.)
template <int HEADS, int C>
__global__ __launch_bounds__(256, (HEADS * C >= 1024) ? 2 : 4) void gat_bwd_src_kernel(
    const float* __restrict__ g, const float* __restrict__ a_src, const float* __restrict__ a_dst,
    const float* __restrict__ escr, const float* __restrict__ delta,
    const int32_t* __restrict__ rowptr_t, const int32_t* __restrict__ col_t,
    const int32_t* __restrict__ pos_t, float* __restrict__ gz, float* __restrict__ gas, int64_t N,
    float ns, const float* __restrict__ att_src, const float* __restrict__ att_dst, const float* __restrict__ gad) {
    // att_src != NULL: the logits were formed from z inside the operator (gat_logits_kernel), so their gradient
    // flows back into grad_z here: + grad_a_src[j,h] att_src[h,:] + grad_a_dst[j,h] att_dst[h,:]
    using G = GatCfg<HEADS, C>;
    constexpr int EB = G::NV >= 4 ? 2 : 3;
    constexpr int HC4 = G::HC / 4;
    __shared__ float4 cache[G::RPB * EB * HC4];
    __shared__ float2 lcache[EB * G::NV * 256];          // (a_dst, delta) of the cached rows, [row][v][thread]
    const int sub = threadIdx.x % G::TPR;
    const int slot = G::TPR == 64 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (int)(threadIdx.x / G::TPR);
    float4* mine = cache + (size_t)slot * EB * HC4;
    const int blk = xcd_block(blockIdx.x, gridDim.x);
    const int64_t chunk = ((N + gridDim.x - 1) / gridDim.x + G::RPB - 1) / G::RPB * G::RPB;
    const int64_t r0 = (int64_t)blk * chunk;
    const int64_t r1 = (r0 + chunk < N) ? r0 + chunk : N;
    const int64_t run = chunk / G::RPB;
    const int64_t j0 = r0 + (int64_t)slot * run;
    const int64_t j1 = (j0 + run < r1) ? j0 + run : r1;
    int hh[G::NV];
#pragma unroll
    for (int v = 0; v < G::NV; ++v) hh[v] = (4 * (sub + G::TPR * v)) / C;
    const int elast = rowptr_t[N] - 1;
    auto ptrs = [&](int64_t r, int& b_, int& e_) {
        const int64_t rc = r < N ? r : N - 1;
        b_ = rowptr_t[rc];
        e_ = rowptr_t[rc + 1];
    };
    auto dsts = [&](int b_, int e_, int (&ii)[EB], int (&pp)[EB]) {
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            int q = (b_ + e < e_) ? b_ + e : e_ - 1;
            q = q < elast ? q : elast;
            q = q > 0 ? q : 0;
            ii[e] = elast >= 0 ? col_t[q] : 0;
            pp[e] = elast >= 0 ? pos_t[q] : 0;
        }
    };
    float asn[G::NV];                                   // the source's own logit, one source ahead
    float2 ec[EB][G::NV];                               // (alpha, dalpha) of the source's first EB out-edges
    auto edge_pairs = [&](const int (&pp)[EB]) {
#pragma unroll
        for (int e = 0; e < EB; ++e)
#pragma unroll
            for (int v = 0; v < G::NV; ++v)
                ec[e][v] = *reinterpret_cast<const float2*>(escr + ((int64_t)pp[e] * HEADS + hh[v]) * 2);
    };
    int beg, end, nbeg, nend, b2, e2;
    int ic[EB], in_[EB], pc[EB], pn[EB], cid[EB];
    int pfid = -1;
    float4 pf[G::NV];
    float2 pfa[G::NV];
#pragma unroll
    for (int v = 0; v < G::NV; ++v) {
        pf[v] = f4zero();
        pfa[v] = make_float2(0.f, 0.f);
        asn[v] = a_src[(j0 < N ? j0 : N - 1) * HEADS + hh[v]];
    }
#pragma unroll
    for (int e = 0; e < EB; ++e) cid[e] = -1;
    ptrs(j0, beg, end);
    ptrs(j0 + 1, nbeg, nend);
    ptrs(j0 + 2, b2, e2);
    dsts(beg, end, ic, pc);
    dsts(nbeg, nend, in_, pn);
    edge_pairs(pc);
    for (int64_t j = j0; j < j1; ++j) {
        int b3, e3, i2[EB], p2[EB];
        ptrs(j + 3, b3, e3);
        dsts(b2, e2, i2, p2);
        int pid = -1;
        if (j + 1 < j1) {
#pragma unroll
            for (int e = EB - 1; e >= 0; --e) {
                bool have = nbeg + e >= nend;
#pragma unroll
                for (int k = 0; k < EB; ++k) have = have || (beg + k < end && in_[e] == ic[k]);
                if (!have) pid = in_[e];
            }
        }
        const bool any = beg < end;
        float as[G::NV], sds[G::NV];
        float4 acc[G::NV];
#pragma unroll
        for (int v = 0; v < G::NV; ++v) {
            as[v] = asn[v];
            asn[v] = a_src[(j + 1 < N ? j + 1 : j) * HEADS + hh[v]];
            sds[v] = 0.f;
            acc[v] = f4zero();
        }
        for (int t = beg; t < end; t += EB) {
            float4 gg[EB][G::NV];
            float2 dd[EB][G::NV];                           // (a_dst, delta) of the edge's destination
            const bool first = t == beg;
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                const int64_t i = ic[e];
                int where = -1;
                if (first) {
#pragma unroll
                    for (int k = 0; k < EB; ++k) where = (ic[e] == cid[k]) ? k : where;
                    where = (ic[e] == pfid) ? EB : where;
                }
                if (where < 0) {
#pragma unroll
                    for (int v = 0; v < G::NV; ++v) {
                        gg[e][v] = ld4(g + i * G::HC + 4 * (sub + G::TPR * v));
                        dd[e][v] = make_float2(a_dst[i * HEADS + hh[v]], delta[i * HEADS + hh[v]]);
                    }
                } else if (where == EB) {
#pragma unroll
                    for (int v = 0; v < G::NV; ++v) { gg[e][v] = pf[v]; dd[e][v] = pfa[v]; }
                } else {
#pragma unroll
                    for (int v = 0; v < G::NV; ++v) {
                        gg[e][v] = mine[where * HC4 + sub + G::TPR * v];
                        dd[e][v] = lcache[(where * G::NV + v) * 256 + threadIdx.x];
                    }
                }
            }
            if (!first) edge_pairs(pc);                    // (later batches of a wide source: read when needed)
            if (first && pid >= 0) {
#pragma unroll
                for (int v = 0; v < G::NV; ++v) {
                    pf[v] = ld4(g + (int64_t)pid * G::HC + 4 * (sub + G::TPR * v));
                    pfa[v] = make_float2(a_dst[(int64_t)pid * HEADS + hh[v]], delta[(int64_t)pid * HEADS + hh[v]]);
                }
            }
            if (first) {
#pragma unroll
                for (int e = 0; e < EB; ++e) {
                    cid[e] = (t + e < end) ? ic[e] : -1;
#pragma unroll
                    for (int v = 0; v < G::NV; ++v) {
                        mine[e * HC4 + sub + G::TPR * v] = gg[e][v];
                        lcache[(e * G::NV + v) * 256 + threadIdx.x] = dd[e][v];
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                if (e > 0 && t + e >= end) break;
#pragma unroll
                for (int v = 0; v < G::NV; ++v) {
                    const float a = ec[e][v].x;
                    const float da = ec[e][v].y;
                    const float raw = as[v] + dd[e][v].x;
                    const float lk = raw > 0.f ? 1.0f : ns;
                    sds[v] += a * (da - dd[e][v].y) * lk;
                    acc[v] = fma4(a, gg[e][v], acc[v]);
                }
            }
            const int tnext = t + EB;
            if (tnext < end) {
#pragma unroll
                for (int e = 0; e < EB; ++e) {
                    const int q = (tnext + e < end) ? tnext + e : tnext;
                    ic[e] = col_t[q];
                    pc[e] = pos_t[q];
                }
            }
        }
        if (!any) {
#pragma unroll
            for (int e = 0; e < EB; ++e) cid[e] = -1;
        }
        if (j + 1 < j1) edge_pairs(pn);                    // the next source's pairs: this one's have been used
#pragma unroll
        for (int v = 0; v < G::NV; ++v) {
            const int c = 4 * (sub + G::TPR * v);
            float4 o = acc[v];
            if (att_src) {
                o = fma4(sds[v], ld4(att_src + c), o);
                o = fma4(gad[j * HEADS + hh[v]], ld4(att_dst + c), o);
            }
            st4(gz + j * G::HC + c, o);
            if (c % C == 0) gas[j * HEADS + hh[v]] = sds[v];
        }
        beg = nbeg; end = nend; nbeg = b2; nend = e2; b2 = b3; e2 = e3;
        pfid = any ? pid : -1;
#pragma unroll
        for (int e = 0; e < EB; ++e) { ic[e] = in_[e]; in_[e] = i2[e]; pc[e] = pn[e]; pn[e] = p2[e]; }
    }
}

// grad of the attention vectors: gatt_src[h,c] = sum_n grad_a_src[n,h] z[n,h,c] (gatt_dst with grad_a_dst): per-workgroup
// column partials [blk][2][HC], summed in a fixed order by gat_att_grad_final_kernel.
template <int HEADS, int C>
__global__ __launch_bounds__(256) void gat_att_grad_kernel(const float* __restrict__ z, const float* __restrict__ gas,
                                                           const float* __restrict__ gad, float* __restrict__ partials,
                                                           int64_t N) {
    using G = GatCfg<HEADS, C>;
    __shared__ float4 red1[256];
    __shared__ float4 red2[256];
    const int sub = threadIdx.x % G::TPR;
    const int slot = threadIdx.x / G::TPR;
    const int64_t chunk = ((N + gridDim.x - 1) / gridDim.x + G::RPB - 1) / G::RPB * G::RPB;
    const int64_t r0 = (int64_t)blockIdx.x * chunk;
    const int64_t r1 = (r0 + chunk < N) ? r0 + chunk : N;
    float4 s1[G::NV], s2[G::NV];
#pragma unroll
    for (int v = 0; v < G::NV; ++v) { s1[v] = f4zero(); s2[v] = f4zero(); }
    for (int64_t i = r0 + slot; i < r1; i += G::RPB) {
#pragma unroll
        for (int v = 0; v < G::NV; ++v) {
            const int c = 4 * (sub + G::TPR * v);
            const float4 zv = ld4(z + i * G::HC + c);
            s1[v] = fma4(gas[i * HEADS + c / C], zv, s1[v]);
            s2[v] = fma4(gad[i * HEADS + c / C], zv, s2[v]);
        }
    }
#pragma unroll
    for (int v = 0; v < G::NV; ++v) {
        __syncthreads();
        red1[threadIdx.x] = s1[v];
        red2[threadIdx.x] = s2[v];
        __syncthreads();
        if (slot == 0) {
            float4 a = red1[sub], b = red2[sub];
            for (int k = 1; k < G::RPB; ++k) { a = add4(a, red1[k * G::TPR + sub]); b = add4(b, red2[k * G::TPR + sub]); }
            const int c = 4 * (sub + G::TPR * v);
            st4(partials + ((int64_t)blockIdx.x * 2) * G::HC + c, a);
            st4(partials + ((int64_t)blockIdx.x * 2 + 1) * G::HC + c, b);
        }
    }
}

__global__ void gat_bias_final_kernel(const float* __restrict__ partials, int nblk, int HC, float* __restrict__ gbias) {
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);      // one wave per column
    if (c >= HC) return;
    const int lane = threadIdx.x & 63;
    double a = 0.0;
    for (int k = lane; k < nblk; k += 64) a += (double)partials[(int64_t)k * HC + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) gbias[c] = (float)a;
}

__global__ void gat_att_grad_final_kernel(const float* __restrict__ partials, int nblk, int HC, float* __restrict__ gsrc,
                                          float* __restrict__ gdst) {
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);      // one wave per column
    if (c >= HC) return;
    const int lane = threadIdx.x & 63;
    double a = 0.0, b = 0.0;
    for (int k = lane; k < nblk; k += 64) {
        a += (double)partials[((int64_t)k * 2) * HC + c];
        b += (double)partials[((int64_t)k * 2 + 1) * HC + c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if (lane == 0) { gsrc[c] = (float)a; gdst[c] = (float)b; }
}

}  // namespace qot

using namespace qot;

#define QOT_DISPATCH_GAT(heads, C, ...)                                    \
    if ((heads) != 4) return QOT_ERR_UNSUPPORTED;                          \
    switch (C) {                                                           \
        case 4:   { constexpr int kC = 4;   __VA_ARGS__; } break;          \
        case 8:   { constexpr int kC = 8;   __VA_ARGS__; } break;          \
        case 16:  { constexpr int kC = 16;  __VA_ARGS__; } break;          \
        case 32:  { constexpr int kC = 32;  __VA_ARGS__; } break;          \
        case 64:  { constexpr int kC = 64;  __VA_ARGS__; } break;          \
        case 128: { constexpr int kC = 128; __VA_ARGS__; } break;          \
        case 256: { constexpr int kC = 256; __VA_ARGS__; } break;          \
        default: return QOT_ERR_UNSUPPORTED;                               \
    }

// Workgroups of the chunked kernels (forward, att gradient) = rows of their partials buffers: ONE resident round
// (what the occupancy of the forward kernel admits per CU x CUs; a second, partly filled round cost 30 %), at most 2048.
template <int C>
static int gat_blocks_for(int64_t N) {
    using G = GatCfg<4, C>;
    static int per_cu = 0;
    if (!per_cu) {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gat_fwd_kernel<4, C>, 256, 0) != hipSuccess || occ <= 0) occ = 4;
        per_cu = occ > 8 ? 8 : occ;
    }
    int64_t b = (N + G::RPB - 1) / G::RPB;
    int64_t cap = (int64_t)per_cu * num_cus();
    if (cap > 2048) cap = 2048;
    return (int)(b < cap ? (b > 0 ? b : 1) : cap);
}

// workgroups of a backward walk: one resident round of its kernel (occupancy x CUs), no more than there are row groups
template <int C>
static int gat_walk_blocks(const void* kernel, int64_t N, size_t dyn_lds = 0) {
    using G = GatCfg<4, C>;
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, 256, dyn_lds) != hipSuccess || occ <= 0) occ = 2;
    int64_t b = (N + G::RPB - 1) / G::RPB;
    const int64_t cap = (int64_t)(occ > 8 ? 8 : occ) * num_cus();
    return (int)(b < cap ? (b > 0 ? b : 1) : cap);
}

extern "C" size_t qot_gat_bn_partials_floats(int64_t N, int heads, int C) {
    return (size_t)2048 * 2 * (size_t)(heads * C);
}

// rows per workgroup of qot_gat_fwd's chunked walk (workgroup b holds rows [b * chunk, min((b + 1) * chunk, N)))
extern "C" int64_t qot_gat_chunk_rows(int64_t N, int heads, int C) {
    int64_t grid = 0;
    QOT_DISPATCH_GAT(heads, C, { grid = gat_blocks_for<kC>(N); using G = GatCfg<4, kC>;
                                 return ((N + grid - 1) / grid + G::RPB - 1) / G::RPB * G::RPB; });
    return 0;
}

// number of workgroups qot_gat_fwd / qot_gat_att_grad launch = rows of their partials buffers
extern "C" int qot_gat_blocks(int64_t N, int heads, int C) {
    QOT_DISPATCH_GAT(heads, C, { return gat_blocks_for<kC>(N); });
    return 0;
}

extern "C" int qot_gat_logits(const float* z, const float* att_src, const float* att_dst, float* a_src, float* a_dst,
                              int64_t N, int heads, int C, qot_stream_t stream) {
    if (N < 0) return QOT_ERR_BADARG;
    if (N == 0) return QOT_OK;
    if (!z || !att_src || !att_dst || !a_src || !a_dst) return QOT_ERR_BADARG;
    QOT_DISPATCH_GAT(heads, C, {
        using G = GatCfg<4, kC>;
        gat_logits_kernel<4, kC><<<grid_for(N, G::RPB), 256, 0, (hipStream_t)stream>>>(z, att_src, att_dst, a_src, a_dst, N);
    });
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_gat_fwd(const float* z, const float* a_src, const float* a_dst, const float* bias,
                           const int32_t* rowptr, const int32_t* col, float* out, float* stats,
                           int64_t N, int heads, int C, float neg_slope, float* bn_partials,
                           qot_stream_t stream) {
    if (N < 0 || !rowptr) return QOT_ERR_BADARG;
    if (N == 0) return QOT_OK;
    if (!z || !a_src || !a_dst || !bias || !col || !out || !stats) return QOT_ERR_BADARG;
    if ((reinterpret_cast<uintptr_t>(stats) & 7) || ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(out)) & 15))
        return QOT_ERR_BADARG;
    QOT_DISPATCH_GAT(heads, C, {
        using G = GatCfg<4, kC>;
        const int nblk = gat_blocks_for<kC>(N);
        gat_fwd_kernel<4, kC><<<nblk, 256, 0, (hipStream_t)stream>>>(
            z, a_src, a_dst, bias, rowptr, col, out, stats, N, neg_slope, bn_partials);
    });
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

// The thin forms: z = x W^T (x [N, K], K <= 8 input features; W [heads*C, K] = GATConv.lin.weight) is formed inside the
// kernels instead of being read; same outputs, same partials layout (qot_gat_blocks workgroups) as the dense forms.
// QOT_ERR_UNSUPPORTED when K is not in 1..8 or W^T does not fit next to the walk's LDS image (64 KB per workgroup).
static bool gat_thin_fits(int heads, int C, int K, size_t* dyn) {
    if (K < 1 || K > 8 || heads != 4) return false;
    const size_t HC = (size_t)heads * C;
    const int EB = HC >= 1024 ? 2 : 3;
    const size_t rpb = 256 / (HC / 4 < 64 ? HC / 4 : 64);
    const size_t fixed = rpb * EB * HC * 4 + (size_t)EB * (HC >= 1024 ? 4 : (HC >= 512 ? 2 : 1)) * 256 * 4;
    *dyn = (size_t)K * HC * 4;
    return fixed + *dyn <= 64 * 1024;
}

extern "C" int qot_gat_thin_supported(int heads, int C, int K) {
    size_t dyn = 0;
    return gat_thin_fits(heads, C, K, &dyn) ? 1 : 0;
}

extern "C" int qot_gat_fwd_thin(const float* x, int K, const float* w, const float* a_src, const float* a_dst,
                                const float* bias, const int32_t* rowptr, const int32_t* col, float* out, float* stats,
                                int64_t N, int heads, int C, float neg_slope, float* bn_partials, qot_stream_t stream) {
    if (N < 0 || !rowptr) return QOT_ERR_BADARG;
    size_t dyn = 0;
    if (!gat_thin_fits(heads, C, K, &dyn)) return QOT_ERR_UNSUPPORTED;
    if (N == 0) return QOT_OK;
    if (!x || !w || !a_src || !a_dst || !bias || !col || !out || !stats) return QOT_ERR_BADARG;
    if ((reinterpret_cast<uintptr_t>(stats) & 7) || (reinterpret_cast<uintptr_t>(out) & 15)) return QOT_ERR_BADARG;
    QOT_DISPATCH_GAT(heads, C, {
        const int nblk = gat_blocks_for<kC>(N);
        gat_fwd_kernel<4, kC, true><<<nblk, 256, dyn, (hipStream_t)stream>>>(
            nullptr, a_src, a_dst, bias, rowptr, col, out, stats, N, neg_slope, bn_partials, x, w, K);
    });
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_gat_bwd_dst(const float* grad_out, const float* z, const float* a_src,
                               const float* a_dst, const float* stats, const int32_t* rowptr,
                               const int32_t* col, float* grad_a_dst, float* escr, float* delta,
                               int64_t N, int heads, int C, float neg_slope, float* grad_bias, float* workspace,
                               qot_stream_t stream) {
    if (N < 0 || !rowptr) return QOT_ERR_BADARG;
    if (grad_bias && !workspace) return QOT_ERR_BADARG;
    if (N == 0) {
        if (grad_bias && heads > 0 && C > 0) QOT_HIP(hipMemsetAsync(grad_bias, 0, sizeof(float) * heads * C, (hipStream_t)stream));
        return QOT_OK;
    }
    if (!grad_out || !z || !a_src || !a_dst || !stats || !col || !grad_a_dst || !escr || !delta)
        return QOT_ERR_BADARG;
    if (((reinterpret_cast<uintptr_t>(stats) | reinterpret_cast<uintptr_t>(escr)) & 7) ||
        ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(grad_out)) & 15))
        return QOT_ERR_BADARG;
    QOT_DISPATCH_GAT(heads, C, {
        using G = GatCfg<4, kC>;
        const int nblk = gat_walk_blocks<kC>(reinterpret_cast<const void*>(gat_bwd_dst_kernel<4, kC>), N);
        gat_bwd_dst_kernel<4, kC><<<nblk, 256, 0, (hipStream_t)stream>>>(
            grad_out, z, a_src, a_dst, stats, rowptr, col, grad_a_dst, escr, delta, N, neg_slope,
            grad_bias ? workspace : nullptr);
        if (grad_bias) {
            QOT_LAUNCH_CHECK();
            gat_bias_final_kernel<<<grid_for(G::HC, 4), 256, 0, (hipStream_t)stream>>>(workspace, nblk, G::HC, grad_bias);
        }
    });
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_gat_bwd_dst_thin(const float* grad_out, const float* x, int K, const float* w, const float* a_src,
                                    const float* a_dst, const float* stats, const int32_t* rowptr, const int32_t* col,
                                    float* grad_a_dst, float* escr, float* delta, int64_t N, int heads, int C,
                                    float neg_slope, float* grad_bias, float* workspace, qot_stream_t stream) {
    if (N < 0 || !rowptr) return QOT_ERR_BADARG;
    if (grad_bias && !workspace) return QOT_ERR_BADARG;
    size_t dyn = 0;
    if (!gat_thin_fits(heads, C, K, &dyn)) return QOT_ERR_UNSUPPORTED;
    if (N == 0) {
        if (grad_bias && heads > 0 && C > 0) QOT_HIP(hipMemsetAsync(grad_bias, 0, sizeof(float) * heads * C, (hipStream_t)stream));
        return QOT_OK;
    }
    if (!grad_out || !x || !w || !a_src || !a_dst || !stats || !col || !grad_a_dst || !escr || !delta) return QOT_ERR_BADARG;
    if (((reinterpret_cast<uintptr_t>(stats) | reinterpret_cast<uintptr_t>(escr)) & 7) ||
        (reinterpret_cast<uintptr_t>(grad_out) & 15))
        return QOT_ERR_BADARG;
    QOT_DISPATCH_GAT(heads, C, {
        using G = GatCfg<4, kC>;
        const int nblk = gat_walk_blocks<kC>(reinterpret_cast<const void*>(gat_bwd_dst_kernel<4, kC, true>), N, dyn);
        gat_bwd_dst_kernel<4, kC, true><<<nblk, 256, dyn, (hipStream_t)stream>>>(
            grad_out, nullptr, a_src, a_dst, stats, rowptr, col, grad_a_dst, escr, delta, N, neg_slope,
            grad_bias ? workspace : nullptr, x, w, K);
        if (grad_bias) {
            QOT_LAUNCH_CHECK();
            gat_bias_final_kernel<<<grid_for(G::HC, 4), 256, 0, (hipStream_t)stream>>>(workspace, nblk, G::HC, grad_bias);
        }
    });
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_gat_bwd_src(const float* grad_out, const float* a_src, const float* a_dst,
                               const float* escr, const float* delta, const int32_t* rowptr_t,
                               const int32_t* col_t, const int32_t* pos_t, float* grad_z,
                               float* grad_a_src, int64_t N, int heads, int C, float neg_slope,
                               const float* att_src, const float* att_dst, const float* grad_a_dst,
                               qot_stream_t stream) {
    if (N < 0 || !rowptr_t) return QOT_ERR_BADARG;
    if (N == 0) return QOT_OK;
    if (!grad_out || !a_src || !a_dst || !escr || !delta || !col_t || !pos_t || !grad_z || !grad_a_src)
        return QOT_ERR_BADARG;
    if (att_src && (!att_dst || !grad_a_dst)) return QOT_ERR_BADARG;
    if ((reinterpret_cast<uintptr_t>(escr) & 7) || ((reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(grad_z)) & 15))
        return QOT_ERR_BADARG;
    QOT_DISPATCH_GAT(heads, C, {
        using G = GatCfg<4, kC>;
        gat_bwd_src_kernel<4, kC><<<gat_walk_blocks<kC>(reinterpret_cast<const void*>(gat_bwd_src_kernel<4, kC>), N), 256, 0,
                                    (hipStream_t)stream>>>(
            grad_out, a_src, a_dst, escr, delta, rowptr_t, col_t, pos_t, grad_z, grad_a_src, N,
            neg_slope, att_src, att_dst, grad_a_dst);
    });
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

// grad of att_src / att_dst ([heads*C] each) from z and the logit gradients; workspace: qot_gat_bn_partials_floats.
extern "C" int qot_gat_att_grad(const float* z, const float* grad_a_src, const float* grad_a_dst, float* grad_att_src,
                                float* grad_att_dst, float* workspace, int64_t N, int heads, int C, qot_stream_t stream) {
    if (N < 0) return QOT_ERR_BADARG;
    if (!grad_att_src || !grad_att_dst || !workspace) return QOT_ERR_BADARG;
    if (N > 0 && (!z || !grad_a_src || !grad_a_dst)) return QOT_ERR_BADARG;
    QOT_DISPATCH_GAT(heads, C, {
        using G = GatCfg<4, kC>;
        const int nblk = gat_blocks_for<kC>(N);
        gat_att_grad_kernel<4, kC><<<nblk, 256, 0, (hipStream_t)stream>>>(z, grad_a_src, grad_a_dst, workspace, N);
        QOT_LAUNCH_CHECK();
        gat_att_grad_final_kernel<<<grid_for(G::HC, 4), 256, 0, (hipStream_t)stream>>>(workspace, nblk, G::HC, grad_att_src, grad_att_dst);
    });
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}
