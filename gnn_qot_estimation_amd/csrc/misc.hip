// Embedding lookup, leaky_relu+dropout, global mean pool, BatchNorm(+ReLU), LUT row
// gather/scatter.  Reference sites: topological_training/models.py:12,51-52 (Embedding),
// :54-55,58-59 (leaky_relu + Dropout), :61 ([PyG-ext] global_mean_pool);
// lightpath_training/models.py:14,31-32 ([PyG-ext] BatchNorm + relu), :39-40 (x[lut_mask]).
// All HBM-bound: 16-B accesses per lane, rows contiguous across a lane group.
#include "common.hpp"

namespace qot {

// ----------------------------------------------------------------------------- rows
__global__ void rows_gather_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                   float* __restrict__ out, int64_t n, int C4) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= n * C4) return;
    int64_t r = t / C4;
    int c = (int)(t - r * C4) * 4;
    st4(out + r * C4 * 4 + c, ld4(src + (int64_t)idx[r] * C4 * 4 + c));
}

// Embedding lookup: as rows_gather, but ids outside [0, V) are clamped so that a bad id can never read past the
// table (the Python layer raises IndexError like nn.Embedding before it gets here; this is the memory-safety net
// for direct C-ABI callers).
__device__ __forceinline__ int clamp_id(int id, int V) { return id < 0 ? 0 : (id >= V ? V - 1 : id); }

__global__ void embed_gather_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                    float* __restrict__ out, int64_t n, int C4, int V) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= n * C4) return;
    int64_t r = t / C4;
    int c = (int)(t - r * C4) * 4;
    st4(out + r * C4 * 4 + c, ld4(src + (int64_t)clamp_id(idx[r], V) * C4 * 4 + c));
}

__global__ void rows_scatter_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                    float* __restrict__ out, int64_t n, int C4) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= n * C4) return;
    int64_t r = t / C4;
    int c = (int)(t - r * C4) * 4;
    st4(out + (int64_t)idx[r] * C4 * 4 + c, ld4(src + r * C4 * 4 + c));
}

// Embedding backward.  ids repeat in every graph (node_ids is NOT offset by collate), so
// thousands of rows hit the same V table rows: pre-reduce a block's rows in an LDS copy of
// the table, then one float atomic per touched table element per block.
template <bool LDS_TABLE>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ g,
                                                        const int32_t* __restrict__ ids,
                                                        float* __restrict__ gt, int64_t N, int V, int H,
                                                        int rows_per_block) {
    extern __shared__ float tab[];
    const int VH = V * H;
    if (LDS_TABLE) {
        for (int t = threadIdx.x; t < VH; t += blockDim.x) tab[t] = 0.f;
        __syncthreads();
    }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < N) ? r0 + rows_per_block : N;
    const int64_t total = (r1 - r0) * H;
    for (int64_t t = threadIdx.x; t < total; t += blockDim.x) {
        int64_t r = r0 + t / H;
        int c = (int)(t % H);
        float val = g[r * H + c];
        const int id = clamp_id(ids[r], V);
        if (LDS_TABLE) atomicAdd(&tab[id * H + c], val);
        else atomicAdd(&gt[(int64_t)id * H + c], val);
    }
    if (LDS_TABLE) {
        __syncthreads();
        for (int t = threadIdx.x; t < VH; t += blockDim.x) {
            float val = tab[t];
            if (val != 0.f) atomicAdd(&gt[t], val);
        }
    }
}

// ----------------------------------------------------------------------------- activation
template <bool BWD>
__global__ void act_kernel(const float* __restrict__ a, const float* __restrict__ yref,
                           float* __restrict__ o, int64_t n4, float slope, uint32_t thr16,
                           float keep_scale, uint64_t seed, const int64_t* __restrict__ step_counter) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= n4) return;
    float4 in = ld4(a + 4 * t);
    float vi[4] = {in.x, in.y, in.z, in.w};
    float vr[4];
    if (BWD) { float4 r = ld4(yref + 4 * t); vr[0] = r.x; vr[1] = r.y; vr[2] = r.z; vr[3] = r.w; }
    uint64_t z = 0;
    if (thr16) z = act_hash64(seed, (uint64_t)step_counter[0], (uint64_t)t);
    float vo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        bool keep = thr16 ? (((uint32_t)(z >> (16 * c)) & 0xFFFFu) >= thr16) : true;
        float ks = keep ? keep_scale : 0.f;
        if (BWD) vo[c] = vi[c] * ks * (vr[c] > 0.f ? 1.0f : slope);
        else     vo[c] = (vi[c] > 0.f ? vi[c] : slope * vi[c]) * ks;
    }
    st4(o + 4 * t, make_float4(vo[0], vo[1], vo[2], vo[3]));
}

// ----------------------------------------------------------------------------- pool
__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ x,
                                                       const int32_t* __restrict__ ptr,
                                                       float* __restrict__ out, int H) {
    __shared__ float4 red[256];
    const int TPR = H / 4;
    const int RPB = 256 / TPR;
    const int sub = threadIdx.x % TPR;
    const int slot = threadIdx.x / TPR;
    const int64_t b = blockIdx.x;
    const int beg = ptr[b], end = ptr[b + 1];
    float4 acc = f4zero();
    if (slot < RPB)
        for (int64_t r = beg + slot; r < end; r += RPB) acc = add4(acc, ld4(x + r * H + 4 * sub));
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < TPR) {
        float4 s = red[threadIdx.x];
        for (int k = 1; k < RPB; ++k) s = add4(s, red[k * TPR + threadIdx.x]);
        int cnt = end - beg;
        st4(out + b * H + 4 * threadIdx.x, scale4(1.0f / (float)(cnt > 1 ? cnt : 1), s));
    }
}

__global__ void pool_bwd_kernel(const float* __restrict__ go, const int32_t* __restrict__ ptr,
                                const int32_t* __restrict__ batch, float* __restrict__ gx, int64_t N,
                                int H4) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= N * H4) return;
    int64_t r = t / H4;
    int c = (int)(t - r * H4) * 4;
    int b = batch[r];
    int cnt = ptr[b + 1] - ptr[b];
    st4(gx + r * H4 * 4 + c, scale4(1.0f / (float)(cnt > 1 ? cnt : 1), ld4(go + (int64_t)b * H4 * 4 + c)));
}

// ----------------------------------------------------------------------------- batch norm
// Column statistics in two stages: each block reduces a contiguous row chunk into
// partials[blk, 2, C]; a one-block finalize combines the partials in fp64.  Sums are taken
// on x - x[0,:] (shifted-data form) so E[x^2]-E[x]^2 cancellation stays benign in fp32.
constexpr int kBnMaxBlocks = 2048;
inline int bn_blocks(int64_t N) {
    int64_t b = (N + 63) / 64;
    if (b < 1) b = 1;
    return (int)(b > kBnMaxBlocks ? kBnMaxBlocks : b);
}

// MODE 0: s1 = sum(x - shift), s2 = sum((x-shift)^2)
// MODE 1: s1 = sum(dy'), s2 = sum(dy' * xhat)   (dy' = dy * [y > 0] when relu)
template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ y,
    const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ partials,
    int64_t N, int C, int relu, const float* __restrict__ wgt, const float* __restrict__ bia,
    const int32_t* __restrict__ rows = nullptr) {
    // rows != NULL (MODE 1): dy is [N, C] compact and its row r belongs to row rows[r] of x (qot_bn_bwd_reduce_rows)
    // relu with y == NULL: the mask is recomputed from x with the forward's own expression (bn_apply_kernel) -- one
    // [N, C] read fewer than taking it from the saved output
    __shared__ float4 r1[256];
    __shared__ float4 r2[256];
    const int C4 = C / 4;
    const int RPB = 256 / C4;
    const int sub = threadIdx.x % C4;
    const int slot = threadIdx.x / C4;
    const int nblk = gridDim.x;
    const int64_t chunk = (N + nblk - 1) / nblk;
    const int64_t r0 = (int64_t)blockIdx.x * chunk;
    const int64_t r1e = (r0 + chunk < N) ? r0 + chunk : N;
    float4 s1 = f4zero(), s2 = f4zero();
    if (slot < RPB) {
        const int c = 4 * sub;
        float4 sh, mu, rs, wv = f4zero(), bv = f4zero();
        if (MODE == 0) sh = ld4(x + c);
        else { mu = ld4(mean + c); rs = ld4(rstd + c); if (relu && !y) { wv = ld4(wgt + c); bv = ld4(bia + c); } }
        for (int64_t r = r0 + slot; r < r1e; r += RPB) {
            float4 xv = ld4(x + ((MODE == 1 && rows) ? (int64_t)rows[r] : r) * C + c);
            if (MODE == 0) {
                float4 d = sub4(xv, sh);
                s1 = add4(s1, d);
                s2 = make_float4(fmaf(d.x, d.x, s2.x), fmaf(d.y, d.y, s2.y), fmaf(d.z, d.z, s2.z), fmaf(d.w, d.w, s2.w));
            } else {
                float4 g = ld4(dy + r * C + c);
                float4 xh = make_float4((xv.x - mu.x) * rs.x, (xv.y - mu.y) * rs.y, (xv.z - mu.z) * rs.z, (xv.w - mu.w) * rs.w);
                if (relu) {
                    float4 yv = y ? ld4(y + r * C + c)
                                  : make_float4(fmaf(xh.x, wv.x, bv.x), fmaf(xh.y, wv.y, bv.y), fmaf(xh.z, wv.z, bv.z), fmaf(xh.w, wv.w, bv.w));
                    g = make_float4(yv.x > 0.f ? g.x : 0.f, yv.y > 0.f ? g.y : 0.f, yv.z > 0.f ? g.z : 0.f, yv.w > 0.f ? g.w : 0.f);
                }
                s1 = add4(s1, g);
                s2 = make_float4(fmaf(g.x, xh.x, s2.x), fmaf(g.y, xh.y, s2.y), fmaf(g.z, xh.z, s2.z), fmaf(g.w, xh.w, s2.w));
            }
        }
    }
    r1[threadIdx.x] = s1;
    r2[threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < C4) {
        float4 a = r1[threadIdx.x], b = r2[threadIdx.x];
        for (int k = 1; k < RPB; ++k) { a = add4(a, r1[k * C4 + threadIdx.x]); b = add4(b, r2[k * C4 + threadIdx.x]); }
        st4(partials + ((int64_t)blockIdx.x * 2) * C + 4 * threadIdx.x, a);
        st4(partials + ((int64_t)blockIdx.x * 2 + 1) * C + 4 * threadIdx.x, b);
    }
}

// one wave per column: lanes stride over the per-block partials (fp64), fixed shuffle tree
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ void bn_finalize_stats_kernel(const float* __restrict__ x, const float* __restrict__ partials,
                                         int nblk, int64_t N, int C, float eps, float momentum,
                                         float* __restrict__ mean, float* __restrict__ rstd,
                                         float* __restrict__ rmean, float* __restrict__ rvar) {
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= C) return;
    const int lane = threadIdx.x & 63;
    double s1 = 0.0, s2 = 0.0;
    for (int b = lane; b < nblk; b += 64) {
        s1 += (double)partials[((int64_t)b * 2) * C + c];
        s2 += (double)partials[((int64_t)b * 2 + 1) * C + c];
    }
    s1 = wave_sum_f64(s1);
    s2 = wave_sum_f64(s2);
    if (lane) return;
    double dn = (double)N;
    double m1 = s1 / dn;
    double var = s2 / dn - m1 * m1;
    if (var < 0.0) var = 0.0;
    double mu = (double)x[c] + m1;
    mean[c] = (float)mu;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) {
        double unb = (N > 1) ? var * dn / (dn - 1.0) : var;
        rmean[c] = (float)((1.0 - momentum) * (double)rmean[c] + momentum * mu);
        rvar[c] = (float)((1.0 - momentum) * (double)rvar[c] + momentum * unb);
    }
}

// Statistics from per-workgroup (mean, M2) pairs of x - shift (GATConv's epilogue): workgroup b holds rows
// [b * chunk, min((b + 1) * chunk, N)).  mean = sum n_b mean_b / N; M2 = sum (M2_b + n_b (mean_b - mean)^2) (Chan's
// formula for many groups at once), both in fp64 with a fixed lane-strided order: no E[d^2] - E[d]^2 cancellation.
__global__ void bn_finalize_chan_kernel(const float* __restrict__ shift, const float* __restrict__ partials, int nblk,
                                        int64_t chunk, int64_t N, int C, float eps, float momentum,
                                        float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ rmean,
                                        float* __restrict__ rvar) {
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= C) return;
    const int lane = threadIdx.x & 63;
    auto rows_of = [&](int b) -> double {
        const int64_t r0 = (int64_t)b * chunk;
        const int64_t r1 = (r0 + chunk < N) ? r0 + chunk : N;
        return r1 > r0 ? (double)(r1 - r0) : 0.0;
    };
    double s = 0.0;
    for (int b = lane; b < nblk; b += 64) s += rows_of(b) * (double)partials[((int64_t)b * 2) * C + c];
    const double mu = wave_sum_f64(s) / (double)N;
    double q = 0.0;
    for (int b = lane; b < nblk; b += 64) {
        const double nb = rows_of(b);
        if (nb > 0.0) {
            const double d = (double)partials[((int64_t)b * 2) * C + c] - mu;
            q += (double)partials[((int64_t)b * 2 + 1) * C + c] + nb * d * d;
        }
    }
    q = wave_sum_f64(q);
    if (lane) return;
    const double dn = (double)N;
    const double var = q / dn;
    const double m = (double)shift[c] + mu;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) {
        const double unb = (N > 1) ? var * dn / (dn - 1.0) : var;
        rmean[c] = (float)((1.0 - momentum) * (double)rmean[c] + momentum * m);
        rvar[c] = (float)((1.0 - momentum) * (double)rvar[c] + momentum * unb);
    }
}

__global__ void bn_finalize_bwd_kernel(const float* __restrict__ partials, int nblk, int C,
                                       float* __restrict__ gw, float* __restrict__ gb) {
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= C) return;
    const int lane = threadIdx.x & 63;
    double s1 = 0.0, s2 = 0.0;
    for (int b = lane; b < nblk; b += 64) {
        s1 += (double)partials[((int64_t)b * 2) * C + c];
        s2 += (double)partials[((int64_t)b * 2 + 1) * C + c];
    }
    s1 = wave_sum_f64(s1);
    s2 = wave_sum_f64(s2);
    if (lane == 0) { gb[c] = (float)s1; gw[c] = (float)s2; }
}

__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                const float* __restrict__ rstd, const float* __restrict__ w,
                                const float* __restrict__ b, float* __restrict__ y, int64_t N, int C4,
                                int relu, const int32_t* __restrict__ rows = nullptr) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= N * C4) return;
    int c = (int)(t % C4) * 4;
    const int64_t xt = rows ? (int64_t)rows[t / C4] * C4 + t % C4 : t;     // rows: y is compact, its row r = row rows[r] of x
    float4 xv = ld4(x + 4 * xt), mu = ld4(mean + c), rs = ld4(rstd + c), wv = ld4(w + c), bv = ld4(b + c);
    float4 o = make_float4(fmaf((xv.x - mu.x) * rs.x, wv.x, bv.x), fmaf((xv.y - mu.y) * rs.y, wv.y, bv.y),
                           fmaf((xv.z - mu.z) * rs.z, wv.z, bv.z), fmaf((xv.w - mu.w) * rs.w, wv.w, bv.w));
    if (relu) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
    st4(y + 4 * t, o);
}

__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                    const float* __restrict__ x, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, const float* __restrict__ w,
                                    const float* __restrict__ gw, const float* __restrict__ gb,
                                    float* __restrict__ gx, int64_t N, int C4, int relu, int batch_stats,
                                    const float* __restrict__ bia, const int32_t* __restrict__ rows = nullptr,
                                    int64_t n_total = 0) {
    // rows form (qot_bn_bwd_apply_rows): dy is compact [N, C] and belongs to rows rows[r] of x / gx, the batch has
    // n_total rows; dy == NULL: the rows whose dy is zero (same expression, so both forms give the dense form's bits)
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= N * C4) return;
    int c = (int)(t % C4) * 4;
    const int64_t xt = rows ? (int64_t)rows[t / C4] * C4 + t % C4 : t;
    if (!rows) n_total = N;
    float4 g = dy ? ld4(dy + 4 * t) : f4zero();
    float4 rs = ld4(rstd + c), wv = ld4(w + c);
    float4 xv = f4zero(), mu = f4zero();
    if (batch_stats || (relu && !y)) { xv = ld4(x + 4 * xt); mu = ld4(mean + c); }
    if (relu && dy) {
        float4 yv;
        if (y) yv = ld4(y + 4 * t);
        else {      // the forward's own expression (bn_apply_kernel): same mask, one [N, C] read fewer
            const float4 bv = ld4(bia + c);
            yv = make_float4(fmaf((xv.x - mu.x) * rs.x, wv.x, bv.x), fmaf((xv.y - mu.y) * rs.y, wv.y, bv.y),
                             fmaf((xv.z - mu.z) * rs.z, wv.z, bv.z), fmaf((xv.w - mu.w) * rs.w, wv.w, bv.w));
        }
        g = make_float4(yv.x > 0.f ? g.x : 0.f, yv.y > 0.f ? g.y : 0.f, yv.z > 0.f ? g.z : 0.f, yv.w > 0.f ? g.w : 0.f);
    }
    float4 o;
    if (batch_stats) {
        float4 a = ld4(gw + c), b = ld4(gb + c);
        const float in = 1.0f / (float)n_total;
        o.x = wv.x * rs.x * (g.x - b.x * in - (xv.x - mu.x) * rs.x * a.x * in);
        o.y = wv.y * rs.y * (g.y - b.y * in - (xv.y - mu.y) * rs.y * a.y * in);
        o.z = wv.z * rs.z * (g.z - b.z * in - (xv.z - mu.z) * rs.z * a.z * in);
        o.w = wv.w * rs.w * (g.w - b.w * in - (xv.w - mu.w) * rs.w * a.w * in);
    } else {
        o = make_float4(g.x * wv.x * rs.x, g.y * wv.y * rs.y, g.z * wv.z * rs.z, g.w * wv.w * rs.w);
    }
    st4(gx + 4 * xt, o);
}

// ----------------------------------------------------------------------------- optimizer / reductions
// SGD with momentum over the flat parameter buffer (torch.optim.SGD semantics, dampening 0, no
// nesterov / weight decay -- topological_training/train.py:66): buf = mu*buf + g ; p -= lr*buf.
// first_step: buf = g (torch initialises the momentum buffer with the first gradient).
__global__ void sgd_momentum_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                    int64_t n, float lr, float mu, int first, const float* __restrict__ lr_dev) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (lr_dev) lr = lr_dev[0];          // learning rate read at run time: a captured step follows the schedule
    const float b = first ? g[t] : fmaf(mu, buf[t], g[t]);
    buf[t] = b;
    p[t] = fmaf(-lr, b, p[t]);
}

// The same update reading the gradient of every parameter from its own tensor (pointers by value in the
// kernel arguments): the pack into the flat gradient buffer -- one more launch per step -- rides along (the
// packed value is still written, so the flat buffer stays what an all-reduce or a test would read).
constexpr int kSgdMaxSegs = 48;
struct SgdSegs {
    const float* g[kSgdMaxSegs];
    int64_t off[kSgdMaxSegs + 1];
    int count;
};
__global__ void sgd_momentum_multi_kernel(float* __restrict__ p, SgdSegs segs, float* __restrict__ gflat,
                                          float* __restrict__ buf, int64_t n, float lr, float mu, int first,
                                          const float* __restrict__ lr_dev) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (lr_dev) lr = lr_dev[0];
    int lo = 0, hi = segs.count;               // segment with off[lo] <= t < off[lo + 1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (segs.off[mid] <= t) lo = mid; else hi = mid;
    }
    const float* gp = segs.g[lo];
    const float g = gp ? gp[t - segs.off[lo]] : 0.f;
    if (gflat) gflat[t] = g;
    const float b = first ? g : fmaf(mu, buf[t], g);
    buf[t] = b;
    p[t] = fmaf(-lr, b, p[t]);
}

// column sums of a row-major [N, C] matrix (bias gradients), optionally fused with the backward of
// dropout(leaky_relu(.)): gx = g * act'(y) is written and ITS column sums are produced -- the bias
// gradient of a conv whose epilogue carried the activation.  Block partials, then one wave per column
// sums them in a fixed order (bitwise reproducible).  (A single-launch variant with a last-block
// ticket was measured 4-8x slower: the agent-scope release fence it needs writes back the L2 lines
// the same kernel has just dirtied with gx.)
template <bool ACT>
__global__ __launch_bounds__(256) void colsum_fused_kernel(const float* __restrict__ g, int ld,
                                                           const float* __restrict__ y, float* __restrict__ gx,
                                                           int64_t N, int C, ActParams act,
                                                           float* __restrict__ partials) {
    __shared__ float4 red[256];
    const int C4 = C / 4;
    const int RPB = 256 / C4;
    const int sub = threadIdx.x % C4, slot = threadIdx.x / C4;
    const int64_t per = (N + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per;
    const int64_t r1 = (r0 + per < N) ? r0 + per : N;
    float4 acc = f4zero();
    if (slot < RPB)
        for (int64_t r = r0 + slot; r < r1; r += RPB) {
            float4 v = ld4(g + r * ld + 4 * sub);
            if (ACT) {
                const int64_t flat = r * C + 4 * sub;
                const float4 yy = ld4(y + flat);
                uint64_t z = 0;
                if (act.thr16) z = act_hash64(act.seed, (uint64_t)act.step[0], (uint64_t)flat >> 2);
                float vi[4] = {v.x, v.y, v.z, v.w};
                const float vr[4] = {yy.x, yy.y, yy.z, yy.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool keep = act.thr16 ? (((uint32_t)(z >> (16 * c)) & 0xFFFFu) >= act.thr16) : true;
                    vi[c] = vi[c] * (keep ? act.keep_scale : 0.f) * (vr[c] > 0.f ? 1.0f : act.slope);
                }
                v = make_float4(vi[0], vi[1], vi[2], vi[3]);
                st4(gx + flat, v);
            }
            acc = add4(acc, v);
        }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < C4) {
        float4 s = red[threadIdx.x];
        for (int k = 1; k < RPB; ++k) s = add4(s, red[k * C4 + threadIdx.x]);
        st4(partials + (int64_t)blockIdx.x * C + 4 * threadIdx.x, s);
    }
}
__global__ void colsum_final_kernel(const float* __restrict__ partials, int nblk, int C, float* __restrict__ out) {
    const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= C) return;
    const float s = wave_sum_partials(partials, nblk, C, t);
    if ((threadIdx.x & 63) == 0) out[t] = s;
}
constexpr int kColsumBlocks = 1024;

// Small dense GEMM for the head MLP / embedding-table projection (a few MFLOP each; the library
// GEMM spends 13-16 us per call on them): C[M,N] = op(A)[M,K] op(B)[K,N] (+ bias[N]), generic
// element strides so every transpose variant (forward, grad_x, grad_W) is the same kernel.
// 32x32 tile per 256-thread block, 2x2 per thread, K tiles of 32 through LDS.
__global__ __launch_bounds__(256) void small_gemm_kernel(const float* __restrict__ A, int64_t sam, int64_t sak,
                                                         const float* __restrict__ B, int64_t sbk, int64_t sbn,
                                                         const float* __restrict__ bias, float* __restrict__ C,
                                                         int ldc, int M, int N, int K, int kchunk) {
    __shared__ float As[32][33];      // As[m][k]
    __shared__ float Bs[32][33];      // Bs[k][n]
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int kbeg = blockIdx.z * kchunk;
    const int kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
    float* Cz = C + (int64_t)blockIdx.z * M * ldc;          // split-K: one partial plane per z
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    const bool a_k_fast = (sak == 1), b_n_fast = (sbn == 1);
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        for (int t = threadIdx.x; t < 1024; t += 256) {
            // consecutive threads walk whichever index is contiguous in memory
            const int hi5 = t >> 5, lo5 = t & 31;
            const int ar = a_k_fast ? hi5 : lo5, ac = a_k_fast ? lo5 : hi5;       // (m, k) within the tile
            const int m = m0 + ar, k = k0 + ac;
            As[ar][ac] = (m < M && k < kend) ? A[m * sam + k * sak] : 0.f;
            const int br = b_n_fast ? hi5 : lo5, bc = b_n_fast ? lo5 : hi5;       // (k, n) within the tile
            const int kk = k0 + br, n = n0 + bc;
            Bs[br][bc] = (kk < kend && n < N) ? B[kk * sbk + n * sbn] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            const float a0 = As[ty][k], a1 = As[ty + 16][k];
            const float b0 = Bs[k][tx], b1 = Bs[k][tx + 16];
            acc[0][0] = fmaf(a0, b0, acc[0][0]); acc[0][1] = fmaf(a0, b1, acc[0][1]);
            acc[1][0] = fmaf(a1, b0, acc[1][0]); acc[1][1] = fmaf(a1, b1, acc[1][1]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + ty + 16 * i, n = n0 + tx + 16 * j;
            if (m < M && n < N) Cz[(int64_t)m * ldc + n] = acc[i][j] + ((bias && blockIdx.z == 0) ? bias[n] : 0.f);
        }
}

// out[c] = sum_b x[b*C + c] for WIDE rows (C up to millions, B rows): the sum over graphs of a
// per-node gradient [B, n*W] (TransformerConv table mode).  grid = (C/4/256, S row slices);
// slice partials [S, C] are summed by the second launch.  Coalesced 16-B accesses, 8 loads in flight.
__global__ void rowsum_wide_kernel(const float* __restrict__ x, int64_t B, int64_t C, int S,
                                   float* __restrict__ dst) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;     // float4 column
    if (t * 4 >= C) return;
    const int64_t per = (B + S - 1) / S;
    const int64_t b0 = blockIdx.y * per;
    const int64_t b1 = (b0 + per < B) ? b0 + per : B;
    float4 acc = f4zero();
    int64_t b = b0;
    for (; b + 8 <= b1; b += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ld4(x + (b + u) * C + 4 * t);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = add4(acc, v[u]);
    }
    for (; b < b1; ++b) acc = add4(acc, ld4(x + b * C + 4 * t));
    st4(dst + (int64_t)blockIdx.y * C + 4 * t, acc);
}
constexpr int kRowsumSlices = 16;

// One-launch form for a moderate number of rows: 1024 threads = 64 float4 columns x 16 row slices, slices meet
// in LDS in a fixed order (the two-launch form cost a second ~5 us launch for the 16 x C slice partials).
__global__ __launch_bounds__(1024) void rowsum_wide_1pass_kernel(const float* __restrict__ x, int64_t B, int64_t C,
                                                                 float* __restrict__ dst) {
    __shared__ float4 red[16][64];
    const int c = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int64_t t = blockIdx.x * 64 + c;                                // float4 column
    float4 acc = f4zero();
    if (t * 4 < C) {
        const int64_t per = (B + 15) / 16;
        const int64_t b0 = sg * per, b1 = (b0 + per < B) ? b0 + per : B;
        int64_t b = b0;
        for (; b + 8 <= b1; b += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ld4(x + (b + u) * C + 4 * t);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = add4(acc, v[u]);
        }
        for (; b + 4 <= b1; b += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = ld4(x + (b + u) * C + 4 * t);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = add4(acc, v[u]);
        }
        for (; b < b1; ++b) acc = add4(acc, ld4(x + b * C + 4 * t));
    }
    red[sg][c] = acc;
    __syncthreads();
    if (sg == 0 && t * 4 < C) {
        float4 s4 = red[0][c];
#pragma unroll
        for (int q = 1; q < 16; ++q) s4 = add4(s4, red[q][c]);
        st4(dst + 4 * t, s4);
    }
}

}  // namespace qot

using namespace qot;

extern "C" int qot_small_gemm(const float* A, int64_t stride_am, int64_t stride_ak, const float* B,
                              int64_t stride_bk, int64_t stride_bn, const float* bias, float* C, int ldc, int M,
                              int N, int K, int split_k, qot_stream_t stream) {
    if (M < 0 || N < 0 || K < 0 || split_k < 1) return QOT_ERR_BADARG;
    if (M == 0 || N == 0) return QOT_OK;
    if (!A || !B || !C) return QOT_ERR_BADARG;
    // split_k > 1: C must hold split_k planes of [M, ldc]; plane z gets the partial product of its
    // K chunk (bias in plane 0); the caller sums the planes (fixed order)
    int kchunk = ((K + split_k - 1) / split_k + 31) & ~31;
    if (kchunk < 32) kchunk = 32;
    small_gemm_kernel<<<dim3(grid_for(N, 32), grid_for(M, 32), split_k), 256, 0, (hipStream_t)stream>>>(
        A, stride_am, stride_ak, B, stride_bk, stride_bn, bias, C, ldc, M, N, K, kchunk);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" size_t qot_rowsum_wide_workspace_floats(int64_t C) { return (size_t)kRowsumSlices * (size_t)(C > 0 ? C : 0); }

extern "C" int qot_rowsum_wide(const float* x, int64_t B, int64_t C, float* out, float* workspace,
                               qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (B <= 0 || C <= 0 || !x || !out || !workspace) return QOT_ERR_BADARG;
    if (C & 3) return QOT_ERR_UNSUPPORTED;
    if (B >= 16 && B <= 1024) {
        rowsum_wide_1pass_kernel<<<grid_for(C / 4, 64), 1024, 0, stream>>>(x, B, C, out);
        QOT_LAUNCH_CHECK();
        return QOT_OK;
    }
    const int S = (B >= 4 * kRowsumSlices) ? kRowsumSlices : 1;
    rowsum_wide_kernel<<<dim3(grid_for(C / 4, 256), S), 256, 0, stream>>>(x, B, C, S, S > 1 ? workspace : out);
    QOT_LAUNCH_CHECK();
    if (S > 1) {
        rowsum_wide_kernel<<<dim3(grid_for(C / 4, 256), 1), 256, 0, stream>>>(workspace, S, C, 1, out);
        QOT_LAUNCH_CHECK();
    }
    return QOT_OK;
}

extern "C" int qot_sgd_momentum(float* param, const float* grad, float* momentum_buf, int64_t n, float lr,
                                float momentum, int first_step, qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || (n > 0 && (!param || !grad || !momentum_buf))) return QOT_ERR_BADARG;
    if (n > 0) {
        sgd_momentum_kernel<<<grid_for(n, 256), 256, 0, stream>>>(param, grad, momentum_buf, n, lr, momentum, first_step,
                                                                  nullptr);
        QOT_LAUNCH_CHECK();
    }
    return QOT_OK;
}

extern "C" int qot_sgd_momentum_multi(float* param, const float* const* grads, const int64_t* offsets, int count,
                                      float* grad_flat, float* momentum_buf, int64_t n, float lr,
                                      const float* lr_dev, float momentum, int first_step, qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || count < 0 || (n > 0 && (!param || !momentum_buf || !grads || !offsets))) return QOT_ERR_BADARG;
    if (count > kSgdMaxSegs) return QOT_ERR_UNSUPPORTED;
    if (n == 0 || count == 0) return QOT_OK;
    SgdSegs segs;
    for (int i = 0; i < count; ++i) {
        if (offsets[i] < 0 || offsets[i + 1] < offsets[i]) return QOT_ERR_BADARG;
        segs.g[i] = grads[i];
        segs.off[i] = offsets[i];
    }
    if (offsets[0] != 0 || offsets[count] != n) return QOT_ERR_BADARG;
    segs.off[count] = n;
    for (int i = count; i < kSgdMaxSegs; ++i) { segs.g[i] = nullptr; segs.off[i + 1] = n; }
    segs.count = count;
    sgd_momentum_multi_kernel<<<grid_for(n, 256), 256, 0, stream>>>(param, segs, grad_flat, momentum_buf, n, lr, momentum,
                                                                    first_step, lr_dev);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_sgd_momentum_dev(float* param, const float* grad, float* momentum_buf, int64_t n,
                                    const float* lr_dev, float momentum, int first_step, qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || !lr_dev || (n > 0 && (!param || !grad || !momentum_buf))) return QOT_ERR_BADARG;
    if (n > 0) {
        sgd_momentum_kernel<<<grid_for(n, 256), 256, 0, stream>>>(param, grad, momentum_buf, n, 0.f, momentum, first_step,
                                                                  lr_dev);
        QOT_LAUNCH_CHECK();
    }
    return QOT_OK;
}

extern "C" size_t qot_colsum_workspace_floats(int C) { return (size_t)kColsumBlocks * (size_t)(C > 0 ? C : 0); }

static int colsum_blocks(int64_t N) {
    int blocks = kColsumBlocks;
    if (N < (int64_t)blocks * 4) blocks = (int)((N + 3) / 4);
    return blocks;
}

extern "C" int qot_colsum(const float* x, int ld, int64_t N, int C, float* out, float* workspace,
                          qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N <= 0 || !x || !out || !workspace || (ld & 3)) return QOT_ERR_BADARG;
    if ((C & 3) || C <= 0 || C > 1024) return QOT_ERR_UNSUPPORTED;
    const ActParams none = make_act(0, 0.f, 0.f, 0, nullptr);
    const int blocks = colsum_blocks(N);
    colsum_fused_kernel<false><<<blocks, 256, 0, stream>>>(x, ld, nullptr, nullptr, N, C, none, workspace);
    QOT_LAUNCH_CHECK();
    colsum_final_kernel<<<grid_for(C, 4), 256, 0, stream>>>(workspace, blocks, C, out);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_act_bwd_colsum(const float* grad_y, const float* y, float* grad_x, int64_t N, int C, float slope,
                                  float p, uint64_t seed, const int64_t* step_counter, float* colsum_out,
                                  float* workspace, qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N <= 0 || !grad_y || !y || !grad_x || !colsum_out || !workspace) return QOT_ERR_BADARG;
    if ((C & 3) || C <= 0 || C > 1024) return QOT_ERR_UNSUPPORTED;
    const ActParams ap = make_act(1, slope, p, seed, step_counter);
    const int blocks = colsum_blocks(N);
    colsum_fused_kernel<true><<<blocks, 256, 0, stream>>>(grad_y, C, y, grad_x, N, C, ap, workspace);
    QOT_LAUNCH_CHECK();
    colsum_final_kernel<<<grid_for(C, 4), 256, 0, stream>>>(workspace, blocks, C, colsum_out);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

static const char* kErrNames[] = {"ok", "unsupported width / edge_dim (no CPU fallback)", "bad argument"};

extern "C" int qot_abi_version(void) { return QOT_ABI_VERSION; }

extern "C" const char* qot_error_string(int code) {
    if (code == 0) return kErrNames[0];
    if (code == QOT_ERR_UNSUPPORTED) return kErrNames[1];
    if (code == QOT_ERR_BADARG) return kErrNames[2];
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown qot error";
}

extern "C" int qot_embed_fwd(const float* table, const int32_t* ids, float* out, int64_t N, int V, int H,
                             qot_stream_t stream) {
    if (N < 0 || (H & 3) || H <= 0) return (H & 3) ? QOT_ERR_UNSUPPORTED : QOT_ERR_BADARG;
    if (N == 0) return QOT_OK;
    if (!table || !ids || !out || V <= 0) return QOT_ERR_BADARG;
    embed_gather_kernel<<<grid_for(N * (H / 4), 256), 256, 0, (hipStream_t)stream>>>(table, ids, out, N, H / 4, V);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_embed_bwd(const float* grad_out, const int32_t* ids, float* grad_table, int64_t N,
                             int V, int H, qot_stream_t stream) {
    if (N < 0 || V <= 0 || H <= 0) return QOT_ERR_BADARG;
    if (N == 0) return QOT_OK;
    if (!grad_out || !ids || !grad_table) return QOT_ERR_BADARG;
    const size_t lds = (size_t)V * H * 4;
    if (lds <= 64 * 1024) {
        // enough workgroups to fill the chip (>= 1024 when N allows), but each still folds
        // >= 64 rows into its LDS table before touching global atomics
        int64_t rpb64 = (N + 511) / 512;
        int rpb = (int)(rpb64 < 64 ? 64 : rpb64);
        embed_bwd_kernel<true><<<grid_for(N, rpb), 256, lds, (hipStream_t)stream>>>(grad_out, ids, grad_table, N, V, H, rpb);
    } else {
        int rpb = 256;
        embed_bwd_kernel<false><<<grid_for(N, rpb), 256, 0, (hipStream_t)stream>>>(grad_out, ids, grad_table, N, V, H, rpb);
    }
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

static int act_launch(bool bwd, const float* a, const float* yref, float* o, int64_t n, float slope, float p,
                      uint64_t seed, const int64_t* step_counter, hipStream_t stream) {
    if (n < 0 || (n & 3)) return (n & 3) ? QOT_ERR_UNSUPPORTED : QOT_ERR_BADARG;
    if (n == 0) return QOT_OK;
    if (!a || !o || (bwd && !yref) || p < 0.f || p >= 1.f) return QOT_ERR_BADARG;
    uint32_t thr = 0;
    float ks = 1.0f;
    if (p > 0.f && step_counter) {
        thr = (uint32_t)(p * 65536.0f + 0.5f);
        if (thr > 65535u) thr = 65535u;
        ks = 1.0f / (1.0f - p);
    }
    int64_t n4 = n / 4;
    if (bwd) act_kernel<true><<<grid_for(n4, 256), 256, 0, stream>>>(a, yref, o, n4, slope, thr, ks, seed, step_counter);
    else     act_kernel<false><<<grid_for(n4, 256), 256, 0, stream>>>(a, yref, o, n4, slope, thr, ks, seed, step_counter);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_act_fwd(const float* x, float* y, int64_t n, float slope, float p, uint64_t seed,
                           const int64_t* step_counter, qot_stream_t stream) {
    return act_launch(false, x, nullptr, y, n, slope, p, seed, step_counter, (hipStream_t)stream);
}

extern "C" int qot_act_bwd(const float* grad_y, const float* y_or_x, float* grad_x, int64_t n, float slope,
                           float p, uint64_t seed, const int64_t* step_counter, qot_stream_t stream) {
    return act_launch(true, grad_y, y_or_x, grad_x, n, slope, p, seed, step_counter, (hipStream_t)stream);
}

extern "C" int qot_pool_fwd(const float* x, const int32_t* ptr, float* out, int64_t B, int H,
                            qot_stream_t stream) {
    if (B < 0) return QOT_ERR_BADARG;
    if ((H & 3) || H <= 0 || H > 1024) return QOT_ERR_UNSUPPORTED;
    if (B == 0) return QOT_OK;
    if (!ptr || !out) return QOT_ERR_BADARG;
    pool_fwd_kernel<<<(int)B, 256, 0, (hipStream_t)stream>>>(x, ptr, out, H);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_pool_bwd(const float* grad_out, const int32_t* ptr, const int32_t* batch, float* grad_x,
                            int64_t N, int64_t B, int H, qot_stream_t stream) {
    if (N < 0 || B < 0) return QOT_ERR_BADARG;
    if ((H & 3) || H <= 0) return QOT_ERR_UNSUPPORTED;
    if (N == 0) return QOT_OK;
    if (!grad_out || !ptr || !batch || !grad_x) return QOT_ERR_BADARG;
    pool_bwd_kernel<<<grid_for(N * (H / 4), 256), 256, 0, (hipStream_t)stream>>>(grad_out, ptr, batch, grad_x, N, H / 4);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" size_t qot_bn_partials_floats(int64_t N, int C) {
    if (N < 0 || C <= 0) return 0;
    return (size_t)bn_blocks(N) * 2 * (size_t)C;
}

extern "C" int qot_bn_stats(const float* x, int64_t N, int C, float eps, float momentum, float* mean,
                            float* rstd, float* running_mean, float* running_var, float* partials,
                            qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N <= 0) return QOT_ERR_BADARG;
    if ((C & 3) || C <= 0 || C > 1024) return QOT_ERR_UNSUPPORTED;
    if (!x || !mean || !rstd || !partials || ((running_mean == nullptr) != (running_var == nullptr)))
        return QOT_ERR_BADARG;
    int nblk = bn_blocks(N);
    bn_partial_kernel<0><<<nblk, 256, 0, stream>>>(x, nullptr, nullptr, nullptr, nullptr, partials, N, C, 0, nullptr, nullptr);
    QOT_LAUNCH_CHECK();
    bn_finalize_stats_kernel<<<grid_for(C, 4), 256, 0, stream>>>(x, partials, nblk, N, C, eps, momentum, mean, rstd, running_mean, running_var);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

// Batch statistics from per-workgroup column partials [nblk][2][C] of (x - shift), (x - shift)^2 produced by the
// kernel that wrote x (qot_gat_fwd with bn_partials): mean / rstd / running statistics as qot_bn_stats, without
// another pass over x.
extern "C" int qot_bn_stats_from_partials(const float* shift, const float* partials, int nblk, int64_t chunk_rows,
                                          int64_t N, int C, float eps, float momentum, float* mean, float* rstd,
                                          float* running_mean, float* running_var, qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N <= 0 || C <= 0 || nblk <= 0 || chunk_rows <= 0 || !shift || !partials || !mean || !rstd) return QOT_ERR_BADARG;
    bn_finalize_chan_kernel<<<grid_for(C, 4), 256, 0, stream>>>(shift, partials, nblk, chunk_rows, N, C, eps, momentum, mean,
                                                                rstd, running_mean, running_var);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_bn_apply(const float* x, const float* mean, const float* rstd, const float* w,
                            const float* b, float* y, int64_t N, int C, int relu, qot_stream_t stream) {
    if (N < 0) return QOT_ERR_BADARG;
    if ((C & 3) || C <= 0) return QOT_ERR_UNSUPPORTED;
    if (N == 0) return QOT_OK;
    if (!x || !mean || !rstd || !w || !b || !y) return QOT_ERR_BADARG;
    bn_apply_kernel<<<grid_for(N * (C / 4), 256), 256, 0, (hipStream_t)stream>>>(x, mean, rstd, w, b, y, N, C / 4, relu);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_bn_bwd_reduce(const float* grad_y, const float* y, const float* x, const float* mean,
                                 const float* rstd, float* gw, float* gb, int64_t N, int C, int relu,
                                 float* partials, const float* w, const float* b, qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (N <= 0) return QOT_ERR_BADARG;
    if ((C & 3) || C <= 0 || C > 1024) return QOT_ERR_UNSUPPORTED;
    if (!grad_y || !x || !mean || !rstd || !gw || !gb || !partials || (relu && !y && (!w || !b))) return QOT_ERR_BADARG;
    int nblk = bn_blocks(N);
    bn_partial_kernel<1><<<nblk, 256, 0, stream>>>(x, grad_y, y, mean, rstd, partials, N, C, relu, w, b);
    QOT_LAUNCH_CHECK();
    bn_finalize_bwd_kernel<<<grid_for(C, 4), 256, 0, stream>>>(partials, nblk, C, gw, gb);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_bn_bwd_apply(const float* grad_y, const float* y, const float* x, const float* mean,
                                const float* rstd, const float* w, const float* gw, const float* gb,
                                float* grad_x, int64_t N, int C, int relu, int batch_stats,
                                const float* b, qot_stream_t stream) {
    if (N < 0) return QOT_ERR_BADARG;
    if ((C & 3) || C <= 0) return QOT_ERR_UNSUPPORTED;
    if (N == 0) return QOT_OK;
    if (!grad_y || !rstd || !w || !grad_x || (relu && !y && (!b || !x || !mean))) return QOT_ERR_BADARG;
    if (batch_stats && (!x || !mean || !gw || !gb)) return QOT_ERR_BADARG;
    bn_bwd_apply_kernel<<<grid_for(N * (C / 4), 256), 256, 0, (hipStream_t)stream>>>(
        grad_y, y, x, mean, rstd, w, gw, gb, grad_x, N, C / 4, relu, batch_stats, b);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

// ---- BatchNorm(+ReLU) of which only some rows are consumed: y_rows = act(BatchNorm(x))[idx] (LightpathGNN's LUT rows,
// lightpath_training/models.py:31-32 then :35-40).  The statistics are those of all N rows (computed elsewhere); the
// normalised matrix is never formed, the backward's column sums run over the n rows that carry a gradient, and grad_x
// (dense: every row feels the batch statistics) is written without a zero-filled [N, C] gradient being read.
// idx: n unique row numbers.
extern "C" int qot_bn_apply_rows(const float* x, const int32_t* idx, int64_t n, const float* mean, const float* rstd,
                                 const float* w, const float* b, float* y_rows, int C, int relu, qot_stream_t stream) {
    if (n < 0) return QOT_ERR_BADARG;
    if ((C & 3) || C <= 0) return QOT_ERR_UNSUPPORTED;
    if (n == 0) return QOT_OK;
    if (!x || !idx || !mean || !rstd || !w || !b || !y_rows) return QOT_ERR_BADARG;
    bn_apply_kernel<<<grid_for(n * (C / 4), 256), 256, 0, (hipStream_t)stream>>>(x, mean, rstd, w, b, y_rows, n, C / 4, relu, idx);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

// gb[c] = sum_r dy'[r, c], gw[c] = sum_r dy'[r, c] * xhat[idx[r], c]  (dy' = dy * [y > 0] when relu; mask recomputed
// from x);  partials: qot_bn_partials_floats(n, C) floats.
extern "C" int qot_bn_bwd_reduce_rows(const float* grad_rows, const int32_t* idx, int64_t n, const float* x,
                                      const float* mean, const float* rstd, float* gw, float* gb, int C, int relu,
                                      float* partials, const float* w, const float* b, qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n <= 0) return QOT_ERR_BADARG;
    if ((C & 3) || C <= 0 || C > 1024) return QOT_ERR_UNSUPPORTED;
    if (!grad_rows || !idx || !x || !mean || !rstd || !gw || !gb || !partials || (relu && (!w || !b))) return QOT_ERR_BADARG;
    int nblk = bn_blocks(n);
    bn_partial_kernel<1><<<nblk, 256, 0, stream>>>(x, grad_rows, nullptr, mean, rstd, partials, n, C, relu, w, b, idx);
    QOT_LAUNCH_CHECK();
    bn_finalize_bwd_kernel<<<grid_for(C, 4), 256, 0, stream>>>(partials, nblk, C, gw, gb);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

// grad_x [N, C] of the batch-statistics form: every row from (gw, gb) and x alone, then the n rows of idx again with
// their gradient -- bit for bit qot_bn_bwd_apply on the zero-filled, scattered [N, C] gradient.
extern "C" int qot_bn_bwd_apply_rows(const float* grad_rows, const int32_t* idx, int64_t n, const float* x,
                                     const float* mean, const float* rstd, const float* w, const float* gw,
                                     const float* gb, float* grad_x, int64_t N, int C, int relu, const float* b,
                                     qot_stream_t stream) {
    if (N < 0 || n < 0 || n > N) return QOT_ERR_BADARG;
    if ((C & 3) || C <= 0) return QOT_ERR_UNSUPPORTED;
    if (N == 0) return QOT_OK;
    if (!x || !mean || !rstd || !w || !gw || !gb || !grad_x || (n > 0 && (!grad_rows || !idx)) || (relu && !b))
        return QOT_ERR_BADARG;
    bn_bwd_apply_kernel<<<grid_for(N * (C / 4), 256), 256, 0, (hipStream_t)stream>>>(
        nullptr, nullptr, x, mean, rstd, w, gw, gb, grad_x, N, C / 4, relu, 1, b);
    QOT_LAUNCH_CHECK();
    if (n > 0) {
        bn_bwd_apply_kernel<<<grid_for(n * (C / 4), 256), 256, 0, (hipStream_t)stream>>>(
            grad_rows, nullptr, x, mean, rstd, w, gw, gb, grad_x, n, C / 4, relu, 1, b, idx, N);
        QOT_LAUNCH_CHECK();
    }
    return QOT_OK;
}

extern "C" int qot_rows_gather(const float* x, const int32_t* idx, float* out, int64_t n_idx, int C,
                               qot_stream_t stream) {
    if (n_idx < 0) return QOT_ERR_BADARG;
    if ((C & 3) || C <= 0) return QOT_ERR_UNSUPPORTED;
    if (n_idx == 0) return QOT_OK;
    if (!x || !idx || !out) return QOT_ERR_BADARG;
    rows_gather_kernel<<<grid_for(n_idx * (C / 4), 256), 256, 0, (hipStream_t)stream>>>(x, idx, out, n_idx, C / 4);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_rows_scatter(const float* grad_out, const int32_t* idx, float* grad_x, int64_t n_idx,
                                int C, qot_stream_t stream) {
    if (n_idx < 0) return QOT_ERR_BADARG;
    if ((C & 3) || C <= 0) return QOT_ERR_UNSUPPORTED;
    if (n_idx == 0) return QOT_OK;
    if (!grad_out || !idx || !grad_x) return QOT_ERR_BADARG;
    rows_scatter_kernel<<<grid_for(n_idx * (C / 4), 256), 256, 0, (hipStream_t)stream>>>(grad_out, idx, grad_x, n_idx, C / 4);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}
