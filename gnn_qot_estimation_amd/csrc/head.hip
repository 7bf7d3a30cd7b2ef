// Fused read-out head of TopologicalGNN:  global_mean_pool -> Linear(H,H) -> LeakyReLU(0.01) ->
// Dropout(p) -> Linear(H,O)   (topological_training/models.py:33-38,61-63) and its backward.
// As separate launches this tail was ~25 kernels / ~145 us per step at B = 1024 (nine small GEMMs,
// split-K sums, pool forward/backward, bias column sums, elementwise); the arithmetic is a few
// MFLOP.  Forward: one workgroup per graph.  Backward: a fixed number of workgroups, each walks its
// graphs keeping the partial weight gradients in registers; partials are summed in a fixed order.
#include "common.hpp"

namespace qot {

constexpr int kHeadBwdBlocks = 512;

// dynamic LDS: p[H] | h[H]
template <int H>
__global__ __launch_bounds__(256) void head_fwd_kernel(
    const float* __restrict__ x, const int32_t* __restrict__ ptr, const float* __restrict__ w0,
    const float* __restrict__ b0, const float* __restrict__ w3, const float* __restrict__ b3,
    float* __restrict__ pooled, float* __restrict__ hidden, float* __restrict__ out, int O, ActParams act,
    const float* __restrict__ target, float beta, float inv_n, float* __restrict__ grad_out, float* __restrict__ loss_rows) {
    __shared__ float4 red[256];
    __shared__ float lrow[8];
    __shared__ float p[H];
    __shared__ float h[H];
    constexpr int TPR = H / 4, RPB = 256 / TPR;
    const int sub = threadIdx.x % TPR, slot = threadIdx.x / TPR;
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
        const int cnt = end - beg;
        s = scale4(1.0f / (float)(cnt > 1 ? cnt : 1), s);
        p[4 * threadIdx.x] = s.x; p[4 * threadIdx.x + 1] = s.y; p[4 * threadIdx.x + 2] = s.z; p[4 * threadIdx.x + 3] = s.w;
        st4(pooled + b * H + 4 * threadIdx.x, s);
    }
    __syncthreads();
    if (threadIdx.x < H) {
        const int o = threadIdx.x;
        float v = b0[o];
#pragma unroll 8
        for (int a = 0; a < H; ++a) v = fmaf(w0[o * H + a], p[a], v);
        v = act_apply1(v, act, (uint64_t)(b * H + o));          // LeakyReLU + Dropout
        h[o] = v;
        hidden[b * H + o] = v;
    }
    __syncthreads();
    if (threadIdx.x < O) {
        const int o = threadIdx.x;
        float v = b3[o];
        for (int a = 0; a < H; ++a) v = fmaf(w3[o * H + a], h[a], v);
        out[b * O + o] = v;
        if (target) {          // SmoothL1Loss(reduction = mean, beta) of this graph's row: gradient now, value summed later
            const float d = v - target[b * O + o];
            const float ad = fabsf(d);
            float l, g;
            if (ad < beta) { l = 0.5f * d * d / beta; g = d / beta; }
            else           { l = ad - 0.5f * beta;    g = d > 0.f ? 1.f : -1.f; }
            grad_out[b * O + o] = g * inv_n;
            lrow[o] = l * inv_n;
        }
    }
    if (target) {
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = lrow[0];
            for (int o = 1; o < O; ++o) s += lrow[o];
            loss_rows[b] = s;
        }
    }
}

// Backward.  Per graph: gh = (W3^T gout) * act'(hidden) ; gp = W0^T gh ; grad_x rows = gp / count.
// Weight-gradient partials per block: gW0[H,H] (thread t owns rows (t>>2)..., see below), gb0[H],
// gW3[O,H], gb3[O].  Layout of one partial: [H*H | H | O*H | O].
template <int H>
__global__ __launch_bounds__(256) void head_bwd_kernel(
    const float* __restrict__ gout, const float* __restrict__ pooled, const float* __restrict__ hidden,
    const int32_t* __restrict__ ptr, const float* __restrict__ w0, const float* __restrict__ w3,
    float* __restrict__ gx, float* __restrict__ partials, int64_t B, int O, ActParams act,
    const float* __restrict__ x_in, ActParams in_act) {
    // x_in != NULL: the head's input is y = dropout(leaky_relu(conv)) (the conv kernel's fused epilogue,
    // models.py:58-59) and gx is to be the gradient wrt the conv output: the pool backward multiplies by
    // act'(y) on the fly (mask regenerated from in_act) and the block also accumulates the column sums of
    // that gradient = the conv's bias gradient (partials slot [.. | H]).
    constexpr int PER = H * H / 256;                  // gW0 elements per thread (16 at H = 64)
    __shared__ float4 cred[256];
    float4 acs = f4zero();                            // threads t < H/4: bias-gradient columns 4t..4t+3
    __shared__ float gh[H];
    __shared__ float gp[H];
    __shared__ float pp[H];
    __shared__ float go[8];
    float aw0[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) aw0[e] = 0.f;
    float ab0 = 0.f, aw3[8], ab3 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) aw3[e] = 0.f;
    const int t = threadIdx.x;
    // both weight matrices sit in LDS for the whole kernel: read from global memory inside the graph loop the
    // 64-step matrix-vector product below was 64 dependent L2 round trips per graph
    __shared__ float w0s[H * H];
    __shared__ float w3s[8 * H];
    {   // all of this thread's H*H/256 weights requested before the first LDS store (load -> store per trip: that many
        // dependent round trips at the start of every workgroup)
        float wv[PER];
#pragma unroll
        for (int e = 0; e < PER; ++e) wv[e] = w0[t + 256 * e];
#pragma unroll
        for (int e = 0; e < PER; ++e) w0s[t + 256 * e] = wv[e];
    }
    for (int e = t; e < O * H; e += 256) w3s[e] = w3[e];
    // gW0 element e of thread t: row o = (t*PER + e) / H, col a = (t*PER + e) % H
    for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
        if (t < O) go[t] = gout[b * O + t];
        if (t < H) pp[t] = pooled[b * H + t];
        const int beg = ptr[b], end = ptr[b + 1];       // requested here, used by the pool backward below
        const float hv_pre = (t < H) ? hidden[b * H + t] : 0.f;
        __syncthreads();
        if (t < H) {
            float v = 0.f;
            for (int o = 0; o < O; ++o) v = fmaf(w3s[o * H + t], go[o], v);
            const float hv = hv_pre;
            // d/dpre of dropout(leaky_relu(pre)): hidden == 0 <=> dropped (or pre == 0)
            float d = 0.f;
            if (act.thr16) {
                const uint64_t z = act_hash64(act.seed, (uint64_t)act.step[0], (uint64_t)(b * H + t) >> 2);
                const bool keep = ((uint32_t)(z >> (16 * ((b * H + t) & 3))) & 0xFFFFu) >= act.thr16;
                d = keep ? act.keep_scale : 0.f;
            } else {
                d = 1.0f;
            }
            d *= (hv > 0.f) ? 1.0f : act.slope;
            gh[t] = v * d;
            ab0 += v * d;
#pragma unroll
            for (int o = 0; o < 8; ++o) if (o < O) aw3[o] = fmaf(go[o], hv, aw3[o]);     // gW3[o, t]
        }
        if (t < O) ab3 += go[t];
        __syncthreads();
        if (t < H) {
            float v = 0.f;
#pragma unroll 8
            for (int o = 0; o < H; ++o) v = fmaf(w0s[o * H + t], gh[o], v);
            gp[t] = v;
        }
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int idx = t * PER + e;
            aw0[e] = fmaf(gh[idx / H], pp[idx % H], aw0[e]);
        }
        __syncthreads();
        // pool backward: every node row of the graph gets gp / count
        const float inv = 1.0f / (float)((end - beg) > 1 ? (end - beg) : 1);
        constexpr int TPR = H / 4, RPB = 256 / TPR;
        const int sub = t % TPR, slot = t / TPR;
        float4 cs = f4zero();
        if (slot < RPB) {
            const float4 v = make_float4(gp[4 * sub] * inv, gp[4 * sub + 1] * inv, gp[4 * sub + 2] * inv, gp[4 * sub + 3] * inv);
            if (!x_in) {
                for (int64_t r = beg + slot; r < end; r += RPB) st4(gx + r * H + 4 * sub, v);
            } else {
                // four rows in flight per lane group (one row per trip left a graph of 1000 nodes 125 dependent round trips
                // per group: 255 us at cfg4, 4.1 TB/s)
                constexpr int UR = 4;
                for (int64_t r0 = beg + slot; r0 < end; r0 += (int64_t)UR * RPB) {
                    float4 yy[UR];
#pragma unroll
                    for (int u = 0; u < UR; ++u) {
                        const int64_t r = r0 + (int64_t)u * RPB;
                        yy[u] = ld4(x_in + (r < end ? r : r0) * H + 4 * sub);
                    }
#pragma unroll
                    for (int u = 0; u < UR; ++u) {
                        const int64_t r = r0 + (int64_t)u * RPB;
                        if (r >= end) break;
                        const int64_t flat = r * H + 4 * sub;
                        uint64_t z = 0;
                        if (in_act.thr16) z = act_hash64(in_act.seed, (uint64_t)in_act.step[0], (uint64_t)flat >> 2);
                        float vi[4] = {v.x, v.y, v.z, v.w};
                        const float vr[4] = {yy[u].x, yy[u].y, yy[u].z, yy[u].w};
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const bool keep = in_act.thr16 ? (((uint32_t)(z >> (16 * c)) & 0xFFFFu) >= in_act.thr16) : true;
                            vi[c] = vi[c] * (keep ? in_act.keep_scale : 0.f) * (vr[c] > 0.f ? 1.0f : in_act.slope);
                        }
                        const float4 o4 = make_float4(vi[0], vi[1], vi[2], vi[3]);
                        st4(gx + flat, o4);
                        cs = add4(cs, o4);
                    }
                }
            }
        }
        if (x_in) {
            cred[t] = cs;
            __syncthreads();
            if (t < TPR) {
                float4 s4 = cred[t];
                for (int k2 = 1; k2 < RPB; ++k2) s4 = add4(s4, cred[k2 * TPR + t]);
                acs = add4(acs, s4);
            }
        }
        __syncthreads();
    }
    float* part = partials + (int64_t)blockIdx.x * (H * H + H + O * H + O + (x_in ? H : 0));
    if (x_in && t < H / 4) st4(part + H * H + H + O * H + O + 4 * t, acs);
#pragma unroll
    for (int e = 0; e < PER; ++e) part[t * PER + e] = aw0[e];
    if (t < H) {
        part[H * H + t] = ab0;
        for (int o = 0; o < O; ++o) part[H * H + H + o * H + t] = aw3[o];
    }
    if (t < O) part[H * H + H + O * H + t] = ab3;
}

// ---- the read-out of a TRAIN step in ONE kernel: forward, criterion and backward per graph -------------------------------
// With the criterion formed in the forward kernel (qot_head_fwd_loss) the head's backward follows its forward at once and
// reads the same rows again: pool -> Linear -> LeakyReLU -> Dropout -> Linear -> SmoothL1 -> d/d(all of it) -> pool
// backward (through the producer's activation) is a chain over ONE graph.  A workgroup walks its graphs; a graph's node
// rows (<= kHeadRowsCap of them) are read from HBM once and stay in LDS for the pool backward, pooled / hidden never leave
// the workgroup.  Outputs: out, loss_rows, grad_x and the per-workgroup parameter-gradient partials of head_bwd_kernel
// (same layout, same fixed-order sum afterwards).  Per step at cfg2: one launch and one 26 MB pass less (head_fwd 9.4 +
// head_bwd 17.4 us before).
constexpr int kHeadRowsCap = 128;

template <int H>
__global__ __launch_bounds__(256) void head_train_kernel(
    const float* __restrict__ x, const int32_t* __restrict__ ptr, const float* __restrict__ w0,
    const float* __restrict__ b0, const float* __restrict__ w3, const float* __restrict__ b3,
    const float* __restrict__ target, float beta, float inv_n, float* __restrict__ out, float* __restrict__ grad_out,
    float* __restrict__ loss_rows, float* __restrict__ gx, float* __restrict__ partials, int64_t B, int O, ActParams act,
    int fold, ActParams in_act) {
    constexpr int PER = H * H / 256;
    constexpr int TPR = H / 4, RPB = 256 / TPR;
    // xs[kHeadRowsCap][H] | w0s[H][H + 1] | w3s[8*H].  W0's rows are padded by one float: the forward product reads
    // w0s[t][a] with t across lanes -- at a row stride of H floats every lane of a wave hit the same bank (64-way conflict)
    constexpr int HP = H + 1;
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    float* xs = dyn;
    float* w0s = dyn + kHeadRowsCap * H;
    float* w3s = w0s + H * HP;
    __shared__ float4 cred[256];
    __shared__ float pp[H];
    __shared__ float hh[H];
    __shared__ float gh[H];
    __shared__ float gp[H];
    __shared__ float go[8];
    __shared__ float lrow[8];
    float4 acs = f4zero();
    float aw0[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) aw0[e] = 0.f;
    float ab0 = 0.f, aw3[8], ab3 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) aw3[e] = 0.f;
    const int t = threadIdx.x;
    const int sub = t % TPR, slot = t / TPR;
    {   // all of this thread's H*H/256 weights requested before the first LDS store (load -> store per trip: that many
        // dependent round trips at the start of every workgroup)
        float wv[PER];
#pragma unroll
        for (int e = 0; e < PER; ++e) wv[e] = w0[t + 256 * e];
#pragma unroll
        for (int e = 0; e < PER; ++e) w0s[((t + 256 * e) / H) * HP + (t + 256 * e) % H] = wv[e];
    }
    for (int e = t; e < O * H; e += 256) w3s[e] = w3[e];
    // per-thread constants of the graph loop, read ONCE (a load per use is a global round trip inside a chain of eight
    // barrier-separated phases per graph: the step counters alone were one trip per node row of the pool backward)
    const uint64_t stepv = act.thr16 ? (uint64_t)act.step[0] : 0;
    const uint64_t in_stepv = (fold && in_act.thr16) ? (uint64_t)in_act.step[0] : 0;
    const float b0t = t < H ? b0[t] : 0.f;
    const float b3t = t < O ? b3[t] : 0.f;
    // The NEXT graph's first rows (and its target) are requested before the current graph's phases, its boundaries one graph
    // earlier still: a graph then costs no exposed round trip of its own (first version: boundaries -> rows -> eight phases,
    // back to back, per graph)
    int vz = 0;
    asm volatile("" : "+v"(vz));                 // boundaries through VECTOR loads (a scalar load shares lgkmcnt with LDS)
    int pbeg = 0, pend = 0, nbeg = 0, nend = 0;
    float4 pv[8];
    float ptgt = 0.f;
    {
        const int64_t b = blockIdx.x;
        if (b < B) { pbeg = ptr[b + vz]; pend = ptr[b + 1 + vz]; }
        const int64_t bn = b + gridDim.x;
        if (bn < B) { nbeg = ptr[bn + vz]; nend = ptr[bn + 1 + vz]; }
        const int pc = pend - pbeg;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = slot + u * RPB;
            pv[u] = (slot < RPB && pc > 0) ? ld4(x + (int64_t)(pbeg + (r < pc ? r : 0)) * H + 4 * sub) : f4zero();   // (an empty graph owns no row)
        }
        if (t < O && b < B) ptgt = target[b * O + t];
    }
    for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
        const int beg = pbeg, end = pend;
        const int cnt = end - beg;
        const bool in_lds = cnt <= kHeadRowsCap;
        const float tgt = ptgt;
        float4 v0[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v0[u] = pv[u];
        {   // request the next graph
            const int64_t bn = b + gridDim.x;
            pbeg = nbeg; pend = nend;
            if (bn < B) {
                const int64_t b2 = bn + gridDim.x;
                if (b2 < B) { nbeg = ptr[b2 + vz]; nend = ptr[b2 + 1 + vz]; }
                const int pc = pend - pbeg;
                if (slot < RPB && pc > 0) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int r = slot + u * RPB;
                        pv[u] = ld4(x + (int64_t)(pbeg + (r < pc ? r : 0)) * H + 4 * sub);
                    }
                }
                if (t < O) ptgt = target[bn * O + t];
            }
        }
        // ---- pool (rows kept in LDS for the pool backward when they fit)
        float4 acc = f4zero();
        if (slot < RPB) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {                       // the prefetched batch
                const int r = slot + u * RPB;
                if (r < cnt) {
                    if (in_lds) *reinterpret_cast<float4*>(&xs[r * H + 4 * sub]) = v0[u];
                    acc = add4(acc, v0[u]);
                }
            }
            // further batches (graphs of more than 8 * RPB rows): eight rows requested before the first is used
            for (int r0 = slot + 8 * RPB; r0 < cnt; r0 += 8 * RPB) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = r0 + u * RPB;
                    v[u] = ld4(x + (int64_t)(beg + (r < cnt ? r : r0)) * H + 4 * sub);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = r0 + u * RPB;
                    if (r < cnt) {
                        if (in_lds) *reinterpret_cast<float4*>(&xs[r * H + 4 * sub]) = v[u];
                        acc = add4(acc, v[u]);
                    }
                }
            }
        }
        cred[t] = acc;
        __syncthreads();
        if (t < TPR) {
            float4 s4 = cred[t];
            for (int k2 = 1; k2 < RPB; ++k2) s4 = add4(s4, cred[k2 * TPR + t]);
            s4 = scale4(1.0f / (float)(cnt > 1 ? cnt : 1), s4);
            pp[4 * t] = s4.x; pp[4 * t + 1] = s4.y; pp[4 * t + 2] = s4.z; pp[4 * t + 3] = s4.w;
        }
        __syncthreads();
        // ---- Linear -> LeakyReLU -> Dropout
        float hv = 0.f, dact = 0.f;
        if (t < H) {
            float v = b0t;
#pragma unroll 8
            for (int a = 0; a < H; ++a) v = fmaf(w0s[t * HP + a], pp[a], v);
            // d/dpre of dropout(leaky_relu(pre)), from the same draw the forward applies
            float d = 1.0f;
            if (act.thr16) {
                const uint64_t flat = (uint64_t)(b * H + t);
                const uint64_t z = act_hash64(act.seed, stepv, flat >> 2);
                const bool keep = ((uint32_t)(z >> (16 * (flat & 3))) & 0xFFFFu) >= act.thr16;
                d = keep ? act.keep_scale : 0.f;
            }
            d *= (v > 0.f) ? 1.0f : act.slope;
            hv = v * d;                                   // = dropout(leaky_relu(v))
            dact = d;
            hh[t] = hv;
        }
        __syncthreads();
        // ---- Linear(H, O), criterion
        if (t < O) {
            float v = b3t;
            for (int a = 0; a < H; ++a) v = fmaf(w3s[t * H + a], hh[a], v);
            out[b * O + t] = v;
            const float df = v - tgt;
            const float ad = fabsf(df);
            float l, g;
            if (ad < beta) { l = 0.5f * df * df / beta; g = df / beta; }
            else           { l = ad - 0.5f * beta;      g = df > 0.f ? 1.f : -1.f; }
            g *= inv_n;
            grad_out[b * O + t] = g;
            go[t] = g;
            lrow[t] = l * inv_n;
            ab3 += g;
        }
        __syncthreads();
        if (t == 0) {
            float sl = lrow[0];
            for (int o = 1; o < O; ++o) sl += lrow[o];
            loss_rows[b] = sl;
        }
        // ---- backward of the two Linear layers
        if (t < H) {
            float v = 0.f;
            for (int o = 0; o < O; ++o) v = fmaf(w3s[o * H + t], go[o], v);
            const float g1 = v * dact;
            gh[t] = g1;
            ab0 += g1;
#pragma unroll
            for (int o = 0; o < 8; ++o) if (o < O) aw3[o] = fmaf(go[o], hv, aw3[o]);     // gW3[o, t]
        }
        __syncthreads();
        if (t < H) {
            float v = 0.f;
#pragma unroll 8
            for (int o = 0; o < H; ++o) v = fmaf(w0s[o * HP + t], gh[o], v);
            gp[t] = v;
        }
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            const int idx = t * PER + e;
            aw0[e] = fmaf(gh[idx / H], pp[idx % H], aw0[e]);
        }
        __syncthreads();
        // ---- pool backward: every node row of the graph gets gp / count, through the producer's activation when folded
        const float inv = 1.0f / (float)(cnt > 1 ? cnt : 1);
        float4 cs = f4zero();
        if (slot < RPB) {
            const float4 v = make_float4(gp[4 * sub] * inv, gp[4 * sub + 1] * inv, gp[4 * sub + 2] * inv, gp[4 * sub + 3] * inv);
            if (!fold) {
                for (int r = slot; r < cnt; r += RPB) st4(gx + (int64_t)(beg + r) * H + 4 * sub, v);
            } else {
                for (int r = slot; r < cnt; r += RPB) {
                    const int64_t flat = (int64_t)(beg + r) * H + 4 * sub;
                    const float4 yy = in_lds ? *reinterpret_cast<const float4*>(&xs[r * H + 4 * sub]) : ld4(x + flat);
                    uint64_t z = 0;
                    if (in_act.thr16) z = act_hash64(in_act.seed, in_stepv, (uint64_t)flat >> 2);
                    float vi[4] = {v.x, v.y, v.z, v.w};
                    const float vr[4] = {yy.x, yy.y, yy.z, yy.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const bool keep = in_act.thr16 ? (((uint32_t)(z >> (16 * c)) & 0xFFFFu) >= in_act.thr16) : true;
                        vi[c] = vi[c] * (keep ? in_act.keep_scale : 0.f) * (vr[c] > 0.f ? 1.0f : in_act.slope);
                    }
                    const float4 o4 = make_float4(vi[0], vi[1], vi[2], vi[3]);
                    st4(gx + flat, o4);
                    cs = add4(cs, o4);
                }
            }
        }
        if (fold) {
            cred[t] = cs;
            __syncthreads();
            if (t < TPR) {
                float4 s4 = cred[t];
                for (int k2 = 1; k2 < RPB; ++k2) s4 = add4(s4, cred[k2 * TPR + t]);
                acs = add4(acs, s4);
            }
        }
        __syncthreads();
    }
    float* part = partials + (int64_t)blockIdx.x * (H * H + H + O * H + O + (fold ? H : 0));
    if (fold && t < H / 4) st4(part + H * H + H + O * H + O + 4 * t, acs);
#pragma unroll
    for (int e = 0; e < PER; ++e) part[t * PER + e] = aw0[e];
    if (t < H) {
        part[H * H + t] = ab0;
        for (int o = 0; o < O; ++o) part[H * H + H + o * H + t] = aw3[o];
    }
    if (t < O) part[H * H + H + O * H + t] = ab3;
}

// ---- the same, GB = 256 / H graphs per pass (r04) ------------------------------------------------------------------------
// head_train_kernel walks ONE graph at a time: eight barrier-separated phases per graph with 64 of 256 threads busy in the
// two matrix-vector products, the graph's rows through LDS.  Here a pass takes GB graphs (4 at H = 64): thread (g, o) owns
// output o of graph g in every dense phase (all 256 threads busy, one barrier per phase per GB graphs), a graph's rows belong
// to FOUR lane groups (H / 4 lanes each) that keep them in REGISTERS from the pool to the pool backward (up to 32 rows per
// group = 128 rows per graph; rows beyond are read again), and the per-thread partial sums of the bias / W3 / column-sum
// gradients meet once per kernel instead of once per graph.  Same outputs and partial-row layout as head_train_kernel.
constexpr int kHeadKeep = 32;            // rows FOUR lane groups per graph would keep each (128 rows per graph in registers)
#ifndef QOT_HEAD_THREADS
#define QOT_HEAD_THREADS 512       /* measured at cfg2: 256 threads 20.4 us, 512 17.3 us, 1024 23.8 us (64 B/lane of scratch) */
#endif
constexpr int kHeadThreads = QOT_HEAD_THREADS;
#ifdef QOT_DIAG
__device__ int g_head_variant;           // ablation bits (tools/bench_head.py): 1 no pool backward, 2 no dense phases, 4 no row loads
#define HD_VAR(bit) (hd_var & (bit))
#else
#define HD_VAR(bit) 0
#endif

template <int H, int NT>
__global__ __launch_bounds__(NT) void head_train_batched_kernel(
    const float* __restrict__ x, const int32_t* __restrict__ ptr, const float* __restrict__ w0,
    const float* __restrict__ b0, const float* __restrict__ w3, const float* __restrict__ b3,
    const float* __restrict__ target, float beta, float inv_n, float* __restrict__ out, float* __restrict__ grad_out,
    float* __restrict__ loss_rows, float* __restrict__ gx, float* __restrict__ partials, int64_t B, int O, ActParams act,
    int fold, ActParams in_act) {
    constexpr int TPR = H / 4, RPB = NT / TPR;            // lane groups of the workgroup
    constexpr int GB = 256 / H;                           // graphs per pass (GB * H = 256 dense outputs)
    constexpr int SPG = RPB / GB;                         // lane groups per graph (16 at NT = 1024)
    constexpr int KS = NT / 256;                          // threads per dense output (adjacent lanes: DPP sums)
    constexpr int KR = kHeadKeep * 4 / SPG;               // rows a lane group keeps (8 at NT = 1024): 128 rows per graph
    constexpr int PER = H * H / NT;                       // gW0 entries per thread
    constexpr int HP = H + 1;
    static_assert(KS == 1 || KS == 2 || KS == 4, "dense outputs are summed inside a quad");
    static_assert(H % (4 * KS) == 0 && PER >= 1, "width");
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    float* w0s = dyn;                         // [H][H + 1]
    float* w3s = w0s + H * HP;                // [8][H]
    __shared__ float4 cred[NT > 640 ? NT : 640];          // lane-group partials; at the end [O + 2][256] floats
    __shared__ __attribute__((aligned(16))) float pp[GB][H];
    __shared__ __attribute__((aligned(16))) float hh[GB][H];
    __shared__ __attribute__((aligned(16))) float gh[GB][H];
    __shared__ __attribute__((aligned(16))) float gp[GB][H];
    __shared__ float go[GB][8];
    __shared__ float lrow[GB][8];
    __shared__ int pb[GB + 1];
    const int t = threadIdx.x;
#ifdef QOT_DIAG
    const int hd_var = g_head_variant;
#endif
    const int sub = t % TPR, slot = t / TPR;
    const int gs = slot / SPG, sr = slot % SPG;           // my lane group's graph (of the pass) and its share of the rows
    const int part = t % KS, od = t / KS;                 // dense phases: output od = (g, o), inner-index share `part`
    const int g = od / H, o = od % H;
    const int64_t npass = (B + GB - 1) / GB;
    // the first pass's graph boundaries are requested together with the weights (one round trip instead of two)
    int pb_first = 0;
    if (t <= GB && (int64_t)blockIdx.x < npass) {
        const int64_t bb = (int64_t)blockIdx.x * GB + t;
        pb_first = ptr[bb < B ? bb : B];
    }
    for (int e = t; e < H * H; e += NT) w0s[(e / H) * HP + e % H] = w0[e];
    for (int e = t; e < O * H; e += NT) w3s[e] = w3[e];
    const uint64_t stepv = act.thr16 ? (uint64_t)act.step[0] : 0;
    const uint64_t in_stepv = (fold && in_act.thr16) ? (uint64_t)in_act.step[0] : 0;
    const float b0o = b0[o];
    float aw0[PER];
#pragma unroll
    for (int e = 0; e < PER; ++e) aw0[e] = 0.f;
    float ab0p = 0.f, aw3p[8], ab3p = 0.f;                // (held by the part == 0 lane of an output)
#pragma unroll
    for (int e = 0; e < 8; ++e) aw3p[e] = 0.f;
    float4 acs = f4zero();
    for (int64_t ps = blockIdx.x; ps < npass; ps += gridDim.x) {
        const int64_t b0g = ps * GB;
        __syncthreads();                                  // the previous pass is done with pb / gp
        if (t <= GB) {
            const int64_t bb = b0g + t;
            pb[t] = ps == (int64_t)blockIdx.x ? pb_first : ptr[bb < B ? bb : B];
        }
        __syncthreads();
        // ---- pool: my lane group's rows of its graph, kept in registers
        const bool live = b0g + gs < B;
        const int beg = pb[gs], cnt = live ? pb[gs + 1] - beg : 0;
        float4 v[KR];
        float4 acc = f4zero();
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            const int r = sr + SPG * k;
            v[k] = (r < cnt && !HD_VAR(4)) ? ld4(x + (int64_t)(beg + r) * H + 4 * sub) : f4zero();
        }
#pragma unroll
        for (int k = 0; k < KR; ++k) acc = add4(acc, v[k]);
        for (int r = sr + SPG * KR; r < cnt; r += SPG) acc = add4(acc, ld4(x + (int64_t)(beg + r) * H + 4 * sub));
        cred[t] = acc;
        __syncthreads();
        if (t < GB * TPR) {
            const int gg = t / TPR, sb = t % TPR;
            float4 s4 = cred[(gg * SPG) * TPR + sb];
            for (int q = 1; q < SPG; ++q) s4 = add4(s4, cred[(gg * SPG + q) * TPR + sb]);
            const int c2 = pb[gg + 1] - pb[gg];
            s4 = scale4(1.0f / (float)(c2 > 1 ? c2 : 1), s4);
            pp[gg][4 * sb] = s4.x; pp[gg][4 * sb + 1] = s4.y; pp[gg][4 * sb + 2] = s4.z; pp[gg][4 * sb + 3] = s4.w;
        }
        __syncthreads();
        // ---- Linear -> LeakyReLU -> Dropout: output o of graph g, the inner index split over KS adjacent lanes
        const int64_t bg = b0g + g;
        float hv = 0.f, dact = 0.f;
        {
            float a0 = 0.f;
            if (!HD_VAR(2)) {
#pragma unroll
                for (int a = part * (H / KS); a < (part + 1) * (H / KS); ++a) a0 = fmaf(w0s[o * HP + a], pp[g][a], a0);
            }
            if (KS >= 2) a0 += dpp_move<0xB1>(a0);
            if (KS >= 4) a0 += dpp_move<0x4E>(a0);
            const float vv = a0 + b0o;
            float d = 1.0f;
            if (act.thr16) {
                const uint64_t flat = (uint64_t)(bg * H + o);
                const uint64_t z = act_hash64(act.seed, stepv, flat >> 2);
                const bool keep = ((uint32_t)(z >> (16 * (flat & 3))) & 0xFFFFu) >= act.thr16;
                d = keep ? act.keep_scale : 0.f;
            }
            d *= (vv > 0.f) ? 1.0f : act.slope;
            hv = vv * d;
            dact = d;
            if (part == 0) hh[g][o] = hv;
        }
        __syncthreads();
        // ---- Linear(H, O), criterion: 16 lanes per (graph, output): a DPP row each
        for (int q = t / 16; q < GB * O; q += NT / 16) {
            const int gg = q / O, o3 = q % O, l16 = t % 16;
            const int64_t bb = b0g + gg;
            float pv = 0.f;
#pragma unroll
            for (int a = l16 * (H / 16); a < (l16 + 1) * (H / 16); ++a) pv = fmaf(w3s[o3 * H + a], hh[gg][a], pv);
            pv = group_sum<16>(pv);
            if (l16 == 0) {
                if (bb < B) {
                    const float vv = pv + b3[o3];
                    out[bb * O + o3] = vv;
                    const float df = vv - target[bb * O + o3];
                    const float ad = fabsf(df);
                    float l, gq;
                    if (ad < beta) { l = 0.5f * df * df / beta; gq = df / beta; }
                    else           { l = ad - 0.5f * beta;      gq = df > 0.f ? 1.f : -1.f; }
                    gq *= inv_n;
                    grad_out[bb * O + o3] = gq;
                    go[gg][o3] = gq;
                    lrow[gg][o3] = l * inv_n;
                } else {
                    go[gg][o3] = 0.f;
                    lrow[gg][o3] = 0.f;
                }
            }
        }
        __syncthreads();
        if (t < GB && b0g + t < B) {
            float sl = lrow[t][0];
            for (int o3 = 1; o3 < O; ++o3) sl += lrow[t][o3];
            loss_rows[b0g + t] = sl;
        }
        if (t < GB * O) ab3p += go[t / O][t % O];
        // ---- backward of the two Linear layers
        {
            float vv = 0.f;
            for (int o3 = 0; o3 < O; ++o3) vv = fmaf(w3s[o3 * H + o], go[g][o3], vv);
            const float g1 = bg < B ? vv * dact : 0.f;
            if (part == 0) {
                gh[g][o] = g1;
                ab0p += g1;
#pragma unroll
                for (int o3 = 0; o3 < 8; ++o3) if (o3 < O) aw3p[o3] = fmaf(go[g][o3], hv, aw3p[o3]);       // gW3[o3, o]
            }
        }
        __syncthreads();
        {
            float a0 = 0.f;
            if (!HD_VAR(2)) {
#pragma unroll
                for (int oo = part * (H / KS); oo < (part + 1) * (H / KS); ++oo) a0 = fmaf(w0s[oo * HP + o], gh[g][oo], a0);
            }
            if (KS >= 2) a0 += dpp_move<0xB1>(a0);
            if (KS >= 4) a0 += dpp_move<0x4E>(a0);
            if (part == 0) gp[g][o] = a0;
        }
        if (!HD_VAR(2)) {
#pragma unroll
            for (int gg = 0; gg < GB; ++gg)
#pragma unroll
                for (int e = 0; e < PER; ++e) {
                    const int idx = t * PER + e;
                    aw0[e] = fmaf(gh[gg][idx / H], pp[gg][idx % H], aw0[e]);
                }
        }
        __syncthreads();
        // ---- pool backward: my rows get gp / count, through the producer's activation when folded
        if (cnt > 0 && !HD_VAR(1)) {
            const float inv = 1.0f / (float)(cnt > 1 ? cnt : 1);
            const float4 gv = make_float4(gp[gs][4 * sub] * inv, gp[gs][4 * sub + 1] * inv, gp[gs][4 * sub + 2] * inv,
                                          gp[gs][4 * sub + 3] * inv);
            auto back = [&](int r, float4 yy) {
                const int64_t flat = (int64_t)(beg + r) * H + 4 * sub;
                if (!fold) { st4(gx + flat, gv); return; }
                uint64_t z = 0;
                if (in_act.thr16) z = act_hash64(in_act.seed, in_stepv, (uint64_t)flat >> 2);
                float vi[4] = {gv.x, gv.y, gv.z, gv.w};
                const float vr[4] = {yy.x, yy.y, yy.z, yy.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool keep = in_act.thr16 ? (((uint32_t)(z >> (16 * c)) & 0xFFFFu) >= in_act.thr16) : true;
                    vi[c] = vi[c] * (keep ? in_act.keep_scale : 0.f) * (vr[c] > 0.f ? 1.0f : in_act.slope);
                }
                const float4 o4 = make_float4(vi[0], vi[1], vi[2], vi[3]);
                st4(gx + flat, o4);
                acs = add4(acs, o4);
            };
#pragma unroll
            for (int k = 0; k < KR; ++k) {
                const int r = sr + SPG * k;
                if (r < cnt) back(r, v[k]);
            }
            for (int r = sr + SPG * KR; r < cnt; r += SPG) back(r, fold ? ld4(x + (int64_t)(beg + r) * H + 4 * sub) : f4zero());
        }
    }
    // ---- the workgroup's partial row; the per-thread sums over graphs / lane groups meet in LDS in a fixed order
    float* part_row = partials + (int64_t)blockIdx.x * (H * H + H + O * H + O + (fold ? H : 0));
#pragma unroll
    for (int e = 0; e < PER; ++e) part_row[t * PER + e] = aw0[e];
    __syncthreads();
    float* red = reinterpret_cast<float*>(cred);          // [O + 2][256] floats: one per dense output (g, o) and array
    if (part == 0) {
        red[od] = ab0p;
#pragma unroll
        for (int o3 = 0; o3 < 8; ++o3) if (o3 < O) red[(1 + o3) * 256 + od] = aw3p[o3];
    }
    if (t < GB * O) red[9 * 256 + t] = ab3p;               // threads (graph, output)
    __syncthreads();
    if (t < H) {
        float s2 = red[t];
#pragma unroll
        for (int gg = 1; gg < GB; ++gg) s2 += red[gg * H + t];
        part_row[H * H + t] = s2;
        for (int o3 = 0; o3 < O; ++o3) {
            float s3 = red[(1 + o3) * 256 + t];
#pragma unroll
            for (int gg = 1; gg < GB; ++gg) s3 += red[(1 + o3) * 256 + gg * H + t];
            part_row[H * H + H + o3 * H + t] = s3;
        }
    }
    if (t < O) {
        float s2 = red[9 * 256 + t];
        for (int gg = 1; gg < GB; ++gg) s2 += red[9 * 256 + gg * O + t];
        part_row[H * H + H + O * H + t] = s2;
    }
    if (fold) {
        __syncthreads();
        cred[t] = acs;
        __syncthreads();
        if (t < TPR) {
            float4 s4 = cred[t];
            for (int k2 = 1; k2 < RPB; ++k2) s4 = add4(s4, cred[k2 * TPR + t]);
            st4(part_row + H * H + H + O * H + O + 4 * t, s4);
        }
    }
}

// out[t] = sum over the nblk workgroup partials (fixed order).  1024 threads = 64 consecutive outputs x 16 part
// groups: every load is a 256-B row segment (the one-wave-per-output form read each partial with a stride of
// n floats: 12.8 us for 9 MB), 8 independent loads in flight per thread, part groups meet in LDS.
__global__ __launch_bounds__(1024) void head_partial_sum_kernel(const float* __restrict__ partials, int nblk, int n,
                                                                float* __restrict__ out) {
    __shared__ float red[16][64];
    const int o = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + o;
    float acc = 0.f;
    if (t < n) {
        const int per = (nblk + 15) / 16;
        const int p0 = sg * per, p1 = (p0 + per < nblk) ? p0 + per : nblk;
        int p = p0;
        for (; p + 8 <= p1; p += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partials[(int64_t)(p + u) * n + t];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; p < p1; ++p) acc += partials[(int64_t)p * n + t];
    }
    red[sg][o] = acc;
    __syncthreads();
    if (sg == 0 && t < n) {
        float s = red[0][o];
#pragma unroll
        for (int q = 1; q < 16; ++q) s += red[q][o];
        out[t] = s;
    }
}

}  // namespace qot

using namespace qot;

#define QOT_HEAD_H(H, ...)                                            \
    switch (H) {                                                      \
        case 16:  { constexpr int kH = 16;  __VA_ARGS__; } break;     \
        case 32:  { constexpr int kH = 32;  __VA_ARGS__; } break;     \
        case 64:  { constexpr int kH = 64;  __VA_ARGS__; } break;     \
        case 128: { constexpr int kH = 128; __VA_ARGS__; } break;     \
        default: return QOT_ERR_UNSUPPORTED;                          \
    }

// The read-out of a train step in one kernel (head_train_kernel): forward + SmoothL1(mean, beta) + backward.  out[B,O],
// grad_out[B,O] = d loss / d out, loss_rows[B] (the loss is their sum), grad_x[N,H] (wrt the CONV output when fold != 0: x is
// then y = dropout(leaky_relu(conv)) with the in_* parameters), workspace: the per-workgroup parameter-gradient partials
// [qot_head_bwd_blocks(B)][H*H + H + O*H + O (+ H)] the caller sums (QOT_ROLE_SUM_ROWS), as qot_head_bwd(grads = NULL).
extern "C" int qot_head_train(const float* x, const int32_t* ptr, const float* w0, const float* b0, const float* w3,
                              const float* b3, const float* target, float beta, float* out, float* grad_out,
                              float* loss_rows, float* grad_x, float* workspace, int64_t B, int H, int O, float slope,
                              float p, uint64_t seed, const int64_t* step_counter, int fold, float in_slope, float in_p,
                              uint64_t in_seed, const int64_t* in_step, qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (B <= 0 || O <= 0 || !(beta > 0.f)) return QOT_ERR_BADARG;
    if (O > 8) return QOT_ERR_UNSUPPORTED;
    if (!x || !ptr || !w0 || !b0 || !w3 || !b3 || !target || !out || !grad_out || !loss_rows || !grad_x || !workspace)
        return QOT_ERR_BADARG;
    const ActParams ap = make_act(1, slope, p, seed, step_counter);
    const ActParams in_ap = make_act(fold ? 1 : 0, in_slope, in_p, in_seed, in_step);
    int blocks = kHeadBwdBlocks;
    if (B < blocks) blocks = (int)B;
    const float inv_n = 1.0f / ((float)B * (float)O);
    if (H <= 64) {
        // the batched form: 256 / H graphs per pass, one workgroup per CU's worth of passes
        const int gbn = 256 / H;
        int64_t nb = (B + gbn - 1) / gbn;
        if (nb > kHeadBwdBlocks) nb = kHeadBwdBlocks;
        blocks = (int)nb;
        QOT_HEAD_H(H, {
            if constexpr (kH <= 64) {
                const size_t lds = (size_t)(kH * (kH + 1) + 8 * kH) * sizeof(float);
                constexpr int NT = kH * kH < kHeadThreads ? 256 : kHeadThreads;      // (H = 16: 256 entries of gW0)
                head_train_batched_kernel<kH, NT><<<blocks, NT, lds, stream>>>(
                    x, ptr, w0, b0, w3, b3, target, beta, inv_n, out, grad_out, loss_rows, grad_x, workspace, B, O, ap, fold,
                    in_ap);
            }
        });
        QOT_LAUNCH_CHECK();
        return QOT_OK;
    }
    QOT_HEAD_H(H, {
        const size_t lds = (size_t)(kHeadRowsCap * kH + kH * (kH + 1) + 8 * kH) * sizeof(float);
        static size_t allowed[kMaxDevices];                 // per device: the attribute is
        const int lrc = ensure_dyn_lds(reinterpret_cast<const void*>(head_train_kernel<kH>), lds, allowed);
        if (lrc != QOT_OK) return lrc;
        head_train_kernel<kH><<<blocks, 256, lds, stream>>>(x, ptr, w0, b0, w3, b3, target, beta, inv_n, out, grad_out,
                                                            loss_rows, grad_x, workspace, B, O, ap, fold, in_ap);
    });
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

// workgroups (= partial rows in the workspace) of qot_head_train for B graphs at width H
extern "C" int qot_head_train_blocks(int64_t B, int H) {
    if (B <= 0) return 0;
    if (H <= 64 && H >= 16) {
        const int gbn = 256 / H;
        const int64_t nb = (B + gbn - 1) / gbn;
        return nb > kHeadBwdBlocks ? kHeadBwdBlocks : (int)nb;
    }
    return B < kHeadBwdBlocks ? (int)B : kHeadBwdBlocks;
}

#ifdef QOT_DIAG
extern "C" void qot_debug_head_variant(int v) { (void)hipMemcpyToSymbol(HIP_SYMBOL(qot::g_head_variant), &v, sizeof(int)); }
#endif

extern "C" int qot_head_fwd_loss(const float* x, const int32_t* ptr, const float* w0, const float* b0,
                                 const float* w3, const float* b3, float* pooled, float* hidden, float* out,
                                 int64_t B, int H, int O, float slope, float p, uint64_t seed,
                                 const int64_t* step_counter, const float* target, float beta, float* grad_out,
                                 float* loss_rows, qot_stream_t stream) {
    if (B < 0 || O <= 0 || O > 8) return (O > 8) ? QOT_ERR_UNSUPPORTED : QOT_ERR_BADARG;
    if (B == 0) return QOT_OK;
    if (!x || !ptr || !w0 || !b0 || !w3 || !b3 || !pooled || !hidden || !out) return QOT_ERR_BADARG;
    if (target && (!grad_out || !loss_rows || !(beta > 0.f))) return QOT_ERR_BADARG;
    const ActParams ap = make_act(1, slope, p, seed, step_counter);
    const float inv_n = 1.0f / ((float)B * (float)O);
    QOT_HEAD_H(H, head_fwd_kernel<kH><<<(int)B, 256, 0, (hipStream_t)stream>>>(x, ptr, w0, b0, w3, b3, pooled, hidden,
                                                                              out, O, ap, target, beta, inv_n, grad_out,
                                                                              loss_rows));
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_head_fwd(const float* x, const int32_t* ptr, const float* w0, const float* b0,
                            const float* w3, const float* b3, float* pooled, float* hidden, float* out,
                            int64_t B, int H, int O, float slope, float p, uint64_t seed,
                            const int64_t* step_counter, qot_stream_t stream) {
    return qot_head_fwd_loss(x, ptr, w0, b0, w3, b3, pooled, hidden, out, B, H, O, slope, p, seed, step_counter, nullptr,
                             1.0f, nullptr, nullptr, stream);
}

// workgroups qot_head_bwd launches for B graphs (= rows of its partials workspace)
extern "C" int qot_head_bwd_blocks(int64_t B) { return B < kHeadBwdBlocks ? (int)(B > 0 ? B : 0) : kHeadBwdBlocks; }

extern "C" size_t qot_head_bwd_workspace_floats(int H, int O) {
    return (size_t)kHeadBwdBlocks * (size_t)(H * H + 2 * H + O * H + O);
}

// grads: [gW0 (H*H) | gb0 (H) | gW3 (O*H) | gb3 (O) | column sums of grad_x (H, only with x_in)] contiguous
extern "C" int qot_head_bwd(const float* grad_out, const float* pooled, const float* hidden, const int32_t* ptr,
                            const float* w0, const float* w3, float* grad_x, float* grads, float* workspace,
                            int64_t B, int H, int O, float slope, float p, uint64_t seed,
                            const int64_t* step_counter, const float* x_in, float in_slope, float in_p,
                            uint64_t in_seed, const int64_t* in_step, qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (B <= 0 || O <= 0) return QOT_ERR_BADARG;
    if (O > 8) return QOT_ERR_UNSUPPORTED;
    if (!grad_out || !pooled || !hidden || !ptr || !w0 || !w3 || !grad_x || !workspace) return QOT_ERR_BADARG;
    const ActParams ap = make_act(1, slope, p, seed, step_counter);
    int blocks = kHeadBwdBlocks;
    if (B < blocks) blocks = (int)B;
    const ActParams in_ap = make_act(x_in ? 1 : 0, in_slope, in_p, in_seed, in_step);
    QOT_HEAD_H(H, head_bwd_kernel<kH><<<blocks, 256, 0, stream>>>(grad_out, pooled, hidden, ptr, w0, w3, grad_x,
                                                                  workspace, B, O, ap, x_in, in_ap));
    QOT_LAUNCH_CHECK();
    if (!grads) return QOT_OK;            // deferred: the caller sums the block partials (QOT_ROLE_SUM_ROWS)
    const int n = H * H + H + O * H + O + (x_in ? H : 0);
    head_partial_sum_kernel<<<grid_for(n, 64), 1024, 0, stream>>>(workspace, blocks, n, grads);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}
