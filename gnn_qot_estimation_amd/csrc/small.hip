// Launch-count reducers for the step's tail.  At batch 1024 a HIP-graph replay of the train step
// is ~60 kernels; every dependent launch costs ~5 us whatever it computes, so the small dense
// pieces around the convolutions are each folded into one kernel:
//   * SmoothL1 (mean) loss value AND its gradient in one pass (train.py:69,113-115);
//   * embedding-table projection q|k|v|skip = table @ W^T + b from the four Linear parameters
//     in place (no concatenation), and its whole backward;
//   * one gather from three parameter tensors into the fragment-ordered NNConv operands.
#include "common.hpp"

namespace qot {

// ---- SmoothL1Loss(reduction='mean', beta) : loss and d loss / d pred -------------------------
// partials[gridDim.x]; counter[0] must be 0 on entry and is left 0 (the last block to arrive sums
// the partials in index order -> bitwise reproducible, no memset per call).
__global__ __launch_bounds__(256) void smooth_l1_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                        int64_t n, float beta, float inv_n, float* __restrict__ grad,
                                                        float* __restrict__ loss, float* __restrict__ partials,
                                                        unsigned int* __restrict__ counter) {
    __shared__ float red[4];
    __shared__ bool last;
    float s = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = pred[i] - tgt[i];
        const float ad = fabsf(d);
        float l, g;
        if (ad < beta) { l = 0.5f * d * d / beta; g = d / beta; }
        else           { l = ad - 0.5f * beta;    g = d > 0.f ? 1.f : -1.f; }
        s += l;
        grad[i] = g * inv_n;
    }
    s = group_sum<64>(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
        __threadfence();
        last = atomicAdd(counter, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last && threadIdx.x < 64) {
        __threadfence();
        float t = 0.f;
        for (int b = threadIdx.x; b < (int)gridDim.x; b += 64) t += __hip_atomic_load(partials + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = group_sum<64>(t);
        if (threadIdx.x == 0) { loss[0] = t * inv_n; counter[0] = 0u; }
    }
}

// ---- table projection ------------------------------------------------------------------------
struct Proj4 {
    const float* w[4];      // each [H, H] (out, in)
    const float* b[4];      // each [H]
};

// out[v, s*H + o] = b_s[o] + sum_a table[v, a] * w_s[o, a];  one block per table row, 4H threads
template <int H>
__global__ __launch_bounds__(4 * H) void table_project_fwd_kernel(const float* __restrict__ table, Proj4 p,
                                                                  float* __restrict__ out, int64_t* __restrict__ counter,
                                                                  int64_t* __restrict__ snapshot) {
    __shared__ float row[H];
    const int v = blockIdx.x, c = threadIdx.x;
    if (counter && v == 0 && c == 0) {        // as step_advance_kernel: this is the forward's first launch in table mode
        const int64_t cc = counter[0] + 1;
        counter[0] = cc;
        snapshot[0] = cc;
    }
    if (c < H) row[c] = table[(int64_t)v * H + c];
    __syncthreads();
    const int s = c / H, o = c % H;
    const float* w = p.w[s] + (int64_t)o * H;
    float acc = p.b[s][o];
#pragma unroll
    for (int a = 0; a < H; a += 4) {
        const float4 ww = ld4(w + a);
        acc = fmaf(ww.x, row[a], acc); acc = fmaf(ww.y, row[a + 1], acc);
        acc = fmaf(ww.z, row[a + 2], acc); acc = fmaf(ww.w, row[a + 3], acc);
    }
    out[(int64_t)v * 4 * H + c] = acc;
}

// blocks [0, 4H): weight + bias gradient of packed row c (= s*H + o):  gw[c, a] = sum_v gp[v, c] table[v, a]
// blocks [4H, 4H + V): table gradient row v:                          gt[v, a] = sum_c gp[v, c] w_{s(c)}[o(c), a]
// grads: gw [4H, H] | gb [4H]   (packed q|k|v|skip order).  256 threads = H columns x PH phases of the
// reduction index (the loops are pure latency otherwise), phases meet in LDS in a fixed order.
template <int H>
__global__ __launch_bounds__(256) void table_project_bwd_kernel(const float* __restrict__ gp, const float* __restrict__ table,
                                                                Proj4 p, float* __restrict__ gtable,
                                                                float* __restrict__ gw, float* __restrict__ gb, int V) {
    constexpr int PH = 256 / H;
    __shared__ float red[256];
    __shared__ float redb[256];
    const int a = threadIdx.x % H, ph = threadIdx.x / H;
    float acc = 0.f, sb = 0.f;
    if ((int)blockIdx.x < 4 * H) {
        const int c = blockIdx.x;
#pragma unroll 4
        for (int v = ph; v < V; v += PH) {
            const float g = gp[(int64_t)v * 4 * H + c];
            acc = fmaf(g, table[(int64_t)v * H + a], acc);
            sb += g;
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
        const int v = blockIdx.x - 4 * H;
        __shared__ float g[4 * H];
        for (int c = threadIdx.x; c < 4 * H; c += 256) g[c] = gp[(int64_t)v * 4 * H + c];
        __syncthreads();
#pragma unroll 4
        for (int c = ph; c < 4 * H; c += PH) {
            const int s = c / H, o = c % H;
            acc = fmaf(g[c], p.w[s][(int64_t)o * H + a], acc);
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
__global__ void gather3_kernel(const float* __restrict__ s0, int n0, const float* __restrict__ s1, int n1,
                               const float* __restrict__ s2, const int32_t* __restrict__ idx, float* __restrict__ out,
                               int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int v = idx[i];          // v < 0: a padding slot (zero)
    out[i] = v < 0 ? 0.f : (v < n0 ? s0[v] : (v < n0 + n1 ? s1[v - n0] : s2[v - n0 - n1]));
}

// ---- table-mode index maps in one launch: ids32 = (int) node_ids ; colf = ids[col] ; colf_t = ids[col_t]
__global__ void table_maps_kernel(const int64_t* __restrict__ ids, const int32_t* __restrict__ col,
                                  const int32_t* __restrict__ col_t, int32_t* __restrict__ ids32,
                                  int32_t* __restrict__ colf, int32_t* __restrict__ colf_t, int64_t N, int64_t E) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < N) ids32[t] = (int32_t)ids[t];
    if (t < E) {
        colf[t] = (int32_t)ids[col[t]];
        colf_t[t] = (int32_t)ids[col_t[t]];
    }
}

// ---- dropout step counter: counter += 1 ; snapshot = counter (the draw backward re-reads) ------
__global__ void step_advance_kernel(int64_t* __restrict__ counter, int64_t* __restrict__ snapshot) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int64_t c = counter[0] + 1;
        counter[0] = c;
        snapshot[0] = c;
    }
}

}  // namespace qot

using namespace qot;

extern "C" int qot_table_maps(const int64_t* node_ids, const int32_t* col, const int32_t* col_t, int32_t* ids32,
                              int32_t* colf, int32_t* colf_t, int64_t N, int64_t E, qot_stream_t stream) {
    if (N < 0 || E < 0) return QOT_ERR_BADARG;
    if (N == 0 && E == 0) return QOT_OK;
    if (!node_ids || !ids32 || (E > 0 && (!col || !col_t || !colf || !colf_t))) return QOT_ERR_BADARG;
    const int64_t n = N > E ? N : E;
    table_maps_kernel<<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>(node_ids, col, col_t, ids32, colf, colf_t, N, E);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_step_advance(int64_t* counter, int64_t* snapshot, qot_stream_t stream) {
    if (!counter || !snapshot) return QOT_ERR_BADARG;
    step_advance_kernel<<<1, 64, 0, (hipStream_t)stream>>>(counter, snapshot);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" size_t qot_smooth_l1_workspace_floats(void) { return 1024 + 1; }

// workspace: [0] = arrival counter (zero before the FIRST call; the kernel restores it), [1..] partials
extern "C" int qot_smooth_l1(const float* pred, const float* target, int64_t n, float beta, float* loss, float* grad,
                             float* workspace, qot_stream_t stream) {
    if (n <= 0 || beta <= 0.f) return QOT_ERR_BADARG;
    if (!pred || !target || !loss || !grad || !workspace) return QOT_ERR_BADARG;
    int grid = grid_for(n, 256 * 4);
    if (grid > 1024) grid = 1024;
    smooth_l1_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(pred, target, n, beta, 1.0f / (float)n, grad, loss,
                                                            workspace + 1, reinterpret_cast<unsigned int*>(workspace));
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

#define QOT_TABLE_H(H, ...)                                           \
    switch (H) {                                                      \
        case 16:  { constexpr int kH = 16;  __VA_ARGS__; } break;     \
        case 32:  { constexpr int kH = 32;  __VA_ARGS__; } break;     \
        case 64:  { constexpr int kH = 64;  __VA_ARGS__; } break;     \
        case 128: { constexpr int kH = 128; __VA_ARGS__; } break;     \
        case 256: { constexpr int kH = 256; __VA_ARGS__; } break;     \
        default: return QOT_ERR_UNSUPPORTED;                          \
    }

extern "C" int qot_table_project_fwd(const float* table, const float* wq, const float* bq, const float* wk,
                                     const float* bk, const float* wv, const float* bv, const float* ws,
                                     const float* bs, float* out, int V, int H, int64_t* step_counter,
                                     int64_t* step_snapshot, qot_stream_t stream) {
    if (V < 0 || (step_counter && !step_snapshot)) return QOT_ERR_BADARG;
    if (V == 0) return step_counter ? qot_step_advance(step_counter, step_snapshot, stream) : QOT_OK;
    if (!table || !wq || !bq || !wk || !bk || !wv || !bv || !ws || !bs || !out) return QOT_ERR_BADARG;
    Proj4 p{{wq, wk, wv, ws}, {bq, bk, bv, bs}};
    QOT_TABLE_H(H, table_project_fwd_kernel<kH><<<V, 4 * kH, 0, (hipStream_t)stream>>>(table, p, out, step_counter,
                                                                                       step_snapshot));
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_table_project_bwd(const float* grad_out, const float* table, const float* wq, const float* wk,
                                     const float* wv, const float* ws, float* grad_table, float* grad_w,
                                     float* grad_b, int V, int H, qot_stream_t stream) {
    if (V <= 0) return QOT_ERR_BADARG;
    if (!grad_out || !table || !wq || !wk || !wv || !ws || !grad_table || !grad_w || !grad_b) return QOT_ERR_BADARG;
    Proj4 p{{wq, wk, wv, ws}, {nullptr, nullptr, nullptr, nullptr}};
    QOT_TABLE_H(H, table_project_bwd_kernel<kH><<<4 * kH + V, 256, 0, (hipStream_t)stream>>>(
                       grad_out, table, p, grad_table, grad_w, grad_b, V));
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_gather3(const float* s0, int64_t n0, const float* s1, int64_t n1, const float* s2,
                           const int32_t* idx, float* out, int64_t n, qot_stream_t stream) {
    if (n < 0 || n0 < 0 || n1 < 0 || n0 + n1 > 0x7fffffff) return QOT_ERR_BADARG;
    if (n == 0) return QOT_OK;
    if (!s0 || !s1 || !s2 || !idx || !out) return QOT_ERR_BADARG;
    gather3_kernel<<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>(s0, (int)n0, s1, (int)n1, s2, idx, out, n);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}
