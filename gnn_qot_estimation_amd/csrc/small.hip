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

extern "C" int qot_table_project_fwd(const float* table, const float* wq, const float* bq, const float* wk,
                                     const float* bk, const float* wv, const float* bv, const float* ws,
                                     const float* bs, float* out, int V, int H, int64_t* step_counter,
                                     int64_t* step_snapshot, qot_stream_t stream) {
    if (V < 0 || (step_counter && !step_snapshot)) return QOT_ERR_BADARG;
    if (V == 0) return step_counter ? qot_step_advance(step_counter, step_snapshot, stream) : QOT_OK;
    qot_role_t r{};
    r.kind = QOT_ROLE_TABLE_PROJECT_FWD;
    const void* ptrs[12] = {table, wq, bq, wk, bk, wv, bv, ws, bs, out, step_counter, step_snapshot};
    for (int k = 0; k < 12; ++k) r.p[k] = ptrs[k];
    r.i[0] = V; r.i[1] = H;
    return qot_run_roles(&r, 1, stream);
}

extern "C" int qot_table_project_bwd(const float* grad_out, const float* table, const float* wq, const float* wk,
                                     const float* wv, const float* ws, float* grad_table, float* grad_w,
                                     float* grad_b, int V, int H, qot_stream_t stream) {
    qot_role_t r{};
    r.kind = QOT_ROLE_TABLE_PROJECT_BWD;
    const void* ptrs[9] = {grad_out, table, wq, wk, wv, ws, grad_table, grad_w, grad_b};
    for (int k = 0; k < 9; ++k) r.p[k] = ptrs[k];
    r.i[0] = V; r.i[1] = H;
    return qot_run_roles(&r, 1, stream);
}

extern "C" int qot_gather3(const float* s0, int64_t n0, const float* s1, int64_t n1, const float* s2,
                           const int32_t* idx, float* out, int64_t n, qot_stream_t stream) {
    if (n == 0 && n0 >= 0 && n1 >= 0) return QOT_OK;
    qot_role_t r{};
    r.kind = QOT_ROLE_GATHER3;
    const void* ptrs[5] = {s0, s1, s2, idx, out};
    for (int k = 0; k < 5; ++k) r.p[k] = ptrs[k];
    r.i[0] = n0; r.i[1] = n1; r.i[2] = n;
    return qot_run_roles(&r, 1, stream);
}
