// Graph preparation: int64 edge_index -> int32 CSR-by-destination + CSC-by-source.
// Replaces the per-call index bookkeeping PyG's MessagePassing.propagate does for
// topological_training/models.py:53,57 and lightpath_training/models.py:30 (including
// GATConv's remove_self_loops/add_self_loops rebuild).  Stable LSD radix sort (rocPRIM)
// keeps the original edge order inside every destination, so every later segmented
// reduction has a fixed summation order (bitwise reproducible, no atomics).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "common.hpp"

namespace qot {

static inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

static inline unsigned key_bits(int64_t N) {
    unsigned b = 1;
    while ((int64_t(1) << b) <= N) ++b;  // keys take values 0..N
    return b;
}

__global__ void csr_keys_kernel(const int64_t* __restrict__ ei, int64_t E, int64_t N, int gat,
                                uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, int64_t cap) {
    int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= cap) return;
    if (e < E) {
        int64_t j = ei[e], i = ei[E + e];
        keys[e] = (gat && j == i) ? (uint32_t)N : (uint32_t)i;
    } else {
        keys[e] = (uint32_t)(e - E);  // appended self loop of node e-E
    }
    vals[e] = (uint32_t)e;
}

__global__ void csr_fill_kernel(const int64_t* __restrict__ ei, int64_t E, int64_t N,
                                const uint32_t* __restrict__ skeys, const uint32_t* __restrict__ svals,
                                int32_t* __restrict__ col, int32_t* __restrict__ eid,
                                int32_t* __restrict__ row, uint32_t* __restrict__ keys2,
                                uint32_t* __restrict__ vals2, int64_t cap) {
    int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= cap) return;
    uint32_t key = skeys[p], v = svals[p];
    int32_t c, id;
    if ((int64_t)v < E) { c = (int32_t)ei[v]; id = (int32_t)v; }
    else                { c = (int32_t)((int64_t)v - E); id = -1; }
    bool live = (int64_t)key < N;
    col[p] = live ? c : 0;
    eid[p] = live ? id : -1;
    row[p] = (int32_t)key;
    keys2[p] = live ? (uint32_t)c : (uint32_t)N;
    vals2[p] = (uint32_t)p;
}

// rowptr[i] = first slot whose sorted key is >= i  (i in [0, N])
__global__ void lower_bound_kernel(const uint32_t* __restrict__ skeys, int64_t cap, int64_t N,
                                   int32_t* __restrict__ rowptr) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i > N) return;
    int64_t lo = 0, hi = cap;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((int64_t)skeys[mid] < i) lo = mid + 1; else hi = mid;
    }
    rowptr[i] = (int32_t)lo;
}

__global__ void csc_fill_kernel(const uint32_t* __restrict__ svals2, const int32_t* __restrict__ row,
                                const int32_t* __restrict__ eid, int32_t* __restrict__ col_t,
                                int32_t* __restrict__ pos_t, int32_t* __restrict__ eid_t, int64_t cap,
                                int64_t N) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= cap) return;
    int32_t p = (int32_t)svals2[t];
    int32_t r = row[p];
    pos_t[t] = p;
    eid_t[t] = eid[p];
    col_t[t] = (r < N) ? r : 0;
}

__global__ void invdeg_kernel(const int32_t* __restrict__ rowptr, float* __restrict__ invdeg, int64_t N) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= N) return;
    int d = rowptr[i + 1] - rowptr[i];
    invdeg[i] = 1.0f / (float)(d > 1 ? d : 1);
}

__global__ void i64_to_i32_kernel(const int64_t* __restrict__ in, int32_t* __restrict__ out, int64_t n) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)in[i];
}

__global__ void batch_ptr_kernel(const int32_t* __restrict__ batch, int64_t N, int64_t B,
                                 int32_t* __restrict__ ptr) {
    int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (b > B) return;
    int64_t lo = 0, hi = N;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((int64_t)batch[mid] < b) lo = mid + 1; else hi = mid;
    }
    ptr[b] = (int32_t)lo;
}

static hipError_t sort_temp_bytes(int64_t cap, unsigned bits, size_t& bytes) {
    bytes = 0;
    return rocprim::radix_sort_pairs(nullptr, bytes, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                     (uint32_t*)nullptr, (uint32_t*)nullptr, (unsigned)cap, 0u, bits,
                                     (hipStream_t)0);
}

}  // namespace qot

using namespace qot;

extern "C" size_t qot_csr_workspace_bytes(int64_t E, int64_t N, int gat_self_loops) {
    if (E < 0 || N < 0) return 0;
    int64_t cap = E + (gat_self_loops ? N : 0);
    if (cap <= 0) return 256;
    size_t temp = 0;
    if (sort_temp_bytes(cap, key_bits(N), temp) != hipSuccess) return 0;
    return 4 * align256((size_t)cap * 4) + align256(temp) + 256;
}

extern "C" int qot_csr_build(const int64_t* edge_index, int64_t E, int64_t N, int gat_self_loops,
                             int32_t* rowptr, int32_t* col, int32_t* eid, int32_t* row,
                             int32_t* rowptr_t, int32_t* col_t, int32_t* pos_t, int32_t* eid_t,
                             float* invdeg, void* workspace, size_t workspace_bytes, qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (E < 0 || N < 0 || !rowptr || !rowptr_t || !invdeg) return QOT_ERR_BADARG;
    if (N >= (int64_t(1) << 31) - 1 || E + N >= (int64_t(1) << 31) - 1) return QOT_ERR_UNSUPPORTED;
    int64_t cap = E + (gat_self_loops ? N : 0);
    const int T = 256;
    if (cap == 0) {
        QOT_HIP(hipMemsetAsync(rowptr, 0, (size_t)(N + 1) * 4, stream));
        QOT_HIP(hipMemsetAsync(rowptr_t, 0, (size_t)(N + 1) * 4, stream));
        if (N > 0) {
            invdeg_kernel<<<grid_for(N, T), T, 0, stream>>>(rowptr, invdeg, N);
            QOT_LAUNCH_CHECK();
        }
        return QOT_OK;
    }
    if (!edge_index && E > 0) return QOT_ERR_BADARG;
    if (!col || !eid || !row || !col_t || !pos_t || !eid_t || !workspace) return QOT_ERR_BADARG;
    unsigned bits = key_bits(N);
    size_t temp = 0;
    QOT_HIP(sort_temp_bytes(cap, bits, temp));
    size_t seg = align256((size_t)cap * 4);
    if (workspace_bytes < 4 * seg + align256(temp)) return QOT_ERR_BADARG;
    char* w = (char*)workspace;
    uint32_t* ka = (uint32_t*)(w);
    uint32_t* kb = (uint32_t*)(w + seg);
    uint32_t* va = (uint32_t*)(w + 2 * seg);
    uint32_t* vb = (uint32_t*)(w + 3 * seg);
    void* tmp = (void*)(w + 4 * seg);

    csr_keys_kernel<<<grid_for(cap, T), T, 0, stream>>>(edge_index, E, N, gat_self_loops, ka, va, cap);
    QOT_LAUNCH_CHECK();
    QOT_HIP(rocprim::radix_sort_pairs(tmp, temp, ka, kb, va, vb, (unsigned)cap, 0u, bits, stream));
    lower_bound_kernel<<<grid_for(N + 1, T), T, 0, stream>>>(kb, cap, N, rowptr);
    QOT_LAUNCH_CHECK();
    // reuse ka/va as the second sort's input
    csr_fill_kernel<<<grid_for(cap, T), T, 0, stream>>>(edge_index, E, N, kb, vb, col, eid, row, ka, va, cap);
    QOT_LAUNCH_CHECK();
    QOT_HIP(rocprim::radix_sort_pairs(tmp, temp, ka, kb, va, vb, (unsigned)cap, 0u, bits, stream));
    lower_bound_kernel<<<grid_for(N + 1, T), T, 0, stream>>>(kb, cap, N, rowptr_t);
    QOT_LAUNCH_CHECK();
    csc_fill_kernel<<<grid_for(cap, T), T, 0, stream>>>(vb, row, eid, col_t, pos_t, eid_t, cap, N);
    QOT_LAUNCH_CHECK();
    invdeg_kernel<<<grid_for(N, T), T, 0, stream>>>(rowptr, invdeg, N);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_i64_to_i32(const int64_t* in, int32_t* out, int64_t n, qot_stream_t stream) {
    if (n < 0 || (n > 0 && (!in || !out))) return QOT_ERR_BADARG;
    if (n == 0) return QOT_OK;
    i64_to_i32_kernel<<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>(in, out, n);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_batch_ptr(const int32_t* batch, int64_t N, int64_t B, int32_t* ptr,
                             qot_stream_t stream) {
    if (N < 0 || B < 0 || !ptr || (N > 0 && !batch)) return QOT_ERR_BADARG;
    batch_ptr_kernel<<<grid_for(B + 1, 256), 256, 0, (hipStream_t)stream>>>(batch, N, B, ptr);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}
