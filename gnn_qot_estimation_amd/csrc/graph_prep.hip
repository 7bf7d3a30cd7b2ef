// Graph preparation: int64 edge_index -> int32 CSR-by-destination + CSC-by-source.
// Replaces the per-call index bookkeeping PyG's MessagePassing.propagate does for
// topological_training/models.py:53,57 and lightpath_training/models.py:30 (including
// GATConv's remove_self_loops/add_self_loops rebuild).  Counting sort + per-row ordering by
// edge id keeps the original edge order inside every destination and every source, so every
// later segmented reduction has a fixed summation order (bitwise reproducible).
#include "common.hpp"
#include "graph_prep_dev.hpp"
#include "mfma_tile.hpp"

namespace qot {

static inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

// ---- counting sort by destination (CSR) and by source (CSC) -------------------------------
// 1. degree histograms; the value each int atomic returns is the edge's (arbitrary) rank inside
//    its destination / source row, kept per edge                      (edge-parallel)
// 2. exclusive scans -> rowptr, rowptr_t
// 3. keys placed at rowptr[row] + rank: no second round of atomics    (edge-parallel)
// 4. one thread per row (N destination rows + N source rows in one launch) insertion-sorts its
//    keys by original edge id, so the order inside a row is the caller's edge order whatever the
//    atomics did: every later segmented sum has a fixed order (bitwise run-to-run reproducible).
//    Rows are short (degree ~4, <= 64 in the power-law config); the sort is O(d^2) per row.
// 5. pos_t[t] = CSR slot of out-edge t                                (edge-parallel)
// GAT mode: edges with j == i are dropped, node n gets key E + n (sorts last, eid -1).

__global__ void csr_hist_kernel(const int64_t* __restrict__ ei, int64_t E, int64_t N, int gat,
                                int32_t* __restrict__ cnt_in, int32_t* __restrict__ cnt_out,
                                int32_t* __restrict__ rank_in, int32_t* __restrict__ rank_out, int64_t cap) {
    int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= cap) return;
    int64_t i, j;
    if (e < E) {
        j = ei[e]; i = ei[E + e];
        if (gat && j == i) return;
    } else {
        i = j = e - E;
    }
    rank_in[e] = atomicAdd(&cnt_in[i], 1);
    rank_out[e] = atomicAdd(&cnt_out[j], 1);
}

__global__ void csr_place_kernel(const int64_t* __restrict__ ei, int64_t E, int gat,
                                 const int32_t* __restrict__ rowptr, const int32_t* __restrict__ rowptr_t,
                                 const int32_t* __restrict__ rank_in, const int32_t* __restrict__ rank_out,
                                 int32_t* __restrict__ key_in, int32_t* __restrict__ key_out, int64_t cap) {
    int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= cap) return;
    int64_t i, j;
    if (e < E) {
        j = ei[e]; i = ei[E + e];
        if (gat && j == i) return;
    } else {
        i = j = e - E;
    }
    key_in[rowptr[i] + rank_in[e]] = (int32_t)e;       // keys >= E are the appended self loops
    key_out[rowptr_t[j] + rank_out[e]] = (int32_t)e;
}

// threads [0, N): destination row i -> col / eid / row / slot_of / invdeg
// threads [N, 2N): source row j     -> col_t / eid_t, sorted keys stay in key_out (= pos_t buffer)
__global__ void csr_sort_emit_kernel(const int64_t* __restrict__ ei, int64_t E, int64_t N,
                                     const int32_t* __restrict__ rowptr, int32_t* __restrict__ key_in,
                                     int32_t* __restrict__ col, int32_t* __restrict__ eid, int32_t* __restrict__ row,
                                     int32_t* __restrict__ slot_of, float* __restrict__ invdeg,
                                     const int32_t* __restrict__ rowptr_t, int32_t* __restrict__ key_out,
                                     int32_t* __restrict__ col_t, int32_t* __restrict__ eid_t) {
    int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (r < N) {
        const int beg = rowptr[r], end = rowptr[r + 1];
        sort_row_keys(key_in, beg, end);
        for (int p = beg; p < end; ++p) {
            const int key = key_in[p];
            col[p] = (key < E) ? (int32_t)ei[key] : (int32_t)(key - E);
            eid[p] = (key < E) ? key : -1;
            row[p] = (int32_t)r;
            slot_of[key] = p;
        }
        const int d = end - beg;
        invdeg[r] = 1.0f / (float)(d > 1 ? d : 1);
    } else if (r < 2 * N) {
        const int64_t j = r - N;
        const int beg = rowptr_t[j], end = rowptr_t[j + 1];
        sort_row_keys(key_out, beg, end);
        for (int t = beg; t < end; ++t) {
            const int key = key_out[t];
            col_t[t] = (key < E) ? (int32_t)ei[E + key] : (int32_t)(key - E);
            eid_t[t] = (key < E) ? key : -1;
        }
    }
}

// live slots = rowptr_t[N] (cap minus the self loops GAT mode dropped); the tail is never touched
__global__ void csc_pos_kernel(int32_t* __restrict__ pos_t, const int32_t* __restrict__ slot_of,
                               const int32_t* __restrict__ rowptr_t, int64_t N) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < (int64_t)rowptr_t[N]) pos_t[t] = slot_of[pos_t[t]];
}

// ---- exclusive scan of two int32 arrays (in-degrees, out-degrees), plain kernels only --------
// (the prep runs inside a captured HIP graph every step: no library calls, no memset nodes)
constexpr int kScanChunk = 2048;        // elements per 256-thread block (8 per thread)

__global__ void zero_i32_kernel(int32_t* __restrict__ p, int64_t n) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0;
}

// phase A: per-chunk totals of both arrays -> bsum[which][chunk]
__global__ __launch_bounds__(256) void scan_chunk_sums_kernel(const int32_t* __restrict__ a,
                                                              const int32_t* __restrict__ b, int64_t n,
                                                              int32_t* __restrict__ bsum, int nchunks) {
    const int32_t* src = blockIdx.y ? b : a;
    const int64_t base = (int64_t)blockIdx.x * kScanChunk + threadIdx.x * 8;
    int s = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) if (base + u < n) s += src[base + u];
    int total;
    block_exclusive_scan_256(s, &total);
    if (threadIdx.x == 0) bsum[blockIdx.y * nchunks + blockIdx.x] = total;
}

// phase B: one block per array turns chunk totals into exclusive chunk offsets (in place)
__global__ __launch_bounds__(256) void scan_chunk_offsets_kernel(int32_t* __restrict__ bsum, int nchunks) {
    int32_t* p = bsum + blockIdx.x * nchunks;
    int carry = 0;
    for (int c0 = 0; c0 < nchunks; c0 += 256) {
        int idx = c0 + threadIdx.x;
        int v = idx < nchunks ? p[idx] : 0;
        int total;
        int ex = block_exclusive_scan_256(v, &total);
        if (idx < nchunks) p[idx] = carry + ex;
        carry += total;
    }
}

// phase C: exclusive scan inside each chunk + chunk offset -> out.  totals_only: bsum still holds
// the per-chunk TOTALS (phase B skipped, few chunks) and every block sums the ones before it.
__global__ __launch_bounds__(256) void scan_apply_kernel(const int32_t* __restrict__ a, const int32_t* __restrict__ b,
                                                         int64_t n, const int32_t* __restrict__ bsum, int nchunks,
                                                         int totals_only, int32_t* __restrict__ out_a,
                                                         int32_t* __restrict__ out_b) {
    const int32_t* src = blockIdx.y ? b : a;
    int32_t* dst = blockIdx.y ? out_b : out_a;
    const int64_t base = (int64_t)blockIdx.x * kScanChunk + threadIdx.x * 8;
    int v[8], s = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) { v[u] = (base + u < n) ? src[base + u] : 0; s += v[u]; }
    int offset;
    if (totals_only) {
        int part = 0;
        for (int c = threadIdx.x; c < (int)blockIdx.x; c += 256) part += bsum[blockIdx.y * nchunks + c];
        block_exclusive_scan_256(part, &offset);
    } else {
        offset = bsum[blockIdx.y * nchunks + blockIdx.x];
    }
    int total;
    int ex = block_exclusive_scan_256(s, &total) + offset;
#pragma unroll
    for (int u = 0; u < 8; ++u) { if (base + u < n) dst[base + u] = ex; ex += v[u]; }
}

__global__ void invdeg_kernel(const int32_t* __restrict__ rowptr, float* __restrict__ invdeg, int64_t N) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= N) return;
    int d = rowptr[i + 1] - rowptr[i];
    invdeg[i] = 1.0f / (float)(d > 1 ? d : 1);
}

__global__ void i32_gather_kernel(const int32_t* __restrict__ map, const int32_t* __restrict__ idx,
                                  int32_t* __restrict__ out, int64_t n) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = map[idx[i]];
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

}  // namespace qot

using namespace qot;

static inline int scan_chunks(int64_t n) { return (int)((n + kScanChunk - 1) / kScanChunk); }

// workspace: cnt_in[N+1] cnt_out[N+1] | rank_in rank_out key_in slot_of [cap each] | bsum[2*nchunks]
extern "C" size_t qot_csr_workspace_bytes(int64_t E, int64_t N, int gat_self_loops) {
    if (E < 0 || N < 0) return 0;
    int64_t cap = E + (gat_self_loops ? N : 0);
    return 2 * align256((size_t)(N + 1) * 4) + 4 * align256((size_t)(cap > 0 ? cap : 1) * 4) +
           align256((size_t)2 * scan_chunks(N + 1) * 4) + 256;
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
        zero_i32_kernel<<<grid_for(N + 1, T), T, 0, stream>>>(rowptr, N + 1);
        QOT_LAUNCH_CHECK();
        zero_i32_kernel<<<grid_for(N + 1, T), T, 0, stream>>>(rowptr_t, N + 1);
        QOT_LAUNCH_CHECK();
        if (N > 0) {
            invdeg_kernel<<<grid_for(N, T), T, 0, stream>>>(rowptr, invdeg, N);
            QOT_LAUNCH_CHECK();
        }
        return QOT_OK;
    }
    if (!edge_index && E > 0) return QOT_ERR_BADARG;
    if (!col || !eid || !row || !col_t || !pos_t || !eid_t || !workspace) return QOT_ERR_BADARG;
    const size_t seg = align256((size_t)(N + 1) * 4);
    const size_t cseg = align256((size_t)cap * 4);
    const int nchunks = scan_chunks(N + 1);
    const size_t need = 2 * seg + 4 * cseg + align256((size_t)2 * nchunks * 4);
    if (workspace_bytes < need) return QOT_ERR_BADARG;
    char* w = (char*)workspace;
    int32_t* cnt_in = (int32_t*)(w);
    int32_t* cnt_out = (int32_t*)(w + seg);
    int32_t* rank_in = (int32_t*)(w + 2 * seg);
    int32_t* rank_out = (int32_t*)(w + 2 * seg + cseg);
    int32_t* key_in = (int32_t*)(w + 2 * seg + 2 * cseg);
    int32_t* slot_of = (int32_t*)(w + 2 * seg + 3 * cseg);
    int32_t* bsum = (int32_t*)(w + 2 * seg + 4 * cseg);
    int32_t* key_out = pos_t;            // sorted keys live in the pos_t buffer until csc_pos_kernel

    zero_i32_kernel<<<grid_for((int64_t)(2 * seg / 4), T), T, 0, stream>>>((int32_t*)w, (int64_t)(2 * seg / 4));
    QOT_LAUNCH_CHECK();
    csr_hist_kernel<<<grid_for(cap, T), T, 0, stream>>>(edge_index, E, N, gat_self_loops, cnt_in, cnt_out, rank_in,
                                                        rank_out, cap);
    QOT_LAUNCH_CHECK();
    scan_chunk_sums_kernel<<<dim3(nchunks, 2), T, 0, stream>>>(cnt_in, cnt_out, N + 1, bsum, nchunks);
    QOT_LAUNCH_CHECK();
    const int totals_only = nchunks <= 2048;
    if (!totals_only) {
        scan_chunk_offsets_kernel<<<2, T, 0, stream>>>(bsum, nchunks);
        QOT_LAUNCH_CHECK();
    }
    scan_apply_kernel<<<dim3(nchunks, 2), T, 0, stream>>>(cnt_in, cnt_out, N + 1, bsum, nchunks, totals_only, rowptr,
                                                          rowptr_t);
    QOT_LAUNCH_CHECK();
    csr_place_kernel<<<grid_for(cap, T), T, 0, stream>>>(edge_index, E, gat_self_loops, rowptr, rowptr_t, rank_in,
                                                         rank_out, key_in, key_out, cap);
    QOT_LAUNCH_CHECK();
    csr_sort_emit_kernel<<<grid_for(2 * N, T), T, 0, stream>>>(edge_index, E, N, rowptr, key_in, col, eid, row, slot_of,
                                                               invdeg, rowptr_t, key_out, col_t, eid_t);
    QOT_LAUNCH_CHECK();
    csc_pos_kernel<<<grid_for(cap, T), T, 0, stream>>>(pos_t, slot_of, rowptr_t, N);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_csr_build_by_graph(const int64_t* edge_index, int64_t E, int64_t N, const int64_t* node_ptr,
                                      const int64_t* edge_ptr, int64_t B, int64_t max_nodes, int64_t max_edges,
                                      int32_t* rowptr, int32_t* col, int32_t* eid, int32_t* row,
                                      int32_t* rowptr_t, int32_t* col_t, int32_t* pos_t, int32_t* eid_t,
                                      float* invdeg, int32_t* status, const int64_t* node_ids, int32_t* ids32,
                                      int32_t* colf, int32_t* colf_t, int32_t* ptr32, qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (E < 0 || N < 0 || B < 0 || max_nodes < 0 || max_edges < 0 || !rowptr || !rowptr_t || !invdeg) return QOT_ERR_BADARG;
    if (B == 0) {
        if (N != 0 || E != 0) return QOT_ERR_BADARG;
        zero_i32_kernel<<<1, 64, 0, stream>>>(rowptr, 1);
        zero_i32_kernel<<<1, 64, 0, stream>>>(rowptr_t, 1);
        QOT_LAUNCH_CHECK();
        return QOT_OK;
    }
    // one workgroup per graph: csr_by_graph_body (graph_prep_dev.hpp) through the multi-role launch (roles.hip)
    qot_role_t r{};
    r.kind = QOT_ROLE_CSR_BY_GRAPH;
    const void* ptrs[18] = {edge_index, node_ptr, edge_ptr, rowptr, col, eid, row, rowptr_t, col_t, pos_t, eid_t, invdeg,
                            status, node_ids, ids32, colf, colf_t, ptr32};
    for (int k = 0; k < 18; ++k) r.p[k] = ptrs[k];
    r.i[0] = E; r.i[1] = N; r.i[2] = B; r.i[3] = max_nodes; r.i[4] = max_edges;
    return qot_run_roles(&r, 1, stream_);
}

// ---- GAT mode, block-diagonal batches of SMALL graphs: one wave per graph, one launch ---------------------------
// LightpathGNN's batches are tens of thousands of chain graphs of 2..20 nodes (lightpath_training/dataset.py); PyG
// rebuilds their self-looped edge list in every GATConv call (lightpath_training/models.py:30).  With the graphs' node
// and edge slices known and NO self loop in the input (the collate step's host-side check; one is flagged, bit 4 of
// status, never mis-built) graph b owns slots [e0 + n0, e0 + n0 + m + n) of the self-looped index whatever the other
// graphs hold, so a wave builds it alone from its own LDS: in-row order = edge id, the inserted self loop last (key
// E + i, eid -1) -- bit for bit qot_csr_build(gat_self_loops = 1).  n <= 64 (one lane per node), m <= 256.
constexpr int kGatSmallMaxN = 64, kGatSmallMaxM = 256;
constexpr int kGatSmallLds = 3 * kGatSmallMaxM + 2 * (kGatSmallMaxN + 1);

__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ int wave_exclusive_scan(int v, int lane, int* total) {
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    *total = __shfl(inc, 63);
    return inc - v;
}

__global__ __launch_bounds__(256) void csr_gat_small_kernel(
    const int64_t* __restrict__ ei, int64_t E, int64_t N, const int64_t* __restrict__ node_ptr,
    const int64_t* __restrict__ edge_ptr, int64_t B, int32_t* __restrict__ rowptr, int32_t* __restrict__ col,
    int32_t* __restrict__ eid, int32_t* __restrict__ row, int32_t* __restrict__ rowptr_t, int32_t* __restrict__ col_t,
    int32_t* __restrict__ pos_t, int32_t* __restrict__ eid_t, float* __restrict__ invdeg, int32_t* __restrict__ status,
    int32_t* __restrict__ ptr32) {
    __shared__ int lds[4][kGatSmallLds];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int* lsrc = lds[wave];
    int* ldst = lsrc + kGatSmallMaxM;
    int* rp = ldst + kGatSmallMaxM;
    int* rpt = rp + kGatSmallMaxN + 1;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    // A graph costs two dependent global round trips (its slices, then its edges) and a wave walks ~8 graphs: the NEXT
    // graph's slices and its first 64 edges are requested before the current graph is worked on (raw values only: no
    // arithmetic on them before the wait -- the compiler would put it right behind the load).
    int64_t b = (int64_t)blockIdx.x * 4 + wave;
    int64_t q_n0 = 0, q_n1 = 0, q_e0 = 0, q_e1 = 0, q_j = 0, q_i = 0;
    if (b < B) {
        q_n0 = node_ptr[b]; q_n1 = node_ptr[b + 1]; q_e0 = edge_ptr[b]; q_e1 = edge_ptr[b + 1];
        if (q_e0 >= 0 && q_e0 + lane < q_e1 && q_e1 <= E) { q_j = ei[q_e0 + lane]; q_i = ei[E + q_e0 + lane]; }
    }
    for (; b < B; b += nwaves) {
        const int64_t n0 = q_n0, e0 = q_e0;
        const int64_t nn = q_n1 - n0, mm = q_e1 - e0;
        const int64_t first_j = q_j, first_i = q_i;
        {
            const int64_t bn = b + nwaves;
            if (bn < B) {
                q_n0 = node_ptr[bn]; q_n1 = node_ptr[bn + 1]; q_e0 = edge_ptr[bn]; q_e1 = edge_ptr[bn + 1];
                if (q_e0 >= 0 && q_e0 + lane < q_e1 && q_e1 <= E) { q_j = ei[q_e0 + lane]; q_i = ei[E + q_e0 + lane]; }
            }
        }
        if (lane == 0) {
            if (ptr32) {
                ptr32[b] = (int32_t)n0;
                if (b == B - 1) ptr32[B] = (int32_t)node_ptr[B];
            }
            if (b == B - 1) { rowptr[N] = (int32_t)(E + N); rowptr_t[N] = (int32_t)(E + N); }
        }
        // host-side size bound violated, or slices that do not lie inside the arrays: flag, write nothing
        if (nn > kGatSmallMaxN || mm > kGatSmallMaxM || nn < 0 || mm < 0 || n0 < 0 || e0 < 0 || n0 + nn > N || e0 + mm > E) {
            if (lane == 0 && status) atomicOr(status, 2);
            continue;
        }
        const int n = (int)nn, m = (int)mm;
        const int64_t s0 = e0 + n0;
        bool bad = false, loop = false;
        for (int e = lane; e < m; e += 64) {
            int j = (int)((e < 64 ? first_j : ei[e0 + e]) - n0), i = (int)((e < 64 ? first_i : ei[E + e0 + e]) - n0);
            if (i < 0 || i >= n || j < 0 || j >= n) { bad = true; i = i < 0 ? 0 : (i >= n ? n - 1 : i); j = j < 0 ? 0 : (j >= n ? n - 1 : j); }
            if (i == j) loop = true;
            lsrc[e] = j;
            ldst[e] = i;
        }
        if (status) {
            if (bad) atomicOr(status, 1);
            if (loop) atomicOr(status, 4);
        }
        wave_lds_fence();
        // lane i = node i: degrees (the edge list is read as broadcasts), offsets, its self loop
        int din = 0, dout = 0;
        for (int e = 0; e < m; ++e) {
            din += (ldst[e] == lane) ? 1 : 0;
            dout += (lsrc[e] == lane) ? 1 : 0;
        }
        int tot;
        const int rpi = wave_exclusive_scan(lane < n ? din + 1 : 0, lane, &tot);
        const int rpti = wave_exclusive_scan(lane < n ? dout + 1 : 0, lane, &tot);
        if (lane < n) {
            rp[lane] = rpi;
            rpt[lane] = rpti;
            rowptr[n0 + lane] = (int32_t)(s0 + rpi);
            rowptr_t[n0 + lane] = (int32_t)(s0 + rpti);
            invdeg[n0 + lane] = 1.0f / (float)(din + 1);
            const int64_t p = s0 + rpi + din, pt = s0 + rpti + dout;
            col[p] = (int32_t)(n0 + lane);
            eid[p] = -1;
            row[p] = (int32_t)(n0 + lane);
            col_t[pt] = (int32_t)(n0 + lane);
            eid_t[pt] = -1;
            pos_t[pt] = (int32_t)p;
        }
        wave_lds_fence();
        // lane per edge: rank among the earlier edges of the same row = its place in edge-id order
        for (int e = lane; e < m; e += 64) {
            const int i = ldst[e], j = lsrc[e];
            int rin = 0, rout = 0;
            for (int f = 0; f < e; ++f) {
                rin += (ldst[f] == i) ? 1 : 0;
                rout += (lsrc[f] == j) ? 1 : 0;
            }
            const int64_t p = s0 + rp[i] + rin, pt = s0 + rpt[j] + rout;
            col[p] = (int32_t)(n0 + j);
            eid[p] = (int32_t)(e0 + e);
            row[p] = (int32_t)(n0 + i);
            col_t[pt] = (int32_t)(n0 + i);
            eid_t[pt] = (int32_t)(e0 + e);
            pos_t[pt] = (int32_t)p;
        }
        wave_lds_fence();                                  // the next graph's stores come after these reads
    }
}

extern "C" int qot_csr_gat_by_graph_supported(int64_t max_nodes, int64_t max_edges) {
    return max_nodes >= 0 && max_edges >= 0 && max_nodes <= kGatSmallMaxN && max_edges <= kGatSmallMaxM;
}

extern "C" int qot_csr_build_gat_by_graph(const int64_t* edge_index, int64_t E, int64_t N, const int64_t* node_ptr,
                                          const int64_t* edge_ptr, int64_t B, int64_t max_nodes, int64_t max_edges,
                                          int32_t* rowptr, int32_t* col, int32_t* eid, int32_t* row, int32_t* rowptr_t,
                                          int32_t* col_t, int32_t* pos_t, int32_t* eid_t, float* invdeg, int32_t* status,
                                          int32_t* ptr32, qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (E < 0 || N < 0 || B <= 0 || !edge_index || !node_ptr || !edge_ptr || !rowptr || !col || !eid || !row || !rowptr_t ||
        !col_t || !pos_t || !eid_t || !invdeg)
        return QOT_ERR_BADARG;
    if (!qot_csr_gat_by_graph_supported(max_nodes, max_edges) || E + N > 0x7fffffff) return QOT_ERR_UNSUPPORTED;
    int64_t wgs = (B + 3) / 4;
    const int64_t cap = 8 * (int64_t)num_cus();            // 32 waves per CU (3.6 KB of LDS each)
    if (wgs > cap) wgs = cap;
    csr_gat_small_kernel<<<(int)wgs, 256, 0, stream>>>(edge_index, E, N, node_ptr, edge_ptr, B, rowptr, col, eid, row,
                                                      rowptr_t, col_t, pos_t, eid_t, invdeg, status, ptr32);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_i32_gather(const int32_t* map, const int32_t* idx, int32_t* out, int64_t n, qot_stream_t stream) {
    if (n < 0 || (n > 0 && (!map || !idx || !out))) return QOT_ERR_BADARG;
    if (n == 0) return QOT_OK;
    i32_gather_kernel<<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>(map, idx, out, n);
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
