// Device-side pieces of the graph preparation that more than one launch uses (graph_prep.hip's own kernels and the
// multi-role launch of roles.hip): per-row key sort, 256-thread block scan, and the one-workgroup-per-graph index build.
#pragma once
#include "common.hpp"

namespace qot {

__device__ __forceinline__ void sort_row_keys(int32_t* __restrict__ k, int beg, int end) {
    for (int a = beg + 1; a < end; ++a) {
        int key = k[a];
        int b = a - 1;
        while (b >= beg && k[b] > key) { k[b + 1] = k[b]; --b; }
        k[b + 1] = key;
    }
}


template <int NT>
__device__ __forceinline__ int block_exclusive_scan(int v, int* total) {
    __shared__ int wsum[NT / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
        if (w < wave) base += wsum[w];
        tot += wsum[w];
    }
    *total = tot;
    __syncthreads();
    return base + inc - v;
}
__device__ __forceinline__ int block_exclusive_scan_256(int v, int* total) { return block_exclusive_scan<256>(v, total); }


// ---- block-diagonal batches: the whole index in ONE launch, one workgroup per graph ----------
// A collated batch keeps every graph's nodes AND edges contiguous (PyG keeps the same slices), so
// graph b's CSR/CSC slots are exactly [edge_ptr[b], edge_ptr[b+1]): histogram, scan, placement, the
// per-row ordering by edge id and the CSC mapping all happen in that workgroup's LDS -- no global
// atomics, no cross-workgroup scan, no workspace.  Same output as the general path (same order).
// LDS ints: cin[n] cout[n] rp[n+1] rpt[n+1] | rank_in[m] rank_out[m] key_in[m] key_out[m] slot_of[m]
//           ends[m] (local source << 16 | local destination: the edge list is read from HBM once) | lnid[n] (node ids)
// status (optional): bit 0 set if an edge leaves its graph's node range (caller's slices are wrong).
template <int NT>
__device__ __forceinline__ void block_scan_into(const int* __restrict__ cnt, int* __restrict__ out, int n) {
    // exclusive scan of cnt[0..n) into out[0..n], out[n] = total; all NT threads call it
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += NT) {
        const int idx = c0 + threadIdx.x;
        const int v = idx < n ? cnt[idx] : 0;
        int total;
        const int ex = block_exclusive_scan<NT>(v, &total);
        const int carry = carry_s;
        if (idx < n) out[idx] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = carry_s;
    __syncthreads();
}

// The in- and out-degree scans of one graph as ONE scan of packed pairs (in | out << 16; a graph's edge count is far below
// 65536: its LDS image would not fit otherwise): exclusive scans of cin / cout [0..n) into rp / rpt [0..n], [n] = totals.
// Half the barriers of two block_scan_into calls; for n <= NT (every reference-scale graph) no carry round trip either.
template <int NT>
__device__ __forceinline__ void block_scan_pair_into(const int* __restrict__ cin, const int* __restrict__ cout,
                                                     int* __restrict__ rp, int* __restrict__ rpt, int n) {
    __shared__ int carry_p;
    if (n <= NT) {
        const int idx = threadIdx.x;
        const int v = idx < n ? (cin[idx] | (cout[idx] << 16)) : 0;
        int total;
        const int ex = block_exclusive_scan<NT>(v, &total);
        if (idx < n) { rp[idx] = ex & 0xFFFF; rpt[idx] = (int)((unsigned)ex >> 16); }
        if (idx == 0) { rp[n] = total & 0xFFFF; rpt[n] = (int)((unsigned)total >> 16); }
        __syncthreads();
        return;
    }
    if (threadIdx.x == 0) carry_p = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += NT) {
        const int idx = c0 + threadIdx.x;
        const int v = idx < n ? (cin[idx] | (cout[idx] << 16)) : 0;
        int total;
        const int ex = block_exclusive_scan<NT>(v, &total);
        const int carry = carry_p;
        const int s2 = carry + ex;                       // (both halves stay below 65536: no carry between them)
        if (idx < n) { rp[idx] = s2 & 0xFFFF; rpt[idx] = (int)((unsigned)s2 >> 16); }
        __syncthreads();
        if (threadIdx.x == 0) carry_p = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) { rp[n] = carry_p & 0xFFFF; rpt[n] = (int)((unsigned)carry_p >> 16); }
    __syncthreads();
}

// NT threads per workgroup: 256 inside the multi-role launch; 1024 in a launch of its own for graphs whose LDS image
// leaves one workgroup per CU anyway (csr_by_graph_wide_kernel: every phase is a loop over the graph's edges or rows)
template <int NT = 256>
__device__ __forceinline__ void csr_by_graph_body(
    const int64_t* __restrict__ ei, int64_t E, int64_t N, const int64_t* __restrict__ node_ptr,
    const int64_t* __restrict__ edge_ptr, int64_t B, int32_t* __restrict__ rowptr, int32_t* __restrict__ col,
    int32_t* __restrict__ eid, int32_t* __restrict__ row, int32_t* __restrict__ rowptr_t,
    int32_t* __restrict__ col_t, int32_t* __restrict__ pos_t, int32_t* __restrict__ eid_t,
    float* __restrict__ invdeg, int32_t* __restrict__ status, int cap_n, int cap_m,
    const int64_t* __restrict__ node_ids, int32_t* __restrict__ ids32, int32_t* __restrict__ colf,
    int32_t* __restrict__ colf_t, int32_t* __restrict__ ptr32, int64_t b, int* __restrict__ lds) {
    const int64_t n0 = node_ptr[b], e0 = edge_ptr[b];
    int n = (int)(node_ptr[b + 1] - n0), m = (int)(edge_ptr[b + 1] - e0);
    if (ptr32 && threadIdx.x == 0) {
        ptr32[b] = (int32_t)n0;
        if (b == B - 1) ptr32[B] = (int32_t)node_ptr[B];
    }
    // host-side size bound violated, or slices that do not lie inside the arrays: flag, write nothing
    if (n > cap_n || m > cap_m || n < 0 || m < 0 || n0 < 0 || e0 < 0 || n0 + n > N || e0 + m > E) {
        if (threadIdx.x == 0 && status) atomicOr(status, 2);
        return;
    }
    // Global round trips cost ~1.5 us each here (1024 workgroups start together): the node ids and the first 512 edges are
    // requested TOGETHER, before anything waits (first version: ids, then two edge iterations, then two more id gathers
    // behind the sort: six trips in a row, 14 us for the job)
    int64_t nid0 = 0;
    if (node_ids && (int)threadIdx.x < n) nid0 = node_ids[n0 + threadIdx.x];
    int64_t sj[2] = {0, 0}, si[2] = {0, 0};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = threadIdx.x + NT * u;
        if (e < m) { sj[u] = ei[e0 + e]; si[u] = ei[E + e0 + e]; }
    }
    int* cin = lds;
    int* cout = cin + cap_n;
    int* rp = cout + cap_n;
    int* rpt = rp + cap_n + 1;
    int* rank_in = rpt + cap_n + 1;
    int* rank_out = rank_in + cap_m;
    int* key_in = rank_out + cap_m;
    int* key_out = key_in + cap_m;
    int* slot_of = key_out + cap_m;
    unsigned int* ends = reinterpret_cast<unsigned int*>(slot_of + cap_m);
    int* lnid = reinterpret_cast<int*>(ends + cap_m);          // the graph's node ids (table mode), read back in the slot loop
    for (int t = threadIdx.x; t < n; t += NT) { cin[t] = 0; cout[t] = 0; }
    if (node_ids) {
        if ((int)threadIdx.x < n) { ids32[n0 + threadIdx.x] = (int32_t)nid0; lnid[threadIdx.x] = (int)nid0; }
        for (int t = threadIdx.x + NT; t < n; t += NT) {
            const int v = (int)node_ids[n0 + t];
            ids32[n0 + t] = v;
            lnid[t] = v;
        }
    }
    __syncthreads();
    bool bad = false;
    for (int ec = 0; ec < m; ec += 2 * NT) {
        if (ec > 0) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int e = ec + threadIdx.x + NT * u;
                if (e < m) { sj[u] = ei[e0 + e]; si[u] = ei[E + e0 + e]; }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = ec + threadIdx.x + NT * u;
            if (e < m) {
                int j = (int)(sj[u] - n0), i = (int)(si[u] - n0);
                if (i < 0 || i >= n || j < 0 || j >= n) { bad = true; i = i < 0 ? 0 : (i >= n ? n - 1 : i); j = j < 0 ? 0 : (j >= n ? n - 1 : j); }
                ends[e] = ((unsigned int)j << 16) | (unsigned int)i;
                rank_in[e] = atomicAdd(&cin[i], 1);
                rank_out[e] = atomicAdd(&cout[j], 1);
            }
        }
    }
    if (bad && status) atomicOr(status, 1);
    __syncthreads();
    if (m < 65536) {
        block_scan_pair_into<NT>(cin, cout, rp, rpt, n);
    } else {
        block_scan_into<NT>(cin, rp, n);
        block_scan_into<NT>(cout, rpt, n);
    }
    for (int e = threadIdx.x; e < m; e += NT) {
        const unsigned int ji = ends[e];
        key_in[rp[ji & 0xFFFFu] + rank_in[e]] = e;
        key_out[rpt[ji >> 16] + rank_out[e]] = e;
    }
    __syncthreads();
    // rows: order the keys (LDS only) and write the per-row outputs; rank_in / rank_out are free by now and
    // take the row of every slot, so that the per-slot outputs can be written slot-parallel (coalesced)
    int* row_of = rank_in;
    int* row_of_t = rank_out;
    for (int r = threadIdx.x; r < 2 * n; r += NT) {
        if (r < n) {
            const int beg = rp[r], end = rp[r + 1];
            sort_row_keys(key_in, beg, end);
            for (int p = beg; p < end; ++p) { row_of[p] = r; slot_of[key_in[p]] = p; }
            const int d = end - beg;
            invdeg[n0 + r] = 1.0f / (float)(d > 1 ? d : 1);
            rowptr[n0 + r] = (int32_t)(e0 + beg);
        } else {
            const int jj = r - n;
            const int beg = rpt[jj], end = rpt[jj + 1];
            sort_row_keys(key_out, beg, end);
            for (int t = beg; t < end; ++t) row_of_t[t] = jj;
            rowptr_t[n0 + jj] = (int32_t)(e0 + beg);
        }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < m; p += NT) {
        const int key = key_in[p];
        const int lsrc = (int)(ends[key] >> 16);
        const int64_t src = n0 + (int64_t)lsrc;
        col[e0 + p] = (int32_t)src;
        if (node_ids) colf[e0 + p] = (int32_t)lnid[lsrc];
        eid[e0 + p] = (int32_t)(e0 + key);
        row[e0 + p] = (int32_t)(n0 + row_of[p]);
        const int kt = key_out[p];
        const int ldst = (int)(ends[kt] & 0xFFFFu);
        const int64_t dst = n0 + (int64_t)ldst;
        col_t[e0 + p] = (int32_t)dst;
        if (node_ids) colf_t[e0 + p] = (int32_t)lnid[ldst];
        eid_t[e0 + p] = (int32_t)(e0 + kt);
        pos_t[e0 + p] = (int32_t)(e0 + slot_of[kt]);
    }
    if (b == B - 1 && threadIdx.x == 0) { rowptr[N] = (int32_t)E; rowptr_t[N] = (int32_t)E; }
}


// LDS bytes of csr_by_graph_body for graphs of at most max_nodes / max_edges
static inline size_t by_graph_lds_bytes(int64_t max_nodes, int64_t max_edges) {
    return (size_t)(5 * max_nodes + 2 + 6 * max_edges) * 4;
}
constexpr size_t kByGraphLdsMax = 144 * 1024;     // one workgroup per CU at most
constexpr int64_t kByGraphMaxNodes = 65535;       // local node ids are packed 16 + 16 bits

}  // namespace qot
