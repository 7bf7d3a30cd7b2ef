// Multi-role launch: several INDEPENDENT small jobs of one train step in ONE kernel launch.
//
// A replayed cfg2 step was 19 launches, eleven of them 4-9 us of launch + dependency latency around a few microseconds
// of work ("that tail is now the second-largest item of the step").  Jobs that do not depend on each
// other do not need launches of their own: a role table (passed BY VALUE in the kernel arguments -- no device-side
// descriptor, nothing to copy, capture-safe) gives every job a contiguous range of 256-thread workgroups, and a
// workgroup runs the body of the job that owns its index.  The bodies are the device functions the standalone entry
// points always ran (graph_prep_dev.hpp, small_dev.hpp, reduce_dev.hpp, nnconv_finalize_dev.hpp); those entry points
// are now one-role calls of this launch, so every test of theirs covers this code.
//
// Used by the host side for
//   forward prologue   {graph index of a block-diagonal batch | embedding-table projection (+ dropout counter) |
//                       NNConv operand packing | r04: score matrix M and P of TransformerConv's graph form}
//                                                                      -- three launches before (topological.py)
//   backward epilogue  stage 1 {read-out partial sums | NNConv slab + grad-h sums | lin_edge level-1 sums | table
//                       gradient row sum}, stage 2 {table projection backward | lin_edge level-2 sum}
//                                                                      -- seven launches before (functional.py)
// Jobs inside one launch MUST be independent (no job may read what another job of the same launch writes).
#include <cstdlib>
#include "common.hpp"
#include "graph_prep_dev.hpp"
#include "mfma_tile.hpp"
#include "nnconv_finalize_dev.hpp"
#include "reduce_dev.hpp"
#include "small_dev.hpp"
#include "tconv_graph_dev.hpp"

namespace qot {

struct RoleTable {
    int n;
    int first[QOT_MAX_ROLES + 1];       // workgroup range of role r: [first[r], first[r + 1])
    qot_role_t role[QOT_MAX_ROLES];
};

template <typename T>
__device__ __forceinline__ T* rp(const qot_role_t& r, int k) { return reinterpret_cast<T*>(const_cast<void*>(r.p[k])); }

#define QOT_ROLE_H(H, ...)                                            \
    switch (H) {                                                      \
        case 16:  { constexpr int kH = 16;  __VA_ARGS__; } break;     \
        case 32:  { constexpr int kH = 32;  __VA_ARGS__; } break;     \
        case 64:  { constexpr int kH = 64;  __VA_ARGS__; } break;     \
        case 128: { constexpr int kH = 128; __VA_ARGS__; } break;     \
        case 256: { constexpr int kH = 256; __VA_ARGS__; } break;     \
        default: break;                                               \
    }

__global__ __launch_bounds__(256) void roles_kernel(const RoleTable t_by_value) {
    extern __shared__ __attribute__((aligned(16))) int dyn_lds[];
    // The table is read where it lies -- in the kernel-argument segment (it is the only argument: offset 0) -- through
    // a pointer: indexing the by-value copy with a runtime role index made the compiler move all of it to scratch
    // (72 B/lane); through the pointer the fields a role needs arrive by scalar loads.
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const RoleTable __attribute__((address_space(4))) * KernargTable;
    const RoleTable& t = *(const RoleTable*)(KernargTable)__builtin_amdgcn_kernarg_segment_ptr();
    (void)t_by_value;
#else
    const RoleTable& t = t_by_value;
#endif
    int r = 0;
    while (r + 1 < t.n && (int)blockIdx.x >= t.first[r + 1]) ++r;       // uniform; <= QOT_MAX_ROLES scalar steps
    const qot_role_t& ro = t.role[r];
    const int vb = (int)blockIdx.x - t.first[r];
    switch (ro.kind) {
    case QOT_ROLE_CSR_BY_GRAPH:
        csr_by_graph_body(rp<const int64_t>(ro, 0), ro.i[0], ro.i[1], rp<const int64_t>(ro, 1), rp<const int64_t>(ro, 2),
                          ro.i[2], rp<int32_t>(ro, 3), rp<int32_t>(ro, 4), rp<int32_t>(ro, 5), rp<int32_t>(ro, 6),
                          rp<int32_t>(ro, 7), rp<int32_t>(ro, 8), rp<int32_t>(ro, 9), rp<int32_t>(ro, 10),
                          rp<float>(ro, 11), rp<int32_t>(ro, 12), (int)ro.i[3], (int)ro.i[4], rp<const int64_t>(ro, 13),
                          rp<int32_t>(ro, 14), rp<int32_t>(ro, 15), rp<int32_t>(ro, 16), rp<int32_t>(ro, 17), (int64_t)vb,
                          dyn_lds);
        break;
    case QOT_ROLE_TABLE_PROJECT_FWD: {
        const Proj4 p{{rp<const float>(ro, 1), rp<const float>(ro, 3), rp<const float>(ro, 5), rp<const float>(ro, 7)},
                      {rp<const float>(ro, 2), rp<const float>(ro, 4), rp<const float>(ro, 6), rp<const float>(ro, 8)}};
        QOT_ROLE_H((int)ro.i[1], table_project_fwd_body<kH, 1>(rp<const float>(ro, 0), p, rp<float>(ro, 9), rp<int64_t>(ro, 10),
                                                               rp<int64_t>(ro, 11), (int)ro.i[0], vb,
                                                               reinterpret_cast<float*>(dyn_lds)));
        break;
    }
    case QOT_ROLE_GATHER3:
        gather3_body(rp<const float>(ro, 0), (int)ro.i[0], rp<const float>(ro, 1), (int)ro.i[1], rp<const float>(ro, 2),
                     rp<const int32_t>(ro, 3), rp<float>(ro, 4), ro.i[2], (int64_t)vb);
        break;
    case QOT_ROLE_SUM_ROWS:
        sum_rows_body(rp<const float>(ro, 0), rp<float>(ro, 1), ro.i[0], ro.i[1], ro.i[2], (int)ro.i[3], vb,
                      reinterpret_cast<float*>(dyn_lds));
        break;
    case QOT_ROLE_NNCONV_FINALIZE64:
        nnconv_bwd_finalize64_body(rp<const float>(ro, 0), (int)ro.i[2], ro.i[3], rp<float>(ro, 2), (int)ro.i[4], (int)ro.i[5],
                                   rp<const float>(ro, 1), (int)ro.i[6], (int)ro.i[7], (int)ro.i[4] * (int)ro.i[1],
                                   rp<float>(ro, 3), rp<float>(ro, 4), vb);
        break;
    case QOT_ROLE_TABLE_PROJECT_BWD: {
        const Proj4 p{{rp<const float>(ro, 2), rp<const float>(ro, 3), rp<const float>(ro, 4), rp<const float>(ro, 5)},
                      {nullptr, nullptr, nullptr, nullptr}};
        QOT_ROLE_H((int)ro.i[1], table_project_bwd_body<kH, 1>(rp<const float>(ro, 0), rp<const float>(ro, 1), p, rp<float>(ro, 6),
                                                               rp<float>(ro, 7), rp<float>(ro, 8), (int)ro.i[0], vb,
                                                               reinterpret_cast<float*>(dyn_lds)));
        break;
    }
    case QOT_ROLE_TABLE_SCORES:
        QOT_ROLE_H((int)ro.i[1], table_scores_body<kH>(rp<const float>(ro, 0), rp<const float>(ro, 1), rp<const float>(ro, 2),
                                                       rp<const float>(ro, 3), rp<const float>(ro, 4), rp<const float>(ro, 5),
                                                       rp<float>(ro, 6), rp<float>(ro, 7), (int)ro.i[0], (int)ro.i[2], vb,
                                                       reinterpret_cast<float*>(dyn_lds)));
        break;
    case QOT_ROLE_TABLE_PROJECT_BWD_SCORES: {
        const Proj4 p{{rp<const float>(ro, 4), rp<const float>(ro, 5), rp<const float>(ro, 6), rp<const float>(ro, 7)},
                      {nullptr, nullptr, nullptr, nullptr}};
        QOT_ROLE_H((int)ro.i[2], table_project_bwd_scores_body<kH>(rp<const float>(ro, 0), rp<const float>(ro, 1),
                                                                   rp<const float>(ro, 2), rp<const float>(ro, 3), p,
                                                                   rp<float>(ro, 8), rp<float>(ro, 9), rp<float>(ro, 10),
                                                                   rp<float>(ro, 11), (int)ro.i[0], (int)ro.i[1], (int)ro.i[3], vb,
                                                                   reinterpret_cast<float*>(dyn_lds)));
        break;
    }
    default:
        break;
    }
}

// Table projection of a big table (V >= kTableBlockMinV): kTableBlock rows per workgroup, kernels of their own (their
// register needs would otherwise be every role's; at those sizes a launch more is nothing).
template <int H>
__global__ __launch_bounds__(256) void table_project_fwd_block_kernel(const float* __restrict__ table, Proj4 p,
                                                                      float* __restrict__ out, int64_t* __restrict__ counter,
                                                                      int64_t* __restrict__ snapshot, int V) {
    extern __shared__ __attribute__((aligned(16))) int dyn_lds[];
    table_project_fwd_body<H, kTableBlock>(table, p, out, counter, snapshot, V, (int)blockIdx.x, reinterpret_cast<float*>(dyn_lds));
}
template <int H>
__global__ __launch_bounds__(256) void table_project_bwd_block_kernel(const float* __restrict__ gp, const float* __restrict__ table,
                                                                      Proj4 p, float* __restrict__ gtable, float* __restrict__ gw,
                                                                      float* __restrict__ gb, int V) {
    extern __shared__ __attribute__((aligned(16))) int dyn_lds[];
    table_project_bwd_body<H, kTableBlock>(gp, table, p, gtable, gw, gb, V, (int)blockIdx.x, reinterpret_cast<float*>(dyn_lds));
}

}  // namespace qot

using namespace qot;

static bool width_ok(int64_t H) { return H == 16 || H == 32 || H == 64 || H == 128 || H == 256; }

// Validates role `r` (as the standalone entry point of that job always did), fills the derived fields the body reads
// (marked "derived" in include/qot_gnn.h) and returns its workgroup count and LDS need; < 0: a QOT_ERR_* code.
static int plan_role(qot_role_t& r, int64_t* blocks, size_t* lds) {
    *blocks = 0;
    *lds = 0;
    const void* const* p = r.p;
    int64_t* i = r.i;
    switch (r.kind) {
    case QOT_ROLE_CSR_BY_GRAPH: {
        const int64_t E = i[0], N = i[1], B = i[2], mn = i[3], me = i[4];
        if (p[13] && (!p[14] || (E > 0 && (!p[15] || !p[16])))) return QOT_ERR_BADARG;
        if (E < 0 || N < 0 || B <= 0 || mn < 0 || me < 0 || !p[3] || !p[7] || !p[11]) return QOT_ERR_BADARG;
        if (N >= (int64_t(1) << 31) - 1 || E >= (int64_t(1) << 31) - 1) return QOT_ERR_UNSUPPORTED;
        if (mn > kByGraphMaxNodes || by_graph_lds_bytes(mn, me) > kByGraphLdsMax) return QOT_ERR_UNSUPPORTED;
        if (!p[1] || !p[2] || (E > 0 && (!p[0] || !p[4] || !p[5] || !p[6] || !p[8] || !p[9] || !p[10]))) return QOT_ERR_BADARG;
        *blocks = B;
        *lds = by_graph_lds_bytes(mn, me);
        return QOT_OK;
    }
    case QOT_ROLE_TABLE_PROJECT_FWD: {
        const int64_t V = i[0], H = i[1];
        if (V <= 0 || (p[10] && !p[11])) return QOT_ERR_BADARG;
        if (!width_ok(H)) return QOT_ERR_UNSUPPORTED;
        for (int k = 0; k < 10; ++k) if (!p[k]) return QOT_ERR_BADARG;
        const int R = table_rows_per_block(V);          // R > 1: launched on its own (launch_heavy_role)
        *blocks = (V + R - 1) / R;
        *lds = (size_t)R * H * 4;
        return QOT_OK;
    }
    case QOT_ROLE_GATHER3: {
        const int64_t n0 = i[0], n1 = i[1], n = i[2];
        if (n <= 0 || n0 < 0 || n1 < 0 || n0 + n1 > 0x7fffffff) return QOT_ERR_BADARG;
        for (int k = 0; k < 5; ++k) if (!p[k]) return QOT_ERR_BADARG;
        *blocks = (n + 255) / 256;
        return QOT_OK;
    }
    case QOT_ROLE_SUM_ROWS: {
        const int64_t nblk = i[0], n = i[1];
        int64_t per = i[2];
        if (nblk <= 0 || n <= 0 || per < 0 || !p[0] || !p[1]) return QOT_ERR_BADARG;
        if (per == 0 || per > nblk) per = nblk;
        i[2] = per;
        const int64_t groups = (nblk + per - 1) / per;
        const bool v4 = (n % 4 == 0) && ((uintptr_t)p[0] % 16 == 0) && ((uintptr_t)p[1] % 16 == 0);
        i[3] = v4 ? 1 : 0;                                   // derived: float4 columns
        const int64_t cols = v4 ? n / 4 : n;
        i[4] = (cols + kSumCols - 1) / kSumCols;             // derived: column blocks per group
        *blocks = i[4] * groups;
        *lds = (size_t)256 * 16;
        if (*blocks > 0x7fffffff) return QOT_ERR_UNSUPPORTED;
        return QOT_OK;
    }
    case QOT_ROLE_NNCONV_FINALIZE64: {
        const int64_t N = i[0], D = i[1];
        if (N <= 0 || D < 1) return QOT_ERR_BADARG;
        if (D > 4) return QOT_ERR_UNSUPPORTED;
        for (int k = 0; k < 5; ++k) if (!p[k]) return QOT_ERR_BADARG;
        int64_t agrid = (N + 31) / 32;
        if (agrid > (int64_t)num_cus() * kAdjBlocksPerCu) agrid = (int64_t)num_cus() * kAdjBlocksPerCu;
        int64_t hgrid = (N + 31) / 32;
        if (hgrid > 2 * (int64_t)num_cus()) hgrid = 2 * (int64_t)num_cus();
        const int64_t K = 2 * D;
        const int64_t elems = (K + 2) * 64 * 64;
        const int64_t nb1 = (elems / 4 * 16 + 255) / 256;
        const int64_t hn = K * (D + 1);
        i[2] = agrid; i[3] = elems; i[4] = K; i[5] = nb1; i[6] = hgrid; i[7] = hn;      // derived
        *blocks = nb1 + (hn + 3) / 4;
        return QOT_OK;
    }
    case QOT_ROLE_TABLE_PROJECT_BWD: {
        const int64_t V = i[0], H = i[1];
        if (V <= 0) return QOT_ERR_BADARG;
        if (!width_ok(H)) return QOT_ERR_UNSUPPORTED;
        for (int k = 0; k < 9; ++k) if (!p[k]) return QOT_ERR_BADARG;
        const int R = table_rows_per_block(V);
        *blocks = 4 * H / R + (V + R - 1) / R;
        *lds = (size_t)(512 * R + R * 4 * H) * 4;
        return QOT_OK;
    }
    case QOT_ROLE_TABLE_SCORES: {
        const int64_t n = i[0], H = i[1], D = i[2];
        if (n <= 0 || D <= 0) return QOT_ERR_BADARG;
        if (!width_ok(H) || D > 8) return QOT_ERR_UNSUPPORTED;
        for (int k = 0; k < 8; ++k) if (!p[k]) return QOT_ERR_BADARG;
        *blocks = (n + kScoreRows - 1) / kScoreRows;
        *lds = (size_t)kScoreRows * (3 * H + 1) * 4;
        return QOT_OK;
    }
    case QOT_ROLE_TABLE_PROJECT_BWD_SCORES: {
        const int64_t V = i[0], n = i[1], H = i[2], D = i[3];
        if (V <= 0 || n <= 0 || n > V || D <= 0) return QOT_ERR_BADARG;
        if (!width_ok(H) || D > 8 || table_rows_per_block(V) > 1) return QOT_ERR_UNSUPPORTED;
        if (n > kTgMaxN) return QOT_ERR_UNSUPPORTED;
        for (int k = 0; k < 12; ++k) if (!p[k]) return QOT_ERR_BADARG;
        *blocks = 4 * H + V + 1;
        *lds = (size_t)tg_bwd_scores_lds_floats((int)n, (int)H, (int)D) * 4;
        return QOT_OK;
    }
    default:
        return QOT_ERR_BADARG;
    }
}

// The per-graph index build of LARGE graphs (LDS image above 64 KB: one workgroup per CU whatever its width) in a launch of
// its own with 1024 threads per graph: every phase is a loop over the graph's edges or rows, and at 1000 nodes / 4000 edges
// 256 threads walked each of them in 16 rounds (196 us for 1024 graphs at cfg4).
__global__ __launch_bounds__(1024) void csr_by_graph_wide_kernel(
    const int64_t* __restrict__ ei, int64_t E, int64_t N, const int64_t* __restrict__ node_ptr,
    const int64_t* __restrict__ edge_ptr, int64_t B, int32_t* __restrict__ rowptr, int32_t* __restrict__ col,
    int32_t* __restrict__ eid, int32_t* __restrict__ row, int32_t* __restrict__ rowptr_t, int32_t* __restrict__ col_t,
    int32_t* __restrict__ pos_t, int32_t* __restrict__ eid_t, float* __restrict__ invdeg, int32_t* __restrict__ status,
    int cap_n, int cap_m, const int64_t* __restrict__ node_ids, int32_t* __restrict__ ids32, int32_t* __restrict__ colf,
    int32_t* __restrict__ colf_t, int32_t* __restrict__ ptr32) {
    extern __shared__ int wide_lds[];
    csr_by_graph_body<1024>(ei, E, N, node_ptr, edge_ptr, B, rowptr, col, eid, row, rowptr_t, col_t, pos_t, eid_t, invdeg, status,
                            cap_n, cap_m, node_ids, ids32, colf, colf_t, ptr32, (int64_t)blockIdx.x, wide_lds);
}
constexpr size_t kCsrWideMinLds = 64 * 1024;

static int launch_wide_csr(const qot_role_t& ro, int64_t blocks, size_t lds, hipStream_t stream) {
    static size_t allowed[kMaxDevices];
    const int lrc = ensure_dyn_lds(reinterpret_cast<const void*>(csr_by_graph_wide_kernel), lds, allowed);
    if (lrc != QOT_OK) return lrc;
    auto P = [&](int k) { return const_cast<void*>(ro.p[k]); };
    csr_by_graph_wide_kernel<<<(int)blocks, 1024, lds, stream>>>(
        (const int64_t*)P(0), ro.i[0], ro.i[1], (const int64_t*)P(1), (const int64_t*)P(2), ro.i[2], (int32_t*)P(3), (int32_t*)P(4),
        (int32_t*)P(5), (int32_t*)P(6), (int32_t*)P(7), (int32_t*)P(8), (int32_t*)P(9), (int32_t*)P(10), (float*)P(11),
        (int32_t*)P(12), (int)ro.i[3], (int)ro.i[4], (const int64_t*)P(13), (int32_t*)P(14), (int32_t*)P(15), (int32_t*)P(16),
        (int32_t*)P(17));
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

// a table projection over >= kTableBlockMinV rows: its own launch (see table_project_*_block_kernel)
static int launch_heavy_role(const qot_role_t& ro, int64_t blocks, size_t lds, hipStream_t stream) {
    const int V = (int)ro.i[0];
    auto P = [&](int k) { return const_cast<void*>(ro.p[k]); };
    if (ro.kind == QOT_ROLE_TABLE_PROJECT_FWD) {
        const Proj4 p{{(const float*)P(1), (const float*)P(3), (const float*)P(5), (const float*)P(7)},
                      {(const float*)P(2), (const float*)P(4), (const float*)P(6), (const float*)P(8)}};
        QOT_ROLE_H((int)ro.i[1], (table_project_fwd_block_kernel<kH><<<(int)blocks, 256, lds, stream>>>(
                                      (const float*)P(0), p, (float*)P(9), (int64_t*)P(10), (int64_t*)P(11), V)));
    } else {
        const Proj4 p{{(const float*)P(2), (const float*)P(3), (const float*)P(4), (const float*)P(5)},
                      {nullptr, nullptr, nullptr, nullptr}};
        QOT_ROLE_H((int)ro.i[1], (table_project_bwd_block_kernel<kH><<<(int)blocks, 256, lds, stream>>>(
                                      (const float*)P(0), (const float*)P(1), p, (float*)P(6), (float*)P(7), (float*)P(8), V)));
    }
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}

extern "C" int qot_run_roles(const qot_role_t* roles, int n_roles, qot_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_roles < 0 || n_roles > QOT_MAX_ROLES || (n_roles > 0 && !roles)) return QOT_ERR_BADARG;
    if (n_roles == 0) return QOT_OK;
    RoleTable t;
    t.n = 0;
    size_t lds = 0;
    int64_t total = 0;
    int n_light = 0;
    for (int r = 0; r < n_roles; ++r) {
        qot_role_t ro = roles[r];
        int64_t blocks;
        size_t need;
        const int rc = plan_role(ro, &blocks, &need);
        if (rc != QOT_OK) return rc;
        if ((ro.kind == QOT_ROLE_TABLE_PROJECT_FWD || ro.kind == QOT_ROLE_TABLE_PROJECT_BWD) && table_rows_per_block(ro.i[0]) > 1) {
            const int hrc = launch_heavy_role(ro, blocks, need, stream);     // jobs of one call are independent: any order
            if (hrc != QOT_OK) return hrc;
            continue;
        }
        if (ro.kind == QOT_ROLE_CSR_BY_GRAPH && need > kCsrWideMinLds && !getenv("QOT_NO_WIDE_CSR")) {
            const int hrc = launch_wide_csr(ro, blocks, need, stream);
            if (hrc != QOT_OK) return hrc;
            continue;
        }
        t.role[n_light] = ro;
        t.first[n_light] = (int)total;
        ++n_light;
        total += blocks;
        if (total > 0x7fffffff) return QOT_ERR_UNSUPPORTED;
        if (need > lds) lds = need;
    }
    if (n_light == 0) return QOT_OK;
    t.n = n_light;
    for (int r = n_light; r <= QOT_MAX_ROLES; ++r) t.first[r] = (int)total;
    {   // dynamic LDS above 64 KB has to be allowed once per device (the first call is outside any graph capture)
        static size_t allowed[kMaxDevices];
        const int lrc = ensure_dyn_lds(reinterpret_cast<const void*>(roles_kernel), lds, allowed);
        if (lrc != QOT_OK) return lrc;
    }
    roles_kernel<<<(int)total, 256, lds, stream>>>(t);
    QOT_LAUNCH_CHECK();
    return QOT_OK;
}
