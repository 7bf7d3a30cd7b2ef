/*
 * qot_gnn.h -- C ABI of libqot_gnn.so, the MI355X (gfx950) message-passing engine that
 * replaces the torch_geometric operator layer under the reference's two models.
 *
 * Reference interface each entry point replaces (paths relative to the reference repo;
 * [PyG-ext] = third-party torch_geometric code reached from that line):
 *
 *   qot_csr_build            edge_index handling inside every conv call:
 *                            topological_training/models.py:53,57 and
 *                            lightpath_training/models.py:30 ([PyG-ext] propagate's
 *                            index_select/scatter bookkeeping; GATConv's
 *                            remove_self_loops + add_self_loops)
 *   qot_embed_{fwd,bwd}      topological_training/models.py:12,51-52  (nn.Embedding lookup)
 *   qot_tconv_*              topological_training/models.py:15-17,53 ([PyG-ext] TransformerConv)
 *   qot_nnconv_*             topological_training/models.py:20-30,57 ([PyG-ext] NNConv aggr="mean")
 *   qot_act_{fwd,bwd}        topological_training/models.py:54-55,58-59 (leaky_relu + Dropout)
 *   qot_pool_{fwd,bwd}       topological_training/models.py:61 ([PyG-ext] global_mean_pool)
 *   qot_gat_*                lightpath_training/models.py:13,30 ([PyG-ext] GATConv heads=4)
 *   qot_bn_*                 lightpath_training/models.py:14,31-32 ([PyG-ext] BatchNorm + relu)
 *   qot_rows_gather/scatter  lightpath_training/models.py:39-40 (x[lut_mask] and its adjoint)
 *
 * Conventions
 *   - Stateless and stream-ordered: every call only enqueues work on `stream`
 *     (a hipStream_t passed as void*); the library never allocates, frees or synchronises.
 *     All outputs and workspaces are caller-allocated DEVICE memory.
 *   - Return value: 0 = ok; >0 = hipError_t of the failed launch/runtime call;
 *     <0 = QOT_ERR_* argument error.  qot_error_string() describes either.
 *   - Feature matrices are row-major fp32.  `ld` arguments are row strides in floats, so a
 *     packed [N, 4H] projection output can be passed as four column slices.
 *   - Indices inside the library are int32 (N, E < 2^31); the int64 edge_index of the
 *     reference contract is consumed only by qot_csr_build.
 *   - Supported widths: H (and heads*C) power of two in [16, 256] (heads*C up to 1024),
 *     edge_dim D in {1..8} for TransformerConv and for the NNConv building blocks qot_nnconv_agg /
 *     qot_nnconv_bwd_edge; the FUSED NNConv tile kernels (qot_nnconv_fused, qot_nnconv_adjoint_dw,
 *     qot_nnconv_gradh_fused, qot_nnconv_dw) are built for D <= 4 (K = 2D <= 8 operand blocks per LDS tile)
 *     and return QOT_ERR_UNSUPPORTED above it -- the host side (functional.NNConvFn) then runs
 *     {qot_nnconv_agg, GEMM, qot_nnconv_bwd_edge}.  Anything else returns QOT_ERR_UNSUPPORTED (callers must
 *     fail loudly; there is no CPU fallback).
 */
#ifndef QOT_GNN_H
#define QOT_GNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* qot_stream_t; /* hipStream_t */

#define QOT_ABI_VERSION 10
#define QOT_OK 0
#define QOT_ERR_UNSUPPORTED (-1) /* width / edge_dim not instantiated */
#define QOT_ERR_BADARG (-2)      /* null pointer, negative size, workspace too small */

int qot_abi_version(void);
const char* qot_error_string(int code);

/* ---- graph preparation ------------------------------------------------------------
 * Builds, from edge_index[2,E] (int64, row 0 = source j, row 1 = target i):
 *   CSR by destination : rowptr[N+1], col[cap] (source of each in-edge), eid[cap]
 *                        (original edge id, or -1 for an inserted self loop), row[cap]
 *                        (destination of each CSR slot)
 *   CSC by source      : rowptr_t[N+1], col_t[cap] (destination of each out-edge),
 *                        pos_t[cap] (CSR slot of that edge), eid_t[cap] (its original edge id)
 * cap = E (+ N when gat_self_loops).  Edge order inside a destination follows the
 * original edge order (stable), so results are run-to-run bitwise reproducible.
 * gat_self_loops != 0: edges with j == i are dropped and one (n,n) edge per node is
 * appended (GATConv semantics); the live edge count is rowptr[N] (device side).
 * invdeg[N] = 1 / max(in_degree, 1).
 */
size_t qot_csr_workspace_bytes(int64_t E, int64_t N, int gat_self_loops);
int qot_csr_build(const int64_t* edge_index, int64_t E, int64_t N, int gat_self_loops,
                  int32_t* rowptr, int32_t* col, int32_t* eid, int32_t* row,
                  int32_t* rowptr_t, int32_t* col_t, int32_t* pos_t, int32_t* eid_t, float* invdeg,
                  void* workspace, size_t workspace_bytes, qot_stream_t stream);
/* Same index for a BLOCK-DIAGONAL batch whose graphs keep their nodes and edges contiguous (what a
 * collate produces; PyG's Batch keeps the same slices): node_ptr[B+1], edge_ptr[B+1] (int64, device)
 * give graph b's node / edge ranges, max_nodes / max_edges bound the largest graph (host-known).
 * One launch, one workgroup per graph, everything in LDS: no global atomics, no workspace.  Output
 * identical to qot_csr_build(gat_self_loops = 0).  Returns QOT_ERR_UNSUPPORTED when the largest graph
 * does not fit LDS ((4 max_nodes + 6 max_edges) * 4 bytes > 144 KB, or max_nodes > 65535): use
 * qot_csr_build then.
 * status (optional, device int32, caller-zeroed): bit 0 = an edge leaves its graph's node range,
 * bit 1 = a graph exceeds max_nodes / max_edges.
 * Optional by-products of the same pass (NULL to skip): with node_ids[N] (int64) the TransformerConv
 * table-mode maps ids32[N] = node_ids, colf[E] = node_ids[col], colf_t[E] = node_ids[col_t]
 * (qot_table_maps); ptr32[B+1] = node_ptr narrowed to int32 (read-out boundaries). */
int qot_csr_build_by_graph(const int64_t* edge_index, int64_t E, int64_t N, const int64_t* node_ptr,
                           const int64_t* edge_ptr, int64_t B, int64_t max_nodes, int64_t max_edges,
                           int32_t* rowptr, int32_t* col, int32_t* eid, int32_t* row, int32_t* rowptr_t,
                           int32_t* col_t, int32_t* pos_t, int32_t* eid_t, float* invdeg, int32_t* status,
                           const int64_t* node_ids, int32_t* ids32, int32_t* colf, int32_t* colf_t, int32_t* ptr32,
                           qot_stream_t stream);
/* GATConv's self-looped index (qot_csr_build with gat_self_loops = 1: PyG GATConv rebuilds it in every call,
 * lightpath_training/models.py:30) of a block-diagonal batch of SMALL graphs -- LightpathGNN's chain graphs of 2..20
 * nodes -- in one launch, one wave per graph.  Requires max_nodes <= 64, max_edges <= 256
 * (qot_csr_gat_by_graph_supported; QOT_ERR_UNSUPPORTED otherwise) and an input WITHOUT self loops: then graph b owns the
 * slots [edge_ptr[b] + node_ptr[b], ... + m + n) whatever the other graphs hold.  Output bit for bit qot_csr_build's.
 * status (optional, device int32): bits 0 / 1 as qot_csr_build_by_graph, bit 2 = a self loop was met (the index is
 * then wrong: rebuild with qot_csr_build).  ptr32 (optional): node_ptr narrowed to int32. */
int qot_csr_gat_by_graph_supported(int64_t max_nodes, int64_t max_edges);
int qot_csr_build_gat_by_graph(const int64_t* edge_index, int64_t E, int64_t N, const int64_t* node_ptr,
                               const int64_t* edge_ptr, int64_t B, int64_t max_nodes, int64_t max_edges, int32_t* rowptr,
                               int32_t* col, int32_t* eid, int32_t* row, int32_t* rowptr_t, int32_t* col_t, int32_t* pos_t,
                               int32_t* eid_t, float* invdeg, int32_t* status, int32_t* ptr32, qot_stream_t stream);
/* out[i] = map[idx[i]] (int32): table row of every CSR / CSC slot's source / destination
 * (node_ids[col], node_ids[col_t]) for TransformerConv's table mode.  idx values must be < len(map). */
int qot_i32_gather(const int32_t* map, const int32_t* idx, int32_t* out, int64_t n, qot_stream_t stream);
/* int64 -> int32 narrowing of node_ids / batch vectors */
int qot_i64_to_i32(const int64_t* in, int32_t* out, int64_t n, qot_stream_t stream);
/* ptr[B+1] from a sorted batch vector */
int qot_batch_ptr(const int32_t* batch, int64_t N, int64_t B, int32_t* ptr, qot_stream_t stream);

/* ---- embedding -------------------------------------------------------------------- */
int qot_embed_fwd(const float* table, const int32_t* ids, float* out, int64_t N, int V, int H,
                  qot_stream_t stream);
/* grad_table must be zero-filled by the caller; accumulates with per-workgroup LDS
 * pre-reduction then float atomics. */
int qot_embed_bwd(const float* grad_out, const int32_t* ids, float* grad_table, int64_t N, int V,
                  int H, qot_stream_t stream);

/* ---- TransformerConv (heads=1) ----------------------------------------------------
 * fwd: out_i = sum_e softmax_e(<q_i, k_j + We ea_e>/sqrt(H)) (v_j + We ea_e) + skip_i
 * stats[N,2] = (max logit, denominator incl. 1e-16) saved for backward.
 * edge_attr is in ORIGINAL edge order ([E,D]); the kernels index it through eid.
 * act != 0 fuses dropout(leaky_relu(out, act_slope), act_p) into the epilogue with the mask of
 * qot_act_fwd/qot_act_bwd (same seed / step counter / element indexing), so the backward is
 * qot_act_bwd(grad, out) followed by the conv backward.
 * Table mode (rowmap != NULL): q/k/v/skip are rows of a PROJECTED EMBEDDING TABLE [V, 4H]
 * ((emb W^T + b)[node_ids] == (emb[node_ids]) W^T + b, topological_training/models.py:51-53);
 * rowmap[i] = table row of node i and `col` must then hold the table row of each in-edge's source
 * (node_ids[col]).  NULL: one row per node, as produced by a node-level GEMM.
 */
/* qot_tconv_fwd in table mode with the logits' dense part <q_i, k_j> looked up in scores[V, ld_scores] = T_q T_k^T
 * (unscaled; qot_gemm_nt on the projected table) instead of gathered and multiplied per edge: rowmap (required) = table row
 * of every node, col = table row of every in-edge's source.  Same results up to the order of the H-term dot. */
int qot_tconv_fwd_scores(const float* q, const float* v, const float* skip, int ld, const float* scores, int ld_scores,
                         const float* edge_attr, const float* w_edge, const int32_t* rowptr, const int32_t* col,
                         const int32_t* eid, const int32_t* rowmap, float* out, float* stats, int64_t N, int H, int D,
                         int act, float act_slope, float act_p, uint64_t act_seed, const int64_t* act_step,
                         qot_stream_t stream);
/* Row form of TransformerConv's backward in table mode for large tables (csrc/tconv_rows.hip; autograd of
 * TransformerConv.propagate under loss.backward(), topological_training/train.py:115, with x = emb[node_ids],
 * models.py:51-53, and node_ids == arange(n) in each of the B graphs, dataset.py:78): no per-node grad_q / grad_k --
 * grad M[r_i, r_j] += ds_e on the score matrix of qot_tconv_fwd_scores, from which the caller forms
 * grad T_q = (grad M T_k + grad P W_e^T) / sqrt(H) and grad T_k = grad M^T T_q / sqrt(H).
 * qot_tconv_bwd_dst_rows: a workgroup owns table row r for one of `parts` slices of the graphs; leaves parts * n partial
 * rows (row part * n + r) of qot_tconv_rows_ld(n, H, D) floats, [grad T_skip (H) | grad M (qot_tconv_rows_npad(n): n
 * rounded up to 32, the tail zero) | grad P (D)], parts * n rows of H * D lin_edge partials (sum of g_i[c] p2_i[d]; the
 * T_q part of that gradient is rs * T_q^T grad P), grad_skip [N, H] (gradient wrt the conv output behind the fused
 * activation) and escr / delta for the source pass.  qot_tconv_bwd_src_rows: parts * n partial rows of grad T_v (H).
 * The caller sums the `parts` row blocks in order.  H in {64, 128, 256}; qot_tconv_rows_supported(n, H, D). */
int qot_tconv_rows_supported(int n, int H, int D);
int qot_tconv_rows_npad(int n);
int qot_tconv_rows_ld(int n, int H, int D);
int qot_tconv_bwd_dst_rows(const float* grad_out, const float* q, const float* v, int ld, const float* edge_attr,
                           const float* w_edge, const float* stats, const int32_t* rowptr, const int32_t* colf,
                           const int32_t* eid, const float* scores, int ld_scores, float* grad_skip, float* escr, float* delta,
                           const float* y_act, float act_slope, float act_p, uint64_t act_seed, const int64_t* act_step, int n,
                           int64_t B, int parts, float* part_rows, float* wedge_partials, int H, int D, qot_stream_t stream);
/* forward of the row form: out / stats bit-equal to qot_tconv_fwd_scores; a workgroup owns table row r for one of `parts`
 * slices of the B graphs (row r of the score matrix staged once, q_r W_e and the skip row in registers) */
int qot_tconv_fwd_rows(const float* q, const float* v, const float* skip, int ld, const float* scores, int ld_scores,
                       const float* edge_attr, const float* w_edge, const int32_t* rowptr, const int32_t* colf, const int32_t* eid,
                       float* out, float* stats, int n, int64_t B, int parts, int H, int D, int act, float act_slope, float act_p,
                       uint64_t act_seed, const int64_t* act_step, qot_stream_t stream);
int qot_tconv_bwd_src_rows(const float* grad_skip, const float* escr, const int32_t* rowptr_t, const int32_t* col_t,
                           const int32_t* pos_t, int n, int64_t B, int parts, float* part_rows, int H, qot_stream_t stream);
int qot_tconv_fwd(const float* q, const float* k, const float* v, const float* skip, int ld,
                  const float* edge_attr, const float* w_edge, const int32_t* rowptr,
                  const int32_t* col, const int32_t* eid, const int32_t* rowmap, float* out,
                  float* stats, int64_t N, int H, int D, int act, float act_slope, float act_p,
                  uint64_t act_seed, const int64_t* act_step, qot_stream_t stream);
/* Forward in the TILE form of table mode (rowmap != NULL, node_ids == arange(tile_n) in each of the tile_B graphs, N = tile_n *
 * tile_B; col = table row of every in-edge's source): a workgroup takes node r of qot_tconv_rows_per_block(H) consecutive
 * graphs, forms row r of T_q T_k^T / sqrt(H) once in LDS, and an in-edge costs one lookup there instead of a key-row
 * gather + dot + lane reduction.  Same results as qot_tconv_fwd up to the order of that dot.  tile_n <= 12288. */
int qot_tconv_fwd_tile(const float* q, const float* k, const float* v, const float* skip, int ld,
                       const float* edge_attr, const float* w_edge, const int32_t* rowptr, const int32_t* col,
                       const int32_t* eid, const int32_t* rowmap, float* out, float* stats, int64_t N, int H, int D,
                       int tile_n, int64_t tile_B, int act, float act_slope, float act_p, uint64_t act_seed,
                       const int64_t* act_step, qot_stream_t stream);
/* bwd, destination pass: grad_q[N,H] (ld_g), grad_skip[N,H] (= grad wrt the conv output, same ld_g;
 * may be NULL), per-edge scratch escr[cap,2] = (alpha, dalpha), delta[N], pds[N,D] = sum_e ds_e ea_e,
 * pal[N,D] = sum_e alpha_e ea_e.
 * y_act != NULL: grad_out is the gradient wrt y = dropout(leaky_relu(conv)) (the epilogue qot_tconv_fwd
 * fused, models.py:54-55) and y_act is that output; the kernel goes back through the activation
 * itself, so grad_skip then holds the gradient wrt the conv output -- pass IT (ld_go = ld_g) to
 * qot_tconv_bwd_src.  grad_w_edge != NULL: grad of lin_edge.weight [H,D] is produced as well
 * (fixed-order block partials; workspace qot_tconv_bwd_dst_workspace_floats(N,H,D) floats),
 * in the same launch.
 * Tile mode (grad_part != NULL; table mode with node_ids == arange(tile_n) for each of the tile_B graphs,
 * N = tile_n * tile_B): a workgroup takes node r of RPB = qot_tconv_rows_per_block(H) consecutive graphs and pre-reduces the
 * table gradient over them: grad_part[ceil(tile_B/RPB), tile_n, 4H] receives the partial sums of
 * grad_q (columns 0..H) and grad_skip (3H..4H) here and of grad_k / grad_v (H..3H) in qot_tconv_bwd_src;
 * the caller sums its first axis.  grad_q may then be NULL; grad_skip[N,H] is still written per node
 * (the source pass gathers it).
 * destinations per workgroup of the TransformerConv kernels at width H (1024/H today; callers size grad_part and the
 * workspace from this query, not from the formula); 0 for an unsupported width.
 * grad_w_edge == NULL with workspace != NULL: the per-workgroup partials [qot_tconv_bwd_dst_blocks(...), H*D] are
 * left in the workspace for the caller to sum (QOT_ROLE_SUM_ROWS of the step's backward epilogue). */
int qot_tconv_rows_per_block(int H);
int64_t qot_tconv_bwd_dst_blocks(int64_t N, int H, int tile_n, int64_t tile_B);
size_t qot_tconv_bwd_dst_workspace_floats(int64_t N, int H, int D);
int qot_tconv_bwd_dst(const float* grad_out, const float* q, const float* k, const float* v, int ld,
                      const float* edge_attr, const float* w_edge, const float* stats,
                      const int32_t* rowptr, const int32_t* col, const int32_t* eid,
                      const int32_t* rowmap, float* grad_q, float* grad_skip, int ld_g, float* escr,
                      float* delta, float* pds, float* pal, const float* y_act, float act_slope, float act_p,
                      uint64_t act_seed, const int64_t* act_step, float* grad_w_edge, float* workspace,
                      int tile_n, int64_t tile_B, float* grad_part, int64_t N, int H, int D, qot_stream_t stream);
/* bwd, source pass: grad_k, grad_v [N,H] (ld_g).  qmap_t (table mode, else NULL): table row of
 * each out-edge's destination (node_ids[col_t]) for the q gather; grad_out (row stride ld_go) and
 * delta stay per node. */
int qot_tconv_bwd_src(const float* grad_out, int ld_go, const float* q, int ld, const float* escr,
                      const float* delta, const int32_t* rowptr_t, const int32_t* col_t,
                      const int32_t* pos_t, const int32_t* qmap_t, float* grad_k, float* grad_v,
                      int ld_g, int tile_n, int64_t tile_B, float* grad_part, int64_t N, int H,
                      qot_stream_t stream);

/* ---- TransformerConv, table mode, GRAPH form (csrc/tconv_graph.hip) -----------------
 * The reference's only mode (topological_training/models.py:51-53 with dataset.py:78): x = node_embeddings(node_ids),
 * node_ids == arange(n) in each of the B graphs, so q / k / v / skip rows are rows of the projected table t4 [V, 4H]
 * (ld floats per row) and the dense part of a logit is an entry of the score matrix
 *     M [n, qot_tconv_graph_ldm(n)] = T_q T_k^T / sqrt(H),      P [n, D] = T_q w_edge / sqrt(H)
 * (qot_table_scores forms both from the PARAMETERS: table [V,H], lin_query, lin_key, lin_edge).  A workgroup takes
 * whole graphs (graph b owns nodes [b n, (b+1) n) and the index slots [rowptr[b n], rowptr[(b+1) n)) of the
 * block-diagonal batch) with M and T_v in LDS.  colf = node_ids[col] (= the source's index inside its graph).
 * qot_tconv_graph_supported(n, max_e, H, D): n <= 128 and the LDS images fit (max_e = most edges of one graph);
 * the entry points return QOT_ERR_UNSUPPORTED otherwise and callers use qot_tconv_fwd / qot_tconv_bwd_*.
 * fwd: out [N,H] as qot_tconv_fwd (same fused activation, same dropout mask); row = the index's destination of every
 * CSR slot.  Left behind for the backward (it needs no logits): alpha [E] = the attention weights, ea_csr [E,D] = the
 * edge features, both in CSR slot order, and aa [N,D] = sum_e alpha_e ea_e per destination.
 * bwd: grad_out [N,H] (through the fused activation when y_act != NULL, as qot_tconv_bwd_dst) ->
 * partials [qot_tconv_bwd_graph_blocks(B)][qot_tconv_graph_row_floats(n,H,D)], one row per workgroup:
 *     [ grad T_v n*H | grad T_skip n*H | grad M n*ldm | grad P n*D | grad w_edge (value path) H*D ]   (parts padded to 4 floats)
 * The caller sums the rows in order (QOT_ROLE_SUM_ROWS: bitwise reproducible).
 * qot_table_project_bwd_scores is qot_table_project_bwd fed by that summed row S: it forms
 *     grad T_q = (grad M T_k + grad P w_edge^T) / sqrt(H),   grad T_k = grad M^T T_q / sqrt(H)
 * where it needs them (rows >= n of the table receive zero) -- the [V, 4H] table gradient is never materialised -- and
 * completes grad w_edge [H,D] = S's value-path share + T_q^T grad P / sqrt(H). */
int qot_tconv_graph_supported(int n, int max_e, int H, int D);
int qot_tconv_graph_ldm(int n);
size_t qot_tconv_graph_row_floats(int n, int H, int D);
int qot_tconv_bwd_graph_blocks(int64_t B);
int qot_table_scores(const float* table, const float* wq, const float* bq, const float* wk, const float* bk,
                     const float* w_edge, float* M, float* P, int n, int H, int D, qot_stream_t stream);
int qot_tconv_fwd_graph(const float* t4, int ld, const float* M, const float* P, const float* w_edge,
                        const float* edge_attr, const int32_t* rowptr, const int32_t* colf, const int32_t* eid,
                        const int32_t* row, float* out, float* alpha, float* ea_csr, float* aa, int n, int64_t B, int max_e,
                        int H, int D, int act, float act_slope, float act_p, uint64_t act_seed, const int64_t* act_step,
                        qot_stream_t stream);
int qot_tconv_bwd_graph(const float* grad_out, const float* y_act, float act_slope, float act_p, uint64_t act_seed,
                        const int64_t* act_step, const float* t4, int ld, const float* w_edge, const float* ea_csr,
                        const float* alpha, const float* aa, const int32_t* rowptr, const int32_t* colf,
                        const int32_t* row, const int32_t* rowptr_t, const int32_t* col_t, const int32_t* pos_t,
                        float* partials, int n, int64_t B, int max_e, int H, int D, qot_stream_t stream);
int qot_table_project_bwd_scores(const float* S, const float* t4, const float* w_edge, const float* table,
                                 const float* wq, const float* wk, const float* wv, const float* ws, float* grad_table,
                                 float* grad_w, float* grad_b, float* grad_w_edge, int V, int n, int H, int D,
                                 qot_stream_t stream);

/* ---- NNConv (aggr = mean), factorised ----------------------------------------------
 * h_e = relu(W1 ea_e + b1) in R^K, K = 2D.  Builds the GEMM operand
 *   A[i] = [ invdeg_i * sum_e h_e[0] x_j | ... | invdeg_i * sum_e h_e[K-1] x_j |
 *            invdeg_i * sum_e x_j | x_i ]                       ([N, (K+2) H])
 * so that NNConv(x) = A @ Wcat + bias with Wcat = [W2 blocks; b2 block; W_root^T].
 * transpose != 0 runs the same aggregation over the CSC (out-edges) with the scale taken
 * at the gathered end (invdeg[col]) -- the adjoint used for grad_x.
 */
int qot_nnconv_agg(const float* x, int ld_x, const float* edge_attr, const float* w1,
                   const float* b1, const int32_t* rowptr, const int32_t* col,
                   const int32_t* eid_or_pos, const int32_t* eid_of_pos, const float* invdeg,
                   int transpose, float* A, int64_t N, int H, int D, qot_stream_t stream);
/* Fused form of qot_nnconv_agg + GEMM (+bias) for H == 64: the [32 x (K+2)H] operand tile is
 * built in LDS and multiplied on the matrix cores (v_mfma_f32_32x32x2_f32), A never touches
 * HBM.  w_perm = Wcat ([(K+2)H, H]) permuted into MFMA fragment order (layout documented in
 * csrc/nnconv_mfma.hip).  edge_ids[p] = original edge id of slot p of the index walked (eid for
 * the CSR, eid_t for the CSC).  transpose != 0: adjoint over the CSC with w_perm built from
 * Wcat^T blocks (grad_x).  bias may be NULL.  H == 64: the tuned tile kernel (csrc/nnconv_mfma.hip), w_perm =
 * Wcat in its fragment order.  H in {16, 32, 128, 256} (D <= 4): the per-pass tile kernel of csrc/nnconv_gen.hip,
 * w_perm in ITS layout (one pass per 64 input channels; documented there, built by
 * functional.nnconv_gen_perm_index); transpose = 2 / 3 selects that kernel for H == 64 too (measurements). */
int qot_nnconv_fused(const float* x, int ld_x, const float* edge_attr, const float* w1,
                     const float* b1, const int32_t* rowptr, const int32_t* col,
                     const int32_t* edge_ids, const float* invdeg, int transpose,
                     const float* w_perm, const float* bias, float* out, int64_t N, int H, int D,
                     int act, float act_slope, float act_p, uint64_t act_seed, const int64_t* act_step,
                     qot_stream_t stream);
/* C[KT,64] = A[N,KT]^T @ G[N,64] (fp32 MFMA, operands streamed from HBM in fragment order,
 * deterministic slab reduction).  Weight-gradient GEMM of NNConv (gWcat = A^T g) -- the shape
 * library GEMMs run at 13-28 TFLOP/s.  KT multiple of 128, <= 1280.  workspace:
 * qot_gemm_tn_workspace_floats(KT) floats. */
size_t qot_gemm_tn_workspace_floats(int KT);
int qot_gemm_tn(const float* A, int lda, const float* G, int ldg, int64_t N, int KT, float* C,
                float* workspace, qot_stream_t stream);
/* NNConv backward, data and weight gradients from one gather (H == 64, D <= 4): grad_x = U @ WcatT
 * (as qot_nnconv_fused transpose=1) and gwcat_t = per-block transposes of d/dWcat
 * (gwcat_t[k*64+o][a] = dWcat[k*64+a][o]) computed as X^T U from the same LDS tile.  Replaces
 * {qot_nnconv_fused(transpose=1), qot_nnconv_agg, qot_gemm_tn}.  x = the forward input rows.
 * param_layout = 1 writes the weight gradient in the parameters' own layouts instead:
 * [d nn.2.weight [H*H, K] | d nn.2.bias [H*H] | d lin.weight [H, H]] (same total size).
 * param_layout = 2 (profiling only) launches the main kernel alone: per-workgroup slabs stay in the
 * workspace and gwcat_t is not written -- lets bench.py time exactly the kernel rocprofv3 lists.
 * workspace: qot_nnconv_adjoint_dw_workspace_floats(D) floats. */
size_t qot_nnconv_adjoint_dw_workspace_floats(int D);
int qot_nnconv_adjoint_dw(const float* grad_out, int ld_g, const float* x, int ld_x,
                          const float* edge_attr, const float* w1, const float* b1,
                          const int32_t* rowptr_t, const int32_t* col_t, const int32_t* eid_t,
                          const float* invdeg, const float* w_perm, float* grad_x, float* gwcat_t, int param_layout,
                          float* workspace, int64_t N, int H, int D, qot_stream_t stream);
/* Fused form of {GA = g @ Wk^T ; qot_nnconv_bwd_edge}, D <= 4: the GA tile is produced by MFMA into LDS and
 * consumed there.  b_perm: Wk^T in fragment order (H == 64: csrc/nnconv_mfma.hip; H in {16, 32, 128, 256}:
 * 32 input channels per pass, layout in csrc/nnconv_gen.hip, built by functional.nnconv_gen_gradh_perm_index).
 * workspace: qot_nnconv_gradh_workspace_floats(D) floats.  gw1/gb1 are overwritten. */
size_t qot_nnconv_gradh_workspace_floats(int D);
int qot_nnconv_gradh_fused(const float* grad_out, int ld_g, const float* x, int ld_x,
                           const float* edge_attr, const float* w1, const float* b1,
                           const int32_t* rowptr, const int32_t* col, const int32_t* eid,
                           const float* invdeg, const float* b_perm, float* gw1, float* gb1,
                           float* workspace, int64_t N, int H, int D, qot_stream_t stream);
/* Weight gradient of NNConv for every supported width (H in {16, 32, 64, 128, 256}, D <= 4) without
 * materialising the [N, (K+2)H] operand: grad_params = [d nn.2.weight [H*H, K] | d nn.2.bias [H*H] |
 * d lin.weight [H, H]] (the parameters' own layouts, (2D+2)*H*H floats) = A^T grad_out, with the operand
 * tile A of 32 destinations gathered into LDS 16 input channels at a time (csrc/nnconv_gen.hip).  x = the
 * forward input rows; rowptr/col/eid = CSR by destination.  Replaces {qot_nnconv_agg, A^T g GEMM} of
 * NNConv's autograd (topological_training/models.py:57 under loss.backward(), train.py:115).
 * workspace: qot_nnconv_dw_workspace_floats(N, H, D) floats.  Fixed summation order. */
size_t qot_nnconv_dw_workspace_floats(int64_t N, int H, int D);
int qot_nnconv_dw(const float* x, int ld_x, const float* grad_out, int ld_g, const float* edge_attr,
                  const float* w1, const float* b1, const int32_t* rowptr, const int32_t* col,
                  const int32_t* eid, const float* invdeg, float* grad_params, float* workspace, int64_t N,
                  int H, int D, qot_stream_t stream);
/* H == 64: both second-stage sums of the NNConv backward in ONE launch.  Call qot_nnconv_adjoint_dw with
 * param_layout = 2 (slabs stay in its workspace) and qot_nnconv_gradh_fused with gw1 = gb1 = NULL (block partials stay in
 * its workspace), then this: grad_params as qot_nnconv_adjoint_dw(param_layout = 1), gw1[K,D], gb1[K].  Fixed order. */
int qot_nnconv_bwd_finalize(const float* adj_workspace, const float* gradh_workspace, float* grad_params,
                            float* gw1, float* gb1, int64_t N, int H, int D, qot_stream_t stream);
/* grad of the edge MLP's first layer: GA[N, K*H] = g @ Wcat[:K*H]^T (caller GEMM);
 * gw1[K,D], gb1[K] zero-filled by caller, accumulated with atomics. */
int qot_nnconv_bwd_edge(const float* GA, int ld_ga, const float* x, int ld_x,
                        const float* edge_attr, const float* w1, const float* b1,
                        const int32_t* rowptr, const int32_t* col, const int32_t* eid,
                        const float* invdeg, float* gw1, float* gb1, int64_t N, int H, int D,
                        qot_stream_t stream);

/* ---- activation: y = dropout(leaky_relu(x, slope), p) ------------------------------
 * Counter-based RNG: keep = hash(seed, *step_counter, element) >= p.  step_counter is a
 * DEVICE int64 the caller bumps once per train step (graph-replay safe).  p == 0 or
 * step_counter == NULL disables dropout.  bwd regenerates the mask from the same inputs.
 */
int qot_act_fwd(const float* x, float* y, int64_t n, float slope, float p, uint64_t seed,
                const int64_t* step_counter, qot_stream_t stream);
int qot_act_bwd(const float* grad_y, const float* y_or_x, float* grad_x, int64_t n, float slope,
                float p, uint64_t seed, const int64_t* step_counter, qot_stream_t stream);

/* ---- global mean pool -------------------------------------------------------------- */
int qot_pool_fwd(const float* x, const int32_t* ptr, float* out, int64_t B, int H,
                 qot_stream_t stream);
int qot_pool_bwd(const float* grad_out, const int32_t* ptr, const int32_t* batch, float* grad_x,
                 int64_t N, int64_t B, int H, qot_stream_t stream);

/* ---- GATConv (concat heads) ---------------------------------------------------------
 * z[N, heads*C], a_src/a_dst[N, heads]; graph = CSR built with gat_self_loops.
 * out[N, heads*C] (+bias fused); stats[N, heads, 2].  stats and escr are read and written as (float, float) pairs:
 * 8-byte aligned, as are all row matrices 16-byte (QOT_ERR_BADARG otherwise).
 * qot_gat_logits: a_src[n,h] = <z[n,h,:], att_src[h,:]>, a_dst likewise (App. B.3), one pass over z.
 * qot_gat_fwd, bn_partials != NULL: also writes per-workgroup column (mean, sum of squared deviations) of out - bias,
 *   [qot_gat_blocks(N, heads, C)][2][heads*C] floats (buffer of qot_gat_bn_partials_floats), which
 *   qot_bn_stats_from_partials(shift = bias, ...) turns into the batch statistics of the BatchNorm that follows
 *   (lightpath_training/models.py:30-31) without another pass over out. */
int qot_gat_blocks(int64_t N, int heads, int C);
size_t qot_gat_bn_partials_floats(int64_t N, int heads, int C);
int qot_gat_logits(const float* z, const float* att_src, const float* att_dst, float* a_src, float* a_dst,
                   int64_t N, int heads, int C, qot_stream_t stream);
int qot_gat_fwd(const float* z, const float* a_src, const float* a_dst, const float* bias,
                const int32_t* rowptr, const int32_t* col, float* out, float* stats, int64_t N,
                int heads, int C, float neg_slope, float* bn_partials, qot_stream_t stream);
/* Thin forms (r04): the layer's projection z = x W^T (x [N, K], K <= 8 input features -- LightpathGNN's first layer, K = 5;
 * W [heads*C, K] = GATConv.lin.weight) is formed inside the attention kernels, so that z is never written or read.  Same
 * outputs and partials layout as qot_gat_fwd / qot_gat_bwd_dst; a_src / a_dst are the caller's (x (W_h^T att_h): two
 * qot_skinny_linear_fwd of width 4).  qot_gat_thin_supported: 1 when K is in 1..8 and W^T fits next to the walk's LDS image;
 * the entry points return QOT_ERR_UNSUPPORTED otherwise. */
int qot_gat_thin_supported(int heads, int C, int K);
int qot_gat_fwd_thin(const float* x, int K, const float* w, const float* a_src, const float* a_dst, const float* bias,
                     const int32_t* rowptr, const int32_t* col, float* out, float* stats, int64_t N, int heads, int C,
                     float neg_slope, float* bn_partials, qot_stream_t stream);
int qot_gat_bwd_dst_thin(const float* grad_out, const float* x, int K, const float* w, const float* a_src,
                         const float* a_dst, const float* stats, const int32_t* rowptr, const int32_t* col,
                         float* grad_a_dst, float* escr, float* delta, int64_t N, int heads, int C, float neg_slope,
                         float* grad_bias, float* workspace, qot_stream_t stream);
/* destination pass: grad_a_dst[N,heads], escr[cap, heads, 2] = (alpha, dalpha), delta[N,heads].
 * grad_bias != NULL: also GATConv's bias gradient [heads*C] = column sums of grad_out (every row passes through this
 * kernel once anyway); workspace: qot_gat_bn_partials_floats(N, heads, C) floats. */
int qot_gat_bwd_dst(const float* grad_out, const float* z, const float* a_src, const float* a_dst,
                    const float* stats, const int32_t* rowptr, const int32_t* col, float* grad_a_dst,
                    float* escr, float* delta, int64_t N, int heads, int C, float neg_slope,
                    float* grad_bias, float* workspace, qot_stream_t stream);
/* source pass: grad_z[N, heads*C], grad_a_src[N, heads].  att_src != NULL (logits formed by qot_gat_logits):
 * grad_z also receives grad_a_src[j,h] att_src[h,:] + grad_a_dst[j,h] att_dst[h,:]. */
int qot_gat_bwd_src(const float* grad_out, const float* a_src, const float* a_dst, const float* escr,
                    const float* delta, const int32_t* rowptr_t, const int32_t* col_t,
                    const int32_t* pos_t, float* grad_z, float* grad_a_src, int64_t N, int heads,
                    int C, float neg_slope, const float* att_src, const float* att_dst,
                    const float* grad_a_dst, qot_stream_t stream);
/* grad_att_src[h*C + c] = sum_n grad_a_src[n,h] z[n,h,c] (grad_att_dst with grad_a_dst); fixed summation order.
 * workspace: qot_gat_bn_partials_floats(N, heads, C) floats. */
int qot_gat_att_grad(const float* z, const float* grad_a_src, const float* grad_a_dst, float* grad_att_src,
                     float* grad_att_dst, float* workspace, int64_t N, int heads, int C, qot_stream_t stream);

/* ---- BatchNorm1d (+ fused ReLU) over the node matrix [N, C] -------------------------
 * qot_bn_stats: mean[C], rstd[C] (biased var, eps), and when running_* != NULL the
 * momentum update with the unbiased variance.  partials: workspace of
 * qot_bn_partials_floats(N, C) floats. */
size_t qot_bn_partials_floats(int64_t N, int C);
int qot_bn_stats(const float* x, int64_t N, int C, float eps, float momentum, float* mean,
                 float* rstd, float* running_mean, float* running_var, float* partials,
                 qot_stream_t stream);
/* y = relu?((x - mean) * rstd * w + b) */
/* batch statistics from per-workgroup column (mean, M2 = sum of squared deviations) pairs [nblk][2][C] of x - shift
 * written by the producer of x (qot_gat_fwd: workgroup b holds rows [b * chunk_rows, min((b + 1) * chunk_rows, N)),
 * chunk_rows = qot_gat_chunk_rows(N, heads, C)), merged with Chan's formula in fp64: same results as qot_bn_stats
 * without reading x again, and no E[d^2] - E[d]^2 cancellation for channels whose mean is far from the shift. */
int64_t qot_gat_chunk_rows(int64_t N, int heads, int C);
int qot_bn_stats_from_partials(const float* shift, const float* partials, int nblk, int64_t chunk_rows, int64_t N, int C,
                               float eps, float momentum, float* mean, float* rstd, float* running_mean,
                               float* running_var, qot_stream_t stream);
int qot_bn_apply(const float* x, const float* mean, const float* rstd, const float* w,
                 const float* b, float* y, int64_t N, int C, int relu, qot_stream_t stream);
/* train-mode backward: needs column sums first (qot_bn_bwd_reduce -> gw[C], gb[C]), then
 * qot_bn_bwd_apply.  eval mode (batch_stats == 0) skips the mean-subtraction terms.
 * relu != 0 with y == NULL: the ReLU mask is recomputed from x, mean, rstd, w (weight), b (bias) with the forward's
 * expression instead of being read from the saved output -- one [N, C] read fewer in each kernel. */
int qot_bn_bwd_reduce(const float* grad_y, const float* y, const float* x, const float* mean,
                      const float* rstd, float* gw, float* gb, int64_t N, int C, int relu,
                      float* partials, const float* w, const float* b, qot_stream_t stream);
int qot_bn_bwd_apply(const float* grad_y, const float* y, const float* x, const float* mean,
                     const float* rstd, const float* w, const float* gw, const float* gb,
                     float* grad_x, int64_t N, int C, int relu, int batch_stats, const float* b,
                     qot_stream_t stream);
/* BatchNorm(+ReLU) of which only some rows are consumed: y_rows = act(BatchNorm(x))[idx] -- LightpathGNN normalises the
 * whole node matrix and then keeps the LUT nodes' rows (lightpath_training/models.py:31-32, 35-40).  Statistics over all
 * N rows as before (qot_bn_stats / qot_bn_stats_from_partials); idx: n unique row numbers.
 * qot_bn_apply_rows: the n output rows only.  qot_bn_bwd_reduce_rows: the column sums over the n rows that carry a
 * gradient (partials: qot_bn_partials_floats(n, C) floats).  qot_bn_bwd_apply_rows: grad_x [N, C] (batch-statistics
 * form), bit for bit what qot_bn_bwd_apply gives for the zero-filled [N, C] gradient with grad_rows scattered into it. */
int qot_bn_apply_rows(const float* x, const int32_t* idx, int64_t n, const float* mean, const float* rstd, const float* w,
                      const float* b, float* y_rows, int C, int relu, qot_stream_t stream);
int qot_bn_bwd_reduce_rows(const float* grad_rows, const int32_t* idx, int64_t n, const float* x, const float* mean,
                           const float* rstd, float* gw, float* gb, int C, int relu, float* partials, const float* w,
                           const float* b, qot_stream_t stream);
int qot_bn_bwd_apply_rows(const float* grad_rows, const int32_t* idx, int64_t n, const float* x, const float* mean,
                          const float* rstd, const float* w, const float* gw, const float* gb, float* grad_x, int64_t N,
                          int C, int relu, const float* b, qot_stream_t stream);

/* ---- train-step envelope helpers (topological_training/train.py:109-116) ----------------
 * qot_sgd_momentum: torch.optim.SGD(lr, momentum) update over one flat buffer; first_step != 0
 * initialises the momentum buffer with the gradient (torch semantics).
 * qot_small_gemm: C[M,N] = op(A) op(B) (+bias) with element strides (head MLP, table projection:
 * topological_training/models.py:33-38,63; a few MFLOP each).  split_k > 1: C holds split_k partial
 * planes [M, ldc] (K chunks of ceil(K/split_k) rounded to 32), summed by the caller in a fixed order.
 * qot_colsum: out[C] = column sums of x[N, C] (bias gradients), deterministic two-stage.
 * qot_act_bwd_colsum: grad_x = qot_act_bwd(grad_y, y) AND colsum(grad_x) in the same pass (bias
 * gradient of a conv whose epilogue carried leaky_relu+dropout, models.py:54-58).
 * Both take qot_colsum_workspace_floats(C) floats of workspace. */
int qot_sgd_momentum(float* param, const float* grad, float* momentum_buf, int64_t n, float lr,
                     float momentum, int first_step, qot_stream_t stream);
/* The same update with the gradient pack folded in: parameter i's gradient is read from grads[i]
 * (device pointer, NULL = zero gradient; `grads` and `offsets` are HOST arrays of `count` pointers and
 * count+1 element offsets into the flat buffers, offsets[0] = 0, offsets[count] = n, count <= 48), the
 * packed value is also stored to grad_flat (may be NULL).  lr_dev != NULL: learning rate read from device
 * memory (as qot_sgd_momentum_dev), else `lr`.  Replaces FlatModel.gather_grads' concatenation launch. */
int qot_sgd_momentum_multi(float* param, const float* const* grads, const int64_t* offsets, int count,
                           float* grad_flat, float* momentum_buf, int64_t n, float lr, const float* lr_dev,
                           float momentum, int first_step, qot_stream_t stream);
/* same update with the learning rate read from device memory at run time, so that a step captured in a
 * HIP graph follows the scheduler (StepLR, topological_training/train.py:67,181) without re-capture */
int qot_sgd_momentum_dev(float* param, const float* grad, float* momentum_buf, int64_t n, const float* lr_dev,
                         float momentum, int first_step, qot_stream_t stream);
int qot_small_gemm(const float* A, int64_t stride_am, int64_t stride_ak, const float* B, int64_t stride_bk,
                   int64_t stride_bn, const float* bias, float* C, int ldc, int M, int N, int K,
                   int split_k, qot_stream_t stream);
size_t qot_colsum_workspace_floats(int C);
/* out[C] = sum over the B rows of x[B, C] for wide rows (sum over graphs of a per-node gradient,
 * TransformerConv table mode); workspace: qot_rowsum_wide_workspace_floats(C) floats. */
size_t qot_rowsum_wide_workspace_floats(int64_t C);
int qot_rowsum_wide(const float* x, int64_t B, int64_t C, float* out, float* workspace, qot_stream_t stream);
int qot_colsum(const float* x, int ld, int64_t N, int C, float* out, float* workspace, qot_stream_t stream);
int qot_act_bwd_colsum(const float* grad_y, const float* y, float* grad_x, int64_t N, int C, float slope, float p,
                       uint64_t seed, const int64_t* step_counter, float* colsum_out, float* workspace,
                       qot_stream_t stream);

/* ---- fused read-out head: global_mean_pool -> Linear(H,H) -> LeakyReLU -> Dropout -> Linear(H,O)
 * (topological_training/models.py:33-38,61-63).  fwd saves pooled[B,H] and hidden[B,H] (post dropout).
 * bwd: grad_x[N,H] (pool backward included) and grads = [gW0 | gb0 | gW3 | gb3] contiguous
 * (deterministic partial sums; workspace qot_head_bwd_workspace_floats(H, O) floats).  O <= 8.
 * x_in != NULL (the head's input, i.e. the last conv's output y = dropout(leaky_relu(conv)) with the
 * in_* activation parameters of that conv's epilogue, models.py:58-59): grad_x is then the gradient
 * wrt the CONV output (the activation backward is applied while the pool gradient is written) and
 * grads gets H more floats: its column sums = that conv's bias gradient.
 * grads == NULL: the per-workgroup partials [qot_head_bwd_blocks(B), H*H + H + O*H + O (+ H with x_in)] are left in
 * the workspace for the caller to sum (QOT_ROLE_SUM_ROWS of the step's backward epilogue). */
int qot_head_fwd(const float* x, const int32_t* ptr, const float* w0, const float* b0, const float* w3,
                 const float* b3, float* pooled, float* hidden, float* out, int64_t B, int H, int O,
                 float slope, float p, uint64_t seed, const int64_t* step_counter, qot_stream_t stream);
/* As qot_head_fwd, with the criterion of the train step folded in (topological_training/train.py:69,113-115:
 * SmoothL1Loss(reduction="mean", beta) on out vs target[B,O]): also writes grad_out[B,O] = d loss / d out and
 * loss_rows[B] = each graph's share of the mean loss; the loss value is the sum of loss_rows (the caller sums it
 * where it sums its other partials: QOT_ROLE_SUM_ROWS).  Replaces the separate qot_smooth_l1 launch of a step. */
int qot_head_fwd_loss(const float* x, const int32_t* ptr, const float* w0, const float* b0, const float* w3,
                      const float* b3, float* pooled, float* hidden, float* out, int64_t B, int H, int O,
                      float slope, float p, uint64_t seed, const int64_t* step_counter, const float* target,
                      float beta, float* grad_out, float* loss_rows, qot_stream_t stream);
/* The read-out of a TRAIN step in one kernel: forward + SmoothL1(mean, beta) + backward per graph (a graph's rows are read
 * once and stay in LDS for the pool backward; pooled / hidden never leave the workgroup).  grad_x[N,H] is the gradient wrt
 * the CONV output when fold != 0 (x = dropout(leaky_relu(conv)) with the in_* parameters, as qot_head_bwd(x_in)).
 * workspace: per-workgroup parameter-gradient partials [qot_head_train_blocks(B, H)][H*H + H + O*H + O (+ H when fold)] for
 * the caller to sum (QOT_ROLE_SUM_ROWS); the loss value is the sum of loss_rows[B].  Replaces {qot_head_fwd_loss, qot_head_bwd}
 * of a step (topological_training/train.py:111-115).  H <= 64 (r04): 256 / H graphs per pass of a workgroup, a graph's rows
 * kept in registers from the pool to the pool backward (csrc/head.hip: head_train_batched_kernel). */
int qot_head_train_blocks(int64_t B, int H);
int qot_head_train(const float* x, const int32_t* ptr, const float* w0, const float* b0, const float* w3, const float* b3,
                   const float* target, float beta, float* out, float* grad_out, float* loss_rows, float* grad_x,
                   float* workspace, int64_t B, int H, int O, float slope, float p, uint64_t seed,
                   const int64_t* step_counter, int fold, float in_slope, float in_p, uint64_t in_seed,
                   const int64_t* in_step, qot_stream_t stream);
size_t qot_head_bwd_workspace_floats(int H, int O);
int qot_head_bwd_blocks(int64_t B);
int qot_head_bwd(const float* grad_out, const float* pooled, const float* hidden, const int32_t* ptr,
                 const float* w0, const float* w3, float* grad_x, float* grads, float* workspace, int64_t B,
                 int H, int O, float slope, float p, uint64_t seed, const int64_t* step_counter,
                 const float* x_in, float in_slope, float in_p, uint64_t in_seed, const int64_t* in_step,
                 qot_stream_t stream);

/* ---- SmoothL1Loss(reduction="mean", beta) value and gradient in one launch
 * (criterion at topological_training/train.py:69, used at :113-115).  grad = d loss / d pred.
 * workspace: qot_smooth_l1_workspace_floats() floats whose first word is 0 before the first call
 * (the kernel leaves it 0). */
size_t qot_smooth_l1_workspace_floats(void);
int qot_smooth_l1(const float* pred, const float* target, int64_t n, float beta, float* loss, float* grad,
                  float* workspace, qot_stream_t stream);

/* ---- embedding-table projection for TransformerConv table mode: out[V,4H] = [q|k|v|skip] rows of
 * table[V,H] under lin_query/lin_key/lin_value/lin_skip (topological_training/models.py:51-53 with
 * x = node_embeddings(node_ids)); reads the four Linear parameters in place.
 * step_counter != NULL: the launch also performs qot_step_advance(step_counter, step_snapshot) -- in table
 * mode it is the forward's first kernel, and the dropout counter's own launch was ~5 us of a 0.6 ms step.
 * bwd: grad_table[V,H], grad_w[4H,H] and grad_b[4H] in q|k|v|skip order. */
int qot_table_project_fwd(const float* table, const float* wq, const float* bq, const float* wk, const float* bk,
                          const float* wv, const float* bv, const float* ws, const float* bs, float* out, int V,
                          int H, int64_t* step_counter, int64_t* step_snapshot, qot_stream_t stream);
int qot_table_project_bwd(const float* grad_out, const float* table, const float* wq, const float* wk,
                          const float* wv, const float* ws, float* grad_table, float* grad_w, float* grad_b, int V,
                          int H, qot_stream_t stream);

/* ---- TransformerConv table mode index maps in one launch (x = node_embeddings(node_ids),
 * topological_training/models.py:51): ids32[N] = node_ids, colf[E] = node_ids[col],
 * colf_t[E] = node_ids[col_t]. */
int qot_table_maps(const int64_t* node_ids, const int32_t* col, const int32_t* col_t, int32_t* ids32,
                   int32_t* colf, int32_t* colf_t, int64_t N, int64_t E, qot_stream_t stream);
/* ---- dropout draw counter: *counter += 1 and *snapshot = *counter, on the stream (replay-safe). */
int qot_step_advance(int64_t* counter, int64_t* snapshot, qot_stream_t stream);

/* ---- out[i] = concat(s0[0:n0], s1[0:n1], s2)[idx[i]]: one gather builds the fragment-ordered NNConv
 * operands from nn.2.weight / nn.2.bias / lin.weight (topological_training/models.py:20-25). */
int qot_gather3(const float* s0, int64_t n0, const float* s1, int64_t n1, const float* s2, const int32_t* idx,
                float* out, int64_t n, qot_stream_t stream);

/* ---- dense fp32 projections on the matrix cores (csrc/gemm.hip): GATConv's shared projection z = x W^T and its
 * autograd (lightpath_training/models.py:13,30; train.py:128), the LUT head's first layer (models.py:17-22).
 * qot_gemm_nt: C[M, N] = A'[M, K] . B[N, K]^T (+ bias[N]); scale != NULL: A' = relu(A * scale[k] + shift[k]) applied
 * while the operand is loaded (BatchNorm + ReLU of the previous layer, models.py:31-32, never materialised).
 * K multiple of 32; N, lda, ldb, ldc multiples of 4; operands 16-byte aligned.
 * qot_gemm_tn_planes: Cpart[splits, M, N], plane z = A[Kz, M]^T . B'[Kz, N] over its chunk of K (the weight gradient
 * g^T y, K = number of nodes; B' = relu(B * scale[n] + shift[n]) when scale != NULL); the caller sums the planes in
 * order (QOT_ROLE_SUM_ROWS: bitwise reproducible).  M, N multiples of 4; splits = qot_gemm_tn_splits(M, N, K). */
int qot_gemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K,
                const float* scale, const float* shift, const float* bias, qot_stream_t stream);
/* qot_gemm_nt_logits: qot_gemm_nt for GATConv with 128 channels per head (N = heads * 128: one 128-column output tile per
 * head), which also leaves the attention logits a_src[m, h] = <C[m, h, :], att_src[h, :]>, a_dst likewise ([M, heads];
 * PyG GATConv's alpha_src / alpha_dst, lightpath_training/models.py:13) -- formed in the epilogue instead of by another
 * pass over the [M, 4C] matrix (qot_gat_logits).  att_src / att_dst: [N] floats, 16-byte aligned. */
int qot_gemm_nt_logits(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K,
                       const float* scale, const float* shift, const float* bias, const float* att_src,
                       const float* att_dst, float* a_src, float* a_dst, qot_stream_t stream);
/* qot_gemm_nt for few output tiles and a long inner dimension: Cpart[kslices][M, N], plane s = the product over the s-th
 * slice of K (K / kslices a multiple of 32); the caller sums the planes in order (QOT_ROLE_SUM_ROWS). */
int qot_gemm_nt_planes(const float* A, int64_t lda, const float* B, int64_t ldb, float* Cpart, int64_t M, int N, int K,
                       int kslices, qot_stream_t stream);
int qot_gemm_tn_splits(int M, int N, int64_t K);
/* 1 when qot_gemm_nt / qot_gemm_nt_logits run a product of this size on the 256 x 256 x 32 tiles of csrc/gemm256.hip (one
 * persistent workgroup per CU; at least one tile per CU and N >= 256), 0 for the 128 x 128 tiles of csrc/gemm.hip.  Same
 * results either way up to the order of the K sum inside a tile; for tests and tools. */
int qot_gemm256_takes(int64_t M, int N);
int qot_gemm_tn_planes(const float* A, int64_t lda, const float* B, int64_t ldb, float* Cpart, int M, int N, int64_t K,
                       int splits, const float* scale, const float* shift, qot_stream_t stream);

/* ---- first-layer projection with a handful of input features (lightpath_training/models.py:13: GATConv(in_channels = 5,
 * ...).lin): out[N, C] = x[N, F] . w[C, F]^T, F <= 8, C multiple of 4 (<= 1024) -- pure bandwidth; and its weight gradient
 * g^T x as per-workgroup partials [qot_skinny_linear_dw_blocks(N)][C * F] the caller sums in order (QOT_ROLE_SUM_ROWS). */
int qot_skinny_linear_fwd(const float* x, const float* w, float* out, int64_t N, int F, int C, qot_stream_t stream);
/* ... with the attention logits of qot_gemm_nt_logits (C = heads * 128, C in {128, 256, 512, 1024}) */
int qot_skinny_linear_fwd_logits(const float* x, const float* w, float* out, int64_t N, int F, int C, const float* att_src,
                                 const float* att_dst, float* a_src, float* a_dst, qot_stream_t stream);
int qot_skinny_linear_dw_blocks(int64_t N);
int qot_skinny_linear_dw(const float* g, const float* x, float* partials, int64_t N, int F, int C, qot_stream_t stream);

/* ---- multi-role launch: several INDEPENDENT small jobs of one train step in ONE kernel launch -------------------
 * The reference's step (topological_training/train.py:109-116) reaches ~60 small torch / PyG kernels around the two
 * convolutions; on this engine the convolutions are a handful of launches and what is left are 4-9 us jobs whose
 * cost is the launch itself.  Jobs that do not depend on each other share a launch: role r gets a contiguous
 * range of 256-thread workgroups and runs the body of the standalone entry point of its kind (those entry points
 * are one-role calls of this function).  The table is copied into the kernel arguments: nothing is read from the
 * caller's array after the call returns, nothing is allocated, no synchronisation.
 * Jobs of one call MUST be independent: none may read what another one writes.
 *
 * kind                        p[] (device pointers)                                       i[] (host integers)
 * QOT_ROLE_CSR_BY_GRAPH       0 edge_index, 1 node_ptr, 2 edge_ptr, 3 rowptr, 4 col, 5 eid, 6 row, 7 rowptr_t,
 *                             8 col_t, 9 pos_t, 10 eid_t, 11 invdeg, 12 status, 13 node_ids, 14 ids32, 15 colf,
 *                             16 colf_t, 17 ptr32                                          0 E, 1 N, 2 B (> 0),
 *                             (as qot_csr_build_by_graph)                                  3 max_nodes, 4 max_edges
 * QOT_ROLE_TABLE_PROJECT_FWD  0 table, 1 wq, 2 bq, 3 wk, 4 bk, 5 wv, 6 bv, 7 ws, 8 bs, 9 out,
 *                             10 step_counter, 11 step_snapshot (as qot_table_project_fwd)  0 V (> 0), 1 H
 * QOT_ROLE_GATHER3            0 s0, 1 s1, 2 s2, 3 idx, 4 out (as qot_gather3)               0 n0, 1 n1, 2 n
 * QOT_ROLE_SUM_ROWS           0 partials [nblk, n], 1 out [groups, n]:                     0 nblk, 1 n, 2 rows per
 *                             out[g, t] = sum of partials[b, t] over the rows b of group g,   group (0 = all rows:
 *                             fixed order (bitwise reproducible)                            one output row);
 *                                                                                           3, 4 derived
 * QOT_ROLE_NNCONV_FINALIZE64  0 adj_workspace, 1 gradh_workspace, 2 grad_params, 3 gw1,    0 N, 1 D; 2..7 derived
 *                             4 gb1 (as qot_nnconv_bwd_finalize, H = 64)
 * QOT_ROLE_TABLE_PROJECT_BWD  0 grad_out, 1 table, 2 wq, 3 wk, 4 wv, 5 ws, 6 grad_table,   0 V, 1 H
 *                             7 grad_w, 8 grad_b (as qot_table_project_bwd)
 * QOT_ROLE_TABLE_SCORES       0 table, 1 wq, 2 bq, 3 wk, 4 bk, 5 w_edge, 6 M, 7 P          0 n, 1 H, 2 D
 *                             (as qot_table_scores)
 * QOT_ROLE_TABLE_PROJECT_BWD_SCORES  0 S, 1 t4, 2 w_edge, 3 table, 4 wq, 5 wk, 6 wv, 7 ws,  0 V, 1 n, 2 H, 3 D
 *                             8 grad_table, 9 grad_w, 10 grad_b, 11 grad_w_edge (as qot_table_project_bwd_scores)
 * "derived" fields are filled by the library in its own copy; callers leave them 0. */
#define QOT_MAX_ROLES 12
enum {
    QOT_ROLE_CSR_BY_GRAPH = 1,
    QOT_ROLE_TABLE_PROJECT_FWD = 2,
    QOT_ROLE_GATHER3 = 3,
    QOT_ROLE_SUM_ROWS = 4,
    QOT_ROLE_NNCONV_FINALIZE64 = 5,
    QOT_ROLE_TABLE_PROJECT_BWD = 6,
    QOT_ROLE_TABLE_SCORES = 7,
    QOT_ROLE_TABLE_PROJECT_BWD_SCORES = 8
};
typedef struct qot_role {
    int32_t kind;
    int32_t reserved;
    const void* p[18];
    int64_t i[8];
} qot_role_t;
int qot_run_roles(const qot_role_t* roles, int n_roles, qot_stream_t stream);

/* ---- row gather / scatter (LUT read-out and its adjoint) ---------------------------- */
int qot_rows_gather(const float* x, const int32_t* idx, float* out, int64_t n_idx, int C,
                    qot_stream_t stream);
/* grad_x must be zero-filled; idx values are unique (a boolean-mask selection) */
int qot_rows_scatter(const float* grad_out, const int32_t* idx, float* grad_x, int64_t n_idx, int C,
                     qot_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* QOT_GNN_H */
