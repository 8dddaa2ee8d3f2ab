/*
 * gnndelete_hip.h - C ABI of libgnndelete_hip.so, the MI355X (gfx950) kernels behind the
 * GNNDelete hot path.
 *
 * The reference (mims-harvard/GNNDelete) has no FFI: its hot path runs inside PyTorch and the
 * un-vendored torch_geometric ops.  Each entry below replaces the third-party/PyTorch op chain
 * the reference calls at the cited site (paths relative to /root/reference).  The host-side
 * mirror of the reference's Python operator API (gnndelete_amd/framework) binds these symbols
 * with ctypes; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned storage (PyTorch tensors); nothing
 *     is allocated, freed or retained; scratch is passed in explicitly;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     every call only enqueues work on it and returns (capturable in a hipGraph);
 *   - matrices are row-major fp32 with an explicit leading dimension (elements);
 *   - graph structure is CSR over TARGET rows with int32 indices:
 *       rowptr[n_rows+1], col[nnz] = source node of each in-edge, val[nnz] = edge weight;
 *   - return value: 0 = ok, >0 = argument error (GD_E_*), <0 = -(hipError_t).
 *     Nothing throws; gd_last_error_string() describes the last failure of the calling thread.
 *   - re-entrant per stream.  Process-wide state is limited to: the thread-local error string; the opt-in
 *     gd_set_matrix_split() switch (initialised from the environment, read at launch time; leave it alone and every
 *     call is a pure function of its arguments); tuning knobs read ONCE from the environment at first use
 *     (GD_SPMM_GRID_CAP, GD_ROWS_GEMM_GRID, GD_ROWS_GEMM_QUEUE, GD_ROWS_GEMM_WS, GD_ROWS_GEMM_WS_MIN_ROWS); the lazily resolved RCCL entry points.
 */
#ifndef GNNDELETE_HIP_H
#define GNNDELETE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GD_ABI_VERSION 9   /* 2: round-2 signatures (xcd_bounds, interleaved GAT edge values, tile conv, ...); 3: gd_set_matrix_split;
                              4: gd_spmm_csr_onepass_f32, gd_rows_gemm_wgrad_reduce_f32 (dw = NULL in the wgrad entries),
                                 gd_comm_* / gd_allreduce_f32 / gd_exchange_rows_f32, gd_segment_softmax_f32, gd_rowpair_dot_f32, gd_step_tail_f32;
                              5: gd_del_loss_bwd_wgrad_f32, multi-row items of gd_spmm_csr_onepass_f32;
                              6: gd_rows_gemm_ws_covers (weight-stationary form of the row GEMMs), gd_spmm_csr_rowgroup_f32;
                             7: gd_rgcn_wave_conv_f32 / gd_rgcn_wave_covers;
                             8: gd_build_source_hash (the library carries a stamp of the sources it was built from), gd_typed_wgrad_f32,
                                gd_typed_edge_dot_f32, gd_spmm_csr_onepass_aux_f32, gd_del1_loss_wgrad_f32; gd_rowtarget_mse_f32 accepts dz = NULL;
                                gd_agg_gemm_f32 and gd_spmm_csr_rowgroup_f32 removed (opt-in forms nobody defaulted);
                             9: gd_rows_gemm_accumulate_f32; gd_rows_gemm_select_f32 / gd_rows_gemm_dots_f32 run weight-stationary with a selector
                                AND an index list; the one-launch item kernels abort on a misaligned XCD range table instead of deadlocking */

enum {
  GD_OK = 0,
  GD_E_NULL = 1,      /* required pointer is NULL */
  GD_E_DIM = 2,       /* unsupported / inconsistent dimension */
  GD_E_ALIGN = 3,     /* pointer or leading dimension not 16-byte aligned where required */
  GD_E_WORKSPACE = 4  /* scratch buffer too small */
};

int gd_abi_version(void);
const char* gd_last_error_string(void);
/* sha256 (first 16 hex digits) over the kernel sources (the .hip, .h and .cpp files of csrc) and this header the library was BUILT from,
 * stamped at build time by the Makefile; gnndelete_amd/_lib.py refuses a library whose stamp differs from the sources next
 * to it (a stale .so travelling with newer sources).  Replaces nothing upstream. */
const char* gd_build_source_hash(void);

/* How the dense fp32 products of the row GEMMs are formed (process-wide switch; initial value from the environment
 * variable GD_MATRIX_SPLIT, else GD_MATRIX_SPLIT_DEFAULT):
 *   0  v_mfma_f32_32x32x2_f32 (the fp32 matrix instruction);
 *   6  (opt-in) every fp32 operand is the exact sum of three bf16 pieces; the product is formed from the six largest of
 *      the nine partial products on v_mfma_f32_32x32x16_bf16, each exact in fp32, accumulated in fp32 (the three left out
 *      are <= 2^-26 of the product): same fp32 inputs and outputs, error against an fp64 product not larger than the fp32
 *      instruction's (rows_gemm.hip; tests/test_kernels_gpu.py).  Used where d_in is 64 / 128 and d_out 96 / 128, and for 128 -> 64.
 * Operands are expected finite and below 2^127 (an infinity, or a value that rounds to one in bf16, turns into NaN where
 * the fp32 instruction would propagate the infinity).  The switch is read at launch time: a hipGraph keeps the kernels it was captured with.  gd_gemm_f32 follows it when k is
 * a multiple of 128.  Replaces nothing upstream (torch.mm on fp32 tensors, framework/models/deletion.py:27). */
#define GD_MATRIX_SPLIT_DEFAULT 0
int gd_matrix_split(void);
int gd_set_matrix_split(int n_products);

/* ---------------------------------------------------------------- message passing ----- */

/* COO -> CSR over TARGET rows, sources of a row ascending, multi-edges kept, ties in input order (stable).
 *   src, dst  [n_edges] int64 device arrays: the two rows of the reference's edge_index (flow source -> target)
 *   rowptr    [n_nodes + 1], col [n_edges] int32 out;  order [n_edges] int32 out or NULL: input position of
 *             the edge stored in CSR slot k
 *   status    device int32 out: 0, or 1 if any endpoint was outside [0, n_nodes) (outputs then undefined;
 *             the caller reads it when it next synchronises - no host sync in here)
 *   workspace >= gd_csr_from_coo_workspace(n_nodes, n_edges) bytes, 256-byte aligned
 * Replaces what torch_geometric re-derives from edge_index inside every conv call (framework/models/gcn.py:16-22,
 * gat.py:16-22, gin.py:26-34) and the sort + coalesce of to_undirected in the Df preprocessing
 * (delete_gnn.py:175-182).  Integer only: bit-exact, independent of the order the edges arrive in. */
int64_t gd_csr_from_coo_workspace(int32_t n_nodes, int64_t n_edges);
int gd_csr_from_coo(const int64_t* src, const int64_t* dst, int64_t n_edges, int32_t n_nodes, int32_t* rowptr,
                    int32_t* col, int32_t* order, int32_t* status, void* workspace, int64_t workspace_bytes,
                    void* stream);


/* GCN symmetric normalisation on a CSR that already contains exactly one self loop per node:
 *   val[k] = deg[i]^-1/2 * deg[col[k]]^-1/2,  deg[i] = rowptr[i+1]-rowptr[i]  (k in row i)
 * Replaces torch_geometric gcn_norm inside GCNConv.forward (framework/models/gcn.py:16,19). */
int gd_gcn_norm_f32(const int32_t* rowptr, const int32_t* col, int32_t n_rows, float* val, void* stream);

/* CSR SpMM / gather-scatter-add:
 *   y[i,:] = self_coef * x[i,:] + sum_{k in row i} (val ? val[k] : 1) * x[col[k],:] + (bias ? bias : 0)
 * d must be a multiple of 4, x/y/bias 16-byte aligned, ldx/ldy multiples of 4.
 * Replaces MessagePassing.propagate (gather + scatter_add) of GCNConv / GINConv
 * (framework/models/gcn.py:11-24, gin.py:11-12,26-34); with the transposed CSR it is the
 * backward of the same op. */
int gd_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                    const float* x, int64_t ldx, float* y, int64_t ldy, const float* bias,
                    float self_coef, int32_t n_rows, int32_t d, void* stream);

/* Load-balanced form of gd_spmm_csr_f32 for heavy-tailed graphs.  The CSR rows are pre-cut into
 * work items of at most 64 in-edges: items[4*i..] = {row, start, end, slot}; slot = -1 when the
 * item covers its whole row (y written directly), else the index of a d-float partial in
 * `scratch`.  split[4*i..] = {row, first_slot, n_slots, 0} lists the rows cut into several
 * items; their partials are added in slot order by a second kernel (no atomics: deterministic).
 * nnz = length of col/val (index loads are clamped to it instead of being predicated).
 * x_self (optional, same pitch ldx): the matrix the self_coef term is read from, y[i] += self_coef *
 * x_self[i,:]; NULL = x itself (GIN's (1+eps) x_i).  With another matrix (or other columns of the
 * same buffer) and self_coef = 1 it is SAGEConv's root path, out = mean_j(x_j W_l) + b + x_i W_r.
 * x_rows = an upper bound on the number of rows of x AND of y (every col id and every item row
 * is < x_rows); when both matrices are smaller than 4 GiB the kernel uses 32-bit row offsets
 * (pass 0 if unknown: 64-bit addressing).
 * xcd_bounds (optional, 9 ascending item indices, [0] = 0, [8] = n_items): the item range each of the 8 XCDs
 * sweeps; NULL = equal eighths.  A placement hint only (balances the bytes each XCD's L2 moves), never results.
 * Same arithmetic, same call sites as gd_spmm_csr_f32. */
int gd_spmm_csr_balanced_f32(const int32_t* items, int32_t n_items, const int32_t* split, int32_t n_split,
                             const int32_t* col, const float* val, const float* x, int64_t ldx,
                             float* y, int64_t ldy, const float* bias, float self_coef,
                             const float* x_self, float* scratch, int32_t d, int32_t nnz, int32_t x_rows,
                             const int32_t* xcd_bounds, void* stream);

/* One-launch form of gd_spmm_csr_balanced_f32 (same arithmetic, same call sites): no scratch rows, no fix-up kernel.
 *   items [n_items, 4], n_items a multiple of 4, in four flavours:
 *     {row, start, end, -1}  a row of at most 64 in-edges (one wave);
 *     {row, start, end, -2}  a GROUP member: a row above 64 in-edges is laid out as four consecutive items at a
 *                            4-aligned position, member w = the w-th contiguous share [start, end) of the row's
 *                            in-edges (shares are multiples of 64 edges, trailing shares may be empty); the block's
 *                            four waves take one member each, the four partial rows are added in LDS in member order and
 *                            written once - deterministic, no atomics;
 *     {-1, 0, 0, -1}         padding;
 *     {row, start, end, -(16 + v)}  MULTI-ROW: up to MAXR CONSECUTIVE light rows row .. row + nr - 1 whose in-edges
 *                            [start, end) are adjacent in the CSR and at most 64 together, summed in ONE visit of a wave (the
 *                            sweep is bound by dependent round trips per visit, not by bytes).  v = c1 | c2 << 7 | c3 << 14 |
 *                            (nr - 1) << 21, c_q = the number of edges of rows 0 .. q - 1 of the item; edge j of the item
 *                            belongs to row (j >= c1) + (j >= c2) + (j >= c3) (fields past nr - 1 are ignored).  MAXR follows
 *                            the width the kernel carries accumulators for: 1 (no such items) for d > 64, 2 for 33 .. 64,
 *                            4 for d <= 32; more rows than MAXR in an item is a caller error (the extra rows are not written).
 *                            A row's sum is associated by edge position in the ITEM: equal to fp32 rounding to its one-row form.
 *   xcd_bounds (required, 9 ascending item indices, all multiples of 4, [0] = 0, [8] = n_items): the item range each of
 *   the 8 XCDs sweeps.  Groups may sit anywhere in a range (the planner puts a range's hub rows first, heaviest first).
 *   INVARIANT (the caller's, SplitPlan.onepass keeps it by construction): every limit is a multiple of 4 and every group starts at
 *   a multiple of 4 - the four waves of a block then always hold one aligned quadruple of items, which is what makes the group
 *   path's block barriers uniform.  The table lives in device memory, so the entry cannot check it without a synchronisation;
 *   the KERNEL aborts the launch (trap -> hipErrorLaunchFailure at the next synchronisation) when a limit of its range is not a
 *   multiple of 4 instead of deadlocking.  The same holds for gd_spmm_csr_onepass_aux_f32 and the one-launch form of
 *   gd_gat_edge_grads_balanced_f32.
 * The sum of a hub row is associated differently from the balanced form's (piece partials in slot order), so the two
 * entries agree to fp32 rounding, not bit for bit; each is bit-reproducible run to run. */
int gd_spmm_csr_onepass_f32(const int32_t* items, int32_t n_items, const int32_t* col, const float* val,
                            const float* x, int64_t ldx, float* y, int64_t ldy, const float* bias, float self_coef,
                            const float* x_self, int32_t d, int32_t nnz, int32_t x_rows, const int32_t* xcd_bounds,
                            void* stream);

/* gd_spmm_csr_onepass_f32 with the edge values read THROUGH a permutation from an interleaved (weight, addend) array kept in
 * another edge order (ABI 8): w[k] = aux[2 perm[k]], and aux_sum[row] = the sum of aux[2 perm[k] + 1] over the row's entries
 * (fixed order).  GATConv's backward forms the attention weights and the logit gradients per edge in TARGET-major order and
 * needs them SOURCE-major for the message gradient dh = A_alpha^T dy and for d a_src: this entry is that aggregation, the
 * transposition pass (gd_gat_transpose_edges_f32) disappears into it.  d in {64, 128}; x and y below 4 GiB with row ids /
 * pitches below 2^24; no bias / self term; items / xcd_bounds as for gd_spmm_csr_onepass_f32 (rows per item <= 2).
 * Replaces the index_select / scatter chain of torch_geometric's GATConv backward (framework/models/gat.py:11-24). */
int gd_spmm_csr_onepass_aux_f32(const int32_t* items, int32_t n_items, const int32_t* col, const int32_t* perm, const float* aux,
                                const float* x, int64_t ldx, float* y, int64_t ldy, float* aux_sum, int32_t d, int32_t nnz,
                                int32_t x_rows, const int32_t* xcd_bounds, void* stream);


/* Per-relation mean aggregation of R-GCN ("typed SpMM") over a relation-major CSR:
 *   rowptr[(r*n_rows + i) .. +1] delimit the in-edges of type r into node i;
 *   y[r][i,:] = mean_{k} x[col[k],:]   (0 when the segment is empty),  y is [R, n_rows, ldy].
 * Replaces the relation loop of RGCNConv.forward (framework/models/rgcn.py:17-22,31-33). */
int gd_rgcn_mean_f32(const int32_t* rowptr, const int32_t* col, const float* x, int64_t ldx,
                     float* y, int64_t ldy, int32_t n_rel, int32_t n_rows, int32_t d, void* stream);

/* GAT (heads=1) fused edge-softmax aggregation over a CSR with one self loop per node:
 *   e_k = LeakyReLU(a_src[col[k]] + a_dst[i], slope); alpha = softmax_k(e) (denominator + 1e-16)
 *   y[i,:] = sum_k alpha_k h[col[k],:] + bias;  alpha (nnz floats, optional) is kept for backward.
 * Replaces GATConv.forward's gather/leaky_relu/softmax/scatter chain (framework/models/gat.py:11-24). */
int gd_gat_aggregate_f32(const int32_t* rowptr, const int32_t* col, const float* a_src, const float* a_dst,
                         const float* h, int64_t ldh, float* y, int64_t ldy, const float* bias,
                         float* alpha_out, float slope, int32_t n_rows, int32_t d, void* stream);

/* Backward of gd_gat_aggregate_f32 w.r.t. h (message path), a_src and a_dst, given dy.
 *   rowptr/col/alpha: forward CSR (target-major) and its saved attention;
 *   rowptr_t/col_t/perm_t: the transposed CSR (source-major) and, per transposed entry, the
 *   index of the same edge in the forward CSR.
 * Outputs: dh [n,d] (message path only), da_src[n], da_dst[n]; scratch de[nnz]. */
int gd_gat_aggregate_bwd_f32(const int32_t* rowptr, const int32_t* col, const float* alpha,
                             const int32_t* rowptr_t, const int32_t* col_t, const int32_t* perm_t,
                             const float* a_src, const float* a_dst, const float* h, int64_t ldh,
                             const float* dy, int64_t lddy, float* dh, int64_t lddh,
                             float* da_src, float* da_dst, float* de, float slope,
                             int32_t n_rows, int32_t d, void* stream);

/* Load-balanced forms of the two GAT entries over the work items of gd_spmm_csr_balanced_f32
 * (<= 64 in-edges per item, hub rows spread over many waves).  Forward: every item computes a
 * local softmax, the items of a split row are merged flash-attention style; instead of the
 * per-edge attention it returns the row statistics rowmax[n] / rowsum[n] (alpha_k =
 * exp(e_k - rowmax_i) / (rowsum_i + 1e-16)).  d must be a power of two in [4, 1024];
 * scratch: gd_gat_balanced_scratch(n_slots, d) floats.
 * gd_gat_edge_grads_balanced_f32 is the target-major half of the backward: ade[nnz][2] = per edge
 * (alpha rebuilt, d loss / d score) interleaved - the source-major half gathers both through the edge
 * permutation, one sector per edge this way -, da_dst[n]; t_row[n] and scratch (n_slots floats) are work
 * space.  The source-major half (dh = sum alpha dy, da_src = sum de) is gd_gat_transpose_edges_f32 +
 * gd_spmm_csr_balanced_f32 on the transposed CSR with val = alpha_t. 
 * gd_gat_edge_grads_balanced_f32 with xcd_bounds (ABI 8; NULL = the piece form above): `items` are then the ONE-LAUNCH items of
 * SplitPlan.onepass(d, multirow=1) - slot -1 = a whole row, slot -2 = member of a group (a hub row as four consecutive,
 * 4-aligned items, one share per wave of a block), row -1 = padding - with the nine item-range limits of the eight XCDs; split
 * must be empty: the members sum the row's t in LDS and each finishes the score gradients of its own edges - no second edge
 * pass, no fix-up launches.  (The forward keeps the piece form: its kernel runs at 64 registers / eight waves per SIMD and a
 * member path with several chunks spills there - measured +200 us on the GAT step, NOTES round 5.) */
int64_t gd_gat_balanced_scratch(int32_t n_slots, int32_t d);
int gd_gat_aggregate_balanced_f32(const int32_t* items, int32_t n_items, const int32_t* split, int32_t n_split,
                                  int32_t n_slots, const int32_t* col, const float* a_src, const float* a_dst,
                                  const float* h, int64_t ldh, float* y, int64_t ldy, const float* bias,
                                  float* rowmax, float* rowsum, float* scratch, float slope,
                                  int32_t d, int32_t nnz, int32_t h_rows /* rows of h, 0 = unknown */, void* stream);
int gd_gat_edge_grads_balanced_f32(const int32_t* items, int32_t n_items, const int32_t* split, int32_t n_split,
                                   const int32_t* col, const float* a_src, const float* a_dst,
                                   const float* rowmax, const float* rowsum, const float* h, int64_t ldh,
                                   const float* dy, int64_t lddy, float* ade, float* da_dst,
                                   float* t_row, float* scratch, float slope, int32_t d, int32_t nnz, const int32_t* xcd_bounds,
        void* stream);

/* GAT attention logits: a1[i] = <h[i,:], v1>, a2[i] = <h[i,:], v2> in one pass over h
 * (alpha_src / alpha_dst of GATConv.forward, framework/models/gat.py:11-12). */
int gd_row_dots_f32(const float* h, int64_t ldh, int32_t n, int32_t d, const float* v1, const float* v2,
                    float* a1, float* a2, void* stream);

/* out[i] = sum of x[perm ? perm[k] : k] over k in [rowptr[i], rowptr[i+1]) - deterministic
 * segment sum (the da_src reduction of the GAT backward). */
int gd_segment_sum_f32(const int32_t* rowptr, const int32_t* perm, const float* x, int32_t n, float* out,
                       void* stream);

/* y[i,:] += a[i] * u[:] + b[i] * v[:]  (d % 4 == 0): the gradient that reaches GATConv's linear output
 * through the attention logits alpha_src = <h, att_src>, alpha_dst = <h, att_dst> (gat.py:11-12 /
 * PyG GATConv), both rank-1 terms in one pass. */
int gd_rank1_add2_f32(float* y, int64_t ldy, int32_t n, int32_t d, const float* a, const float* u,
                      const float* b, const float* v, void* stream);

/* GAT backward, edge quantities moved to the transposed (source-major) edge order in one pass:
 * alpha_t[k] = ade[perm[k]][0] (weights of the transposed SpMM that forms dh) and
 * da_src[j] = sum_{k in source row j} ade[perm[k]][1] (gradient of alpha_src); perm = position of
 * transposed edge k in the forward CSR; ade as written by gd_gat_edge_grads_balanced_f32. */
int gd_gat_transpose_edges_f32(const int32_t* rowptr_t, const int32_t* perm, const float* ade, int32_t n,
                               float* alpha_t, float* da_src, void* stream);

/* Fused R-GCN message passing (PyG RGCNConv aggr='mean', framework/models/rgcn.py:16-38) for constant
 * relation weights - no [R, N, d] per-relation aggregate is formed:
 *     y[i,:] += sum over the (i, r) runs of node i:  ( sum_{e in run} w[e] * x[col[e],:] ) @ W_r
 * Node-major typed graph: node_ptr[n+1] indexes the runs of a node, seg_ptr[S+1] the edges of a run,
 * seg_rel[S] its relation; col/w per edge (forward: w = 1/|run|).  weight = [R, n_blocks, ib, ob]
 * (n_blocks = 1: dense [R, d_in, d_out]).  trans != 0 multiplies by W_r^T instead (input gradient on the
 * transposed graph; then d_in is the forward d_out and vice versa).  y must hold the root/bias term (or
 * zeros) on entry; d_in, d_out <= 128.  Replaces the per-relation masked propagate + einsum loop. */
int gd_rgcn_conv_f32(const int32_t* node_ptr, const int32_t* seg_ptr, const int32_t* seg_rel,
                     const int32_t* col, const float* w, const float* x, int64_t ldx, int32_t d_in,
                     const float* weight, int32_t n_blocks, int32_t trans,
                     float* y, int64_t ldy, int32_t d_out, int32_t n_nodes, void* stream);

/* The same conv regrouped by (64-node tile, relation) so that a relation weight is fetched once per tile instead
 * of once per (node, relation) run, with the tile's outputs held in LDS across all relations
 * (csrc/rgcn_tile.hip).  Supported: d_in, d_out in {64, 128}; weight = [R, 4, ib, ob] (the reference's num_blocks = 4,
 * framework/models/rgcn.py:17-22) or dense [R, 1, d_in, d_out]; gd_rgcn_tile_kl() returns the k range per wave (0 =
 * not supported: use gd_rgcn_conv_f32).
 *
 * packed_w: gd_rgcn_pack_weight_f32(weight [R, n_blocks, ib, ob], ...) -> R * (d_out / 16) * (kl / 16) * 256 floats, the weight of
 * this direction in the lane order of the MFMA operand (trans != 0: W_r^T; d_in / d_out are those of the direction).
 *
 * Tile plan of a typed graph (node-major, every (node, relation) run cut into pieces of <= 16 edges; the k-th piece of
 * a run belongs to pass k): steps are the distinct (tile = node / 64, relation, pass) triples in ascending order,
 *     tile_step_ptr[n_tiles + 1]   steps of a tile              step_rel[S]        relation of a step
 *     step_piece_ptr[S + 1]        pieces of a step (at most 32; a crowded (tile, relation, pass) continues in the next step)
 *     piece[P][2]                  {first edge, (node % 64) | length << 8}, at most one piece per node and step
 *     col[E], w[E]                 source node and weight per edge, in (step, node % 64, source) order
 *     tile_order[n_tiles]          launch order of the tiles (most steps first) or NULL
 * Hubs (nodes with so many pieces that their tile would run long after the others): their pieces may be moved to SLICE
 * rows - row ids >= 64 ceil(n_nodes / 64) in extra tiles (n_tiles > ceil(n_nodes / 64)); slice row v writes row
 * v - 64 ceil(n_nodes / 64) of y_ext [(n_tiles - ceil(n_nodes / 64)) * 64, d_out], and y[hub_node[h]] += the rows
 * y_ext[hub_ptr[h] .. hub_ptr[h + 1]) in order afterwards.  n_hubs = 0 and NULLs when the plan has no slices.
 * y must hold the root / bias term (or zeros) on entry, as for gd_rgcn_conv_f32; relations are accumulated in
 * ascending order: bit-reproducible. */
int32_t gd_rgcn_tile_kl(int32_t d_in, int32_t d_out, int32_t n_blocks, int32_t trans);
int gd_rgcn_pack_weight_f32(const float* weight, int32_t n_rel, int32_t n_blocks, int32_t d_in, int32_t d_out,
                            int32_t trans, float* packed, void* stream);
int gd_rgcn_tile_conv_f32(const int32_t* tile_order, const int32_t* tile_step_ptr, const int32_t* step_rel,
                          const int32_t* step_piece_ptr, const int32_t* piece,
                          const int32_t* col, const float* w, int32_t n_tiles, const float* x, int64_t ldx,
                          int32_t d_in, const float* packed_w, int32_t n_blocks, int32_t trans, float* y, int64_t ldy,
                          int32_t d_out, int32_t n_nodes, const int32_t* hub_node, const int32_t* hub_ptr, int32_t n_hubs,
                          float* y_ext, void* stream);

/* The same conv for the reference's FOUR diagonal blocks (weight = [R, 4, ib, ob], framework/models/rgcn.py:17-22; d_in,
 * d_out in {64, 128}; gd_rgcn_wave_covers() = 1), wave-private form (csrc/rgcn_wave.hip): output block t of a node depends on
 * the input features [t d_in / 4, (t + 1) d_in / 4) only, so one WAVE owns (64-node tile, block t) - no block barrier, the
 * outputs of the job in a wave-private LDS accumulator, the next unit's gathers in flight while this one is multiplied.
 * packed_w as for gd_rgcn_tile_conv_f32 (gd_rgcn_pack_weight_f32 of the direction; the transposed direction is this entry
 * on the transposed plan with the transposed pack, d_in / d_out those of the direction).
 *
 * Unit plan of a typed graph (node-major; every (node, relation) run cut into pieces of <= 4 edges; the pieces of one
 * (tile = node / 64, relation), in (node, piece) order, 16 to a unit):
 *     tile_unit_ptr[n_tiles + 1]   units of a tile, relations ascending;  unit_rel[U + 1]   relation of a unit | s << 16, s = the
 *                                  OR of the slot flags of the unit (which steps of the same-node scan it needs)
 *     n_units = U; EVERY UNIT ARRAY HOLDS ONE MORE, EMPTY UNIT AT INDEX U (relation 0, slot words 0, all pairs unused):
 *                                  what the kernel's software pipeline runs on past the end of a tile
 *     unit_row[U + 1][16]          slot word: node % 64 | flags << 8 | last << 12 (0 for an unused slot).  The slots of one
 *                                  node inside a unit are consecutive; flag bit b = the slot 2^b to the left exists and holds
 *                                  the same node; last = no further slot of this node follows in the unit
 *     unit_edges[U + 1][16][4][2]  (source node, weight as float bits) per edge of a slot; unused pairs = (n_nodes, 0.0f)
 *     job_tile[n_tiles]            launch order of the tiles (most units first) or NULL
 * tile must be 64.  y must hold the root / bias term (or zeros) on entry; relations are accumulated in ascending order,
 * the slots of a node by a fixed scan tree: bit-reproducible. 
 * relu_in != 0 (ABI 8): the conv reads relu(x) - RGCN's second layer reads relu(z1) (framework/models/rgcn.py:36-37) - formed
 * where the gathered rows land instead of in a pass of its own over [n_nodes, d_in]. */
int32_t gd_rgcn_wave_covers(int32_t d_in, int32_t d_out, int32_t n_blocks);
int gd_rgcn_wave_conv_f32(const int32_t* job_tile, int32_t n_tiles, int32_t tile, const int32_t* tile_unit_ptr,
                          int32_t n_units, const int32_t* unit_rel, const int32_t* unit_edges, const int32_t* unit_row,
                          const float* x, int64_t ldx, int32_t d_in, const float* packed_w, int32_t n_blocks, float* y,
                          int64_t ldy, int32_t d_out, int32_t n_nodes, int32_t relu_in, void* stream);

/* Random walks for GraphSAINT mini-batches (torch_geometric GraphSAINTRandomWalkSampler / torch_sparse random_walk as
 * used at framework/trainer/gnndelete_nodeemb.py:379-381, :734-736): out[s * n_walks + w] = node of walker w after s
 * steps (s = 0: start[w]); a step draws one out-neighbour of the current node uniformly from the CSR over SOURCE rows
 * (rowptr / col), a node without out-edges keeps the walker.  The draw is a hash of (seed, w, s): reproducible. */
int gd_random_walk(const int32_t* rowptr, const int32_t* col, int32_t n_nodes, const int64_t* start, int32_t n_walks,
                   int32_t walk_length, uint64_t seed, int64_t* out, void* stream);

/* The non-MSE row losses of the reference's loss zoo (framework/trainer/gnndelete_nodeemb.py:18-28), value per row
 * pair and gradient with respect to a, in one pass; rows of a / b optionally gathered through ia / ib (int64, NULL =
 * identity):
 *   kind 0  CosineDistance*:  val[r] = 1 - cos(a_r, b_r)  (each norm clamped at 1e-8 as F.cosine_similarity does)
 *   kind 1  BoundedKLD*:      val[r] = KL(softmax(b_r) || softmax(a_r)),  grad[r,:] = softmax(a_r) - softmax(b_r)
 * The caller reduces val over the rows (mean / sum; 1 - exp(-KL / n) for the bounded KLD) and scales grad. */
int gd_rowpair_loss_f32(int32_t kind, const float* a, int64_t ld_a, const int64_t* ia, const float* b, int64_t ld_b,
                        const int64_t* ib, int32_t n_rows, int32_t d, float* val, float* grad, int64_t ld_g, void* stream);

/* ---------------------------------------------------------------- Del operator --------- */

/* Row-subset GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32):
 *   for s in [0,n_sel):  r = idx ? idx[s] : s
 *       out[r,:] = act( in[r,:] ) @ (trans_w ? W^T : W)       W is [d_in, d_out] ([d_out, d_in] if trans_w)
 *   save_in (optional, compact [n_sel, d_in]) receives the gathered input rows (for the weight
 *   gradient); `in` and `out` may alias (rows are fully read before they are written) when
 *   d_in == d_out.  relu_in applies max(0,.) to the gathered rows first.
 *   bias (optional, [d_out]) is added.
 * With idx = S_Df node list and W = deletion_weight this is DeletionLayer.forward
 * (framework/models/deletion.py:17-29: clone + boolean gather + matmul + index_put); with
 * trans_w it is its input-gradient; with idx = NULL it is a dense Linear.
 * Two kernel forms, same results to fp32 rounding (the k order of the accumulation differs):
 *   - weight-stationary (rows_gemm_ws.hip, v_mfma_f32_16x16x4_f32): one wave per SIMD keeps W in its registers; taken
 *     where gd_rows_gemm_ws_covers() says so - widths in {64, 128}, >= 65,536 rows, no save_in (bias: plain calls only), out not
 *     aliasing in, fp32 products (GD_ROWS_GEMM_WS=0 in the environment turns it off);
 *   - LDS-operand (rows_gemm.hip): everything else, down to a scalar kernel for odd widths. */
int gd_rows_gemm_f32(const float* in, int64_t ld_in, const int32_t* idx, int32_t n_sel,
                     const float* w, int32_t d_in, int32_t d_out, int32_t trans_w,
                     const float* bias, int32_t relu_in,
                     float* out, int64_t ld_out, float* save_in, void* stream);

/* gd_rows_gemm_f32 over a matrix whose rows live in two buffers of the same shape: row r is read from
 * in_alt where sel[r] != 0 and from in otherwise (e.g. z1 = Del-1 output on the S_Df rows, conv1 output
 * elsewhere: deletion.py:17-29 clones the whole matrix to get this; here neither copy is made). */
int gd_rows_gemm_ws_covers(int32_t n_sel, int32_t d_in, int32_t d_out);   /* 1: plain / sign / gate calls of this size run weight-stationary */

int gd_rows_gemm_select_f32(const float* in, const float* in_alt, const uint8_t* sel, int64_t ld_in,
                            const int32_t* idx, int32_t n_sel, const float* w, int32_t d_in, int32_t d_out,
                            int32_t trans_w, const float* bias, int32_t relu_in,
                            float* out, int64_t ld_out, void* stream);

/* (ABI 9) out[r,:] += in[r,:] @ (trans_w ? W^T : W) for r = idx[s] or s, s < n_sel - the weight-stationary form with the
 * accumulators of a 16-row unit started from the rows of `out` (fetched a unit ahead).  GraphSAGE's root term: SAGEConv is
 * W_l mean_j x_j + b_l + W_r x_i (BASELINE config 3's model; the reference has no SAGE) - the aggregation writes the
 * neighbour term + bias, this call adds x W_r^T in place, in a matrix-bound kernel with memory to spare, instead of the
 * aggregation reading a second row stream (134 -> 104 us on the bench graph).  Only where gd_rows_gemm_ws_covers(n_sel, d_in,
 * d_out) holds (widths in {64, 128}, >= 65,536 rows), 16-byte aligned rows, in != out; GD_E_DIM otherwise (no other form exists:
 * callers below the threshold keep the aggregation's self-row operand). */
int gd_rows_gemm_accumulate_f32(const float* in, int64_t ld_in, const int32_t* idx, int32_t n_sel, const float* w, int32_t d_in,
                                int32_t d_out, int32_t trans_w, float* out, int64_t ld_out, void* stream);

/* Dense out = act(in or in_alt) @ W (+ bias) as gd_rows_gemm_select_f32 (sel / in_alt may be NULL), plus the two
 * row dot products o1[r] = <out[r,:], u1>, o2[r] = <out[r,:], u2> from the epilogue: GATConv's lin_src followed by
 * (x_src * att_src).sum(-1), (x_dst * att_dst).sum(-1) (framework/models/gat.py:11-12) without re-reading the rows
 * just written.  MFMA path only (d_in, d_out multiples of 32, d_out <= 128, weight <= 64 KB). */
int gd_rows_gemm_dots_f32(const float* in, const float* in_alt, const uint8_t* sel, int64_t ld_in, const float* w,
                          int32_t d_in, int32_t d_out, int32_t trans_w, const float* bias, int32_t relu_in, float* out,
                          int64_t ld_out, const int32_t* idx /* NULL = rows 0 .. n_rows-1, else the n_rows listed rows:
                          out / o1 / o2 are written at those row ids */, int32_t n_rows, const float* u1, const float* u2,
                          float* o1, float* o2, void* stream);

/* gd_rows_gemm_f32 that also emits the sign pattern of what it wrote, packed one bit per output
 * feature: sign_bits is compact [n_sel, ceil(d_out/32)] words, bit b of word k of entry s is set
 * iff out[idx[s], 32k + b] > 0.  With W = deletion_weight this is the ReLU gate of F.relu(x1)
 * (deletion.py:66-67) recorded while the Del output is still in registers. */
int gd_rows_gemm_signs_f32(const float* in, int64_t ld_in, const int32_t* idx, int32_t n_sel,
                           const float* w, int32_t d_in, int32_t d_out, int32_t trans_w,
                           const float* bias, int32_t relu_in,
                           float* out, int64_t ld_out, float* save_in, uint32_t* sign_bits, void* stream);

/* The product with the ReLU backward folded into its epilogue:
 *       out[r,n] = gate bit (s,n) ? (in[r,:] @ W)[n] : 0,   r = idx[s]
 * gate_bits: what gd_rows_gemm_signs_f32 wrote for the SAME idx list and d_out (input gradient of
 * `relu -> conv2.lin`, deletion.py:67-68: dh = (dt2 @ W2) * [z1 > 0]).  16 B of gate per row
 * instead of a second [n_sel, d_out] fp32 read. */
int gd_rows_gemm_gated_f32(const float* in, int64_t ld_in, const int32_t* idx, int32_t n_sel,
                           const float* w, int32_t d_in, int32_t d_out, int32_t trans_w,
                           const uint32_t* gate_bits, float* out, int64_t ld_out, void* stream);

/* The ReLU backward alone, on a row subset: out[r,:] = src[r,:] where the gate bit (s, n) is set, else 0, r = idx[s]
 * (gate_bits as written by gd_rows_gemm_signs_f32 for the same idx list; src may be out).  Used where the gradient that
 * has to be gated is a sum of several kernels' outputs (R-GCN: root product + typed conv). */
int gd_gate_rows_f32(const float* src, int64_t ld_src, const int32_t* idx, int32_t n_sel, const uint32_t* gate_bits,
                     int32_t d, float* out, int64_t ld_out, void* stream);

/* The same with a rank-2 correction of the product before the gate:
 *       out[r,n] = gate bit (s,n) ? (in[r,:] @ W)[n] + row_a[r] col_p[n] + row_b[r] col_q[n] : 0
 * GAT's input gradient dh2 = A_alpha^T dy + da_src (x) att_src + da_dst (x) att_dst (framework/models/gat.py through
 * PyG's GATConv) feeds this product; with col_p = att_src @ W, col_q = att_dst @ W (constants of a frozen
 * backbone) the rank-1 terms are added here, on the output side, and the separate pass over dh2 disappears.
 * MFMA widths only (d_in, d_out multiples of 32, d_out <= 128). */
int gd_rows_gemm_gated_rank1_f32(const float* in, int64_t ld_in, const int32_t* idx, int32_t n_sel,
                                 const float* w, int32_t d_in, int32_t d_out, int32_t trans_w,
                                 const uint32_t* gate_bits, const float* row_a, const float* col_p,
                                 const float* row_b, const float* col_q, float* out, int64_t ld_out, void* stream);

/* Weight gradient of the row-subset GEMM:  dW[d_a, d_b] (+)= sum_s a[ia(s),:]^T g[ig(s),:]
 *   ia(s) = a_idx ? a_idx[s] : s, likewise g_idx.  Deterministic split-K: `partials` must hold
 *   gd_rows_gemm_wgrad_workspace(n_sel, d_a, d_b) floats.  accumulate != 0 adds into dW.
 *   relu_mask (optional, same indexing as g through g_idx, ld = ld_g): g is multiplied by
 *   (relu_mask > 0) on the fly (backward through F.relu, deletion.py:67).
 *   g_add (optional, same indexing and ld as g): a second upstream gradient added row by row
 *   after the mask, dW = a^T (mask(g) + g_add) - one pass over `a` for a weight that receives
 *   gradient from two losses (both_all / both_layerwise, gnndelete_nodeemb.py:215-262).
 *   dw = NULL (this entry and gd_rows_gemm_wgrad_loss_f32 with param = NULL): only the per-block partial products are
 *   written; gd_rows_gemm_wgrad_reduce_f32 adds them up later - on another stream if the caller orders it after this
 *   one - so that the launch-sized reduction leaves the step's critical path.
 * Replaces autograd's matmul backward for deletion_weight. */
int64_t gd_rows_gemm_wgrad_workspace(int32_t n_sel, int32_t d_a, int32_t d_b);
int gd_rows_gemm_wgrad_f32(const float* a, int64_t ld_a, const int32_t* a_idx,
                           const float* g, int64_t ld_g, const int32_t* g_idx,
                           const float* relu_mask, const float* g_add, int32_t n_sel, int32_t d_a,
                           int32_t d_b, float* dw, int32_t accumulate, float* partials, void* stream);

/* gd_rows_gemm_wgrad_f32 followed by torch.optim.Adam on `param` with the freshly reduced dW,
 * fused into the split-K reduction (t = *iter + 1; `iter` is a shared iteration counter that
 * gd_loss_finalize_f32 advances once per step). */
int gd_rows_gemm_wgrad_adam_f32(const float* a, int64_t ld_a, const int32_t* a_idx,
                                const float* g, int64_t ld_g, const int32_t* g_idx,
                                const float* relu_mask, const float* g_add, int32_t n_sel, int32_t d_a,
                                int32_t d_b, float* dw, int32_t accumulate, float* partials,
                                float* param, float* exp_avg, float* exp_avg_sq, const int32_t* iter,
                                double lr, double beta1, double beta2, double eps, void* stream);

/* The second half of the weight gradient on its own: dW (+)= sum of the gd_rows_gemm_wgrad_blocks(n_sel) partial
 * products a wgrad entry left in `partials` (called with dw = NULL, same n_sel / d_a / d_b), added in block order
 * (deterministic); param != NULL applies torch.optim.Adam with the reduced gradient in the same launch (t = *iter + 1). */
int gd_rows_gemm_wgrad_reduce_f32(const float* partials, int32_t n_sel, int32_t d_a, int32_t d_b, float* dw,
                                  int32_t accumulate, float* param, float* exp_avg, float* exp_avg_sq,
                                  const int32_t* iter, double lr, double beta1, double beta2, double eps, void* stream);

/* The launch-sized tail of a training step in ONE launch: gd_rows_gemm_wgrad_reduce_f32 (with Adam) for BOTH Del weights and
 * gd_loss_finalize_f32 (per-block loss partials of the two layers -> hist[*pos], ring position, iteration counter).  Weight k
 * (k = 1, 2): partials_k as left by a wgrad entry called with dw = NULL for n_sel_k rows of a [d_k, d_k] weight.  Same
 * summation orders, same Adam arithmetic as the separate entries (t = *iter + 1 for both updates).  `arrive`: one device int32,
 * zero before the first call, private to these calls - the blocks that update the weights check in on it after reading *iter,
 * the finalize block advances *iter when all have (no block reads a counter another block of the launch has already written).
 * Replaces, per iteration, optimizer[0].step(), optimizer[1].step() and the three .item() syncs of the logging block
 * (framework/trainer/gnndelete_nodeemb.py:232-262, 304-311). */
int gd_step_tail_f32(const float* partials1, int32_t n_sel1, int32_t d1, int32_t accumulate1, float* dw1, float* param1,
                     float* exp_avg1, float* exp_avg_sq1, const float* partials2, int32_t n_sel2, int32_t d2,
                     int32_t accumulate2, float* dw2, float* param2, float* exp_avg2, float* exp_avg_sq2, double lr,
                     double beta1, double beta2, double eps, const float* loss_partials1, int32_t n1,
                     const float* loss_partials2, int32_t n2, float* hist, int32_t capacity, int32_t* pos, int32_t* iter,
                     int32_t* arrive, void* stream);
/* (ABI 8) The same launch with the NUMBER OF PARTIAL MATRICES of each weight given instead of derived from a row count
 * (gd_del1_loss_wgrad_f32 leaves gd_del1_loss_wgrad_parts(n_sel) of them - one per compute unit - where the weight-gradient
 * entries leave gd_rows_gemm_wgrad_blocks(n_sel)): same summation order over the partials that exist. */
int gd_step_tail_parts_f32(const float* partials1, int32_t n_part1, int32_t d1, int32_t accumulate1, float* dw1, float* param1,
                           float* exp_avg1, float* exp_avg_sq1, const float* partials2, int32_t n_part2, int32_t d2,
                           int32_t accumulate2, float* dw2, float* param2, float* exp_avg2, float* exp_avg_sq2, double lr,
                           double beta1, double beta2, double eps, const float* loss_partials1, int32_t n1,
                           const float* loss_partials2, int32_t n2, float* hist, int32_t capacity, int32_t* pos, int32_t* iter,
                           int32_t* arrive, void* stream);

/* Weight gradient whose upstream gradient is FORMED while it is fetched, from the folded DEC + NI
 * row-target terms of that layer (see gd_rowtarget_mse_f32): for selected row s with loss slot
 * u = loss_slot[s] >= 0,  g_s = coef[u] * (z[z_idx[s],:] - tm[u,:]),  else 0;  then  + g_add.
 *     dW (+)= sum_s a[ia(s),:]^T g_s        loss_partials[2b], [2b+1] = per-block sums of
 *     cnt |z - tm|^2 over the DEC / NI slots (cnt_signed < 0 marks an NI slot),
 * b < gd_rows_gemm_wgrad_blocks(n_sel); reduce them with gd_loss_finalize_f32.  Replaces
 * gd_rowtarget_mse_f32 + gd_rows_gemm_wgrad_adam_f32 for a layer whose loss gradient feeds nothing
 * but its own Del weight (layer 1 with the backbone frozen): the [S, d] gradient is never written.
 * param != NULL applies Adam as gd_rows_gemm_wgrad_adam_f32 does.  Needs d_a, d_b in {32, 64, 128}. */
int32_t gd_rows_gemm_wgrad_blocks(int32_t n_sel);
int gd_rows_gemm_wgrad_loss_f32(const float* a, int64_t ld_a, const int32_t* a_idx,
                                const float* z, int64_t ld_z, const int32_t* z_idx,
                                const int32_t* loss_slot, const float* tm, const float* coef,
                                const float* cnt_signed, const float* g_add,
                                int32_t n_sel, int32_t d_a, int32_t d_b, float* dw, int32_t accumulate,
                                float* partials, float* loss_partials,
                                float* param, float* exp_avg, float* exp_avg_sq, const int32_t* iter,
                                double lr, double beta1, double beta2, double eps, void* stream);

/* Last-layer Del operator + its folded DEC/NI loss + its input gradient in one pass over the S_Df rows
 * (deletion.py:17-29 forward, gnndelete_nodeemb.py:196-210 loss, autograd's matmul backward):
 *     z = p[idx[s],:] @ W_D ;  dz[s,:] = coef[u] (z - tm[u,:]) (u = loss_slot[s] >= 0, else 0) ;
 *     dp[idx[s],:] = dz[s,:] @ W_D^T ;  loss_partials = per-block (DEC, NI) sums of cnt |z - tm|^2
 * for b < gd_del_loss_bwd_blocks(n_sel).  z is not written (no other consumer during training); dz is
 * compact [n_sel, d] for gd_rows_gemm_wgrad_f32.  d in {32, 64}; p, dz, dp must not alias. */
int32_t gd_del_loss_bwd_blocks(int32_t n_sel);
int gd_del_loss_bwd_f32(const float* p, int64_t ld_p, const int32_t* idx, int32_t n_sel, const float* w, int32_t d,
                        const int32_t* loss_slot, const float* tm, const float* coef, const float* cnt_signed,
                        float* dz, int64_t ld_dz, float* dp, int64_t ld_dp, float* loss_partials, void* stream);

/* The same pass + the Del weight's gradient (autograd's  p[idx,:]^T dz  for DeletionLayer.deletion_weight, deletion.py:17-29):
 *     wgrad_partials[b, :, :] (b < gd_rows_gemm_wgrad_blocks(n_sel), each [d, d] row-major) = per-block sums of p[idx[s],:]^T dz[s,:],
 * laid out and counted like the partials of gd_rows_gemm_wgrad_f32 (dw = NULL), so gd_rows_gemm_wgrad_reduce_f32 / gd_step_tail_f32
 * finish it (fixed order: bit-reproducible).  The launch has that many blocks, so loss_partials holds
 * 2 * gd_rows_gemm_wgrad_blocks(n_sel) floats here.  dz may be NULL (nothing else reads it once the gradient is formed here). */
int gd_del_loss_bwd_wgrad_f32(const float* p, int64_t ld_p, const int32_t* idx, int32_t n_sel, const float* w, int32_t d,
                              const int32_t* loss_slot, const float* tm, const float* coef, const float* cnt_signed,
                              float* dz, int64_t ld_dz, float* dp, int64_t ld_dp, float* loss_partials,
                              float* wgrad_partials, void* stream);
/* (ABI 8) The same with the number of partial slots to fill given: gd_del_loss_bwd_wgrad_parts(n_sel, d) (what the launch has
 * blocks for - one per compute unit in the weight-stationary form of d = 64 at >= 65,536 rows; reduce with
 * gd_step_tail_parts_f32) ... gd_rows_gemm_wgrad_blocks(n_sel); loss_partials holds 2 * n_part floats. */
int32_t gd_del_loss_bwd_wgrad_parts(int32_t n_sel, int32_t d);
int gd_del_loss_bwd_wgrad_parts_f32(const float* p, int64_t ld_p, const int32_t* idx, int32_t n_sel, const float* w, int32_t d,
                                    const int32_t* loss_slot, const float* tm, const float* coef, const float* cnt_signed,
                                    float* dz, int64_t ld_dz, float* dp, int64_t ld_dp, float* loss_partials,
                                    float* wgrad_partials, int32_t n_part, void* stream);

/* (ABI 8) FIRST-layer Del operator at d = 128, its folded loss and its weight gradient in one pass over the S_Df rows - what
 * gd_rows_gemm_signs_f32 followed by gd_rows_gemm_wgrad_loss_f32 (dw = NULL) compute in two (p read once, z only written):
 *     z[idx[s],:] = p[idx[s],:] @ W_D;   sign_out[s, 4] = packed [z > 0]  (the layout gd_rows_gemm_gated_f32 reads);
 *     g_s = coef[u] (z - tm[u,:]) for u = loss_slot[s] >= 0, else 0;   + g_add[idx[s],:] when g_add != NULL
 *     wgrad_partials[b] = block b's part of p[idx,:]^T g and loss_partials[2b], [2b+1] = its cnt |z - tm|^2 over the DEC / NI slots,
 *     b < n_part: the caller says how many partial slots it wants filled - gd_del1_loss_wgrad_parts(n_sel) (what the launch has
 *     blocks for: reduce with gd_step_tail_parts_f32) up to gd_rows_gemm_wgrad_blocks(n_sel) (slots beyond the launch's blocks are
 *     written as zeros, so that gd_rows_gemm_wgrad_reduce_f32 / gd_step_tail_f32, which count that many, finish it).
 * z must not alias p or g_add.  gd_del1_loss_wgrad_covers(n_sel, d): 1 where callers should prefer it (d = 128, launches of
 * >= 65,536 rows; GD_DEL1_FUSED=0 turns it off).  Replaces DeletionLayer.forward of deletion1 + the layer-1 terms of the loss
 * + autograd's deletion_weight gradient (framework/models/deletion.py:17-29, trainer/gnndelete_nodeemb.py:188-242). */
int32_t gd_del1_loss_wgrad_covers(int32_t n_sel, int32_t d);
int32_t gd_del1_loss_wgrad_parts(int32_t n_sel);
int gd_del1_loss_wgrad_f32(const float* p, int64_t ld_p, const int32_t* idx, int32_t n_sel, const float* w, int32_t d, float* z,
                           int64_t ld_z, uint32_t* sign_out, const int32_t* loss_slot, const float* tm, const float* coef,
                           const float* cnt_signed, const float* g_add, int64_t ld_gadd, float* loss_partials,
                           float* wgrad_partials, int32_t n_part, void* stream);
/* (ABI 8) The same pass with the second gradient stream FORMED in the kernel instead of read (the GCN / GIN layer-wise step, where it
 * is conv2's input gradient of the previous iteration):  g_add[idx[s],:] = (dt[idx[s], 0:64] @ w_next[64, 128]) (.) the sign
 * pattern sign_io[s, 4] holds when the call starts - i.e. the one the PREVIOUS call stored; the call then overwrites it with this
 * z's.  row_a / col_a / row_b / col_b (all four or none): row_a[idx[s]] col_a[:] + row_b[idx[s]] col_b[:] is added to the product
 * before the gate (GATConv's two rank-1 input-gradient terms).  Replaces gd_rows_gemm_gated_f32 / gd_rows_gemm_gated_rank1_f32 (+ its
 * [S, 128] write and read-back) in front of gd_del1_loss_wgrad_f32; d = 128, d_next = 64 only; sign_io 8-byte aligned. */
int gd_del1_chain_loss_wgrad_f32(const float* p, int64_t ld_p, const int32_t* idx, int32_t n_sel, const float* w, int32_t d, float* z,
                                 int64_t ld_z, uint32_t* sign_io, const int32_t* loss_slot, const float* tm, const float* coef,
                                 const float* cnt_signed, const float* dt, int64_t ld_dt, int32_t d_next, const float* w_next,
                                 const float* row_a, const float* col_a, const float* row_b, const float* col_b,
                                 float* loss_partials, float* wgrad_partials, int32_t n_part, void* stream);

/* ---------------------------------------------------------------- losses --------------- */

/* Fused Deleted-Edge-Consistency + Neighborhood-Influence MSE terms of one layer, value and
 * gradient in one pass (framework/trainer/gnndelete_nodeemb.py:196-210 with nn.MSELoss).
 * Terms are grouped by the row of z they touch:
 *   for u in [0,n_seg): row = seg_row[u]; for t in [seg_ptr[u], seg_ptr[u+1]):
 *       diff = z[row,:] - o[term_o[t],:]
 *       sums[term_kind[t]] += |diff|^2                      (kind 0 = DEC, 1 = NI)
 *       dz[u or row,:] += 2 * term_w[t] * diff
 *   dz_compact != 0 -> dz is [n_seg, ld_dz] indexed by u, else [N, ld_dz] indexed by row
 *   (only the touched rows are written).  sums[2] must be zeroed by the caller; partial sums are
 *   reduced deterministically through `partials` (gd_rowpair_mse_workspace(n_seg) floats). */
int64_t gd_rowpair_mse_workspace(int32_t n_seg);
int gd_rowpair_mse_f32(const float* z, int64_t ld_z, const float* o, int64_t ld_o, int32_t d,
                       const int32_t* seg_ptr, const int32_t* seg_row, int32_t n_seg,
                       const int32_t* term_o, const float* term_w, const int32_t* term_kind,
                       float* dz, int64_t ld_dz, int32_t dz_compact,
                       float* sums, float* partials, void* stream);

/* Pre-folded form of the same losses for runs whose targets never change (full-batch training:
 * z_ori and the negatives are fixed outside the loop, gnndelete_nodeemb.py:177-185).  For every
 * touched row u (row_idx[u], ascending) the caller folds its terms once into the mean target
 * tm[u,:] (compact [n_rows, d]), cnt[u] = number of terms, coef[u] = 2 * w * cnt and a constant
 * K (sum_t |o_t|^2 - cnt |tbar|^2, added on the host):
 *       diff = z[row,:] - tm[u,:];  dz[row,:] = coef[u] * diff;  sums[kind[u]] += cnt[u] * |diff|^2
 * Pure streaming, deterministic; `partials` holds gd_rowtarget_mse_workspace(n_rows) floats;
 * sums[2] is accumulated into (zero it first).  row_idx need not be ascending.  dz = NULL: the loss sums only (rows whose
 * gradient feeds no trainable weight - the layer-1 DEC rows outside the Del-1 mask of a knowledge-graph request). */
int64_t gd_rowtarget_mse_workspace(int32_t n_rows);
int gd_rowtarget_mse_f32(const float* z, int64_t ld_z, const float* tm, int32_t d,
                         const int32_t* row_idx, const float* coef, const float* cnt, const int32_t* kind,
                         int32_t n_rows, float* dz, int64_t ld_dz, float* sums, float* partials, void* stream);
/* (ABI 8) Two such jobs of widths in {64, 128} in ONE launch (sums = NULL form: the per-block partials of each job go to its own
 * array, gd_rowtarget_mse_blocks(n_rows) pairs each): the two stand-alone loss launches a knowledge-graph step needs for its DEC
 * rows (layer 1: loss sums only, dz_a = NULL; layer 2: gradient rows written) are launch-sized on their own. */
int32_t gd_rowtarget_mse_pair_covers(int32_t d_a, int32_t d_b);
int gd_rowtarget_mse_pair_f32(const float* z_a, int64_t ld_z_a, const float* tm_a, int32_t d_a, const int32_t* row_idx_a,
                              const float* coef_a, const float* cnt_a, const int32_t* kind_a, int32_t n_rows_a, float* dz_a,
                              int64_t ld_dz_a, float* partials_a, const float* z_b, int64_t ld_z_b, const float* tm_b, int32_t d_b,
                              const int32_t* row_idx_b, const float* coef_b, const float* cnt_b, const int32_t* kind_b,
                              int32_t n_rows_b, float* dz_b, int64_t ld_dz_b, float* partials_b, void* stream);

/* End-of-step bookkeeping in one launch: reduce the per-block partials of the two layers' loss
 * kernels (called with sums = NULL; n1/n2 = gd_rowtarget_mse_blocks(n_rows)) in a fixed order,
 * add extra_sums[4] (optional), write (r1, l1, r2, l2) to hist[*pos], advance the ring position
 * (mod capacity) and the iteration counter *iter (the Adam step number source). */
int32_t gd_rowtarget_mse_blocks(int32_t n_rows);
int gd_loss_finalize_f32(const float* partials1, int32_t n1, const float* partials2, int32_t n2,
                         const float* extra_sums, float* hist, int32_t capacity, int32_t* pos, int32_t* iter,
                         void* stream);

/* Dense transform with a reduction dimension of any width: out[row(s), :] = in[row(s), :] @ W (+ bias), s < n_rows,
 * row(s) = idx[s] or s; in [*, k] with k % 32 == 0 (zero-pad the columns of `in` and the rows of W), W [k, n] row-major
 * (= the TRANSPOSE of a torch Linear weight), n in {32, 64, 96, 128}.  The layer-1 transform x W1^T of the wide
 * bag-of-words inputs (framework/models/gcn.py:11-12, gat.py / gin.py alike; F = 1,639 / 8,710) - W is streamed
 * through LDS in 32-row chunks and the k range is split over several blocks per row group; `workspace` holds
 * gd_gemm_f32_workspace(n_rows, k, n) floats for their partial products (may be NULL when that is 0), which are
 * added in split order: deterministic.  Replaces torch's matmul -> rocBLAS on the path. */
int64_t gd_gemm_f32_workspace(int32_t n_rows, int32_t k, int32_t n);
int gd_gemm_f32(const float* in, int64_t ld_in, const int32_t* idx, int32_t n_rows, const float* w, int32_t k,
                int32_t n, const float* bias, float* out, int64_t ld_out, float* workspace, void* stream);

/* Link decoders.  dot: out[m] = <z[e0[m]], z[e1[m]]>  (framework/models/gcn.py:26-36);
 * distmult: out[m] = sum_c z[e0[m],c] * rel[etype[m],c] * z[e1[m],c]  (rgcn.py:40-47). */
int gd_edge_dot_f32(const float* z, int64_t ld_z, int32_t d, const int64_t* e0, const int64_t* e1,
                    const float* rel, int64_t ld_rel, const int64_t* etype,
                    int64_t n_edges, float* out, void* stream);

/* Input gradient of the decoders (autograd of gcn.py:26-36 / rgcn.py:40-47: two scatter-adds with atomics):
 *     dz[v,:] = sum_{k in [inc_ptr[v], inc_ptr[v+1])} w[k] * z[other[k],:]  (* rel[etype[k],:] for DistMult)
 * over a node-major incidence list of the decoded edges: for every edge m = (a, b) with upstream gradient dout[m] the
 * list of a holds (other = b, w = dout[m], etype = type of m) and the list of b holds (other = a, ...).  Every row of
 * dz is written (zeros where nothing is incident); incidences are added in list order, hubs (> 64 incidences) by the
 * whole wave in a fixed pattern: deterministic, no atomics.  d % 4 == 0, 16-byte aligned rows; dz must not alias z. */
int gd_edge_dot_bwd_f32(const float* z, int64_t ld_z, int32_t d, const int32_t* other, const float* w,
                        const float* rel, int64_t ld_rel, const int32_t* etype, const int64_t* inc_ptr,
                        int64_t n_nodes, float* dz, int64_t ld_dz, void* stream);

/* Edge-probability Neighborhood-Influence term of GNNDeleteTrainer
 * (framework/trainer/gnndelete.py:174-193 pair mask, :239-241 loss) over the n_s x n_s block of
 * 2-hop S_Df nodes, value and gradient in one pass, without forming z z^T:
 *     loss = inv_count * sum_{i>j, T[i][j] >= 0} (sigmoid(z[nodes[i]] . z[nodes[j]]) - T[i][j])^2
 *     dz[i,:] = d loss / d z[nodes[i],:]                      (compact [n_s, d])
 * target: [n_s, n_s] row-major (ld_t), only the strictly lower triangle is read; a negative entry
 * excludes the pair (the Df pairs, :186-190).  inv_count = 1 / number of included pairs.
 * workspace: gd_pairs_sigmoid_mse_workspace(n_s, d) floats.  Replaces the N x N `z @ z.t()`,
 * the N x N boolean mask indexing and their autograd backward. */
int64_t gd_pairs_sigmoid_mse_workspace(int32_t n_s, int32_t d);
int gd_pairs_sigmoid_mse_f32(const float* z, int64_t ld_z, const int32_t* nodes, int32_t n_s, int32_t d,
                             const float* target, int64_t ld_t, float inv_count,
                             float* loss, float* dz, float* workspace, void* stream);

/* ---------------------------------------------------------------- optimizer ------------ */

/* torch.optim.Adam (no amsgrad, weight_decay 0) on one tensor (delete_gnn.py:221-226).
 * `step` is a device int32 counter incremented by the kernel (graph-capturable). */
int gd_adam_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int32_t* step,
                int64_t n, double lr, double beta1, double beta2, double eps, void* stream);

/* gd_adam_f32 with the step number read from a shared iteration counter (t = *iter + 1). */
int gd_adam_at_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const int32_t* iter,
                   int64_t n, double lr, double beta1, double beta2, double eps, void* stream);

/* ---------------------------------------------------------------- relational attention -- */

/* Softmax over the entries of every CSR row (torch_geometric.utils.softmax as the reference's RGATConv calls it across
 * the relations of a target node, framework/models/rgat.py:322-337):
 *     alpha[k] = exp(e[k] - max_row) / (sum_row exp(e - max_row) + 1e-16),   k in [rowptr[i], rowptr[i+1])
 * and its backward  de[k] = alpha[k] (dalpha[k] - sum_row alpha dalpha).  Empty rows are skipped. */
int gd_segment_softmax_f32(const int32_t* rowptr, const float* e, int32_t n_rows, float* alpha, void* stream);
int gd_segment_softmax_bwd_f32(const int32_t* rowptr, const float* alpha, const float* dalpha, int32_t n_rows,
                               float* de, void* stream);

/* out[k] = < a[ia[k], :], b[ib[k], :] > for k < n: the gradient of an edge weight of a weighted typed aggregation,
 * d alpha_e = < dm[(relation, target) row of e, :], x[source of e, :] > (autograd of the alpha_e x_j messages,
 * rgat.py:322-337).  float4 path for d % 4 == 0 with 16-byte aligned rows, any d otherwise. */
int gd_rowpair_dot_f32(const float* a, int64_t ld_a, const int32_t* ia, const float* b, int64_t ld_b, const int32_t* ib,
                       int64_t n, int32_t d, float* out, void* stream);

/* Gradients of TRAINABLE relation weights and of per-edge coefficients, straight from relation-major edge lists (no [R, N, d]
 * tensor of per-relation aggregates; ABI 8):
 *   gd_typed_wgrad_f32     dW[r, b] = sum over the edges e of relation r (rel_ptr[r] <= e < rel_ptr[r + 1] of src / dst / w) of
 *                          w[e] * x[src[e], block b]^T dy[dst[e], block b];  dW is [n_rel, n_blocks, d_in / n_blocks, d_out / n_blocks]
 *                          (n_blocks = 1: dense [d_in, d_out] per relation); w = NULL: weight 1.  Edges of a relation are added
 *                          in array order by ONE workgroup per 16 x 16 output tile: bit-reproducible, no atomics.
 *   gd_typed_edge_dot_f32  out[e] = < x[src[e], :], dy[dst[e], :] W_(rel[e])^T > (the gradient of an edge's coefficient when
 *                          the message is coefficient * x_src W_rel), weight as for gd_typed_wgrad_f32's dW.
 * Replace torch.einsum('rnbi,rbio->nbo') over the [R, N, in] mean aggregates and its autograd in the reference's trainable
 * RGCNConv (framework/models/rgcn.py:17-38 under trainer/base.py:394-493, retrain.py:235-339) and the per-edge
 * transforms of RGATConv (framework/models/rgat.py:188-206, :322-337). */
int gd_typed_wgrad_f32(const int32_t* rel_ptr, int32_t n_rel, const int32_t* src, const int32_t* dst, const float* w,
                       const float* x, int64_t ldx, const float* dy, int64_t ldy, int32_t n_blocks, int32_t d_in, int32_t d_out,
                       float* dw, void* stream);
int gd_typed_edge_dot_f32(const int32_t* src, const int32_t* dst, const int32_t* rel, int64_t n_edges, const float* x, int64_t ldx,
                          const float* dy, int64_t ldy, const float* weight, int32_t n_blocks, int32_t d_in, int32_t d_out,
                          float* out, void* stream);

/* ---------------------------------------------------------------- collectives (RCCL) ---- */

/* The two exchanges of the row-partitioned multi-GPU step (the reference has no distributed code at all: SURVEY
 * section 5; north_star adds "RCCL all-reduce of partial aggregates and Del-operator gradients over xGMI").  One process
 * per GPU; a communicator is an RCCL ncclComm_t behind a void*.  RCCL is looked up at first use (dlsym among the loaded
 * libraries - a PyTorch-ROCm process carries its own librccl - then librccl.so by name): without it these entries return
 * GD_E_NULL and everything else in the library works.  Failures of RCCL itself come back as -(1000 + ncclResult_t).
 *   gd_comm_unique_id   rank 0 fills 128 bytes (ncclUniqueId, HOST memory) and ships them to every rank out of band
 *   gd_comm_init        collective over the n_ranks processes: *comm <- the communicator of `rank` on the CURRENT device
 *   gd_comm_destroy     NULL is accepted
 *   gd_allreduce_f32    buf[0..n) <- sum over ranks, in place, enqueued on `stream`: the packed [dW_D1 | dW_D2 | loss sums]
 *                       buffer of a step (80 KiB, latency-bound) - replaces nothing upstream
 *   gd_exchange_rows_f32  sparse all-to-all of feature rows: send_rows[p] / recv_rows[p] (HOST arrays of n_ranks int64 row
 *                       counts) rows of row_elems floats go to / come from peer p, packed back to back in peer order in
 *                       `send` / `recv` (device); one grouped ncclSend / ncclRecv per non-empty pair on `stream` - xGMI is
 *                       point-to-point, so every pair moves over its own link concurrently.  The halo rows of a layer's
 *                       aggregation (forward) and of its transpose (backward). */
int gd_comm_unique_id(void* id128);
int gd_comm_init(const void* id128, int32_t n_ranks, int32_t rank, void** comm);
int gd_comm_destroy(void* comm);
int gd_allreduce_f32(float* buf, int64_t n, void* comm, void* stream);
int gd_exchange_rows_f32(const float* send, const int64_t* send_rows, float* recv, const int64_t* recv_rows,
                         int32_t row_elems, int32_t n_ranks, void* comm, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GNNDELETE_HIP_H */
