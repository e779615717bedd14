/* fsw_hip.h -- C ABI of libfsw_hip.so: the MI355X (gfx950) native hot path of FSW_conv / FSW_embedding.
 *
 * Drop-in boundary.  The reference's only native code is libfsw_embedding.so, loaded with ctypes
 * (reference fsw_embedding.py:94-99, 195-206) and called with raw tensor.data_ptr() device pointers
 * from segcumsum_cuda (fsw_embedding.py:2878-3012).  This library is loaded the same way and keeps the
 * same conventions -- plain pointers and sizes, caller-owned pre-allocated buffers, no torch types --
 * with three deliberate changes:
 *   - every entry point is stream-ordered (takes a hipStream_t) and never device-synchronises
 *     (the reference wrappers cudaDeviceSynchronize() before and after each launch,
 *      fsw_embedding.cu:197, 208, 215, 227, and use the default stream);
 *   - every entry point returns an int status (0 = ok); nothing prints-and-exit(1)s
 *     (fsw_embedding.cu:20-27).  fsw_last_error() returns the message of the last failure;
 *   - besides the segmented cumsum, the whole chain the reference builds out of ~15 E*S-sized COO
 *     tensors (fsw_embedding.py:894-1112: projection, per-slice sort, weight permutation, segmented
 *     cumsum, sinc/cos readout, reduction) is exported as fused kernels on a CSR adjacency.
 *
 * The three legacy symbols at the end keep the reference's exact signatures, so the reference's own
 * fsw_embedding.py can dlopen this library in place of libfsw_embedding.so unchanged.
 *
 * All pointers are DEVICE pointers unless stated otherwise.  Index type is int32: num_rows, num_cols
 * and num_edges must each be < 2^31.
 */
#ifndef FSW_HIP_H
#define FSW_HIP_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* fsw_stream_t; /* a hipStream_t (torch.cuda.current_stream().cuda_stream) */

#define FSW_ABI_VERSION 5

/* Degree classes of the fused neighbourhood kernels.  Rows are binned by in-degree:
 *   bin b, 0 <= b <= FSW_REG_MAX_DEG : rows of degree exactly b (register path, one wave per row and
 *                                      64-slice chunk, exact-size sorting network)
 *   bin FSW_BIN_MID0 + i             : fsw_mid_size(i-1) < degree <= fsw_mid_size(i), i < FSW_NUM_MID_BINS
 *                                      (register path with the network of size fsw_mid_size(i), +inf padding)
 *   bin FSW_BIN_LDS0 + i             : 256 << i < degree <= 512 << i, i < FSW_NUM_LDS_BINS (wave-sort path: the
 *                                      neighbourhood is transposed through LDS, every wavefront sorts one slice's
 *                                      line held across its lanes' registers)
 *   bin FSW_BIN_HUB0 + i             : 2048 << i < degree <= 4096 << i, i < FSW_NUM_HUB_BINS (hub path: a workgroup of 2 << i
 *                                      wavefronts holds ONE slice's line in its registers, 2048 keys per wavefront; the merge
 *                                      levels above one wavefront exchange registers through LDS)
 *   bin FSW_BIN_GLOBAL               : degree > FSW_HUB_MAX_DEG (global-scratch bitonic path, any degree)
 * General (non-unit) weights carry a weight next to every key and a line holds one more element (the reference's pad element), so
 * their per-lane register path ends at FSW_MID_MAX_DEG_WEIGHTED; above it (key, weight) lines of 64 x {3 .. 32} keys per lane on
 * one, two or four wavefronts take rows of up to 8191 neighbours, sorted blocks of 8192 + merge-path levels the rest
 * (csrc/embed_hub.hip, csrc/merge_path.h); with edge features the LDS-staged / scratch-line kernels of csrc/embed_wsort.hip.      */
#define FSW_REG_MAX_DEG 32
#define FSW_NUM_MID_BINS 9
#define FSW_MID_SIZES {40, 48, 64, 80, 96, 128, 160, 192, 256}
#define FSW_MID_MAX_DEG 256
#define FSW_MID_MAX_DEG_WEIGHTED 128
#define FSW_LDS_MAX_DEG 2048
#define FSW_HUB_MAX_DEG 32768
#define FSW_BIN_MID0 (FSW_REG_MAX_DEG + 1)
#define FSW_NUM_LDS_BINS 3
#define FSW_BIN_LDS0 (FSW_BIN_MID0 + FSW_NUM_MID_BINS)
#define FSW_NUM_HUB_BINS 4
#define FSW_BIN_HUB0 (FSW_BIN_LDS0 + FSW_NUM_LDS_BINS)
#define FSW_BIN_GLOBAL (FSW_BIN_HUB0 + FSW_NUM_HUB_BINS)
#define FSW_NUM_BINS (FSW_BIN_GLOBAL + 1)

/* stats[] words written by fsw_graph_build / fsw_project_f32 (device int32[FSW_NUM_STATS]) */
#define FSW_STAT_FLAGS 0        /* OR of FSW_FLAG_* */
#define FSW_STAT_MAX_DEGREE 1
#define FSW_STAT_NUM_ZERO_DEG 2
#define FSW_STAT_NUM_REG 3      /* rows with 1 <= degree <= FSW_REG_MAX_DEG */
#define FSW_STAT_NUM_LDS 4      /* rows with FSW_REG_MAX_DEG < degree <= FSW_LDS_MAX_DEG (mid bins + LDS bin) */
#define FSW_STAT_NUM_GLOBAL 5   /* rows with degree > FSW_LDS_MAX_DEG (hub bins + global bin) */
#define FSW_STAT_NNZ 6          /* number of CSR entries in use = rowptr[num_rows] (after coalescing, without invalid edges) */
#define FSW_STAT_USER 7         /* never written by the library after the build zeroes it: the Python side parks the bits of
                                   the total-mass scale here so that ONE device->host copy per forward fetches everything */
#define FSW_NUM_STATS 8

#define FSW_FLAG_INDEX_RANGE 1   /* an edge endpoint outside [0, num_rows) x [0, num_cols) */
#define FSW_FLAG_W_NONFINITE 2   /* reference assert fsw_embedding.py:678-679 */
#define FSW_FLAG_W_NEGATIVE 4    /* reference assert fsw_embedding.py:680 */
#define FSW_FLAG_X_NONFINITE 8   /* reference assert fsw_embedding.py:652-653 */

int fsw_abi_version(void);
const char* fsw_arch(void);       /* "gfx950" */
const char* fsw_last_error(void); /* host string, valid until the next failing call on this thread */

/* ---- adjacency: edge list -> CSR by recipient, plus degree bins ---------------------------------
 * Replaces FSW_conv.edge_index_to_adj (reference fsw_conv.py:384-447: torch.sparse_coo_tensor().coalesce()
 * + sp.get_slice_info) and the sp.get_slice_info(W, -1) / concat_sparse sorts of FSW_embedding.forward
 * (fsw_embedding.py:778-821).  Parallel edges are kept as separate elements: k parallel unit edges
 * j->i contribute exactly what the reference's single coalesced entry of weight k contributes, because
 * the Fourier readout telescopes over equal projections (DESIGN.md "duplicates").
 *   recipients/senders : int64[num_edges]  (edge_index row 1 / row 0; COO indices row 0 / row 1)
 *   edge_w             : float[num_edges] or NULL for unit weights
 *   rowptr int32[num_rows+1], col int32[num_edges], w float[num_edges] (ignored if edge_w NULL)
 *   perm int32[num_rows]  rows ordered by degree bin; bin_start int32[FSW_NUM_BINS+1] offsets into perm
 *   invperm int32[num_rows] or NULL: position of every row in perm (perm[invperm[r]] == r)
 *   stats int32[FSW_NUM_STATS] (zeroed by this call)
 *   chunk_rows : 0, or a positive multiple of FSW_BIN_BLOCK_ROWS.  With chunk_rows > 0 the rows are binned separately
 *                inside every chunk of chunk_rows consecutive rows: perm lists chunk 0's rows by degree bin, then chunk
 *                1's, ...; bin_start then holds ceil(num_rows / chunk_rows) (<= FSW_MAX_ROW_CHUNKS) rows of
 *                FSW_NUM_BINS + 1 offsets.  Passing row c of bin_start (and num_rows = chunk_rows) to fsw_embed_f32 /
 *                fsw_conv_fused_f32 restricts that call to the recipients of chunk c -- the multi-GPU path overlaps
 *                the collective of one node range with the kernels of the next this way.                          */
#define FSW_BIN_BLOCK_ROWS 2048
#define FSW_MAX_ROW_CHUNKS 256
size_t fsw_graph_workspace_bytes(int64_t num_rows, int64_t num_edges);
int fsw_graph_build(const int64_t* recipients, const int64_t* senders, const float* edge_w,
                    int64_t num_edges, int64_t num_rows, int64_t num_cols, int64_t chunk_rows,
                    int32_t* rowptr, int32_t* col, float* w, int32_t* perm, int32_t* invperm, int32_t* bin_start,
                    int32_t* stats, void* workspace, size_t workspace_bytes, fsw_stream_t stream);
/* The same build with the same arguments and the same result, entry for entry; one partition pass over the high row bits +
 * one workgroup per bucket of 2048 rows instead of three LSD passes when the shape suits it (>= 32768 rows, <= 65536 edges
 * per bucket on average; the LSD passes otherwise).  Faster on graphs without hub rows (0.38 -> ~0.2 ms at 1M rows / 10M
 * edges); a bucket that holds hub rows serialises on its workgroup, so callers keep fsw_graph_build for skewed graphs.   */
int fsw_graph_build_two_level(const int64_t* recipients, const int64_t* senders, const float* edge_w,
                    int64_t num_edges, int64_t num_rows, int64_t num_cols, int64_t chunk_rows,
                    int32_t* rowptr, int32_t* col, float* w, int32_t* perm, int32_t* invperm, int32_t* bin_start,
                    int32_t* stats, void* workspace, size_t workspace_bytes, fsw_stream_t stream);

/* Coalescing variant: entries sorted by (recipient, sender), parallel edges merged into ONE entry whose
 * weight (edge_w or 1 per edge) and edge-feature vector are the sums over the duplicates -- exactly what
 * torch.sparse_coo_tensor(...).coalesce() does to adj and X_edge in the reference (fsw_conv.py:397-398,
 * 436-437).  Required with edge features (the key of an element then depends on its feature vector, so
 * parallel edges are no longer equivalent to separate elements).  col/w/ef are sized for num_edges
 * entries, the first stats[FSW_STAT_NNZ] are used.  slot_of_edge int32[num_edges] (nullable): CSR
 * position of every input edge (-1 for rejected edges) -- the backward routes feature gradients with it. */
int fsw_graph_build_coalesced(const int64_t* recipients, const int64_t* senders, const float* edge_w,
                              const float* edge_feat, int d_edge, int64_t num_edges, int64_t num_rows, int64_t num_cols,
                              int32_t* rowptr, int32_t* col, float* w, float* ef, int32_t* slot_of_edge, int32_t* perm,
                              int32_t* invperm, int32_t* bin_start, int32_t* stats, void* workspace, size_t workspace_bytes,
                              fsw_stream_t stream);

/* ---- projection: Xp[n, ldp] = X[n, ldx] . V[S, ldv]^T on the matrix cores -------------------------
 * Default: every operand split into three bf16 planes (x = x1 + x2 + x3 carries fp32's 24 significand bits), six
 * v_mfma_f32_32x32x16_bf16 per k-step with fp32 accumulation: < 5e-7 against float64, 2.7x fewer matrix cycles than the exact
 * fp32 MFMA (v_mfma_f32_32x32x2_f32), which FSW_PROJECT_EXACT_FP32=1 in the environment selects instead (d <= 128).
 * Replaces torch.tensordot(X, projVecs) (reference fsw_embedding.py:909-913).  Sets
 * FSW_FLAG_X_NONFINITE in stats[FSW_STAT_FLAGS] if X holds a NaN/Inf (stats may be NULL).
 * x_copy (nullable): the kernel also stores the X rows it stages to x_copy[i*ld_copy + c] -- this is
 * the right half of FSW_conv's torch.cat((emb, vertex_features)) (reference fsw_conv.py:357-358),
 * written while X is on chip anyway instead of by a separate copy kernel.                          */
int fsw_project_f32(const float* X, int64_t n, int d, int64_t ldx, const float* V, int S, int64_t ldv,
                    float* Xp, int64_t ldp, float* x_copy, int64_t ld_copy, int32_t* stats, fsw_stream_t stream);

/* ---- C[M, N] = beta * C + A^T . B for tall row-major A [K, lda >= M], B [K, ldb >= N] (K in the millions, M x N at most 16 blocks
 * of 64 x 64): the weight-gradient shape of the backward pass -- gV = gXp^T . X (reference fsw_embedding.py:909-913 differentiated by
 * torch.autograd) and the first Linear layer's gW (fsw_conv.py:361).  Exact fp32 products and sums (v_mfma_f32_32x32x2_f32); the K
 * axis is split over the grid, the partial results are summed in a fixed order (no float atomics: bitwise reproducible).
 * workspace: fsw_gemm_tn_workspace_bytes(M, N).                                                                            */
size_t fsw_gemm_tn_workspace_bytes(int M, int N);
int fsw_gemm_tn_f32(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t K, int M, int N, float* C, int64_t ldc, float beta,
                    void* workspace, size_t workspace_bytes, fsw_stream_t stream);

/* ---- unit-weight readout coefficients ------------------------------------------------------------
 * table[(D*(D-1)/2 + t) * ldt + k] = (1+xi_k) * [sin(2 pi xi_k (t+1)/D) - sin(2 pi xi_k t/D)] / (pi xi_k)
 * for 1 <= D <= max_deg, 0 <= t < D (the reference's Delta_t of fsw_embedding.py:1047-1075 times the
 * (1+xi) of :1109 for weights 1/D), evaluated in float64 and rounded once.                        */
size_t fsw_unit_table_rows(int max_deg);
int fsw_unit_coeff_table(const float* freqs, int S, int max_deg, float* table, int64_t ldt, fsw_stream_t stream);

/* ---- fused neighbourhood sort + segmented cumsum + Fourier readout --------------------------------
 * Replaces forward_helper's sparse branch (reference fsw_embedding.py:917-1112) and the epilogue
 * (:853-888) for the 'plain' total-mass method.  For every row r and slice k:
 *   out[r*ldo + has_mass + k] = out_scale * ( (1+xi_k) * sum_t Delta_t p_(t) + bias[has_mass + k] )
 *   out[r*ldo]                = out_scale * ( f(m_r) * mass_scale + bias[0] )        if has_mass     */
typedef struct {
  /* adjacency (from fsw_graph_build) */
  const int32_t* rowptr;
  const int32_t* col;
  const float* w; /* NULL = unit weights */
  const int32_t* perm;
  const int32_t* bin_start;
  int64_t num_rows;
  /* projected features and slice parameters */
  const float* Xp;
  int64_t ldp;
  const float* freqs; /* [S] */
  int32_t S;
  float tau;                 /* total_mass_pad_thresh */
  const float* unit_table;   /* from fsw_unit_coeff_table, required when w == NULL and tau <= 1 */
  int64_t ldt;
  /* output */
  float* out;
  int64_t ldo;
  const float* bias; /* NULL or [has_mass + S] */
  float out_scale;
  int32_t has_mass;  /* 1: column 0 of out carries the encoded total mass */
  int32_t mass_fn;   /* 0 identity, 1 sqrt: 2m/(sqrt(m+1)+1), 2 log1p      (fsw_embedding.py:857-865) */
  float mass_scale;
  /* row counts per path, host values read back from stats (num_lds_rows = stats[FSW_STAT_NUM_LDS]: every row of
   * FSW_REG_MAX_DEG < degree <= FSW_LDS_MAX_DEG, the library picks the kernel per degree bin); -1 = unknown (launch
   * every path) */
  int64_t num_reg_rows, num_lds_rows, num_global_rows, num_zero_rows;
  int64_t max_degree; /* host value of stats[FSW_STAT_MAX_DEGREE]; required when num_global_rows != 0 */
  /* scratch for rows above FSW_LDS_MAX_DEG (sorted in a scratch line per wavefront, any degree): at least
   * fsw_embed_scratch_bytes(max_degree) bytes, or NULL if num_global_rows == 0 */
  void* scratch;
  size_t scratch_bytes;
  /* edge features (reference fsw_embedding.py:934-968): the key of CSR entry e of slice k is
   * Xp[col[e], k] + sum_q efeat[e*d_edge + q] * Ve[k*ldve + q]   (Ve = projVecs[:, d_in:]).  Needs w != NULL
   * (a graph from fsw_graph_build_coalesced).  efeat == NULL / d_edge == 0: no edge features.          */
  const float* efeat;
  const float* Ve;
  int64_t ldve;
  int32_t d_edge;
  int32_t reserved;
  /* HOST copy of the FSW_NUM_BINS + 1 words bin_start points at (nullable): with it the library knows the rows of every
   * degree bin -- empty bins are not launched and every grid is sized exactly (without it the grids are sized by the class
   * totals above and surplus workgroups leave at once: 4.5 of 81 ms on a 64M-edge RMAT graph).            */
  const int32_t* bin_start_host;
} fsw_embed_args;

size_t fsw_embed_scratch_bytes(int64_t max_degree);
int fsw_embed_f32(const fsw_embed_args* args, fsw_stream_t stream);

/* ---- FSW_conv fast path: embedding fused with the first Linear layer of the MLP ---------------------
 * Y[i, :] = act( [ out_scale * E(N(i)) , x_i ] . W^T + b ),  W = [W1 | W2]  -- reference fsw_conv.py:355-362
 * (self.fsw_embed -> torch.cat((mw*emb, vertex_features)) -> mlp[0] -> activation) without writing the
 * embedding to HBM, in two calls:
 *   fsw_project_linear_f32  the projection GEMM with a second output block Y2 = X . W2^T + b2 (the
 *                           vertex-feature half of the Linear layer; W2 [H2, ldw2] row-major).  Row i of
 *                           the block is stored at Y2[row_map[i]] when row_map != NULL -- pass the graph's
 *                           invperm so that every workgroup of the fused kernel reads one contiguous run;
 *   fsw_conv_fused_f32      neighbourhood kernel + E . W1^T on the fp32 matrix cores; with Yin != NULL it
 *                           adds row invperm-position p of Yin [n, ldyin] -- or, with yin_by_node != 0, row `node` of a Yin kept
 *                           in node order (a block that a BLAS GEMM produced: no row permutation) -- else lin_bias, applies the
 *                           activation (0 none, 1 relu, 2 leaky relu with `slope`) and stores Y [n, ldy].
 * Preconditions of fsw_conv_fused_f32: unit weights (args->w == NULL), tau <= 1, fsw_conv_fused_lds_bytes() <= 64 KiB.
 * It computes the rows of in-degree 0 .. FSW_REG_MAX_DEG; rows above that are left untouched in Y -- the caller runs
 * fsw_embed_f32 for them (num_reg_rows = num_zero_rows = 0 restricts that call to the long rows) and applies the Linear
 * layer to those rows itself (fsw_gnn_amd/fsw_conv.py does: a graph with a few hubs keeps the fused kernel for the rest).
 * args->out / ldo are ignored.  Wq: W1^T packed for 16-byte operand loads, zero padded:
 *   Wq[((g*ldw + j)*8) + 4*h + i] = W1[j][8g + 2i + h],  g < ceil(K/8) + 16 (the tail groups are zero:
 *   the kernel prefetches past the end), j < ldw (Hout rounded up to 32), K = has_mass + S, h in {0,1},
 *   i in {0..3}; 16-byte aligned.                                                                      */
size_t fsw_conv_fused_lds_bytes(int S, int has_mass);
/* Packs K columns of W [Hout, ldw_in] starting at column col0 (= the W1 block that multiplies the embedding columns
 * a call of fsw_conv_fused_f32 produces: the whole embedding on one GPU, one rank's [mass |] slice block under slice
 * sharding) into Wq (fsw_packed_linear_floats(K, Hout) floats, 16-byte aligned, layout above) and, when W2out != NULL,
 * copies the d2 columns of W2 (a pointer INTO W, row stride ldw_in) to W2out [Hout, ldw2out].  One launch per
 * forward: nothing is cached on the host, so in-place edits of the weights are always seen.                        */
size_t fsw_packed_linear_floats(int K, int Hout);
int fsw_pack_linear_f32(const float* W, int64_t ldw_in, int Hout, int col0, int K, float* Wq, const float* W2, int d2,
                        float* W2out, int64_t ldw2out, fsw_stream_t stream);
int fsw_project_linear_f32(const float* X, int64_t n, int d, int64_t ldx, const float* V, int S, int64_t ldv,
                           float* Xp, int64_t ldp, const float* W2, int H2, int64_t ldw2, const float* b2, float* Y2,
                           int64_t ldy2, const int32_t* row_map, int32_t* stats, fsw_stream_t stream);
int fsw_conv_fused_f32(const fsw_embed_args* args, const float* Wq, int64_t ldw, const float* lin_bias, int Hout,
                       const float* Yin, int64_t ldyin, int yin_by_node, int act, float slope, float* Y, int64_t ldy,
                       fsw_stream_t stream);
/* R[r, c] = act(R[r, c] + Yin[r, c] + bias[c]) for rows x H floats in place, one launch (Yin, bias may be NULL; act as in
 * fsw_conv_fused_f32).  Epilogue of the slice-sharded layer forms (fsw_gnn_amd/dist.py): the rows a rank owns after the
 * reduce-scatter of the partial sums receive x . W2^T + b (reference fsw_conv.py:357-362, the vertex-feature half of mlp[0]) and
 * the activation.                                                                                                          */
int fsw_add_bias_act_f32(float* R, int64_t ldr, const float* Yin, int64_t ldyin, const float* bias, int64_t rows, int H, int act,
                         float slope, fsw_stream_t stream);

/* ---- backward of the fused neighbourhood kernels (every weight mode and degree class) -----------------
 * Replaces the reverse-mode chain of the reference's sparse autograd Functions (ag.*.backward, reference
 * fsw_embedding.py:1284-2257; reverse segcumsum :2158-2172; scatter-unsort :2055-2070).  Given the output
 * gradient g [num_rows, ldg] (column has_mass + k belongs to slice k; the same args as the forward, args->out
 * ignored) it ACCUMULATES
 *   gXp[j, k]  += out_scale * sum over edges j->i of g[i,k] * C[D_i][rank_ik(j)][k]      (gXp zeroed by the caller)
 *   gfreq[k]   += out_scale * sum_i g[i,k] * sum_t dC[D_i][t][k] * p_(t)                  (nullable)
 * with dC = dC/dxi from fsw_unit_dcoeff_table (same layout as fsw_unit_coeff_table; used on the unit-weight
 * register path, may be NULL otherwise -- weighted rows and rows above FSW_REG_MAX_DEG evaluate the
 * coefficients in float64 on the fly; args->scratch as for fsw_embed_f32 when num_global_rows != 0).  The
 * weights are constants (no gradient w.r.t. w).  The gradients of X and projVecs follow as two plain GEMMs:
 * gX = gXp . projVecs, gprojVecs = gXp^T . X.                                                              */
int fsw_unit_dcoeff_table(const float* freqs, int S, int max_deg, float* dtable, int64_t ldt, fsw_stream_t stream);
int fsw_embed_backward_f32(const fsw_embed_args* args, const float* dtable, const float* g, int64_t ldg, float* gXp,
                           int64_t ldgp, float* gfreq, fsw_stream_t stream);
/* Edge-feature graphs: the same backward, but the gradient of every KEY is stored instead of being accumulated:
 * gkey[e*ldk + k] = out_scale * g[i,k] * C(entry e, slice k) for CSR entry e of row i (gkey zeroed by the caller).
 * From it: gXp = scatter-add of gkey rows by col, g_efeat = gkey . Ve, gVe = gkey^T . efeat.               */
int fsw_embed_backward_keys_f32(const fsw_embed_args* args, const float* dtable, const float* g, int64_t ldg, float* gkey,
                                int64_t ldk, float* gfreq, fsw_stream_t stream);
/* Store-and-sum backward (no float atomics on the register rows; the reference's own backward is an index_add of the same
 * terms, ag.permute_sparse.backward fsw_embedding.py:1286): fsw_embed_backward_keys_f32 with dtable != NULL stores the key
 * gradients of a unit-weight graph, fsw_graph_transpose lists the CSR entries sender by sender (cptr [num_cols + 1], order [nnz];
 * workspace of fsw_graph_workspace_bytes(num_cols, nnz) bytes; entries whose col is out of range sort last and belong to no
 * sender), and fsw_segment_sum_rows_f32 forms out[j, 0:S] = sum over q in cptr[j] .. cptr[j+1]-1 of src[order[q], 0:S].
 * `out` must be zeroed by the caller (senders without out-edges are not written; a sender whose list crosses a 256-entry
 * segment boundary is accumulated).  Sums of up to two segments are bitwise reproducible.                    */
int fsw_graph_transpose(const int32_t* col, int64_t nnz, int64_t num_cols, int32_t* cptr, int32_t* order, void* workspace,
                        size_t workspace_bytes, fsw_stream_t stream);
int fsw_segment_sum_rows_f32(const float* src, int64_t lds, const int32_t* ptr, const int32_t* order, int64_t num_out, int64_t nnz,
                             int S, float* out, int64_t ldo, fsw_stream_t stream);

/* ---- generic neighbourhood kernels: any in-degree, float32 or float64 storage, float64 arithmetic ------------------------
 * (csrc/embed_generic.hip)  Two uses:
 *   value_dtype 1 (float64): the float64 build of the path -- FSW_embedding / FSW_conv(dtype=torch.float64), which the
 *       reference's own test_conv.py runs (test_conv.py:24); forward (g == NULL) and backward (g != NULL);
 *   value_dtype 0 (float32): gradients with respect to the weights for the float32 path (gw), which the tuned backward
 *       kernels above treat as constants (reference ag.div_sparse_dense.backward fsw_embedding.py:1656,
 *       ag.cumsum_sparse.backward :2160, ag.permute_sparse.backward :1286).
 * The graph is a plain CSR (rowptr / col / w in CSR order, no degree bins).  All value pointers have the type selected by
 * value_dtype.  Ke (nullable): the edge-feature term of every key, Ke[e * ldke + k] = <efeat_e, projVecs[k, d_in:]>
 * (reference fsw_embedding.py:934-968), added to Xp[col[e], k].
 * Forward:  out[r * ldo + has_mass + k] = out_scale * ((1 + xi_k) sum_t Delta_t p_(t) + bias[has_mass + k]), and with has_mass
 *           out[r * ldo] = out_scale * (f(m_r) * mass_scale + bias[0]).
 * Backward: for the output gradient g [num_rows, ldg] (column has_mass + k belongs to slice k)
 *           gkey[e * ldk + k]  = out_scale * g[r, k] * d out[r, k] / d key_e              (stored; nullable)
 *           gfreq[k]          += out_scale * sum_r g[r, k] * d out[r, k] / d xi_k          (accumulated; nullable)
 *           gw[e]             += out_scale * sum_k g[r, k] * d out[r, k] / d w_e           (accumulated; nullable; the
 *                                total-mass column's dependence on w is NOT included).
 * scratch: fsw_embed_generic_scratch_bytes(max_degree, num_rows) bytes.                                                */
typedef struct {
  int32_t value_dtype;   /* 0 float32, 1 float64 */
  int32_t S;
  const int32_t* rowptr;
  const int32_t* col;
  const void* w;         /* [nnz] raw weights, NULL = unit */
  int64_t num_rows;
  int64_t max_degree;    /* host value: an upper bound of the longest row */
  const void* Xp;
  int64_t ldp;
  const void* Ke;
  int64_t ldke;
  const void* freqs;
  double tau;
  void* out;
  int64_t ldo;
  const void* bias;
  double out_scale;
  int32_t has_mass;
  int32_t mass_fn;
  double mass_scale;
  const void* g;
  int64_t ldg;
  void* gkey;
  int64_t ldk;
  void* gfreq;
  void* gw;
  void* scratch;
  size_t scratch_bytes;
} fsw_generic_args;

size_t fsw_embed_generic_scratch_bytes(int64_t max_degree, int64_t num_rows);
int fsw_embed_generic(const fsw_generic_args* args, fsw_stream_t stream);
/* Xp [n, ldp] = X [n, ldx] . V [S, ldv]^T in float64 on the matrix cores (v_mfma_f64_16x16x4_f64); sets
 * FSW_FLAG_X_NONFINITE in stats[FSW_STAT_FLAGS] (stats nullable) */
int fsw_project_f64(const double* X, int64_t n, int d, int64_t ldx, const double* V, int S, int64_t ldv, double* Xp,
                    int64_t ldp, int32_t* stats, fsw_stream_t stream);

/* ---- stand-alone segmented cumulative sum --------------------------------------------------------
 * Replaces segcumsum / segcumsum_cuda (reference fsw_embedding.py:2795-3012): inclusive scan of
 * values restarted wherever consecutive segment ids differ, ONE streaming pass (chained scan with decoupled
 * look-back: every element read once and written once), in place allowed (out == values).  value_dtype: 0 float32, 1 float64 (the reference's torch_dtype
 * enum, fsw_embedding.cu:14-17).  id_bytes: 4 or 8.  reverse != 0 scans from the end (the backward
 * pass of cumsum_sparse, fsw_embedding.py:2158-2172).  workspace: fsw_segcumsum_workspace_bytes(n). */
size_t fsw_segcumsum_workspace_bytes(int64_t n);
int fsw_segcumsum(int value_dtype, const void* values, void* out, const void* segment_ids, int id_bytes,
                  int64_t n, int reverse, void* workspace, size_t workspace_bytes, fsw_stream_t stream);

/* ---- legacy entry points: exact signatures of reference fsw_embedding.cu:194, 212, 231 ------------
 * Same semantics as the reference kernels (block-local segmented scan + carry add), launched on the
 * default stream with a stream synchronise after the launch, void return.                           */
void segcumsum_wrapper(int dtype, void* values, const int64_t* segment_ids, int64_t size, int64_t max_seg_size,
                       void* block_sums_out, int64_t* block_last_ids_out, bool return_next_level, int64_t num_blocks,
                       int64_t threads_per_block, size_t shared_memory_size);
void add_block_sums_wrapper(int dtype, void* output, const void* block_sums, const int64_t* segment_ids,
                            const int64_t* block_last_id, int64_t size, int64_t num_blocks, int64_t threads_per_block);
int get_max_threads_per_block(int device_index);
/* the launch helpers the reference library exports next to its wrappers (fsw_embedding.cu:125-183): default stream, no
 * synchronisation, void return */
void launch_segcumsum_kernel_float(float* values, const int64_t* segment_ids, int64_t size, int64_t max_seg_size,
                                   float* block_sums_out, int64_t* block_last_ids_out, bool return_next_level,
                                   int64_t num_blocks, int64_t threads_per_block, int64_t shared_memory_size);
void launch_segcumsum_kernel_double(double* values, const int64_t* segment_ids, int64_t size, int64_t max_seg_size,
                                    double* block_sums_out, int64_t* block_last_ids_out, bool return_next_level,
                                    int64_t num_blocks, int64_t threads_per_block, int64_t shared_memory_size);
void launch_add_block_sums_kernel_float(float* output, const float* block_sums, const int64_t* segment_ids,
                                        const int64_t* block_last_id, int64_t size, int64_t num_blocks, int64_t threads_per_block);
void launch_add_block_sums_kernel_double(double* output, const double* block_sums, const int64_t* segment_ids,
                                         const int64_t* block_last_id, int64_t size, int64_t num_blocks, int64_t threads_per_block);

#ifdef __cplusplus
}
#endif
#endif /* FSW_HIP_H */
