/*
 * maxsim.h -- C ABI of libmaxsim.so: MI355X (gfx950) ColBERT late-interaction (MaxSim) rerank path.
 *
 * This is the drop-in boundary for ONE hot path of wuyaoxuehun/colbert (a pure-Python/torch code base, so
 * there is no reference FFI to mirror; each entry point names the reference Python it replaces, paths
 * relative to the reference checkout):
 *
 *   maxsim_score_dense  <- BaseModel.score(Q, D, q_mask, d_mask)      colbert/modeling/BaseModel.py:39-46
 *   maxsim_score_dense_fwd/_bwd <- the same operator under autograd    colbert/modeling/colbert_model.py:87-96
 *   maxsim_rerank       <- the gather/pad/mask/score body of
 *                          ColbertRanker.rank_forward                 colbert/ranking/colbert_ranker.py:88-118
 *                          (batched over queries: replaces the per-query loop
 *                           colbert/training/dense_server_client.py:44-48)
 *   maxsim_topk         <- sort(descending)+[:depth]                  colbert/ranking/colbert_ranker.py:128-130
 *                          (also the per-query merge after the doc-sharded RCCL all-gather)
 *   maxsim_embedding_ids_to_pids <- ColbertIndex.embedding_ids_to_pids colbert/ranking/colbert_ranker.py:212-229
 *   maxsim_rerank_ex    <- maxsim_rerank + the per-query keep_nonzero   colbert/training/training_utils.py:48-53,
 *                          of the batched driver loop                   colbert/training/dense_server_client.py:44-45
 *   maxsim_rank_forward <- ONE call of ColbertRanker.rank_forward       colbert/ranking/colbert_ranker.py:75-137
 *                          (the reference's online call shape, colbert/indexing/faiss_indexers.py:234)
 *   maxsim_build_doc_table <- index state set up by init_ranker         colbert/ranking/colbert_ranker.py:31-43
 *   maxsim_rerank_counted / maxsim_topk_counted <- the same two steps on partly-live candidate rows (a doc shard's share of
 *                          every list; the per-query distinct pids of colbert_ranker.py:212-229), scheduled from a device-built
 *                          work list
 *   maxsim_shard_candidates <- no reference counterpart (its rerank is single-GPU, colbert_ranker.py:154): the
 *                          per-rank candidate filter of the doc-sharded path
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. torch tensors); the library allocates
 *     nothing and keeps no state between calls; it is re-entrant per stream;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is asynchronous on it;
 *   - return value: MAXSIM_OK (0) or a negative MAXSIM_E* code; no exceptions cross the ABI;
 *   - all tensors are dense row-major; any alignment is accepted (token matrices and queries that are not 16-byte
 *     aligned are scored by the generic kernel instead of the streaming kernels).
 */
#ifndef MAXSIM_H
#define MAXSIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAXSIM_VERSION 123 /* 0.1.1: maxsim_index_view, maxsim_rerank_ex (q_mask, doc table), maxsim_rank_forward,
                              maxsim_shard_candidates, maxsim_build_doc_table; 111: maxsim_score_dense_kernel;
                              120: counted candidate rows (maxsim_rerank_counted, maxsim_topk_counted);
                              121: maxsim_index_view.uniform_len, read-ceiling probes, maxsim_host_alloc_coherent;
                              122: maxsim_embedding_ids_to_pids_ex (tok_keep, id_base, row-block table:
                                   maxsim_row_blocks_bytes / maxsim_build_row_blocks), maxsim_index_view_bytes,
                                   maxsim_index_view.struct_size in place of `reserved` (checked by maxsim_rerank_ex,
                                   maxsim_rerank_counted and maxsim_rank_forward: 0 or the caller's sizeof, else MAXSIM_EINVAL);
                              123: no new symbol -- uniform_len = 4 / 8 / 16 now also selects a fixed-length kernel on an
                                   fp16 / bf16 index (the reference's multi-view storage); same results, bit for bit */

/* element types of Q / D / index */
#define MAXSIM_F32 0
#define MAXSIM_F16 1
#define MAXSIM_BF16 2
/* maxsim_rerank index_dtype only: fp32 storage, contraction on the 16-bit matrix pipe with BOTH operands split
   into fp16 pieces on the fly (|error| ~1e-6 on a score; needs |x| < 65504, e.g. L2-normalised embeddings) */
#define MAXSIM_F32_FAST 3
/* maxsim_rerank index_dtype only: fp32 storage, "3 x bf16" contraction: both operands cut exactly into three bf16
   pieces, six piece products on the bf16 matrix pipe, fp32 accumulation: fp32-class accuracy, no magnitude limit */
#define MAXSIM_F32_BF16X3 4

/* element types of the mask tensors of maxsim_score_dense */
#define MAXSIM_MASK_NONE 0 /* both mask pointers ignored: all ones */
#define MAXSIM_MASK_I64 1  /* what rank_forward passes, colbert_ranker.py:111-112 */
#define MAXSIM_MASK_I32 2
#define MAXSIM_MASK_F32 3 /* what test_score passes, BaseModel.py:73 */
#define MAXSIM_MASK_U8 4  /* torch.bool / uint8 */

/* error codes */
#define MAXSIM_OK 0
#define MAXSIM_EINVAL -1  /* bad argument (null pointer, negative size, unknown dtype) */
#define MAXSIM_EEMPTY -2  /* empty candidate list / empty doc axis: the reference asserts or raises there
                             (colbert_ranker.py:76; max over an empty dim at BaseModel.py:44) */
#define MAXSIM_ERANGE -3  /* size outside what the kernels support (see each function) */
#define MAXSIM_ELAUNCH -4 /* the HIP runtime refused the launch (hipGetLastError != hipSuccess) */

int maxsim_version(void);
const char* maxsim_strerror(int code);

/*
 * All-pairs MaxSim, BaseModel.py:39-46:
 *   out[q, d] = sum_m max_n  dot( Q[q,m,:] * q_mask[q,m] ,  D[d,n,:] * d_mask[d,n] )
 * Masked tokens are zeroed (they contribute similarity 0 to the max -- a floor -- and 0 to the sum).
 *
 *   Q       [nq, Lq, h]  element type `dtype`
 *   D       [nd, Ld, h]  element type `dtype`
 *   q_mask  [nq, Lq], d_mask [nd, Ld]  element type `mask_dtype` (both), any numeric values
 *   out     [nq, nd] float32 (the arithmetic is fp32 whatever `dtype` is)
 * nq == 0 or nd == 0 is a no-op; Lq == 0 writes zeros; Ld == 0 -> MAXSIM_EEMPTY; h >= 0.
 * Fast paths (MFMA + LDS-DMA streaming): 16 <= h <= 1024 with 16-byte-aligned rows, any of the three dtypes (h == 128 fp32
 * keeps the query tile in registers); queries longer than 32 tokens take one launch per 32 tokens (the sum over
 * query tokens is additive).  Every other shape runs the generic kernel.
 */
int maxsim_score_dense(const void* Q, const void* D, const void* q_mask, const void* d_mask, int nq, int nd,
                       int Lq, int Ld, int h, int dtype, int mask_dtype, float* out, void* stream);

/*
 * Training form of the same operator (its second caller: colbert/modeling/colbert_model.py:87-96 differentiates
 * BaseModel.score through torch autograd and materialises the [q,d,m,n] similarity tensor).  The forward also
 * records, per (q, d, query token m), the arg-max doc token (torch.max semantics: the first maximal index); the
 * backward routes the incoming gradient through those tokens:
 *     dQ[q,m,:] = q_mask[q,m] * sum_d g[q,d] * d_mask[d,i] * D[d,i,:]        i = argmax[q,d,m]
 *     dD[d,i,:] += d_mask[d,i] * g[q,d] * q_mask[q,m] * Q[q,m,:]
 *   argmax [nq, nd, Lq] int32;  grad_out [nq, nd] float32;  dQ [nq, Lq, h], dD [nd, Ld, h] float32 (either may be NULL;
 *   both are fully overwritten).  fp32 accumulation in a fixed order (reproducible) except in the global-atomics fallback.
 * h <= 1024 for the backward.
 */
/* Which kernel serves an all-pairs problem of this shape with 16-byte aligned operands: 1 = the GEMM-blocked kernel
 * (16-bit Q and D of one type, h % 64 == 0, h >= 128, Lq <= 32, Ld <= 384, float32 masks or none, Q and D below 3.75 GB
 * each, at least 128 (doc, 8-query) tiles), 0 = the streaming / generic kernels (everything else).  With 0/1 masks (what
 * the reference passes: tokenizers.py:57 prefix masks, q_mask of ones) the two give the same scores and the same arg-max.
 * With general float masks they agree to 16-bit rounding only: BaseModel.score multiplies Q and D by their masks in the
 * operand dtype before the matmul, as the streaming kernel does, while the GEMM-blocked kernel weighs the finished fp32
 * similarities (d_mask) and the maximum (non-negative q_mask) -- more accurate, not bit-identical, and a tie created by
 * the weighting can move the first-max arg-max.  Tests and benchmarks use this function to say what they measured.
 * Negative: MAXSIM_EINVAL. */
int maxsim_score_dense_kernel(int nq, int nd, int Lq, int Ld, int h, int dtype, int mask_dtype);

int maxsim_score_dense_fwd(const void* Q, const void* D, const void* q_mask, const void* d_mask, int nq, int nd,
                           int Lq, int Ld, int h, int dtype, int mask_dtype, float* out, int32_t* argmax,
                           void* stream);
int maxsim_score_dense_bwd(const void* Q, const void* D, const void* q_mask, const void* d_mask,
                           const int32_t* argmax, const float* grad_out, int nq, int nd, int Lq, int Ld, int h,
                           int dtype, int mask_dtype, float* dQ, float* dD, void* workspace, int64_t workspace_bytes,
                           void* stream);
/* Bytes of device scratch with which maxsim_score_dense_bwd computes dD through a per-doc inverse index (fastest,
 * reproducible).  workspace may be NULL / smaller: dD then falls back to an LDS slab per doc (or, for very long docs,
 * global float atomics). */
int64_t maxsim_score_dense_bwd_workspace(int nq, int nd, int Lq, int Ld);

/*
 * Fused ragged rerank, the body of rank_forward (colbert_ranker.py:88-118) for a batch of queries, with the
 * token index resident in HBM and no padded copy of D:
 *   scores[q, c] = sum_{m < q_len[q]}  max( floor_c , max_{t < doclens[pid]} dot(Q[q,m,:], index[tok_offsets[pid]+t,:]) )
 *   pid = cand_pids[q, c];   floor_c = 0 if pad_len != NULL && pad_len[pid] > doclens[pid] else -inf
 * `pad_len[pid]` is the stride S_g of the length bucket the reference would gather the doc at
 * (colbert_ranker.py:90): the reference's zero-masked padding slots floor the max at 0 exactly when
 * doclen < S_g.  pad_len == NULL means "no padding anywhere" (no floor).
 *
 *   index       [n_tokens, h]  element type index_dtype (the reference stores fp16, colbert_ranker.py:62)
 *   tok_offsets [n_docs] int64  first token row of each doc (doclens prefix sum, colbert_ranker.py:32)
 *   doclens     [n_docs] int32
 *   pad_len     [n_docs] int32 or NULL
 *   Q           [nq, Lq, h] element type q_dtype (F32, or F16/BF16 when the encoder already emits 16-bit
 *               queries); token-major (the Python shim undoes the reference's [1,h,Lq] permute)
 *   q_len       [nq] int32 or NULL (= Lq for every query); tokens m >= q_len[q] are dropped
 *               (what keep_nonzero does before search(), training_utils.py:48-53)
 *   cand_pids   [nq, ncand] int64; an entry < 0 or >= n_docs is a padding slot: its score is -inf
 *   scores      [nq, ncand] float32
 * A doc with doclens == 0 scores 0.  ncand == 0 -> MAXSIM_EEMPTY (colbert_ranker.py:76).
 * Fast paths (MFMA + LDS-DMA streaming; queries longer than 32 tokens: one launch per 32): h == 128 (query tile in registers; index F32: f32-input
 * MFMA, exact; F16/BF16: 16-bit MFMA with the fp32 query split into 2/3 pieces, no query bits dropped) or
 * any 16 <= h <= 1024 whose rows are a multiple of 16 bytes (query tile staged in LDS; e.g. the reference's default
 * dim 768; widths that are not a multiple of 128 pad the last 128-dim block with zeros on the query side).
 * n_tokens must be < 2^32 for the fast paths.  Everything else runs the generic kernel.
 */
int maxsim_rerank(const void* index, int index_dtype, int64_t n_tokens, const int64_t* tok_offsets,
                  const int32_t* doclens, const int32_t* pad_len, int64_t n_docs, const void* Q, int q_dtype,
                  const int32_t* q_len, const int64_t* cand_pids, int nq, int ncand, int Lq, int h,
                  float* scores, void* stream);

/*
 * Per-query top-k by score, descending (colbert_ranker.py:128-130); ties broken by lower position in the
 * candidate list (the reference's torch.sort is unstable, so any tie order is conforming).
 *   scores [nq, ncand] float32; pids [nq, ncand] int64 or NULL (then positions 0..ncand-1 are returned)
 *   out_scores [nq, k] float32, out_pids [nq, k] int64; if k > ncand the tail is (-inf, -1).
 * 1 <= ncand <= 16384 (the reference's BSIZE, colbert_ranker.py:11), k >= 1.
 */
int maxsim_topk(const float* scores, const int64_t* pids, int nq, int ncand, int k, float* out_scores,
                int64_t* out_pids, void* stream);

/*
 * Candidate-side glue: ANN result (embedding ids) -> per-query distinct pid lists, the GPU form of
 * ColbertIndex.embedding_ids_to_pids (colbert_ranker.py:212-229: emb2pid lookup + per-query set()).  The pid of a
 * token row is found by binary search in tok_offsets (no separate emb2pid table, colbert_ranker.py:163-174).
 *   emb_ids   [nq, n] int64 token rows (FAISS ids); entries < 0 or >= n_tokens are dropped
 *   out_pids  [nq, n] int64: the query's distinct pids ascending, then -1 padding (directly usable as cand_pids)
 *   out_count [nq] int32: number of distinct pids
 * 1 <= n <= 16384 (= 32 query tokens x faiss_depth 512, the reference's BSIZE); n_docs < 2^32 - 1.
 */
int maxsim_embedding_ids_to_pids(const int64_t* emb_ids, int nq, int n, const int64_t* tok_offsets, int64_t n_docs,
                                 int64_t n_tokens, int64_t* out_pids, int32_t* out_count, void* stream);

/*
 * The same step with the driver's two elementwise passes folded in and a faster row -> pid lookup:
 *   ids_per_token, tok_keep [nq, n / ids_per_token] uint8 or NULL : slot i of a query holds a neighbour of its token
 *               i / ids_per_token (the layout of colbert_ranker.py:178); the neighbours of a token with tok_keep == 0 are
 *               ignored -- the query tokens keep_nonzero drops before the search (training_utils.py:48-53,
 *               dense_server_client.py:45).  n must be a multiple of ids_per_token when tok_keep is given.
 *   id_base   : subtracted from every id first; ids outside [id_base, id_base + n_tokens) are dropped.  A doc shard passes
 *               the global token row of its first token and gets LOCAL pids of the rows that are its own.
 *   row_blocks: NULL, or the table maxsim_build_row_blocks wrote for this (tok_offsets, n_docs, n_tokens), 8-byte aligned:
 *               8 bytes per 64 token rows -- the doc of the block's first row and where the next doc starts inside it (the
 *               reference keeps 4 bytes for EVERY row: emb2pid, colbert_ranker.py:163-174).  With it a lookup is one 8-byte
 *               load in the common case instead of a ~log2(n_docs)-step binary search.
 * Same output contract as maxsim_embedding_ids_to_pids.
 */
int maxsim_embedding_ids_to_pids_ex(const int64_t* emb_ids, int nq, int n, int ids_per_token, const uint8_t* tok_keep,
                                    int64_t id_base, const int64_t* tok_offsets, int64_t n_docs, int64_t n_tokens,
                                    const void* row_blocks, int64_t* out_pids, int32_t* out_count, void* stream);
/* Bytes of the row-block table of an index with n_tokens rows; its builder (one launch at index load time). */
int64_t maxsim_row_blocks_bytes(int64_t n_tokens);
int maxsim_build_row_blocks(const int64_t* tok_offsets, int64_t n_docs, int64_t n_tokens, void* row_blocks, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Index view: everything the rerank entry points need to know about one HBM-resident (shard of an) index, passed as
 * one struct in HOST memory (read during the call, not retained).  The first seven fields are the index arguments of
 * maxsim_rerank; `doc_table` is optional.
 */
typedef struct maxsim_index_view {
  const void* index;          /* [n_tokens, h] token matrix, element type index_dtype */
  int32_t index_dtype;        /* MAXSIM_F32 / F16 / BF16 / F32_FAST / F32_BF16X3 */
  int32_t h;                  /* embedding width */
  int64_t n_tokens;
  const int64_t* tok_offsets; /* [n_docs] */
  const int32_t* doclens;     /* [n_docs] */
  const int32_t* pad_len;     /* [n_docs] or NULL */
  int64_t n_docs;
  const void* doc_table;      /* NULL, or n_docs packed 16-byte rows written by maxsim_build_doc_table from the three
                                 arrays above: the kernels then read one cache line per candidate instead of three */
  int32_t uniform_len;        /* 0, or L > 0: EVERY doc has exactly L tokens (doclens[pid] == L, tok_offsets[pid] == pid * L:
                                 the prefix sum) and no doc is padded (pad_len NULL or == L everywhere) -- the reference's
                                 multi-view configuration, where every doc keeps d_view viewer tokens
                                 (proj_conf/dense.yaml:31-32).  A promise by the caller, who knows the doclens on the
                                 host; for h == 128, an fp32 index and L in {4, 8, 16} the rerank then runs a kernel
                                 with the doc length compiled in, which does not consult the per-doc arrays at all
                                 (bit-identical scores, ~10 % faster). */
  int32_t struct_size;        /* 0, or sizeof(maxsim_index_view) as the CALLER compiled it: the library refuses a size it
                                 does not know (MAXSIM_EINVAL) instead of reading fields the caller's struct does not
                                 have.  (The field sits where `reserved` (0) was up to version 121.  A caller built
                                 against a header older than 120 has a shorter struct and no such field: compare
                                 maxsim_version() with the MAXSIM_VERSION it was compiled with, or
                                 maxsim_index_view_bytes() with its sizeof, at start-up.) */
} maxsim_index_view;
/* sizeof(maxsim_index_view) in THIS build of the library. */
int64_t maxsim_index_view_bytes(void);

/* Bytes of the packed descriptor table of n_docs docs (16 per doc). */
int64_t maxsim_doc_table_bytes(int64_t n_docs);
/* table[pid] = {int64 tok_offsets[pid], int32 doclens[pid], int32 pad_len ? pad_len[pid] : doclens[pid]}.
 * `table` must be 16-byte aligned device memory of maxsim_doc_table_bytes(n_docs) bytes.  Done once per index
 * (colbert_ranker.py:31-43 computes the same per-doc state at load time). */
int maxsim_build_doc_table(const int64_t* tok_offsets, const int32_t* doclens, const int32_t* pad_len,
                           int64_t n_docs, void* table, void* stream);

/*
 * maxsim_rerank with the index passed as a view and one more optional argument:
 *   q_mask [nq, Lq] uint8 or NULL: query token m of query q is scored iff m < q_len[q] (when given) AND q_mask[q,m] != 0.
 * This is the reference's per-query `keep_nonzero` (training_utils.py:48-53, applied at dense_server_client.py:45 to
 * the tokenizer's q_active_padding, which zeroes punctuation and [SEP] MID-sequence, tokenizers.py:36) for a whole
 * batch without compacting Q: a dropped token behaves as a zero query row, whose similarities are all exactly 0 and
 * add 0 to the sum -- the value the reference gets by removing the token.
 * Everything else (shapes, padding slots, the 0-floor, error codes, fast paths) is as maxsim_rerank.
 */
int maxsim_rerank_ex(const maxsim_index_view* iv, const void* Q, int q_dtype, const int32_t* q_len,
                     const uint8_t* q_mask, const int64_t* cand_pids, int nq, int ncand, int Lq, float* scores,
                     void* stream);

/*
 * One ColbertRanker.rank_forward (colbert_ranker.py:75-137) in one call: rerank of ONE query against n candidate
 * pids followed by the descending top-`depth` -- the reference's online call (faiss_indexers.py:234), where per-call
 * host overhead and launch latency, not bandwidth, dominate.  Two kernels are enqueued back to back on `stream` (the
 * rerank -- with the docs of a 16-bit index split over several waves when the launch is small -- and the top-k, which
 * for n <= 2048 ranks by counting on n / 16 workgroups); nothing is allocated, copied or synchronised in between.
 *   Q          [Lq, h] token-major, element type q_dtype (the shim undoes the reference's [1,h,Lq] permute)
 *   pids       [n] int64, any memory the GPU can read: device memory, or PINNED host memory (hipHostMalloc /
 *              torch pin_memory), which saves the H2D copy call
 *   workspace  device scratch of maxsim_rank_forward_workspace_bytes(n) bytes, 16-byte aligned; its first 64 bytes
 *              must be ZERO before the first call and are left zero by every call (counters of the fused epilogue);
 *              the n floats that follow receive the full score vector (colbert_ranker.py:122).  One workspace serves
 *              one call at a time.
 *   out_pids   [k] int64 and out_scores [k] float32, k = min(depth, n): device memory or pinned host memory (the kernel
 *              then writes the result straight to the host; no D2H copy call)
 *   done_flag  NULL, or one uint32 in host-COHERENT pinned memory (hipHostMallocCoherent) owned by the caller for this
 *              workspace: with sync != 0 and n <= 2048 the call then waits by polling this word, which the top-k kernel's
 *              last workgroup stores to after every output is written (a fraction of the cost of a stream
 *              synchronisation); out_pids / out_scores must then be host-coherent too
 *   sync       != 0: the results are complete and visible to the host when the call returns
 * n == 0 -> MAXSIM_EEMPTY (assert len(pids) > 0, colbert_ranker.py:76); n <= 16384 (BSIZE, colbert_ranker.py:11).
 */
int64_t maxsim_rank_forward_workspace_bytes(int n);
int maxsim_rank_forward(const maxsim_index_view* iv, const void* Q, int q_dtype, int Lq, const int64_t* pids, int n,
                        int depth, void* workspace, int64_t* out_pids, float* out_scores, uint32_t* done_flag,
                        int sync, void* stream);

/*
 * Doc-sharded rerank, per-rank candidate filter (the reference reranks on one GPU only): this rank owns
 * the global pid range [lo, hi).  Per query, the in-range entries of cand_global [nq, ncand] are moved to the front of
 * the row in list order (stable) -- as local pids (pid - lo) in out_local and unchanged in out_global (may be NULL) --
 * and the rest of both rows is -1 (a padding slot for maxsim_rerank: score -inf, no tokens read).  out_count [nq]
 * (may be NULL) receives the number of in-range entries.  out_local may alias cand_global (in-place).
 * The row width stays ncand, so nothing has to be read back to size the rerank launch.
 */
int maxsim_shard_candidates(const int64_t* cand_global, int nq, int ncand, int64_t lo, int64_t hi,
                            int64_t* out_local, int64_t* out_global, int32_t* out_count, void* stream);

/*
 * Counted candidate rows.  A candidate matrix whose rows are only partly live -- one rank's share of a doc-sharded step
 * (maxsim_shard_candidates: about 1/N of every row), the distinct pids of an ANN search (maxsim_embedding_ids_to_pids) --
 * comes with a per-row live count in DEVICE memory.  Precondition for both entry points below: the live entries of row q
 * are its first cand_count[q] slots and every slot after them holds a negative pid (what those two functions write).
 * Results are then identical to maxsim_rerank_ex / maxsim_topk on the same matrices; what changes is the schedule:
 *
 * maxsim_rerank_counted: the device builds a dense list of work items from the counts (two small kernels: a scan and a
 *   fill, which also writes the -inf tail of `scores`) and the streaming kernel runs as a fixed grid that walks that list
 *   -- no host synchronisation to size the launch, no all-padding workgroups, every wave of every workgroup busy.  One
 *   rank's share of an 8-way sharded step then costs what the same docs cost as dense rows.  Served (what rerank_impl
 *   dispatches, maxsim.hip): every h == 128 index with 16-byte-aligned rows and Lq <= 32 per query slice, whatever its
 *   doc lengths and index dtype (wave-sized items; a uniform 4 / 8 / 16-token fp32 index (uniform_len) runs the
 *   fixed-length kernel's list form); 16 <= h <= 1024 with 16-byte-aligned rows, h != 128, whenever the query image and
 *   at least 4 one-sub-tile rings fit the 160 KiB of LDS (workgroup-sized items: the waves share the staged query) --
 *   including widths with a partial last 128-dim block (96, 192, 320 ...).  Other shapes (unaligned rows, h < 16,
 *   h > 1024) take maxsim_rerank_ex's path.
 *   worklist: 16-byte aligned device scratch of maxsim_worklist_bytes(nq, ncand) bytes (contents need not survive the
 *   call; NULL or too small = maxsim_rerank_ex's path).  ncand < 2^20 for the list form.
 * maxsim_topk_counted: maxsim_topk that ranks the live slots only (rows longer than 2048: sorted as the next power of
 *   two >= the row's count instead of >= its width).
 */
int64_t maxsim_worklist_bytes(int nq, int ncand);
int maxsim_rerank_counted(const maxsim_index_view* iv, const void* Q, int q_dtype, const int32_t* q_len,
                          const uint8_t* q_mask, const int64_t* cand_pids, const int32_t* cand_count, int nq, int ncand,
                          int Lq, float* scores, void* worklist, int64_t worklist_bytes, void* stream);
int maxsim_topk_counted(const float* scores, const int64_t* pids, const int32_t* counts, int nq, int ncand, int k,
                        float* out_scores, int64_t* out_pids, void* stream);

/*
 * Host-COHERENT pinned memory for the buffers of maxsim_rank_forward that the host reads while the kernel may still be
 * running (out_pids, out_scores, done_flag): hipHostMalloc(hipHostMallocCoherent) from the HIP runtime THIS library is
 * linked against (a caller that resolved the runtime by name could load a second copy whose allocations the first does
 * not know).  The pointer is valid on the host and on every device.  NULL on failure.  Free with maxsim_host_free.
 */
void* maxsim_host_alloc_coherent(int64_t bytes);
void maxsim_host_free(void* p);

/*
 * Measurement aid ("measure achievable with a copy/read microbench on the box"): streams the first
 * `bytes` (rounded down to a multiple of 2 MiB, returned in *bytes_read when given) of a device buffer through the rerank
 * kernels' own fetch path -- non-temporal LDS-DMA into per-wave LDS rings -- and consumes nothing.  Timing this launch
 * with HIP events gives the read rate the memory system delivers to that access pattern on this box, the ceiling
 * bench.py reports as roofline.read_ceiling next to the 8 TB/s spec peak.  variant: 0 = one 16 KiB tile per wave (the fp32
 * kernel's ring), 1 = two 8 KiB tiles (the 16-bit kernels' ring), 2 = two 16 KiB tiles.  buf must be 16-byte aligned.
 */
int maxsim_hbm_read_probe(const void* buf, int64_t bytes, int variant, int64_t* bytes_read, void* stream);
/* The same rings fed with SCATTERED pieces: `read_bytes` (a multiple of 1 MiB) are read as granules of `granule` bytes (a
 * power of two, 1 KiB .. 1 MiB: one short doc) taken from hashed positions of the first `bytes` of buf -- the ceiling of
 * a rerank over short docs (C4: 4 KiB docs).  variant: 0 = one 16 KiB tile per wave, 1 = two 8 KiB tiles, 2 = one 8 KiB
 * tile per wave (the short-doc kernel: 16 waves per CU), 3 = the same with twice the bytes per wave. */
int maxsim_hbm_read_probe_scattered(const void* buf, int64_t bytes, int granule, int variant, int64_t read_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MAXSIM_H */
