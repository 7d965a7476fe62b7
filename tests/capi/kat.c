/* kat.c -- libmaxsim.so driven from plain C through include/maxsim.h (no Python, no torch): the reference's
 * known-answer test (colbert/modeling/BaseModel.py:70-75 -> [[21, 41]]), a masked variant, the zero-floor pair (SURVEY 8c golden 2),
 * a three-doc ragged rerank + top-k, the view-based entry points (doc table, q_mask, one-call rank_forward with pinned
 * host buffers, doc-shard filter), and the error codes.  Built by __graft_entry__.build() with hipcc (the HIP
 * runtime is used only for hipMalloc/hipMemcpy); run by tests/test_gpu_parity.py::test_c_abi_from_plain_c.
 * Prints one line per check; exit status 0 iff all pass. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "maxsim.h"

static int failures = 0;
#define CHECK(cond, name)                                     \
  do {                                                        \
    printf("%s %s\n", (cond) ? "ok  " : "FAIL", name);        \
    if (!(cond)) ++failures;                                  \
  } while (0)

static void* to_dev(const void* host, size_t bytes) {
  void* d = NULL;
  if (hipMalloc(&d, bytes ? bytes : 1) != hipSuccess) return NULL;
  if (bytes) hipMemcpy(d, host, bytes, hipMemcpyHostToDevice);
  return d;
}

int main(void) {
  CHECK(maxsim_version() == MAXSIM_VERSION, "version");

  /* the reference's known-answer test, BaseModel.test_score (BaseModel.py:70-75): all-ones float masks -> [[21, 41]] */
  {
    const float Q[6] = {1, 5, 4, 2, 8, 1};
    const float D[12] = {0, 0, 0, 1, 1, 1, 3, 2, 1, 1, 1, 3};
    const float qm[2] = {1, 1}, dm[4] = {1, 1, 1, 1};
    float *dQ = to_dev(Q, sizeof Q), *dD = to_dev(D, sizeof D), *dqm = to_dev(qm, sizeof qm), *ddm = to_dev(dm, sizeof dm);
    float* dout = NULL;
    hipMalloc((void**)&dout, 2 * sizeof(float));
    int rc = maxsim_score_dense(dQ, dD, dqm, ddm, 1, 2, 2, 2, 3, MAXSIM_F32, MAXSIM_MASK_F32, dout, NULL);
    float out[2] = {0, 0};
    hipDeviceSynchronize();
    hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost);
    printf("     score_dense -> [%g, %g] (rc %d)\n", out[0], out[1], rc);
    CHECK(rc == MAXSIM_OK && out[0] == 21.0f && out[1] == 41.0f, "test_score KAT = [[21, 41]]");
    /* a masked query token contributes 0, a masked doc token similarity 0 (BaseModel.py:41-42) */
    const float qm2[2] = {1, 0}, dm2[4] = {1, 1, 1, 0};
    hipMemcpy(dqm, qm2, sizeof qm2, hipMemcpyHostToDevice);
    hipMemcpy(ddm, dm2, sizeof dm2, hipMemcpyHostToDevice);
    rc = maxsim_score_dense(dQ, dD, dqm, ddm, 1, 2, 2, 2, 3, MAXSIM_F32, MAXSIM_MASK_F32, dout, NULL);
    hipDeviceSynchronize();
    hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost);
    CHECK(rc == MAXSIM_OK && out[0] == 10.0f && out[1] == 17.0f, "masks: query token 1 and doc 1's second token masked -> [[10, 17]]");
    /* error behaviour: empty doc axis (the reference's max over an empty dim raises), bad dtype */
    CHECK(maxsim_score_dense(dQ, dD, dqm, ddm, 1, 2, 2, 0, 3, MAXSIM_F32, MAXSIM_MASK_F32, dout, NULL) == MAXSIM_EEMPTY, "Ld == 0 -> EEMPTY");
    CHECK(maxsim_score_dense(dQ, dD, dqm, ddm, 1, 2, 2, 2, 3, 77, MAXSIM_MASK_F32, dout, NULL) == MAXSIM_EINVAL, "unknown dtype -> EINVAL");
    CHECK(maxsim_score_dense(NULL, dD, dqm, ddm, 1, 2, 2, 2, 3, MAXSIM_F32, MAXSIM_MASK_F32, dout, NULL) == MAXSIM_EINVAL, "NULL Q -> EINVAL");
    CHECK(strlen(maxsim_strerror(MAXSIM_EEMPTY)) > 0, "strerror");
    /* which kernel an all-pairs shape gets (host logic only): the reference's training step -> the GEMM-blocked kernel */
    CHECK(maxsim_score_dense_kernel(272, 544, 32, 384, 768, MAXSIM_BF16, MAXSIM_MASK_F32) == 1, "dense_kernel: training step");
    CHECK(maxsim_score_dense_kernel(272, 544, 32, 384, 768, MAXSIM_F32, MAXSIM_MASK_F32) == 0, "dense_kernel: fp32 operands");
    CHECK(maxsim_score_dense_kernel(1, 10, 32, 180, 128, MAXSIM_F16, MAXSIM_MASK_NONE) == 0, "dense_kernel: too few tiles");
    CHECK(maxsim_score_dense_kernel(1, 10, 0, 180, 128, MAXSIM_F16, MAXSIM_MASK_NONE) == MAXSIM_EINVAL, "dense_kernel: Lq < 1");
  }

  /* (the reference's own KAT numbers, [[21, 41]], are checked from Python against the golden file) the zero-floor pair: */
  {
    const float Q[2] = {1, 0};
    const float D[4] = {-1, 0, -2, 0};
    const int64_t qm[1] = {1}, dm_full[2] = {1, 1}, dm_floor[2] = {1, 0};
    float *dQ = to_dev(Q, sizeof Q), *dD = to_dev(D, sizeof D);
    void *dqm = to_dev(qm, sizeof qm), *d1 = to_dev(dm_full, sizeof dm_full), *d2 = to_dev(dm_floor, sizeof dm_floor);
    float* dout;
    hipMalloc((void**)&dout, sizeof(float));
    float a = 9, b = 9;
    maxsim_score_dense(dQ, dD, dqm, d1, 1, 1, 1, 2, 2, MAXSIM_F32, MAXSIM_MASK_I64, dout, NULL);
    hipDeviceSynchronize();
    hipMemcpy(&a, dout, 4, hipMemcpyDeviceToHost);
    maxsim_score_dense(dQ, dD, dqm, d2, 1, 1, 1, 2, 2, MAXSIM_F32, MAXSIM_MASK_I64, dout, NULL);
    hipDeviceSynchronize();
    hipMemcpy(&b, dout, 4, hipMemcpyDeviceToHost);
    CHECK(a == -1.0f && b == 0.0f, "zero-floor pair: full mask -> -1, masked slot -> 0");
  }

  /* ragged rerank on a 3-doc index of one-hot tokens, dim 128 (fast path), + top-k */
  {
    enum { H = 128, NTOK = 6, LQ = 2 };
    static float index[NTOK * H], Q[LQ * H];
    /* doc 0 = tokens {e0, e1}, doc 1 = {e2}, doc 2 = {e0, e3, e1} (e_i = unit vector i) */
    const int tok_dim[NTOK] = {0, 1, 2, 0, 3, 1};
    for (int t = 0; t < NTOK; ++t) index[t * H + tok_dim[t]] = 1.0f;
    Q[0 * H + 0] = 1.0f;  /* query token 0 = e0 */
    Q[1 * H + 1] = 0.5f;  /* query token 1 = 0.5 e1 */
    const int64_t offs[3] = {0, 2, 3};
    const int32_t lens[3] = {2, 1, 3};
    const int64_t cand[4] = {1, 2, -1, 0};
    void *dI = to_dev(index, sizeof index), *dQ = to_dev(Q, sizeof Q), *doff = to_dev(offs, sizeof offs),
         *dlen = to_dev(lens, sizeof lens), *dc = to_dev(cand, sizeof cand);
    float* dsc;
    hipMalloc((void**)&dsc, 4 * sizeof(float));
    int rc = maxsim_rerank(dI, MAXSIM_F32, NTOK, (const int64_t*)doff, (const int32_t*)dlen, NULL, 3, dQ, MAXSIM_F32, NULL,
                           (const int64_t*)dc, 1, 4, LQ, H, dsc, NULL);
    float sc[4];
    hipDeviceSynchronize();
    hipMemcpy(sc, dsc, sizeof sc, hipMemcpyDeviceToHost);
    printf("     rerank -> [%g, %g, %g, %g] (rc %d)\n", sc[0], sc[1], sc[2], sc[3], rc);
    CHECK(rc == MAXSIM_OK && sc[0] == 0.0f && sc[1] == 1.5f && isinf(sc[2]) && sc[2] < 0 && sc[3] == 1.5f, "rerank scores");
    float* dts;
    int64_t* dtp;
    hipMalloc((void**)&dts, 3 * sizeof(float));
    hipMalloc((void**)&dtp, 3 * sizeof(int64_t));
    rc = maxsim_topk(dsc, (const int64_t*)dc, 1, 4, 3, dts, dtp, NULL);
    float ts[3];
    int64_t tp[3];
    hipDeviceSynchronize();
    hipMemcpy(ts, dts, sizeof ts, hipMemcpyDeviceToHost);
    hipMemcpy(tp, dtp, sizeof tp, hipMemcpyDeviceToHost);
    CHECK(rc == MAXSIM_OK && tp[0] == 2 && tp[1] == 0 && tp[2] == 1 && ts[0] == 1.5f && ts[2] == 0.0f, "topk: ties by list position");
    CHECK(maxsim_rerank(dI, MAXSIM_F32, NTOK, (const int64_t*)doff, (const int32_t*)dlen, NULL, 3, dQ, MAXSIM_F32, NULL,
                        (const int64_t*)dc, 1, 0, LQ, H, dsc, NULL) == MAXSIM_EEMPTY, "ncand == 0 -> EEMPTY (colbert_ranker.py:76)");

    /* the view-based entry points on the same index: packed doc table, query-token mask (keep_nonzero,
       training_utils.py:48-53), one-call rank_forward with pinned host buffers, the doc-shard candidate filter */
    void* dtab = NULL;
    hipMalloc(&dtab, (size_t)maxsim_doc_table_bytes(3));
    CHECK(maxsim_build_doc_table((const int64_t*)doff, (const int32_t*)dlen, NULL, 3, dtab, NULL) == MAXSIM_OK, "build_doc_table");
    maxsim_index_view iv;
    memset(&iv, 0, sizeof iv);
    iv.index = dI; iv.index_dtype = MAXSIM_F32; iv.h = H; iv.n_tokens = NTOK;
    iv.tok_offsets = (const int64_t*)doff; iv.doclens = (const int32_t*)dlen; iv.pad_len = NULL; iv.n_docs = 3;
    iv.doc_table = dtab;
    const uint8_t keep[LQ] = {1, 0};  /* drop query token 1: doc 0 -> 1, doc 2 -> 1, doc 1 -> 0 */
    void* dkeep = to_dev(keep, sizeof keep);
    rc = maxsim_rerank_ex(&iv, dQ, MAXSIM_F32, NULL, (const uint8_t*)dkeep, (const int64_t*)dc, 1, 4, LQ, dsc, NULL);
    hipDeviceSynchronize();
    hipMemcpy(sc, dsc, sizeof sc, hipMemcpyDeviceToHost);
    CHECK(rc == MAXSIM_OK && sc[0] == 0.0f && sc[1] == 1.0f && isinf(sc[2]) && sc[3] == 1.0f, "rerank_ex: doc table + q_mask");
    rc = maxsim_rerank_ex(&iv, dQ, MAXSIM_F32, NULL, NULL, (const int64_t*)dc, 1, 4, LQ, dsc, NULL);
    hipDeviceSynchronize();
    hipMemcpy(sc, dsc, sizeof sc, hipMemcpyDeviceToHost);
    CHECK(rc == MAXSIM_OK && sc[0] == 0.0f && sc[1] == 1.5f && sc[3] == 1.5f, "rerank_ex without mask = rerank");

    int64_t* hp = NULL;  /* pinned host: pids in; coherent pinned: top-k and completion word out */
    char* hout = NULL;
    hipHostMalloc((void**)&hp, 4 * sizeof(int64_t), hipHostMallocDefault);
    hipHostMalloc((void**)&hout, 256, hipHostMallocCoherent);
    const int64_t list[3] = {1, 2, 0};
    memcpy(hp, list, sizeof list);
    memset(hout, 0, 256);
    void* ws = NULL;
    hipMalloc(&ws, (size_t)maxsim_rank_forward_workspace_bytes(3));
    hipMemset(ws, 0, (size_t)maxsim_rank_forward_workspace_bytes(3));
    hipDeviceSynchronize();
    int64_t* op = (int64_t*)hout;
    float* os = (float*)(hout + 64);
    uint32_t* flag = (uint32_t*)(hout + 128);
    for (int rep = 0; rep < 3; ++rep) {  /* repeated calls on one workspace: its counter returns to zero */
      op[0] = op[1] = -7;
      rc = maxsim_rank_forward(&iv, dQ, MAXSIM_F32, LQ, hp, 3, 2, ws, op, os, flag, 1, NULL);
      CHECK(rc == MAXSIM_OK && op[0] == 2 && op[1] == 0 && os[0] == 1.5f && os[1] == 1.5f, "rank_forward: one call, pinned in/out, top-2");
    }
    CHECK(maxsim_rank_forward(&iv, dQ, MAXSIM_F32, LQ, hp, 0, 2, ws, op, os, flag, 1, NULL) == MAXSIM_EEMPTY, "rank_forward: no pids -> EEMPTY");

    const int64_t glob[6] = {7, 12, 10, 3, 11, 10};  /* this shard owns global pids [10, 13) */
    void* dg = to_dev(glob, sizeof glob);
    int64_t *dl = NULL, *dgo = NULL;
    int32_t* dcnt = NULL;
    hipMalloc((void**)&dl, sizeof glob); hipMalloc((void**)&dgo, sizeof glob); hipMalloc((void**)&dcnt, 4);
    rc = maxsim_shard_candidates((const int64_t*)dg, 1, 6, 10, 13, dl, dgo, dcnt, NULL);
    int64_t loc[6], gl[6];
    int32_t cnt = -1;
    hipDeviceSynchronize();
    hipMemcpy(loc, dl, sizeof loc, hipMemcpyDeviceToHost);
    hipMemcpy(gl, dgo, sizeof gl, hipMemcpyDeviceToHost);
    hipMemcpy(&cnt, dcnt, 4, hipMemcpyDeviceToHost);
    CHECK(rc == MAXSIM_OK && cnt == 4 && loc[0] == 2 && loc[1] == 0 && loc[2] == 1 && loc[3] == 0 && loc[4] == -1 && loc[5] == -1 &&
              gl[0] == 12 && gl[1] == 10 && gl[2] == 11 && gl[3] == 10 && gl[4] == -1, "shard_candidates: stable in-range compaction");

    /* counted rows: the compacted row (local pids 2 0 1 0, then -1 -1) with its device-side count through
       maxsim_rerank_counted + maxsim_topk_counted == maxsim_rerank_ex + maxsim_topk on the same row */
    const int64_t wlb = maxsim_worklist_bytes(1, 6);
    void* dwl = NULL;
    float *ds6 = NULL, *ds6c = NULL, *dts3 = NULL, *dtsc = NULL;
    int64_t *dtp3 = NULL, *dtpc = NULL;
    hipMalloc(&dwl, (size_t)wlb);
    hipMalloc((void**)&ds6, 24); hipMalloc((void**)&ds6c, 24); hipMalloc((void**)&dts3, 12); hipMalloc((void**)&dtsc, 12);
    hipMalloc((void**)&dtp3, 24); hipMalloc((void**)&dtpc, 24);
    rc = maxsim_rerank_ex(&iv, dQ, MAXSIM_F32, NULL, NULL, dl, 1, 6, LQ, ds6, NULL);
    int rc2 = maxsim_rerank_counted(&iv, dQ, MAXSIM_F32, NULL, NULL, dl, dcnt, 1, 6, LQ, ds6c, dwl, wlb, NULL);
    int rc3 = maxsim_topk(ds6, dgo, 1, 6, 3, dts3, dtp3, NULL);
    int rc4 = maxsim_topk_counted(ds6c, dgo, dcnt, 1, 6, 3, dtsc, dtpc, NULL);
    float s6[6], s6c[6], ts3[3], ts3c[3];
    int64_t tp3[3], tp3c[3];
    hipDeviceSynchronize();
    hipMemcpy(s6, ds6, 24, hipMemcpyDeviceToHost); hipMemcpy(s6c, ds6c, 24, hipMemcpyDeviceToHost);
    hipMemcpy(ts3, dts3, 12, hipMemcpyDeviceToHost); hipMemcpy(ts3c, dtsc, 12, hipMemcpyDeviceToHost);
    hipMemcpy(tp3, dtp3, 24, hipMemcpyDeviceToHost); hipMemcpy(tp3c, dtpc, 24, hipMemcpyDeviceToHost);
    CHECK(rc == MAXSIM_OK && rc2 == MAXSIM_OK && rc3 == MAXSIM_OK && rc4 == MAXSIM_OK && wlb > 0, "counted rows: return codes");
    CHECK(memcmp(s6, s6c, 24) == 0 && isinf(s6c[4]) && isinf(s6c[5]) && s6c[0] == 1.5f, "rerank_counted == rerank_ex (incl. the -inf tail)");
    CHECK(memcmp(ts3, ts3c, 12) == 0 && memcmp(tp3, tp3c, 24) == 0 && tp3c[0] == 12, "topk_counted == topk");

    /* ANN token rows -> distinct pids (colbert_ranker.py:212-229) through the row-block table, with a dropped query token
       and a shard's id_base: three docs of 2, 0, 3 rows (tok_offsets 0 2 2); 2 query tokens x 3 neighbours each */
    const int64_t offs3[3] = {0, 2, 2};
    const int64_t ids6[6] = {104, 100, -1, 101, 103, 999};   /* id_base 100 -> rows 4 0 . | 1 3 (999: another shard) */
    const uint8_t keep2[2] = {1, 1}, keep1[2] = {0, 1};
    void *doffs3 = to_dev(offs3, sizeof offs3), *dids6 = to_dev(ids6, sizeof ids6);
    void *dkeep2 = to_dev(keep2, 2), *dkeep1 = to_dev(keep1, 2);
    const int64_t rbb = maxsim_row_blocks_bytes(5);
    void* drb = NULL;
    int64_t* dout6 = NULL;
    int32_t* dn6 = NULL;
    hipMalloc(&drb, (size_t)rbb); hipMalloc((void**)&dout6, 48); hipMalloc((void**)&dn6, 4);
    rc = maxsim_build_row_blocks((const int64_t*)doffs3, 3, 5, drb, NULL);
    rc2 = maxsim_embedding_ids_to_pids_ex((const int64_t*)dids6, 1, 6, 3, (const uint8_t*)dkeep2, 100, (const int64_t*)doffs3, 3, 5,
                                          drb, dout6, dn6, NULL);
    int64_t out6[6];
    int32_t n6 = -1;
    hipDeviceSynchronize();
    hipMemcpy(out6, dout6, 48, hipMemcpyDeviceToHost); hipMemcpy(&n6, dn6, 4, hipMemcpyDeviceToHost);
    CHECK(rc == MAXSIM_OK && rc2 == MAXSIM_OK && rbb == 16 && n6 == 2 && out6[0] == 0 && out6[1] == 2 && out6[2] == -1 && out6[5] == -1,
          "embedding_ids_to_pids_ex: row blocks + id_base (the empty doc 1 owns no row)");
    rc2 = maxsim_embedding_ids_to_pids_ex((const int64_t*)dids6, 1, 6, 3, (const uint8_t*)dkeep1, 100, (const int64_t*)doffs3, 3, 5,
                                          NULL, dout6, dn6, NULL);
    hipDeviceSynchronize();
    hipMemcpy(out6, dout6, 48, hipMemcpyDeviceToHost); hipMemcpy(&n6, dn6, 4, hipMemcpyDeviceToHost);
    CHECK(rc2 == MAXSIM_OK && n6 == 2 && out6[0] == 0 && out6[1] == 2 && out6[2] == -1,
          "embedding_ids_to_pids_ex: first query token dropped, no table (rows 1 and 3 -> docs 0 and 2)");
    /* the struct-size guard of maxsim_index_view: this build's size is accepted (as is 0), anything else is refused */
    maxsim_index_view iv2 = iv;
    iv2.struct_size = (int32_t)sizeof(maxsim_index_view);
    rc = maxsim_rerank_ex(&iv2, dQ, MAXSIM_F32, NULL, NULL, dl, 1, 6, LQ, ds6, NULL);
    iv2.struct_size = (int32_t)sizeof(maxsim_index_view) - 8;
    rc2 = maxsim_rerank_ex(&iv2, dQ, MAXSIM_F32, NULL, NULL, dl, 1, 6, LQ, ds6, NULL);
    hipDeviceSynchronize();
    CHECK(rc == MAXSIM_OK && rc2 == MAXSIM_EINVAL && maxsim_index_view_bytes() == (int64_t)sizeof(maxsim_index_view),
          "index view: struct_size guard");
    CHECK(maxsim_embedding_ids_to_pids_ex((const int64_t*)dids6, 1, 6, 4, (const uint8_t*)dkeep2, 0, (const int64_t*)doffs3, 3, 5,
                                          NULL, dout6, dn6, NULL) == MAXSIM_EINVAL, "ids_per_token must divide n -> EINVAL");
  }
  printf("%s\n", failures ? "FAILED" : "ALL OK");
  return failures ? 1 : 0;
}
