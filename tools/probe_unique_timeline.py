"""Timing build only (-DMAXSIM_DIAG -DMAXSIM_STAMP_UNIQUE, MAXSIM_UNIQUE_BLOCKS=0): where one workgroup of k_unique_pids spends its time."""
import os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_amd
from colbert_amd import _lib
dev = "cuda"
nd = 1000000
idx = torch.zeros(nd * 180, 8, dtype=torch.float16, device=dev)
r = colbert_amd.ColbertRanker.from_device_tensor(idx, [180] * nd)
g = torch.Generator(device=dev).manual_seed(0)
for nq in (1, 256):
    n, hot = 16384, 1500
    docs = torch.randint(0, nd, (nq, hot), generator=g, device=dev)
    ids = docs.gather(1, torch.randint(0, hot, (nq, n), generator=g, device=dev)) * 180 + torch.randint(0, 180, (nq, n), generator=g, device=dev)
    out = torch.empty(nq, n, dtype=torch.int64, device=dev)
    cnt = torch.zeros(nq + 64, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    rows = []
    for it in range(12):
        rc = _lib.lib.maxsim_embedding_ids_to_pids_ex(ids.data_ptr(), nq, n, 1, None, 0, r.d_offsets.data_ptr(), r.n_docs, r.num_embeddings,
                                                      r.d_row_blocks.data_ptr(), out.data_ptr(), cnt.data_ptr(), st)
        assert rc == 0
        torch.cuda.synchronize()
        s = cnt[(nq + 1) & ~1:][:12].view(torch.int64).tolist()
        rows.append([(s[i + 1] - s[i]) / 100.0 for i in range(5)])
    rows = rows[2:]
    med = [sorted(c)[len(c) // 2] for c in zip(*rows)]
    print(f"nq {nq}: us per phase of workgroup 0 (median of 10): init {med[0]:.2f} | lookups+inserts {med[1]:.2f} | compact {med[2]:.2f} | sort {med[3]:.2f} | write {med[4]:.2f} | total {sum(med):.2f}")
