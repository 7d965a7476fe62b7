"""On the GPU box: `sharded.load_shard` at a realistic size -- an index of PARTS part files of ~1.4 GB each (fp16, dim 128,
ragged docs) written in the reference's format to /tmp, then every shard of WORLD loaded one after another from the files
(memory-mapped: only the overlapping rows are read) and compared with the unsharded ColbertRanker(index_path=...).
Prints seconds and GB/s per shard.  env: PARTS (6), WORLD (4)"""
import os, shutil, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_amd
from colbert_amd.index_io import save_index
from colbert_amd.sharded import load_shard
parts_n, world = int(os.environ.get("PARTS", 6)), int(os.environ.get("WORLD", 4))
path = "/tmp/maxsim_probe_index"
shutil.rmtree(path, ignore_errors=True)
g = torch.Generator().manual_seed(3)
t0 = time.perf_counter()
parts, dls = [], []
for i in range(parts_n):
    dl = (torch.randn(46000, generator=g) * 40 + 120).round().clamp(8, 180).long().tolist()
    parts.append(F.normalize(torch.randn(sum(dl), 128, generator=g), dim=-1).half())
    dls.append(dl)
save_index(path, parts, dls)
nbytes = sum(p.numel() * 2 for p in parts)
print("wrote %d parts, %.2f GB in %.1f s" % (parts_n, nbytes / 1e9, time.perf_counter() - t0), flush=True)
del parts
t0 = time.perf_counter()
whole = colbert_amd.ColbertRanker(index_path=path, device="cuda:0")
torch.cuda.synchronize()
print("unsharded load: %.2f s (%.2f GB/s)" % (time.perf_counter() - t0, nbytes / 1e9 / (time.perf_counter() - t0)), flush=True)
offs = whole.doclens_pfxsum
ok = True
for r in range(world):
    t0 = time.perf_counter()
    sh = load_shard(path, r, world, device="cuda:0")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rows = sh.local.num_embeddings
    same = torch.equal(sh.local.tensor[:rows], whole.tensor[sh.tok_lo:sh.tok_hi]) and sh.local.strides == whole.strides \
        and torch.equal(sh.local.d_pad_len, whole.d_pad_len[sh.lo:sh.hi])
    ok = ok and same
    print("shard %d of %d: docs [%d, %d), %.2f GB in %.2f s (%.2f GB/s), equal to the unsharded rows: %s"
          % (r, world, sh.lo, sh.hi, rows * 256 / 1e9, dt, rows * 256 / 1e9 / dt, same), flush=True)
    del sh
shutil.rmtree(path, ignore_errors=True)
print("ALL EQUAL" if ok else "MISMATCH")
