import sys, time, torch, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/query-recommendation-system_amd")
import qrlsh
from qrlsh import ops, pipeline
nq=int(sys.argv[1]); D=32768
off, rows = qrlsh.synth_csr(nq, D, seed=0, device="cuda")
table = ops.perm_table(ops.legacy_permutations(128, D, seed=42), "cuda")
K = pipeline.max_candidates(nq)
for _ in range(5): res = pipeline.query_similarities(off, rows, table, 32, K, validate=False)
torch.cuda.synchronize(); t=time.perf_counter()
n = 100 if nq <= 2_000_000 else 60
for _ in range(n): res = pipeline.query_similarities(off, rows, table, 32, K, validate=False)
torch.cuda.synchronize(); print("nq", nq, "OVERLAP", os.environ.get("QRLSH_OVERLAP"), "GROUPS", os.environ.get("QRLSH_EMIT_GROUPS"), "ms/step %.3f" % ((time.perf_counter()-t)/n*1e3), res.pairs.numel())
