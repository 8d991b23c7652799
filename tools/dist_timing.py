import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "query-recommendation-system_amd"))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29577")
torch.cuda.set_device(0); dev=torch.device("cuda",0)
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
import qrlsh
from qrlsh import ops, pipeline, dist as qd
nq=1_000_000; P=128; b=32; D=32768; K=34
off,rows=qrlsh.synth_csr(nq,D,seed=0,device=dev)
table=ops.perm_table(ops.legacy_permutations(P,D,seed=42),dev)
# monkeypatch timing points
marks=[]
orig_a2a=qd._all_to_all; orig_ag=qd._all_gather; orig_ex=qd._exchange_var
def T(name):
    torch.cuda.synchronize(); marks.append((name,time.perf_counter()))
class B(qd.HipBackend):
    def minhash(self,*a): T("start"); r=super().minhash(*a); T("minhash"); return r
    def emit_pairs(self,*a): T("pre-emit"); r=super().emit_pairs(*a); T("emit"); return r
    def sort_words(self,*a): r=super().sort_words(*a); T("sort_words"); return r
    def sort_unique(self,*a): r=super().sort_unique(*a); T("sort_unique"); return r
    def score_only(self,*a): T("pre-score"); r=super().score_only(*a); T("score"); return r
    def topk(self,*a): T("pre-topk"); r=super().topk(*a); T("topk"); return r
for it in range(3):
    marks.clear()
    res=qd.query_similarities_sharded(off,rows,table,b,K,nq,backend=B())
    T("end")
for (n0,t0),(n1,t1) in zip(marks[:-1],marks[1:]): print("%-12s -> %-12s %.3f ms"%(n0,n1,(t1-t0)*1e3))
print("total %.3f ms"%((marks[-1][1]-marks[0][1])*1e3))
dist.destroy_process_group()
