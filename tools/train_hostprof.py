import os, sys, cProfile, pstats, time
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd"))
sys.argv = ["x", "--steps", "3"]
src = open(os.path.join(ROOT, "tools/train_bench.py")).read().replace('ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))', 'ROOT = "/root/repo"')
g = {"__name__": "__main__", "__file__": os.path.join(ROOT, "tools/train_bench.py")}
exec(compile(src, "train_bench", "exec"), g)
model, batches = g["model"], g["batches"]
import torch
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for i in range(5): model.train_emb(batches[i % 4])
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
t0=time.perf_counter()
for i in range(10): model.train_emb(batches[i % 4])
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print("host issue time per step %.2f ms, total %.2f ms" % ((t1-t0)*100, (t2-t0)*100))
