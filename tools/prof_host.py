import cProfile, pstats, sys, os, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from zeroshotsemanticsegmentation_amd import engine, models, synth
E,K,H,B=300,59,512,1
emb=synth.make_embeddings(K,E)
m=models.FCN32s(E); m.load_synthetic(1337, device=torch.device("cuda")); m.train()
ts=engine.TrainStep(m, emb, precision=torch.bfloat16)
x=torch.from_numpy(synth.make_images(B,H,H)).cuda(); t=torch.from_numpy(synth.make_labels(B,H,H,K)).cuda()
for _ in range(5): ts.step(x,t)
torch.cuda.synchronize()
pr=cProfile.Profile(); pr.enable()
for _ in range(20): ts.step(x,t)
torch.cuda.synchronize()
pr.disable()
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
