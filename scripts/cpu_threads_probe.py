"""how do the oracle's legs scale with torch's thread count on this host?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnrma_amd import synth
from oracle import rma_oracle as O
nt = int(sys.argv[1])
torch.set_num_threads(nt)
t0 = time.time()
sc = synth.make_scene((1, 32, 120, 160, (192, 192, 80), 4), seed=0)
proj, feat, tsdf = sc["projection"][:, 0], sc["features"][:, 0], sc["tsdf"][0, 0]
print(nt, "threads; scene %.1fs" % (time.time() - t0), flush=True)
for r in range(2):
    t0 = time.time(); O.backproject_view(sc["dims"], 0.04, sc["origin"], O.scale_projection(proj[0], 4), feat[0]); t1 = time.time()
    pts = O.aggregate_rma(proj, feat, tsdf, sc["dims"], 0.04, sc["origin"], 4); t2 = time.time()
    print(nt, "dense %.2fs rma %.2fs" % (t1 - t0, t2 - t1), flush=True)
