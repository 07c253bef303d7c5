"""distribution of distinct neighbour rows per 128-row tile (feasibility of an LDS-resident halo)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from cnrma_amd import pipeline, rma, synth
from cnrma_amd import sparse as S
dev = torch.device("cuda:0")
WL = sys.argv[1] if len(sys.argv) > 1 else "S"
V, C, H, W, dims, stride = synth.SHAPES[WL]
sc = synth.make_scene(WL, seed=0, boxes=3, device=dev)
feat, proj, tsdf = sc["features"][:, 0].to(dev), sc["projection"][:, 0], sc["tsdf"][0, 0].to(dev)
backbone, head = bench.build_model(C, dev)
cfg = pipeline.SceneConfig(dims, stride=stride, max_points=500000, sampler="device")
seen = []
orig = S.CoordSet.neighbours
def hook(self, out_set, k, off, method="auto"):
    nbr = orig(self, out_set, k, off, method)
    if k == 3 and id(nbr) not in [id(x[0]) for x in seen]:
        seen.append((nbr, self.n, out_set.n, self.stride, out_set.stride))
    return nbr
S.CoordSet.neighbours = hook
pipeline.forward_scene(cfg, backbone, head, feat, proj, tsdf)
for nbr, nin, nout, s_in, s_out in seen:
    for BM in (64, 128):
        n = nbr.shape[0]
        nt = (n + BM - 1) // BM
        pad = torch.full((nt * BM - n, 27), -1, dtype=nbr.dtype, device=dev)
        t = torch.cat((nbr, pad)).view(nt, BM * 27)
        ts, _ = torch.sort(t, dim=1)
        uniq = ((ts[:, 1:] != ts[:, :-1]) & (ts[:, 1:] >= 0)).sum(dim=1) + (ts[:, 0] >= 0).long()
        pairs = (nbr >= 0).sum().item()
        q = torch.quantile(uniq.float(), torch.tensor([0.5, 0.9, 0.99, 1.0], device=dev)).tolist()
        print(f"in={nin:7d} out={nout:7d} s={s_in}->{s_out} BM={BM} density={pairs/(n*27):.2f} U median/p90/p99/max = {q}")
