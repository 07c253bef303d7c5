"""How many MFMA row tiles the 3x3x3 stride-1 convolutions of a scene execute per granularity of the offset mask:
per (64-row tile, offset) -- what the gather-once kernels did up to round 5 -- per (32-row sub-tile, offset) and per
(16-row sub-tile, offset), next to the algorithmic pair count.  Unit: 32-row MFMA tiles x offsets (one unit = the row half of
a 64-row tile at one offset); `pairs / 32` is the floor no output-stationary 32-row form can beat.
    python scripts/subtile_stats.py S|NS"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cnrma_amd import pipeline, synth
from cnrma_amd import sparse as S

dev = torch.device("cuda:0")
WL = sys.argv[1] if len(sys.argv) > 1 else "S"
V, C, H, W, dims, stride = synth.SHAPES[WL]
sc = synth.make_scene(WL, seed=0, boxes=3, device=dev, channels_last=True)
feat, proj, tsdf = sc["features"][:, 0], sc["projection"][:, 0], sc["tsdf"][0, 0].to(dev)
backbone, head = bench.build_model(C, dev)
cfg = pipeline.SceneConfig(dims, stride=stride, max_points=500000, sampler="device", sample_seed=0)
seen = []
uses = {}
orig = S.CoordSet.neighbours


def hook(self, out_set, k, off, method="auto"):
    nbr = orig(self, out_set, k, off, method)
    if k == 3 and self is out_set:
        if id(nbr) not in uses:
            seen.append((nbr, out_set))
            uses[id(nbr)] = 0
        uses[id(nbr)] += 1
    return nbr


S.CoordSet.neighbours = hook
with torch.no_grad():
    pipeline.forward_scene(cfg, backbone, head, feat, proj, tsdf, dense=False)
tot = {"pairs32": 0.0, 64: 0, 32: 0, 16: 0, "perm32": 0, "side32": 0}
for nbr, cs in seen:
    n = int(cs.n)
    nb = nbr[:n]
    nt = (n + 63) // 64
    pad = torch.full((nt * 64 - n, 27), -1, dtype=nb.dtype, device=dev)
    v = (torch.cat((nb, pad)) >= 0)
    pairs = int(v.sum().item())
    row = {}
    for g in (64, 32, 16):
        act = v.view(-1, g, 27).any(dim=1)                 # [sub-tiles][27]
        row[g] = int(act.sum().item()) * (g // 16)         # in 16-row units
    # what a permutation of the rows INSIDE each 64-row tile could buy (the tile's union is unchanged by it): rows sorted by how many
    # neighbours they have, the 32 fullest in one half -- a cheap stand-in for the best 2-way partition
    v64 = v.view(-1, 64, 27)
    order = torch.argsort(v64.sum(dim=2), dim=1, descending=True)
    vs = torch.gather(v64, 1, order.unsqueeze(-1).expand(-1, -1, 27))
    row["perm32"] = int((vs[:, :32].any(dim=1).sum() + vs[:, 32:].any(dim=1).sum()).item()) * 2
    # and by the direction they lack most neighbours in (sign of the mean missing offset along the axis of largest imbalance)
    offs = torch.tensor([[dx, dy, dz] for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1)], dtype=torch.float32, device=dev)
    miss = (~v64).float() @ offs                               # [tiles, 64, 3]: where a row's missing neighbours point
    axis = miss.abs().sum(dim=1).argmax(dim=1)                 # per tile: the axis along which the rows differ most
    key = torch.gather(miss, 2, axis.view(-1, 1, 1).expand(-1, 64, 1)).squeeze(-1)
    order = torch.argsort(key, dim=1)
    vs = torch.gather(v64, 1, order.unsqueeze(-1).expand(-1, -1, 27))
    row["side32"] = int((vs[:, :32].any(dim=1).sum() + vs[:, 32:].any(dim=1).sum()).item()) * 2
    u = uses[id(nbr)]
    print(f"rows={n:7d} stride={cs.stride:3d} used x{u}  pairs/16={pairs / 16:10.0f}  executed 16-row units: "
          f"mask64={row[64]:9d} ({row[64] * 16 / pairs:.2f}x)  mask32={row[32]:9d} ({row[32] * 16 / pairs:.2f}x)  "
          f"mask16={row[16]:9d} ({row[16] * 16 / pairs:.2f}x)   rows permuted inside the tile, mask32: fullest-first {row['perm32'] * 16 / pairs:.2f}x  "
          f"by side {row['side32'] * 16 / pairs:.2f}x   density={pairs / (n * 27):.3f}")
    tot["pairs32"] += u * pairs / 16
    for g in (64, 32, 16, "perm32", "side32"):
        tot[g] += u * row[g]
print(f"{WL}: weighted by uses (not by channels): mask64 {tot[64] / tot['pairs32']:.3f}x  mask32 {tot[32] / tot['pairs32']:.3f}x  "
      f"mask16 {tot[16] / tot['pairs32']:.3f}x of the pair count; mask32 with the rows permuted inside each tile: fullest-first "
      f"{tot['perm32'] / tot['pairs32']:.3f}x, by side {tot['side32'] / tot['pairs32']:.3f}x")
