"""2D stack (ResNet-50 FPN + AtlasFPNFeature, torch / MIOpen) over one scene's views in NCHW and in torch.channels_last:
time per scene and the memory format of the maps it hands to the hot path (the saving of the
layout pass must not be paid back upstream)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import projects.mvsdetection  # noqa: F401
from projects.mvsdetection.registry import build_backbone
from test_backbone2d_cpu import CFG_FPN, CFG_HEAD

dev = torch.device("cuda:0")
V = int(sys.argv[1]) if len(sys.argv) > 1 else 40
x = torch.randn(V, 3, 480, 640, device=dev)
for fmt_name, fmt in (("NCHW", torch.contiguous_format), ("channels_last", torch.channels_last), ("NCHW", torch.contiguous_format),
                      ("channels_last", torch.channels_last)):
    torch.manual_seed(0)
    fpn, head = build_backbone(dict(CFG_FPN)).to(dev).eval(), build_backbone(dict(CFG_HEAD)).to(dev).eval()
    fpn.to(memory_format=fmt); head.to(memory_format=fmt)
    xi = x.contiguous(memory_format=fmt)
    with torch.no_grad():
        for _ in range(2):
            y = head(fpn(xi))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            y = head(fpn(xi))
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print(f"{fmt_name:14s} {ms:8.1f} ms per {V}-view scene; output {tuple(y.shape)} channels-last in memory: "
          f"{y.permute(0, 2, 3, 1).is_contiguous()}; checksum {float(y.double().abs().mean()):.6f}", flush=True)
