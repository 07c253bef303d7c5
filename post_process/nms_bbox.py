"""Offline NMS over the raw boxes written by the detector ({scene}_bbox_raw.npz -> {scene}{postfix}).
CLI-compatible with the reference's post_process/nms_bbox.py (same arguments, same output keys boxes/scores/labels)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnrma_amd import postprocess  # noqa: E402


def nms_bboxes(args):
    for scene_id in sorted(os.listdir(args.result_path)):
        raw = np.load(os.path.join(args.result_path, scene_id, scene_id + "_bbox_raw.npz"))
        boxes, scores, labels = postprocess.nms(torch.tensor(raw["bboxes"]).cuda(), torch.tensor(raw["scores"]).cuda())
        np.savez(os.path.join(args.result_path, scene_id, scene_id + args.postfix),
                 boxes=postprocess.to_saved_layout(boxes).cpu().numpy(), scores=scores.cpu().numpy(),
                 labels=labels.cpu().numpy())
        print("Saved", scene_id)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--result_path", type=str, required=True)
    ap.add_argument("--postfix", type=str, default="_atlas_bbox.npz")
    nms_bboxes(ap.parse_args())
