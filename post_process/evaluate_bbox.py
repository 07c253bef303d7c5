"""mAP@0.25 / mAP@0.5 of the NMS'ed boxes against {data_path}/{dataset}_instance_data/{scene}_aligned_bbox.npy.
CLI-compatible with the reference's post_process/evaluate_bbox.py."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnrma_amd import postprocess  # noqa: E402

SCANNET_CAT_IDS = [3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39]


def evaluate_bbox(dataset, data_path, result_path, postfix):
    cat2cls = {c: i for i, c in enumerate(SCANNET_CAT_IDS if dataset == "scannet" else range(17))}
    gts, dts = [], []
    for scene_id in sorted(os.listdir(result_path)):
        d = np.load(os.path.join(result_path, scene_id, scene_id + postfix + ".npz"))
        dts.append(dict(boxes=d["boxes"], scores=d["scores"], labels=d["labels"]))
        g = np.load(os.path.join(data_path, f"{dataset}_instance_data", scene_id + "_aligned_bbox.npy"))
        gts.append(dict(boxes=g[:, :-1], labels=np.array([cat2cls[int(c)] for c in g[:, -1]], dtype=np.int64)))
    res = postprocess.indoor_eval(gts, dts, (0.25, 0.5), n_classes=len(cat2cls))
    print({k: v for k, v in res.items() if not k.startswith("AP_")})
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", type=str, default="scannet")
    ap.add_argument("--data_path", type=str, required=True)
    ap.add_argument("--result_path", type=str, required=True)
    ap.add_argument("--postfix", type=str, default="_atlas_bbox")
    a = ap.parse_args()
    evaluate_bbox(a.dataset, a.data_path, a.result_path, a.postfix)
