"""What the two detectors of the plugin share (reference: the common halves of models/atlas.py:71-405 and
models/ray_marching.py:113-257, :547-682): the 2D feature extractor over all views, the dense unprojection-accumulate
state (`volume`, `valid`), reconstruction outputs, and the mmcv runner protocol (forward / train_step / val_step /
parse_losses / data_converter)."""
import os
from collections import OrderedDict

import torch
import torch.distributed as dist
from torch import nn

from cnrma_amd import rma

from ..datasets.tsdf import TSDF
from ..registry import build_backbone, build_head


class MultiViewBase(nn.Module):
    # The 2D feature extractor runs in torch.channels_last: its output [V,C,H',W'] is then channels-last IN MEMORY, which is
    # the layout the aggregation kernels gather from (one 128-byte line per pixel and 32 channels) -- the hot path reads it
    # in place (cnrma_amd.rma.is_channels_last), no NCHW -> NHWC pass (25 GB of traffic at the north-star shape).  The
    # tensor's logical shape and every value are unchanged (reference: backbone2d.py:28-67 / ray_marching.py:211-213).
    channels_last_2d = True

    def __init__(self, pixel_mean, pixel_std, voxel_size, n_scales, voxel_dim_train, voxel_dim_test, origin,
                 backbone2d_stride, backbone2d, feature_2d, backbone_3d, tsdf_head, save_path):
        super().__init__()
        self.fp16_enabled = False
        # a sub-network given as None is not built: its OUTPUT then comes in as an input (`features`, `tsdf`) -- that
        # is how the hot path is measured, with the 2D / 3D CNN results resident in HBM
        self.fpn = build_backbone(backbone2d) if backbone2d is not None else None
        self.feature_2d = build_backbone(feature_2d) if feature_2d is not None else None
        self.backbone3d = build_backbone(backbone_3d) if backbone_3d is not None else None
        self.tsdf_head = build_head(tsdf_head) if tsdf_head is not None else None
        if self.channels_last_2d:
            for m in (self.fpn, self.feature_2d):
                if m is not None:
                    m.to(memory_format=torch.channels_last)
        self.pixel_mean = torch.Tensor(pixel_mean).view(-1, 1, 1)
        self.pixel_std = torch.Tensor(pixel_std).view(-1, 1, 1)
        self.voxel_size, self.n_scales = voxel_size, n_scales
        self.voxel_dim_train, self.voxel_dim_test = voxel_dim_train, voxel_dim_test
        self.voxel_dim = voxel_dim_test
        self.save_path = save_path
        if save_path is not None:
            os.makedirs(save_path, exist_ok=True)
        self.origin = torch.tensor(origin, dtype=torch.float32).view(1, 3)
        self.backbone2d_stride = backbone2d_stride

    # ---- state (reference ray_marching.py:200-209) --------------------------------------------------------------------
    def initialize_volume(self):
        self.volume = 0
        self.valid = 0
        self._views = []          # (projection [B,3,4], feature [B,C,H,W]) collected by aggregate_2d_features

    def normalizer(self, x):
        return (x - self.pixel_mean.type_as(x)) / self.pixel_std.type_as(x)

    def backbone2d(self, image):
        if not self.channels_last_2d:
            return self.feature_2d(self.fpn(image))
        y = self.feature_2d(self.fpn(image.contiguous(memory_format=torch.channels_last)))
        return y.contiguous(memory_format=torch.channels_last)      # a no-op when the network kept the format (it does)

    def init_weights(self):
        """called by the reference's train.py:219; every sub-module initialises itself at construction"""

    def _features(self, inputs, batched):
        """feature maps of all views [V,B,C,H',W']: through the 2D network (all views in one batch when the BatchNorm
        statistics are to be shared, else view by view), or taken from the inputs when there is no 2D network"""
        if self.fpn is None:
            f = inputs["features"]
            if isinstance(f, (list, tuple)):       # one sample per GPU (the reference's structural limit): a view, no 12.6-GB stack copy
                return f[0].unsqueeze(1) if len(f) == 1 else torch.stack(f, dim=1)
            return f
        images = inputs["imgs"].transpose(0, 1)
        if batched:
            x = self.backbone2d(self.normalizer(images.reshape(-1, *images.shape[2:])))
            return x.view(images.shape[0], images.shape[1], *x.shape[1:])
        return torch.stack([self.backbone2d(self.normalizer(im)) for im in images], dim=0)

    # ---- dense unprojection (reference ray_marching.py:220-257 / atlas.py:120-153) ---------------------------------------
    def aggregate_2d_features(self, projection, feature):
        """Collect one view.  The reference adds a full C x G volume per call; here the views are only recorded and
        clear_3d_features() runs ONE kernel over all of them (sum in view order + mean), which is bit-identical."""
        self._views.append((projection, feature))

    def _whole(self, views_of, hint):
        """the [V, ...] tensor the per-view slices handed to aggregate_2d_features were cut from -- `hint`, when the slices
        are exactly its rows (the detectors' own loop): no re-stacking copy (12.6 GB at the north-star shape)"""
        if hint is not None and hint.shape[0] == len(views_of) and tuple(hint.shape[1:]) == tuple(views_of[0].shape) and \
                all(v.data_ptr() == hint[i].data_ptr() and v.stride() == hint[i].stride() for i, v in enumerate(views_of)):
            return hint
        return torch.stack(views_of, dim=0)

    def clear_3d_features(self):
        hint = getattr(self, "_view_source", (None, None))
        projs = self._whole([p for p, _ in self._views], hint[0])      # [V,B,3,4]
        feats = self._whole([f for _, f in self._views], hint[1])      # [V,B,C,H,W]
        self._view_source = (None, None)
        projs_cpu = projs.detach().cpu()                               # ONE device->host copy for all views
        vols, valids = [], []
        org = self.origin.view(-1).tolist()
        self._nhwc = {}                # channels-last copies of this scene's feature maps, shared with the ray marching
        for b in range(feats.shape[1]):
            if torch.is_grad_enabled() and feats.requires_grad:      # training: gradient of the volume -> feature maps
                vol, cnt = rma.BackprojectAccum.apply(feats[:, b], projs_cpu[:, b], self.voxel_dim, self.voxel_size, org,
                                                      self.backbone2d_stride)
            else:
                nhwc = rma.to_nhwc(feats[:, b])
                self._nhwc[(feats.data_ptr(), b)] = nhwc
                vol, cnt = rma.backproject_accum(nhwc, projs_cpu[:, b], self.voxel_dim, self.voxel_size, org,
                                                 self.backbone2d_stride)
            vols.append(vol)
            valids.append((cnt > 0).unsqueeze(0))
        self.volume = torch.stack(vols)
        self.valid = torch.stack(valids)
        self._views = []

    # ---- reconstruction outputs (reference ray_marching.py:500-545 / atlas.py:232-268) ------------------------------------
    def post_process(self, outputs, inputs):
        outs = []
        for i, vol in enumerate(outputs["scene_tsdf_004"]):
            tsdf = TSDF(self.voxel_size, self.origin, vol.squeeze(0))
            tsdf.origin = inputs["offset"][i].view(1, 3)
            outs.append(dict(scene=inputs["scene"][i], scene_tsdf=tsdf))
        return outs

    def save_reconstruction(self, outputs, inputs):
        """{save_path}/{scene}/{scene}.npz for every sample, + {scene}.ply when scikit-image / trimesh are installed"""
        if "offset" not in inputs or "scene" not in inputs:
            return []
        results = self.post_process(outputs, inputs)
        for r in results:
            d = os.path.join(self.save_path, r["scene"])
            os.makedirs(d, exist_ok=True)
            r["scene_tsdf"].save(os.path.join(d, r["scene"] + ".npz"))
            try:
                r["scene_tsdf"].get_mesh().export(os.path.join(d, r["scene"] + ".ply"))
            except ImportError:
                pass
        return results

    # ---- runner protocol (reference ray_marching.py:547-682) --------------------------------------------------------------
    def forward(self, return_loss=True, rescale=False, **kwargs):
        if return_loss:
            return self.forward_train(kwargs)
        return self.forward_test(self.data_converter(kwargs))

    def data_converter(self, data):
        """stack the per-sample lists of the DataContainer scatter (reference :653-682)"""
        for key in ("imgs", "projection", "offset", "axis_align_matrix"):
            if key in data and isinstance(data[key], (list, tuple)):
                data[key] = torch.stack(list(data[key]), dim=0)
        if "tsdf_dict" in data:
            names = list(data["tsdf_dict"][0].keys())
            dev = data["projection"].device
            data["tsdf_list"] = {n: torch.stack([d[n].tsdf_vol.unsqueeze(0) for d in data["tsdf_dict"]], 0).to(dev)
                                 for n in names}
            data.pop("tsdf_dict")
        data.pop("axis_align_matrix", None)
        return data

    def parse_losses(self, losses):
        log_vars = OrderedDict()
        for name, value in losses.items():
            if isinstance(value, torch.Tensor):
                log_vars[name] = value.mean()
            elif isinstance(value, list):
                log_vars[name] = sum(v.mean() for v in value)
            else:
                raise TypeError(f"{name} is not a tensor or list of tensors")
        loss = sum(v for k, v in log_vars.items() if "loss" in k)
        log_vars["total_loss"] = loss
        for name, value in log_vars.items():
            if dist.is_available() and dist.is_initialized():
                value = value.data.clone()
                dist.all_reduce(value.div_(dist.get_world_size()))
            log_vars[name] = value.item() if isinstance(value, torch.Tensor) else float(value)
        return loss, log_vars

    def train_step(self, data, optimizer):
        data = self.data_converter(data)
        loss, log_vars = self.parse_losses(self(**data))
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data["projection"]))

    def val_step(self, data, optimizer=None):
        return self(**data, return_loss=False)
