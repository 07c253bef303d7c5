"""Stand-in for mmcv.parallel.DataContainer (used by AtlasCollectData, reference atlas_transforms.py:38-56): a tagged
box around one sample's field.  With mmcv installed the real class is used, so mmcv's collate / scatter see what they
expect; without it `collate` below does what the reference's batch-size-1 loaders need (stack tensors, list the rest)."""
try:
    from mmcv.parallel import DataContainer            # noqa: F401
    HAVE_MMCV = True
except Exception:
    HAVE_MMCV = False

    class DataContainer:
        def __init__(self, data, stack=False, padding_value=0, cpu_only=False, pad_dims=2):
            self._data, self.stack, self.cpu_only = data, stack, cpu_only

        @property
        def data(self):
            return self._data

        def __repr__(self):
            return f"DataContainer({self._data!r})"


def collate(samples, device=None):
    """list of AtlasCollectData results -> the kwargs of model(return_loss=..., **batch): per key a LIST with one entry
    per sample (what mmcv's scatter hands the model for non-stacked containers); tensors moved to `device` unless the
    container is cpu_only"""
    import torch
    out = {}
    for key in samples[0]:
        vals = []
        for s in samples:
            v = s[key]
            cpu_only = getattr(v, "cpu_only", False)
            v = v.data if hasattr(v, "data") and not isinstance(v, torch.Tensor) else v
            if isinstance(v, torch.Tensor) and device is not None and not cpu_only:
                v = v.to(device)
            vals.append(v)
        out[key] = vals
    return out
