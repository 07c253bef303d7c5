"""Voxel index grid (reference: projects/mvsdetection/datasets/tsdf.py:14-29).  The TSDF container class of the
reference (npz I/O, marching cubes) is out of the hot-path scope."""
import torch


def coordinates(voxel_dim, device=None):
    """int64 [3, nx*ny*nz] voxel indices, x slowest / z fastest.  The HIP dense kernel derives the same indices from
    the lane id and never materialises this tensor; it is kept for API compatibility."""
    nx, ny, nz = voxel_dim
    g = torch.arange(nx * ny * nz, dtype=torch.long, device=device)
    return torch.stack((g // (ny * nz), (g // nz) % ny, g % nz))
