"""TSDF container + voxel index grid (reference: projects/mvsdetection/datasets/tsdf.py:14-180).

`coordinates` is the dense voxel index list of the reference (:14-29; the HIP dense kernel derives the same indices from
the lane id and never materialises it).  `TSDF` holds a truncated signed distance volume with the metadata needed to
interpret it -- npz I/O in the reference's file format, device moves, and the resampling under a rigid transform that the
space-transform pipelines use (:112-180).  Mesh extraction needs scikit-image + trimesh (third-party, absent in this
image): get_mesh() raises ImportError without them and the detectors then skip the .ply export only."""
import numpy as np
import torch


def coordinates(voxel_dim, device=None):
    """int64 [3, nx*ny*nz] voxel indices, x slowest / z fastest."""
    nx, ny, nz = voxel_dim
    g = torch.arange(nx * ny * nz, dtype=torch.long, device=device)
    return torch.stack((g // (ny * nz), (g // nz) % ny, g % nz))


class TSDF:
    def __init__(self, voxel_size, origin, tsdf_vol):
        """voxel_size in metres, origin [1,3] = position of voxel (0,0,0), tsdf_vol [nx,ny,nz]"""
        self.voxel_size = voxel_size
        self.origin = origin
        self.tsdf_vol = tsdf_vol
        self.device = tsdf_vol.device

    def save(self, fname):
        np.savez_compressed(fname, origin=self.origin.cpu().numpy(), voxel_size=self.voxel_size,
                            tsdf=self.tsdf_vol.detach().cpu().numpy())

    @classmethod
    def load(cls, fname):
        with np.load(fname) as z:
            return cls(z["voxel_size"].item(), torch.as_tensor(z["origin"]).view(1, 3), torch.as_tensor(z["tsdf"]))

    def to(self, device):
        self.origin = self.origin.to(device)
        self.tsdf_vol = self.tsdf_vol.to(device)
        self.device = device
        return self

    def get_mesh(self):
        """marching cubes on the sign-flipped volume (unobserved voxels -> outside); needs scikit-image and trimesh"""
        from skimage import measure            # noqa: F401  (ImportError when absent: callers skip the mesh)
        import trimesh
        vol = -self.tsdf_vol.detach().clone()
        vol[vol == -1] = 1
        vol = vol.clamp(-1, 1).cpu().numpy()
        if vol.min() >= 0 or vol.max() <= 0:
            return trimesh.Trimesh(vertices=np.zeros((0, 3)))
        mc = getattr(measure, "marching_cubes", None) or measure.marching_cubes_lewiner
        verts, faces, norms, _ = mc(vol, level=0)
        return trimesh.Trimesh(vertices=verts * self.voxel_size + self.origin.cpu().numpy(), faces=faces, vertex_normals=norms)

    def transform(self, transform=None, voxel_dim=None, origin=None, align_corners=False):
        """resample under the rigid map `transform` (new world -> old world, 4x4 or 3x4) into a volume of `voxel_dim`
        voxels whose voxel (0,0,0) sits at `origin`: nearest sample everywhere, trilinear where the nearest sample is
        inside the truncation band (|tsdf| < 1), +1 outside the old volume (reference :112-180).  Kept as the reference
        has it: the grid is normalised the align_corners=True way, 2 i / (n - 1) - 1, while grid_sample runs with
        align_corners=False, so the old volume is read half a voxel low and stretched by n / (n - 1)."""
        dev = self.tsdf_vol.device
        old_dim = list(self.tsdf_vol.shape)
        transform = torch.eye(4, device=dev) if transform is None else transform.to(dev)
        voxel_dim = old_dim if voxel_dim is None else list(voxel_dim)
        origin = self.origin if origin is None else torch.tensor(origin, dtype=torch.float, device=dev).view(1, 3)
        world = coordinates(voxel_dim, dev).float() * self.voxel_size + origin.T
        world = transform[:3, :] @ torch.cat((world, torch.ones_like(world[:1])), dim=0)
        g = (world - self.origin.to(dev).T) / self.voxel_size
        g = 2 * g / (torch.tensor(old_dim, device=dev) - 1).view(3, 1) - 1           # [-1, 1] over the old volume
        grid = g[[2, 1, 0]].T.view([1] + voxel_dim + [3])                             # grid_sample wants (z, y, x) last
        src = self.tsdf_vol.view([1, 1] + old_dim)
        near = torch.nn.functional.grid_sample(src, grid, mode="nearest", align_corners=align_corners).squeeze()
        lin = torch.nn.functional.grid_sample(src, grid, mode="bilinear", align_corners=align_corners).squeeze()
        band = near.abs() < 1
        near[band] = lin[band]
        near[(grid.abs() >= 1).squeeze(0).any(3)] = 1
        return TSDF(self.voxel_size, origin, near)
