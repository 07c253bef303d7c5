"""Ground-truth boxes in the depth convention of mmdet3d 0.15's DepthInstance3DBoxes, for use WITHOUT mmdet3d (absent
in this image): the attributes the FCAF3D assigner / losses read (tensor [m,7] = (x, y, z_bottom, dx, dy, dz, yaw),
gravity_center, volume, corners) and the in-place augmentations TransformFeaturesBBoxes and AtlasTransformSpaceDetection
call (flip / rotate / scale / translate: fcaf3d_transforms.py:128-146, :258).  Third-party semantics restated from the
published mmdet3d source -- parity unpinned, like the other mmdet3d pieces."""
import torch


class GTBoxes:
    def __init__(self, tensor, box_dim=None, with_yaw=None, origin=(0.5, 0.5, 0.0)):
        t = torch.as_tensor(tensor, dtype=torch.float32).reshape(-1, tensor.shape[-1] if hasattr(tensor, "shape") and len(tensor.shape) > 1 else 7).clone()
        if t.shape[1] == 6:
            t = torch.cat((t, t.new_zeros(len(t), 1)), dim=1)
            with_yaw = False if with_yaw is None else with_yaw
        self.with_yaw = True if with_yaw is None else bool(with_yaw)
        if tuple(origin) != (0.5, 0.5, 0.0):          # e.g. (.5,.5,.5): centres given at the gravity centre
            t[:, :3] += t[:, 3:6] * (t.new_tensor((0.5, 0.5, 0.0)) - t.new_tensor(origin))
        self.tensor = t

    # ---- what the assigner / losses read ---------------------------------------------------------------------------
    @property
    def gravity_center(self):
        c = self.tensor[:, :3].clone()
        c[:, 2] += self.tensor[:, 5] / 2
        return c

    @property
    def volume(self):
        return self.tensor[:, 3] * self.tensor[:, 4] * self.tensor[:, 5]

    @property
    def corners(self):
        """[m,8,3]: corner (i,j,k) in {0,1}^3 order x-major, rotated by yaw about z around the bottom centre"""
        t = self.tensor
        unit = torch.tensor([[i, j, k] for i in (0, 1) for j in (0, 1) for k in (0, 1)], dtype=t.dtype, device=t.device)
        rel = (unit - t.new_tensor((0.5, 0.5, 0.0))) * t[:, None, 3:6]
        c, s = torch.cos(t[:, 6]), torch.sin(t[:, 6])
        x = rel[..., 0] * c[:, None] - rel[..., 1] * s[:, None]
        y = rel[..., 0] * s[:, None] + rel[..., 1] * c[:, None]
        return torch.stack((x, y, rel[..., 2]), dim=-1) + t[:, None, :3]

    def __len__(self):
        return len(self.tensor)

    def to(self, device):
        out = GTBoxes(self.tensor.to(device))
        out.with_yaw = self.with_yaw
        return out

    def clone(self):
        out = GTBoxes(self.tensor.clone())
        out.with_yaw = self.with_yaw
        return out

    # ---- augmentations (in place, like mmdet3d) ----------------------------------------------------------------------
    def translate(self, vec):
        self.tensor[:, :3] += torch.as_tensor(vec, dtype=self.tensor.dtype, device=self.tensor.device).view(-1)[:3]

    def scale(self, factor):
        self.tensor[:, :6] *= factor

    def flip(self, direction="horizontal"):
        assert direction in ("horizontal", "vertical")
        if direction == "horizontal":
            self.tensor[:, 0] = -self.tensor[:, 0]
            if self.with_yaw:
                self.tensor[:, 6] = -self.tensor[:, 6] + torch.pi
        else:
            self.tensor[:, 1] = -self.tensor[:, 1]
            if self.with_yaw:
                self.tensor[:, 6] = -self.tensor[:, 6]

    def rotate(self, angle):
        """rotate about +z with the matrix the point transform uses (fcaf3d_transforms.py:152-167); boxes without a yaw
        stay axis-aligned: their footprint becomes the bounding rectangle of the rotated one"""
        a = torch.as_tensor(angle, dtype=self.tensor.dtype)
        c, s = torch.cos(a), torch.sin(a)
        rot_t = torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]], dtype=self.tensor.dtype).T.to(self.tensor.device)
        if self.with_yaw:
            self.tensor[:, :3] = self.tensor[:, :3] @ rot_t
            self.tensor[:, 6] -= a
        else:
            corners = self.corners @ rot_t
            self.tensor[:, :3] = self.tensor[:, :3] @ rot_t
            self.tensor[:, 3] = corners[..., 0].max(dim=1)[0] - corners[..., 0].min(dim=1)[0]
            self.tensor[:, 4] = corners[..., 1].max(dim=1)[0] - corners[..., 1].min(dim=1)[0]
