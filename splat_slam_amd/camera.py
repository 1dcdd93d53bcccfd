"""Camera of the mapping loop -- mirror of /root/reference/thirdparty/monogs/utils/camera_utils.py:13-148 and of the
matrix helpers in /root/reference/thirdparty/gaussian_splatting/utils/graphics_utils.py:33-46,72-101.

Same attribute names and conventions (W2C as R,T; transposed matrices; tau deltas; exposure a,b).  The reference
re-inverts a 4x4 twice per property access (graphics_utils.py:41-45, camera_utils.py:94-108: 4 linalg.inv launches per
render); here the three matrices are computed the same way ONCE per update_RT and cached -- bit-identical values, no
launches on the hot path.
"""
import math

import torch
from torch import nn


def getWorld2View2(R, t, translate=None, scale=1.0):
    """World-to-camera 4x4 of graphics_utils.py:33-46: assembled from (R, t), taken to camera-to-world space where the
    camera centre may be shifted / scaled, and inverted back.  The reference only calls it with the defaults, where the
    result is [R|t] up to the rounding of the two inversions -- which are performed (not short-circuited), so the
    matrices equal the reference's bit for bit on the same device.  (Cached by Camera: no launches on the hot path.)"""
    w2c = torch.eye(4, device=R.device, dtype=torch.float32)
    w2c[:3, :3] = R
    w2c[:3, 3] = t
    c2w = torch.linalg.inv(w2c)
    if translate is not None or scale != 1.0:
        centre = c2w[:3, 3] if translate is None else c2w[:3, 3] + translate.to(R.device)
        c2w[:3, 3] = centre * scale
    return torch.linalg.inv(c2w)


def getProjectionMatrix2(znear, zfar, cx, cy, fx, fy, W, H):
    """Pinhole projection with principal-point offset (graphics_utils.py:72-93), row-major, w_clip = z_view.
    The frustum edges on the near plane are those of the pixel rectangle [0, W] x [0, H] shifted by (cx, cy)."""
    edge = lambda c, n, f, sign: znear / f * ((((2 * c - n) / n) + sign) * n / 2.0)
    left, right = edge(cx, W, fx, -1.0), edge(cx, W, fx, 1.0)
    bottom, top = edge(cy, H, fy, -1.0), edge(cy, H, fy, 1.0)
    P = torch.zeros(4, 4)
    P[0, 0], P[0, 2] = 2.0 * znear / (right - left), (right + left) / (right - left)
    P[1, 1], P[1, 2] = 2.0 * znear / (top - bottom), (top + bottom) / (top - bottom)
    P[2, 2], P[2, 3] = zfar / (zfar - znear), -(zfar * znear) / (zfar - znear)
    P[3, 2] = 1.0
    return P


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


class Camera(nn.Module):
    def __init__(self, uid, color, depth, gt_T, projection_matrix, fx, fy, cx, cy, fovx, fovy, image_height,
                 image_width, device="cuda:0"):
        super().__init__()
        self.uid = uid
        self.device = device
        T = torch.eye(4, device=device)
        self.R = T[:3, :3]
        self.T = T[:3, 3]
        self.R_gt = gt_T[:3, :3]
        self.T_gt = gt_T[:3, 3]
        self.original_image = color
        self.depth = depth              # reference keeps numpy here (slam_utils.py:87-89); a device tensor also works
        self.grad_mask = None
        self.fx, self.fy, self.cx, self.cy = fx, fy, cx, cy
        self.FoVx, self.FoVy = fovx, fovy
        self.image_height, self.image_width = image_height, image_width
        self.cam_rot_delta = nn.Parameter(torch.zeros(3, requires_grad=True, device=device))
        self.cam_trans_delta = nn.Parameter(torch.zeros(3, requires_grad=True, device=device))
        # (zeros(): a fill on the device; tensor([0.0], device=...) is a pageable host-to-device copy that waits for the stream)
        self.exposure_a = nn.Parameter(torch.zeros(1, requires_grad=True, device=device))
        self.exposure_b = nn.Parameter(torch.zeros(1, requires_grad=True, device=device))
        self.projection_matrix = projection_matrix.to(device=device).contiguous()   # raw pointers are handed to the C ABI
        self._cache = None
        self._version = 0

    @staticmethod
    def init_from_dataset(dataset, data, projection_matrix):
        """camera_utils.py:74-92."""
        return Camera(data["idx"], data["gt_color"], data["glorie_depth"], data["glorie_pose"], projection_matrix,
                      dataset.fx, dataset.fy, dataset.cx, dataset.cy, dataset.fovx, dataset.fovy, dataset.H_out,
                      dataset.W_out, device=dataset.device)

    def _matrices(self):
        if self._cache is None:
            with torch.no_grad():
                w2c = getWorld2View2(self.R, self.T)
                view = w2c.transpose(0, 1).contiguous()
                full = (view.unsqueeze(0).bmm(self.projection_matrix.unsqueeze(0))).squeeze(0).contiguous()
                center = torch.linalg.inv(view)[3, :3].contiguous()      # camera_utils.py:106-108
            self._cache = (view, full, center)
        return self._cache

    @property
    def world_view_transform(self):
        return self._matrices()[0]

    @property
    def full_proj_transform(self):
        return self._matrices()[1]

    @property
    def camera_center(self):
        return self._matrices()[2]

    def update_RT(self, R, t):
        self.R = R.to(device=self.device)
        self.T = t.to(device=self.device)
        self._cache = None
        self._version = getattr(self, "_version", 0) + 1

    def clean(self):
        self.original_image = None
        self.depth = None
        self.grad_mask = None
        self.cam_rot_delta = None
        self.cam_trans_delta = None
        self.exposure_a = None
        self.exposure_b = None
