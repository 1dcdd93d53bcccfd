"""CPU oracles for the small C-ABI entry points -- TEST INFRASTRUCTURE ONLY (same rules as raster_oracle.py).

PARITY UNPINNED for distCUDA2 and lietorch: both are un-vendored submodules of the reference
(/root/reference/.gitmodules:1-3,10-12; no sources, no tests).  The restatements follow their published definitions:
  knn_mean_dist2   simple-knn distCUDA2: mean squared distance to the 3 nearest neighbours (self excluded), exact.
                   Call site: /root/reference/thirdparty/gaussian_splatting/scene/gaussian_model.py:194-200.
  se3_*            SE3 group in (t, q=(x,y,z,w)) storage with tangent (rho, theta); exp / log through the closed forms
                   that /root/reference/thirdparty/monogs/utils/pose_utils.py:30-78 also uses (pinned by golden G2).
"""
import torch


def knn_mean_dist2(xyz):
    x = xyz.double()
    d2 = torch.cdist(x, x) ** 2
    d2.fill_diagonal_(float("inf"))
    k = min(3, x.shape[0] - 1)
    if k <= 0:
        return torch.zeros(x.shape[0], dtype=torch.float64)
    v = torch.topk(d2, k, dim=1, largest=False).values
    return v.sum(dim=1) / 3.0


def _hat(v):
    z = torch.zeros_like(v[..., 0])
    return torch.stack([torch.stack([z, -v[..., 2], v[..., 1]], -1), torch.stack([v[..., 2], z, -v[..., 0]], -1),
                        torch.stack([-v[..., 1], v[..., 0], z], -1)], -2)


def quat_to_R(q):
    x, y, z, w = q.unbind(-1)
    return torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
                        torch.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
                        torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)], -2)


def se3_matrix(pose):
    T = torch.eye(4, dtype=pose.dtype).repeat(*pose.shape[:-1], 1, 1)
    T[..., :3, :3] = quat_to_R(pose[..., 3:])
    T[..., :3, 3] = pose[..., :3]
    return T


def se3_exp_matrix(tau):
    """4x4 of exp(tau), tau = (rho, theta): pose_utils.py:30-78 in fp64."""
    rho, th = tau[..., :3], tau[..., 3:]
    W = _hat(th)
    W2 = W @ W
    a = th.norm(dim=-1)[..., None, None]
    I = torch.eye(3, dtype=tau.dtype).expand_as(W)
    small = a < 1e-8
    a_s = torch.where(small, torch.ones_like(a), a)
    R = torch.where(small, I + W + 0.5 * W2, I + torch.sin(a_s) / a_s * W + (1 - torch.cos(a_s)) / a_s ** 2 * W2)
    V = torch.where(small, I + 0.5 * W + W2 / 6, I + (1 - torch.cos(a_s)) / a_s ** 2 * W + (a_s - torch.sin(a_s)) / a_s ** 3 * W2)
    T = torch.eye(4, dtype=tau.dtype).repeat(*tau.shape[:-1], 1, 1)
    T[..., :3, :3] = R
    T[..., :3, 3] = (V @ rho[..., None])[..., 0]
    return T


def adjT(T, a):
    """(Ad_T)^T a for a = (a_rho, a_theta), Ad_T = [[R, [t]x R], [0, R]]."""
    R, t = T[..., :3, :3], T[..., :3, 3]
    ar, at = a[..., :3], a[..., 3:]
    Rt = R.transpose(-1, -2)
    o1 = (Rt @ ar[..., None])[..., 0]
    o2 = (Rt @ (at - torch.linalg.cross(t, ar))[..., None])[..., 0]
    return torch.cat([o1, o2], -1)
