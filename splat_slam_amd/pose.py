"""Pose update of the mapping loop.  Same contract as /root/reference/thirdparty/monogs/utils/pose_utils.py:66-98
(`SE3_exp(tau)` with tau = [rho, theta], left perturbation `W2C <- exp(tau) @ W2C`, deltas zeroed afterwards), written
as one batched closed form: for w = theta, K = [w]x, t = |w|,

    R = I + a K + b K^2        V = I + b K + c K^2        a = sin t / t,  b = (1 - cos t) / t^2,  c = (t - sin t) / t^3

with the coefficients replaced by their limits (1, 1/2, 1/6) below t = 1e-5 -- the same switch-over point as the
reference (:38-45, :55-62), so golden vectors G2 / G3 (tests/test_golden_host.py) hold on both sides of it.
The device path of the same map is the C ABI's se3_exp / se3_matrix (include/splat_hip.h), checked against the same
vectors in tests/test_gpu_aux.py.
"""
import torch

_SMALL_ANGLE = 1e-5


def _cross_matrix(w):
    """[..., 3] -> [..., 3, 3] with  _cross_matrix(w) @ x == cross(w, x)."""
    eye = torch.eye(3, device=w.device, dtype=w.dtype)
    return torch.linalg.cross(eye.expand(*w.shape[:-1], 3, 3), w.unsqueeze(-2).expand(*w.shape[:-1], 3, 3), dim=-1)


def _rodrigues_coefficients(t):
    """(a, b, c) of the module docstring for angles t (any shape), limits below the switch-over."""
    small = t < _SMALL_ANGLE
    ts = torch.where(small, torch.ones_like(t), t)          # keep the unused branch finite
    a = torch.where(small, torch.ones_like(t), torch.sin(ts) / ts)
    b = torch.where(small, torch.full_like(t, 0.5), (1.0 - torch.cos(ts)) / (ts * ts))
    c = torch.where(small, torch.full_like(t, 1.0 / 6.0), (ts - torch.sin(ts)) / (ts * ts * ts))
    return a, b, c


def SE3_exp(tau):
    """tau [..., 6] = (rho, theta) -> homogeneous [..., 4, 4]."""
    rho, w = tau[..., :3], tau[..., 3:]
    K = _cross_matrix(w)
    K2 = K @ K
    a, b, c = (x[..., None, None] for x in _rodrigues_coefficients(torch.linalg.norm(w, dim=-1)))
    eye = torch.eye(3, device=tau.device, dtype=tau.dtype)
    out = torch.zeros(*tau.shape[:-1], 4, 4, device=tau.device, dtype=tau.dtype)
    out[..., :3, :3] = eye + a * K + b * K2
    out[..., :3, 3] = ((eye + b * K + c * K2) @ rho.unsqueeze(-1)).squeeze(-1)
    out[..., 3, 3] = 1.0
    return out


def update_pose(camera, converged_threshold=1e-4):
    """Folds the camera's optimised deltas into its pose (pose_utils.py:81-98): [R|T] <- exp([trans_delta, rot_delta]) [R|T].
    Returns whether the step was below the convergence threshold."""
    tau = torch.cat([camera.cam_trans_delta.detach(), camera.cam_rot_delta.detach()])
    step = SE3_exp(tau)
    R_new = step[:3, :3] @ camera.R
    T_new = step[:3, :3] @ camera.T + step[:3, 3]
    camera.update_RT(R_new, T_new)
    with torch.no_grad():
        camera.cam_rot_delta.zero_()
        camera.cam_trans_delta.zero_()
    return tau.norm() < converged_threshold
