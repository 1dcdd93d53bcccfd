"""SE3 exponential + pose update of the mapping loop -- mirror of
/root/reference/thirdparty/monogs/utils/pose_utils.py:16-98 (tau = [rho, theta], left perturbation)."""
import torch


def skew_sym_mat(x):
    ssm = torch.zeros(3, 3, device=x.device, dtype=x.dtype)
    ssm[0, 1], ssm[0, 2] = -x[2], x[1]
    ssm[1, 0], ssm[1, 2] = x[2], -x[0]
    ssm[2, 0], ssm[2, 1] = -x[1], x[0]
    return ssm


def SO3_exp(theta):
    W = skew_sym_mat(theta)
    W2 = W @ W
    angle = torch.norm(theta)
    I = torch.eye(3, device=theta.device, dtype=theta.dtype)
    if angle < 1e-5:
        return I + W + 0.5 * W2
    return I + (torch.sin(angle) / angle) * W + ((1 - torch.cos(angle)) / (angle ** 2)) * W2


def V(theta):
    I = torch.eye(3, device=theta.device, dtype=theta.dtype)
    W = skew_sym_mat(theta)
    W2 = W @ W
    angle = torch.norm(theta)
    if angle < 1e-5:
        return I + 0.5 * W + (1.0 / 6.0) * W2
    return I + W * ((1.0 - torch.cos(angle)) / (angle ** 2)) + W2 * ((angle - torch.sin(angle)) / (angle ** 3))


def SE3_exp(tau):
    rho, theta = tau[:3], tau[3:]
    T = torch.eye(4, device=tau.device, dtype=tau.dtype)
    T[:3, :3] = SO3_exp(theta)
    T[:3, 3] = V(theta) @ rho
    return T


def update_pose(camera, converged_threshold=1e-4):
    tau = torch.cat([camera.cam_trans_delta, camera.cam_rot_delta], axis=0)
    T_w2c = torch.eye(4, device=tau.device)
    T_w2c[0:3, 0:3] = camera.R
    T_w2c[0:3, 3] = camera.T
    new_w2c = SE3_exp(tau) @ T_w2c
    converged = tau.norm() < converged_threshold
    camera.update_RT(new_w2c[0:3, 0:3], new_w2c[0:3, 3])
    camera.cam_rot_delta.data.fill_(0)
    camera.cam_trans_delta.data.fill_(0)
    return converged
