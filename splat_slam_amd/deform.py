"""Map deformation after the tracker moved a keyframe -- mirror of Mapper.update_mapping_points,
/root/reference/src/mapper.py:154-255, and the quaternion helpers it uses
(/root/reference/thirdparty/gaussian_splatting/utils/general_utils.py:138-175).  Pure torch (device agnostic)."""
import torch


def rotation_matrix_to_quaternion(R):
    q = torch.zeros((R.size(0), 4), device=R.device)
    zero = torch.tensor(0.0, device=R.device)
    q[:, 0] = torch.sqrt(torch.max(zero, 1 + R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2])) / 2
    q[:, 1] = torch.sqrt(torch.max(zero, 1 + R[:, 0, 0] - R[:, 1, 1] - R[:, 2, 2])) / 2
    q[:, 2] = torch.sqrt(torch.max(zero, 1 - R[:, 0, 0] + R[:, 1, 1] - R[:, 2, 2])) / 2
    q[:, 3] = torch.sqrt(torch.max(zero, 1 - R[:, 0, 0] - R[:, 1, 1] + R[:, 2, 2])) / 2
    q[:, 1] *= torch.sign(q[:, 1] * (R[:, 2, 1] - R[:, 1, 2]))
    q[:, 2] *= torch.sign(q[:, 2] * (R[:, 0, 2] - R[:, 2, 0]))
    q[:, 3] *= torch.sign(q[:, 3] * (R[:, 1, 0] - R[:, 0, 1]))
    return q


def quaternion_multiply(q1, q2):
    w1, x1, y1, z1 = q1[..., 0], q1[..., 1], q1[..., 2], q1[..., 3]
    w2, x2, y2, z2 = q2[..., 0], q2[..., 1], q2[..., 2], q2[..., 3]
    w = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2
    x = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2
    y = w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2
    z = w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2
    return torch.stack((w, x, y, z), dim=-1)


@torch.no_grad()
def update_mapping_points(gaussians, frame_idx, w2c, w2c_old, depth, depth_old, intrinsics, method=None):
    """Moves (and, unless method == "rigid", depth-rescales) the Gaussians anchored to keyframe `frame_idx`.
    Adam moments of the touched tensors are reset exactly like the reference (replace_tensor_to_optimizer)."""
    frame_mask = gaussians.unique_kfIDs == frame_idx
    if frame_mask.sum() == 0:
        return
    dev = gaussians.get_xyz.device
    frame_mask = frame_mask.to(dev)
    transformation = torch.linalg.inv(torch.linalg.inv(w2c_old) @ w2c)
    if method == "rigid":
        means = gaussians.get_xyz.detach()
        ones = torch.ones(int(frame_mask.sum()), 1, device=dev).float()
        pts4 = torch.cat((means[frame_mask], ones), dim=1)
        means[frame_mask] = (transformation @ pts4.T).T[:, :3]
        gaussians._xyz = gaussians.replace_tensor_to_optimizer(means, "xyz")["xyz"]
        rots = gaussians.get_rotation.detach()
        tq = rotation_matrix_to_quaternion(transformation.unsqueeze(0))
        rots[frame_mask] = quaternion_multiply(tq.expand_as(rots[frame_mask]), rots[frame_mask])
        gaussians._rotation = gaussians.replace_tensor_to_optimizer(rots, "rotation")["rotation"]
        return
    depth = depth.to(dev)
    depth_old = depth_old.to(dev)
    means = gaussians.get_xyz.detach()[frame_mask]
    ones = torch.ones(means.shape[0], 1, device=dev).float()
    pts4 = torch.cat((means, ones), dim=1)
    pix = (intrinsics @ (w2c_old @ pts4.T)[:3, :]).T
    pix[:, 0] /= pix[:, 2]
    pix[:, 1] /= pix[:, 2]
    pix = pix[:, :2].long()
    height, width = depth.shape
    pix[:, 0] = torch.clamp(pix[:, 0], min=0, max=width - 1)
    pix[:, 1] = torch.clamp(pix[:, 1], min=0, max=height - 1)
    d_new = depth[pix[:, 1], pix[:, 0]]
    d_old = depth_old[pix[:, 1], pix[:, 0]]
    means_cam = (w2c_old @ pts4.T).T[:, :3]
    rescale = (1 + 1 / (means_cam[:, 2]) * (d_new - d_old)).unsqueeze(-1)
    rescale[torch.logical_or(d_new == 0, d_old == 0)] = 1
    rescale[rescale <= 0.0] = 1
    means_cam = rescale.repeat(1, 3) * means_cam
    pts4 = torch.cat((means_cam, ones), dim=1)
    means = (torch.linalg.inv(w2c_old) @ pts4.T).T[:, :3]
    pts4 = torch.cat((means, ones), dim=1)
    means = (transformation @ pts4.T).T[:, :3]
    global_means = gaussians.get_xyz.detach()
    global_means[frame_mask] = means
    gaussians._xyz = gaussians.replace_tensor_to_optimizer(global_means, "xyz")["xyz"]
    rots = gaussians.get_rotation.detach()
    tq = rotation_matrix_to_quaternion(transformation.unsqueeze(0))
    rots[frame_mask] = quaternion_multiply(tq.expand_as(rots[frame_mask]), rots[frame_mask])
    gaussians._rotation = gaussians.replace_tensor_to_optimizer(rots, "rotation")["rotation"]
    scales = gaussians._scaling.detach()
    scales[frame_mask] = scales[frame_mask] + torch.log(rescale)
    gaussians._scaling = gaussians.replace_tensor_to_optimizer(scales, "scaling")["scaling"]
