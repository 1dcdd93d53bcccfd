"""Map deformation after the tracker moved a keyframe -- mirror of Mapper.update_mapping_points,
/root/reference/src/mapper.py:154-255, and the quaternion helpers it uses
(/root/reference/thirdparty/gaussian_splatting/utils/general_utils.py:138-175).  Torch formulation (device agnostic,
used on CPU and as the checker); a model that lives on the GPU goes through ONE pass of sgr_deform_points instead of
the ~25 torch kernels below."""
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
    if dev.type == "cuda" and USE_HIP:
        return _update_mapping_points_hip(gaussians, frame_idx, w2c, w2c_old, depth, depth_old, intrinsics, method)
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


USE_HIP = True


def _update_mapping_points_hip(gaussians, frame_idx, w2c, w2c_old, depth, depth_old, intrinsics, method=None):
    """Same update through the C ABI: the 4x4 algebra stays on the host side (three tiny matrices), the per-Gaussian work
    is one kernel in place on the parameter storage; the Parameter / Adam-state replacement is the reference's."""
    import ctypes as C
    from splat_slam_amd import _native as nat
    dev = gaussians.get_xyz.device
    w2c_old = w2c_old.to(dev).float()
    w2c = w2c.to(dev).float()
    c2w_old = torch.linalg.inv(w2c_old)
    transformation = torch.linalg.inv(c2w_old @ w2c)
    tq = rotation_matrix_to_quaternion(transformation[:3, :3].unsqueeze(0))[0]
    f = nat.SgrDeformFrame()
    f.frame_idx, f.rigid = int(frame_idx), int(method == "rigid")
    for name, t in (("w2c_old", w2c_old), ("c2w_old", c2w_old), ("transform", transformation)):
        getattr(f, name)[:] = t.detach().cpu().reshape(-1).tolist()
    f.quat_wxyz[:] = tq.detach().cpu().tolist()
    f.intrinsics[:] = intrinsics.detach().float().cpu().reshape(-1).tolist()
    keep = []
    if not f.rigid:
        dn = depth.to(device=dev, dtype=torch.float32).contiguous()
        do = depth_old.to(device=dev, dtype=torch.float32).contiguous()
        keep += [dn, do]
        f.height, f.width = int(dn.shape[0]), int(dn.shape[1])
        f.depth_new, f.depth_old = dn.data_ptr(), do.data_ptr()
    ids = gaussians.unique_kfIDs.to(device=dev, dtype=torch.int32).contiguous()
    xyz = gaussians._xyz.detach().contiguous()
    rot = gaussians._rotation.detach().contiguous()
    sc = gaussians._scaling.detach().contiguous()
    with torch.cuda.device(dev):
        nat.check(nat.lib().sgr_deform_points(xyz.shape[0], ids.data_ptr(), C.byref(f), xyz.data_ptr(), rot.data_ptr(),
                                              sc.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "sgr_deform_points")
    gaussians._xyz = gaussians.replace_tensor_to_optimizer(xyz, "xyz")["xyz"]
    gaussians._rotation = gaussians.replace_tensor_to_optimizer(rot, "rotation")["rotation"]
    if not f.rigid:
        gaussians._scaling = gaussians.replace_tensor_to_optimizer(sc, "scaling")["scaling"]
