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


def _rigid_delta(w2c_new, w2c_old):
    """World-space motion of the points anchored to a keyframe whose pose went w2c_old -> w2c_new
    (mapper.py:167,231: inv(inv(w2c_old) @ w2c_new)) = c2w_new @ w2c_old, and its rotation as a (w, x, y, z) quaternion."""
    delta = torch.linalg.inv(torch.linalg.inv(w2c_old) @ w2c_new)
    return delta, rotation_matrix_to_quaternion(delta.unsqueeze(0))


def ray_rescale_factors(p_cam, K, depth_new, depth_old):
    """Per-point factor along the old camera's ray (mapper.py:196-228): 1 + (d_new - d_old) / z at the pixel the point
    projects to (truncated to integers, clamped to the image border); 1 where either depth map is empty there or the
    factor would not be positive."""
    h, w = depth_new.shape
    uvw = p_cam @ K.T
    u = (uvw[:, 0] / uvw[:, 2]).long().clamp(0, w - 1)
    v = (uvw[:, 1] / uvw[:, 2]).long().clamp(0, h - 1)
    d_new, d_old = depth_new[v, u], depth_old[v, u]
    f = 1 + (d_new - d_old) / p_cam[:, 2]
    keep = (d_new == 0) | (d_old == 0) | (f <= 0)
    return torch.where(keep, torch.ones_like(f), f)


@torch.no_grad()
def update_mapping_points(gaussians, frame_idx, w2c, w2c_old, depth, depth_old, intrinsics, method=None):
    """Moves (and, unless method == "rigid", depth-rescales) the Gaussians anchored to keyframe `frame_idx`
    (Mapper.update_mapping_points, mapper.py:154-255).  Adam moments of the touched tensors are reset exactly like the
    reference (replace_tensor_to_optimizer).  Torch formulation: CPU models and the checker of sgr_deform_points."""
    sel = gaussians.unique_kfIDs == frame_idx
    if sel.sum() == 0:
        return
    dev = gaussians.get_xyz.device
    if dev.type == "cuda" and USE_HIP:
        return _update_mapping_points_hip(gaussians, frame_idx, w2c, w2c_old, depth, depth_old, intrinsics, method)
    sel = sel.to(dev)
    delta, dq = _rigid_delta(w2c, w2c_old)
    xyz = gaussians.get_xyz.detach()
    pts = xyz[sel]
    log_f = None
    if method != "rigid":
        c2w_old = torch.linalg.inv(w2c_old)
        p_cam = pts @ w2c_old[:3, :3].T + w2c_old[:3, 3]
        f = ray_rescale_factors(p_cam, intrinsics, depth.to(dev), depth_old.to(dev))
        pts = (f[:, None] * p_cam) @ c2w_old[:3, :3].T + c2w_old[:3, 3]
        log_f = torch.log(f)[:, None]
    xyz[sel] = pts @ delta[:3, :3].T + delta[:3, 3]
    gaussians._xyz = gaussians.replace_tensor_to_optimizer(xyz, "xyz")["xyz"]
    rots = gaussians.get_rotation.detach()               # activated (normalised) rotations are written back, :240-250
    rots[sel] = quaternion_multiply(dq.expand_as(rots[sel]), rots[sel])
    gaussians._rotation = gaussians.replace_tensor_to_optimizer(rots, "rotation")["rotation"]
    if log_f is not None:
        scales = gaussians._scaling.detach()
        scales[sel] = scales[sel] + log_f
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
    gaussians.invalidate_activations()         # (raw-pointer writes: no _version bump)
    gaussians._xyz = gaussians.replace_tensor_to_optimizer(xyz, "xyz")["xyz"]
    gaussians._rotation = gaussians.replace_tensor_to_optimizer(rot, "rotation")["rotation"]
    if not f.rigid:
        gaussians._scaling = gaussians.replace_tensor_to_optimizer(sc, "scaling")["scaling"]
