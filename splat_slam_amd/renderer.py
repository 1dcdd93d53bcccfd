"""render() of the mapping loop -- mirror of
/root/reference/thirdparty/gaussian_splatting/gaussian_renderer/__init__.py:24-153 (same arguments, same result dict),
calling the drop-in `diff_gaussian_rasterization` package (HIP)."""
import math

import torch

from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, mask=None):
    if pc.get_xyz.shape[0] == 0:
        return None
    # the dummy whose .grad receives dL/d(mean2D) (:43-52).  The reference builds `zeros_like(...) + 0` and retains the grad of
    # that non-leaf; a leaf of zeros is the same tensor with the same .grad and one kernel instead of two
    xyz = pc.get_xyz
    screenspace_points = torch.zeros(xyz.shape, dtype=xyz.dtype, device=xyz.device, requires_grad=True)
    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
        projmatrix_raw=viewpoint_camera.projection_matrix, sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center, prefiltered=False, debug=False)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    means3D = xyz
    means2D = screenspace_points
    opacity = pc.get_opacity
    scales = pc.get_scaling
    if scales.shape[-1] == 1:
        scales = scales.repeat(1, 3)
    rotations = pc.get_rotation
    shs, colors_precomp = (None, override_color) if override_color is not None else (pc.get_features, None)
    # `mask` (a boolean over the Gaussians) renders a subset; the rasterizer sees ordinary tensors either way
    pick = (lambda t: t) if mask is None else (lambda t: None if t is None else t[mask])
    rendered_image, radii, depth, opacity, n_touched = rasterizer(
        means3D=pick(means3D), means2D=pick(means2D), shs=pick(shs), colors_precomp=pick(colors_precomp), opacities=pick(opacity),
        scales=pick(scales), rotations=pick(rotations), cov3D_precomp=None, theta=viewpoint_camera.cam_rot_delta,
        rho=viewpoint_camera.cam_trans_delta)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii, "depth": depth, "opacity": opacity, "n_touched": n_touched}
