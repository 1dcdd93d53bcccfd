"""Mapping loss -- mirror of /root/reference/thirdparty/monogs/utils/slam_utils.py:71-119 (SSIM branch off by default,
/root/reference/configs/splat_slam.yaml:36) plus a fused HIP variant (`sgr_mapping_loss`) with identical values."""

import torch

from splat_slam_amd import _native as nat


def _gt_depth(viewpoint, device):
    d = viewpoint.depth
    if not torch.is_tensor(d):
        d = torch.from_numpy(d)
    return d.to(dtype=torch.float32, device=device)[None]


def get_loss_mapping_rgbd(config, image, depth, viewpoint, initialization=False):
    alpha = config["Training"]["alpha"] if "alpha" in config["Training"] else 0.95
    rgb_boundary_threshold = config["Training"]["rgb_boundary_threshold"]
    gt_image = viewpoint.original_image.to(image.device)
    _, h, w = gt_image.shape
    gt_depth = _gt_depth(viewpoint, image.device)
    rgb_pixel_mask = (gt_image.sum(dim=0) > rgb_boundary_threshold).view(1, h, w)
    l1_rgb = torch.abs(image * rgb_pixel_mask - gt_image * rgb_pixel_mask)
    depth_pixel_mask = (gt_depth > 0.01).view(*depth.shape)
    l1_depth = torch.abs(depth * depth_pixel_mask - gt_depth * depth_pixel_mask)
    return alpha * l1_rgb.mean() + (1 - alpha) * l1_depth.mean()


def get_loss_mapping(config, image, depth, viewpoint, opacity, initialization=False):
    if initialization:
        image_ab = image
    else:
        image_ab = (torch.exp(viewpoint.exposure_a)) * image + viewpoint.exposure_b
    return get_loss_mapping_rgbd(config, image_ab, depth, viewpoint)


def get_median_depth(depth, opacity=None, mask=None, return_std=False):
    """Median (and optionally std) of the rendered depth over pixels that are covered (depth > 0), nearly opaque
    (opacity > 0.95) and inside `mask` -- slam_utils.py:108-119; scales the keyframe baseline test (mapper.py:984)."""
    depth = depth.detach()
    use = depth > 0
    for extra in ((opacity.detach() > 0.95) if opacity is not None else None, mask):
        if extra is not None:
            use = use & extra
    picked = depth[use]
    return (picked.median(), picked.std(), use) if return_std else picked.median()


class _FusedMappingLoss(torch.autograd.Function):
    """One pass over the image: loss value + dL/dimage, dL/ddepth, dL/da, dL/db (fixed-order reduction)."""

    @staticmethod
    def forward(ctx, image, depth, exp_a, exp_b, gt_image, gt_depth, alpha, thr):
        lib = nat.lib()
        dev = image.device
        _, H, W = image.shape
        image, depth = image.contiguous(), depth.contiguous()
        # one arena per call: dL/dimage | dL/ddepth | loss, d/da, d/db | 16 KiB of reduction scratch
        hw = H * W
        arena = torch.empty(4 * hw + 4 + 4096, dtype=torch.float32, device=dev)
        d_img, d_dep = arena[:3 * hw].view(3, H, W), arena[3 * hw:4 * hw].view(depth.shape)
        loss, d_a, d_b = arena[4 * hw:4 * hw + 1], arena[4 * hw + 1:4 * hw + 2], arena[4 * hw + 2:4 * hw + 3]
        scratch = arena[4 * hw + 4:]
        if dev.index is not None and dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        nat.check(lib.sgr_mapping_loss(H, W, image.data_ptr(), depth.data_ptr(), gt_image.data_ptr(), gt_depth.data_ptr(),
                                       nat.ptr(exp_a), nat.ptr(exp_b), alpha, thr, 1.0, loss.data_ptr(), d_img.data_ptr(),
                                       d_dep.data_ptr(), d_a.data_ptr(), d_b.data_ptr(), scratch.data_ptr(), 4 * scratch.numel(),
                                       torch.cuda.current_stream().cuda_stream), "sgr_mapping_loss")
        ctx.save_for_backward(d_img, d_dep, d_a, d_b)
        ctx.has_exp = exp_a is not None
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        d_img, d_dep, d_a, d_b = ctx.saved_tensors
        return (d_img * g, d_dep * g, d_a * g if ctx.has_exp else None, d_b * g if ctx.has_exp else None,
                None, None, None, None)


def get_loss_mapping_fused(config, image, depth, viewpoint, opacity, initialization=False):
    alpha = config["Training"]["alpha"] if "alpha" in config["Training"] else 0.95
    thr = config["Training"]["rgb_boundary_threshold"]
    gt_image = viewpoint.original_image
    gt_depth = getattr(viewpoint, "_depth_dev", None)
    if gt_depth is None:
        gt_depth = _gt_depth(viewpoint, image.device).contiguous()
        viewpoint._depth_dev = gt_depth     # keep the ground-truth depth resident instead of re-uploading per call
    a, b = (None, None) if initialization else (viewpoint.exposure_a, viewpoint.exposure_b)
    return _FusedMappingLoss.apply(image, depth, a, b, gt_image.contiguous(), gt_depth, float(alpha), float(thr))
