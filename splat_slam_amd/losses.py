"""Mapping loss -- mirror of /root/reference/thirdparty/monogs/utils/slam_utils.py:71-119 (SSIM branch off by default,
/root/reference/configs/splat_slam.yaml:36) plus a fused HIP variant (`sgr_mapping_loss`) with identical values."""

import torch

from splat_slam_amd import _native as nat


def _gt_depth(viewpoint, device):
    d = viewpoint.depth
    if not torch.is_tensor(d):
        d = torch.from_numpy(d)
    return d.to(dtype=torch.float32, device=device)[None]


_WINDOWS = {}


def ssim(img1, img2, window_size=11):
    """Mean structural similarity with an 11x11 Gaussian window of sigma 1.5, per channel, zero padding
    (/root/reference/thirdparty/gaussian_splatting/utils/loss_utils.py:36-101); C1 = 0.01^2, C2 = 0.03^2."""
    import torch.nn.functional as F
    ch = img1.size(-3)
    key = (window_size, ch, img1.device, img1.dtype)
    win = _WINDOWS.get(key)
    if win is None:
        x = torch.arange(window_size, dtype=torch.float32) - window_size // 2
        g1 = torch.exp(-(x * x) / (2 * 1.5 ** 2))
        g1 = (g1 / g1.sum()).unsqueeze(1)
        win = (g1 @ g1.t()).unsqueeze(0).unsqueeze(0).expand(ch, 1, window_size, window_size).contiguous().to(img1.device, img1.dtype)
        _WINDOWS[key] = win
    blur = lambda t: F.conv2d(t, win, padding=window_size // 2, groups=ch)
    mu1, mu2 = blur(img1), blur(img2)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1, s2, s12 = blur(img1 * img1) - mu1_sq, blur(img2 * img2) - mu2_sq, blur(img1 * img2) - mu12
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu12 + c1) * (2 * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2))).mean()


def uses_ssim(mapping_config):
    """`ssim_loss: True` (slam_utils.py:89-98; off in configs/splat_slam.yaml:36): the loss is no longer a per-pixel L1, so the
    fused kernels (sign-code epilogue, sgr_mapping_loss) do not apply: the loops run the torch formulation through autograd."""
    return bool(mapping_config["Training"].get("ssim_loss", False))


def get_loss_mapping_rgbd(config, image, depth, viewpoint, initialization=False):
    alpha = config["Training"]["alpha"] if "alpha" in config["Training"] else 0.95
    rgb_boundary_threshold = config["Training"]["rgb_boundary_threshold"]
    gt_image = viewpoint.original_image.to(image.device)
    _, h, w = gt_image.shape
    gt_depth = _gt_depth(viewpoint, image.device)
    rgb_pixel_mask = (gt_image.sum(dim=0) > rgb_boundary_threshold).view(1, h, w)
    l1_rgb = torch.abs(image * rgb_pixel_mask - gt_image * rgb_pixel_mask)
    if uses_ssim(config):                      # slam_utils.py:89-98: (1 - lambda) |.| + lambda (1 - ssim), the scalar broadcast per pixel
        lam = config["opt_params"]["lambda_dssim"]
        l1_rgb = (1.0 - lam) * l1_rgb + lam * (1.0 - ssim(image, gt_image))
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


def _native_nodes():
    import diff_gaussian_rasterization as drg
    return getattr(drg, "native_extension", lambda: None)()       # (tests inject the oracle under this module name)


class _FusedMappingLoss(torch.autograd.Function):
    """One pass over the image: loss value + dL/dimage, dL/ddepth, dL/da, dL/db (fixed-order reduction)."""

    @staticmethod
    def forward(ctx, image, depth, exp_a, exp_b, gt_image, gt_depth, alpha, thr):
        lib = nat.lib()
        dev = image.device
        _, H, W = image.shape
        image, depth = image.contiguous(), depth.contiguous()
        # one arena per call: dL/dimage | dL/ddepth | d/da, d/db | loss | 16 KiB of reduction scratch
        hw = H * W
        arena = torch.empty(4 * hw + 4 + 4096, dtype=torch.float32, device=dev)
        d_img, d_dep = arena[:3 * hw].view(3, H, W), arena[3 * hw:4 * hw].view(depth.shape)
        d_a, d_b, loss = arena[4 * hw:4 * hw + 1], arena[4 * hw + 1:4 * hw + 2], arena[4 * hw + 2:4 * hw + 3]
        scratch = arena[4 * hw + 4:]
        if dev.index is not None and dev.index != torch.cuda.current_device():
            torch.cuda.set_device(dev)
        nat.check(lib.sgr_mapping_loss(H, W, image.data_ptr(), depth.data_ptr(), gt_image.data_ptr(), gt_depth.data_ptr(),
                                       nat.ptr(exp_a), nat.ptr(exp_b), alpha, thr, 1.0, loss.data_ptr(), d_img.data_ptr(),
                                       d_dep.data_ptr(), d_a.data_ptr(), d_b.data_ptr(), scratch.data_ptr(), 4 * scratch.numel(),
                                       torch.cuda.current_stream().cuda_stream), "sgr_mapping_loss")
        ctx.save_for_backward(arena)
        ctx.has_exp, ctx.shape = exp_a is not None, (H, W, tuple(depth.shape))
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        arena, = ctx.saved_tensors
        H, W, dshape = ctx.shape
        hw = H * W
        s = arena[:4 * hw + 2] * g               # the four gradients are contiguous: ONE scaling launch
        return (s[:3 * hw].view(3, H, W), s[3 * hw:4 * hw].view(dshape), s[4 * hw:4 * hw + 1] if ctx.has_exp else None,
                s[4 * hw + 1:4 * hw + 2] if ctx.has_exp else None, None, None, None, None)


def get_loss_mapping_fused(config, image, depth, viewpoint, opacity, initialization=False):
    if uses_ssim(config):                      # not an L1 per pixel: the torch formulation
        return get_loss_mapping(config, image, depth, viewpoint, opacity, initialization)
    alpha = config["Training"]["alpha"] if "alpha" in config["Training"] else 0.95
    thr = config["Training"]["rgb_boundary_threshold"]
    gt_image = viewpoint.original_image
    gt_depth = getattr(viewpoint, "_depth_dev", None)
    if gt_depth is None:
        gt_depth = _gt_depth(viewpoint, image.device).contiguous()
        viewpoint._depth_dev = gt_depth     # keep the ground-truth depth resident instead of re-uploading per call
    a, b = (None, None) if initialization else (viewpoint.exposure_a, viewpoint.exposure_b)
    ext = _native_nodes()
    if ext is not None and image.is_cuda:        # the same launch, its autograd node in C++ (diff_gaussian_rasterization/csrc/dgr_native.cpp)
        import diff_gaussian_rasterization as drg
        return ext.mapping_loss(image, depth, a, b, gt_image, gt_depth, float(alpha), float(thr), bool(drg.DEFER_POSE_GRADS))
    return _FusedMappingLoss.apply(image, depth, a, b, gt_image.contiguous(), gt_depth, float(alpha), float(thr))
