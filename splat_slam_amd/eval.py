"""Rendering metric of the hot path -- PSNR as computed by eval_rendering,
/root/reference/src/utils/eval_utils.py:90-123 with psnr() of
/root/reference/thirdparty/gaussian_splatting/utils/image_utils.py:19-21.  (SSIM / LPIPS / mesh metrics need
third-party evaluation libraries and are out of scope)"""
import torch

from splat_slam_amd.renderer import render


def psnr(img1, img2):
    mse = ((img1 - img2) ** 2).view(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


@torch.no_grad()
def eval_rendering_psnr(frames, gaussians, pipe, background):
    """frames: list of Camera in keyframe order; exposure compensation is applied to every frame but the first
    (eval_utils.py:96-99); PSNR over pixels where the ground truth is > 0 (:109,123)."""
    scores = []
    for k, frame in enumerate(frames):
        rendering = render(frame, gaussians, pipe, background)["render"].detach()
        image = torch.exp(frame.exposure_a.detach()) * rendering + frame.exposure_b.detach() if k > 0 else rendering
        image = torch.clamp(image, 0.0, 1.0)
        gt = frame.original_image
        mask = gt > 0
        scores.append(psnr(image[mask].unsqueeze(0), gt[mask].unsqueeze(0)).item())
    return scores
