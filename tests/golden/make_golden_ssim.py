#!/usr/bin/env python3
"""Generates tests/golden/reference_ssim.npz by IMPORTING the reference's mapping loss with `ssim_loss: True`
(/root/reference/thirdparty/monogs/utils/slam_utils.py:80-105, ssim of gaussian_splatting/utils/loss_utils.py:36-101):
loss value + autograd gradients wrt image, depth and the exposure parameters.  Build container only; the output is data.

    python tests/golden/make_golden_ssim.py
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
torch.Tensor.cuda = lambda self, *a, **k: self

from thirdparty.gaussian_splatting.utils.loss_utils import ssim  # noqa: E402
from thirdparty.monogs.utils.slam_utils import get_loss_mapping  # noqa: E402

g = torch.Generator().manual_seed(47)
out = {}


class _VP:
    pass


cfg = {"Training": {"alpha": 0.8, "rgb_boundary_threshold": 0.01, "ssim_loss": True}, "opt_params": {"lambda_dssim": 0.2}}
H, W = 40, 56
image = torch.rand(3, H, W, generator=g).requires_grad_(True)
depth = (3 * torch.rand(1, H, W, generator=g)).requires_grad_(True)
gt = (0.7 * image.detach() + 0.3 * torch.rand(3, H, W, generator=g)).clamp(0, 1)      # correlated with the render: SSIM well away from 0
gt[:, :4, :7] = 0.0
gtd = 3 * torch.rand(H, W, generator=g)
gtd[10:14, 20:30] = 0.0
vp = _VP()
vp.original_image = gt
vp.depth = gtd.numpy()
vp.exposure_a = torch.tensor([0.05], requires_grad=True)
vp.exposure_b = torch.tensor([-0.02], requires_grad=True)
loss = get_loss_mapping(cfg, image, depth, vp, None, initialization=False)
loss.backward()
out["image"], out["depth"], out["gt"], out["gtd"] = image.detach().numpy(), depth.detach().numpy(), gt.numpy(), gtd.numpy()
out["loss"] = np.array(loss.item())
out["dimage"], out["ddepth"] = image.grad.numpy(), depth.grad.numpy()
out["da"], out["db"] = np.array(vp.exposure_a.grad.item()), np.array(vp.exposure_b.grad.item())
a, b = torch.rand(3, 24, 30, generator=g), torch.rand(3, 24, 30, generator=g)
out["ssim_a"], out["ssim_b"], out["ssim"] = a.numpy(), b.numpy(), np.array(ssim(a, b).item())
np.savez_compressed(os.path.join(HERE, "reference_ssim.npz"), **out)
print({k: (v.shape if v.ndim else float(v)) for k, v in out.items()})
