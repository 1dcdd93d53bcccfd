"""Minimal drop-in for the part of `lietorch` the Splat-SLAM mapping path touches
(/root/reference/thirdparty/glorie_slam/depth_video.py:17,327-330: `SE3(pose).inv().matrix()`), plus the group
operations the reference's tracker glue uses, backed by the gfx950 `se3_*` entry points of
include/splat_hip.h.  Pose layout (tx,ty,tz,qx,qy,qz,qw); tangent (rho, theta) -- the same order as
/root/reference/thirdparty/monogs/utils/pose_utils.py:81-98.  Forward only (the mapping path never differentiates it).
GPU tensors only: there is no CPU path.
"""
import torch

from splat_slam_amd import _native as nat

__all__ = ["SE3", "Sim3"]


def _call(fn_name, n, *tensors_then_out):
    lib = nat.lib()
    dev = tensors_then_out[0].device
    with torch.cuda.device(dev):
        args = [t.data_ptr() for t in tensors_then_out[:-1]] + [n, tensors_then_out[-1].data_ptr(),
                                                                  torch.cuda.current_stream(dev).cuda_stream]
        nat.check(getattr(lib, fn_name)(*args), fn_name)
    return tensors_then_out[-1]


def _prep(t, width):
    if not t.is_cuda:
        raise RuntimeError("lietorch (MI355X build): tensors must live on the GPU; there is no CPU path")
    return t.detach().float().reshape(-1, width).contiguous()


class SE3:
    manifold_dim = 6
    embedded_dim = 7

    def __init__(self, data):
        self.data = data

    @staticmethod
    def Identity(*batch, device="cuda", dtype=torch.float32):
        d = torch.zeros(*batch, 7, device=device, dtype=dtype)
        d[..., 6] = 1.0
        return SE3(d)

    @staticmethod
    def exp(tau):
        a = _prep(tau, 6)
        out = torch.empty(a.shape[0], 7, device=a.device)
        return SE3(_call("se3_exp", a.shape[0], a, out).reshape(*tau.shape[:-1], 7))

    def log(self):
        a = _prep(self.data, 7)
        out = torch.empty(a.shape[0], 6, device=a.device)
        return _call("se3_log", a.shape[0], a, out).reshape(*self.data.shape[:-1], 6)

    def inv(self):
        a = _prep(self.data, 7)
        out = torch.empty_like(a)
        return SE3(_call("se3_inv", a.shape[0], a, out).reshape(self.data.shape))

    def matrix(self):
        a = _prep(self.data, 7)
        out = torch.empty(a.shape[0], 16, device=a.device)
        return _call("se3_matrix", a.shape[0], a, out).reshape(*self.data.shape[:-1], 4, 4)

    def vec(self):
        return self.data

    def __mul__(self, other):
        if isinstance(other, SE3):
            a, b = torch.broadcast_tensors(self.data, other.data)
            a, b = _prep(a, 7), _prep(b, 7)
            out = torch.empty_like(a)
            lib = nat.lib()
            with torch.cuda.device(a.device):
                nat.check(lib.se3_mul(a.data_ptr(), b.data_ptr(), a.shape[0], out.data_ptr(),
                                      torch.cuda.current_stream(a.device).cuda_stream), "se3_mul")
            return SE3(out.reshape(torch.broadcast_shapes(self.data.shape, other.data.shape)))
        return self.act(other)

    def act(self, pts):
        shape = torch.broadcast_shapes(self.data.shape[:-1], pts.shape[:-1])
        a = _prep(self.data.expand(*shape, 7), 7)
        p = _prep(pts.expand(*shape, 3), 3)
        out = torch.empty_like(p)
        lib = nat.lib()
        with torch.cuda.device(a.device):
            nat.check(lib.se3_act(a.data_ptr(), p.data_ptr(), a.shape[0], out.data_ptr(),
                                  torch.cuda.current_stream(a.device).cuda_stream), "se3_act")
        return out.reshape(*shape, 3)

    def adjT(self, a6):
        shape = torch.broadcast_shapes(self.data.shape[:-1], a6.shape[:-1])
        a = _prep(self.data.expand(*shape, 7), 7)
        v = _prep(a6.expand(*shape, 6), 6)
        out = torch.empty_like(v)
        lib = nat.lib()
        with torch.cuda.device(a.device):
            nat.check(lib.se3_adjT(a.data_ptr(), v.data_ptr(), a.shape[0], out.data_ptr(),
                                   torch.cuda.current_stream(a.device).cuda_stream), "se3_adjT")
        return out.reshape(*shape, 6)

    def retr(self, dx):
        """exp(dx) * self  (thirdparty/glorie_slam/geom/ba.py:45)."""
        return SE3.exp(dx) * self

    def __getitem__(self, idx):
        return SE3(self.data[idx])


class Sim3:
    """Placeholder: the reference only does `isinstance(x, Sim3)` (thirdparty/glorie_slam/geom/projective_ops.py:98)."""
