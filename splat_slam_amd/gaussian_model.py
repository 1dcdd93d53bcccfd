"""Gaussian map store + optimiser surgery -- mirror of
/root/reference/thirdparty/gaussian_splatting/scene/gaussian_model.py:34-742 (same public names and semantics).

What is identical: the six parameter tensors and their activations (:76-101), the Adam groups and learning rates
(:264-313), the xyz lr schedule (:315-329, general_utils.py:79-94), new-point seeding from an RGB-D frame (:134-219),
optimiser-state surgery on cat / prune (:519-593), clone / split / prune rules (:639-736), densification statistics
(:738-742), opacity resets (:382-395).

What differs, on purpose (MI355X-first, documented in DESIGN.md):
  * no Open3D: back-projection and random down-sampling are torch ops (torch.randperm instead of Open3D's RNG,
    SURVEY.md 3.6) and the 3-NN distance comes from the HIP `sknn_dist2` (or an injected function);
  * `unique_kfIDs` / `n_obs` live on the same device as the parameters (the reference keeps them on the CPU and pays
    a D2H mask copy per densify, :556-557,666-667).
"""

import numpy as np
import weakref

import torch
from torch import nn

C0 = 0.28209479177387814


def RGB2SH(rgb):
    return (rgb - 0.5) / C0


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def helper(step, lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """general_utils.py:79-94."""
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    if lr_delay_steps > 0:
        delay_rate = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
    else:
        delay_rate = 1.0
    t = np.clip(step / max_steps, 0, 1)
    log_lerp = np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)
    return delay_rate * log_lerp


def build_rotation(r):
    """general_utils.py:113-136 (normalises, then (w,x,y,z) -> R)."""
    q = r / torch.sqrt((r * r).sum(dim=1))[:, None]
    R = torch.zeros((q.size(0), 3, 3), device=r.device, dtype=r.dtype)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - w * z)
    R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y)
    R[:, 2, 1] = 2 * (y * z + w * x)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


class OptParams:
    """opt_params block of /root/reference/configs/splat_slam.yaml:63-79."""

    def __init__(self, d=None):
        d = d or {}
        self.position_lr_init = d.get("position_lr_init", 0.00016)
        self.position_lr_final = d.get("position_lr_final", 0.0000016)
        self.position_lr_delay_mult = d.get("position_lr_delay_mult", 0.01)
        self.position_lr_max_steps = d.get("position_lr_max_steps", 30000)
        self.feature_lr = d.get("feature_lr", 0.0025)
        self.opacity_lr = d.get("opacity_lr", 0.05)
        self.scaling_lr = d.get("scaling_lr", 0.001)
        self.rotation_lr = d.get("rotation_lr", 0.001)
        self.percent_dense = d.get("percent_dense", 0.01)
        self.lambda_dssim = d.get("lambda_dssim", 0.2)
        self.densify_from_iter = d.get("densify_from_iter", 500)
        self.densify_grad_threshold = d.get("densify_grad_threshold", 0.0002)


class GaussianModel:
    def __init__(self, sh_degree: int, config=None, device="cuda", knn_fn=None):
        self.device = torch.device(device)
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        e = lambda: torch.empty(0, device=self.device)
        self._xyz, self._features_dc, self._features_rest = e(), e(), e()
        self._scaling, self._rotation, self._opacity = e(), e(), e()
        self.max_radii2D, self.xyz_gradient_accum = e(), e()
        self.unique_kfIDs = torch.empty(0, device=self.device).int()
        self.n_obs = torch.empty(0, device=self.device).int()
        self.optimizer = None
        self.config = config
        self.isotropic = False
        self.spatial_lr_scale = 1.0
        self._knn_fn = knn_fn
        self.use_hip_compaction = True   # CUDA models prune through sgr_keep_list / sgr_gather_rows (bit-identical to torch)
        # The <= 12 renders of one mapping iteration (mapper.py:426-485) see the same parameters: on the GPU the activated
        # tensors (and their autograd nodes) are shared between them instead of being recomputed -- and differentiated --
        # per view (8 + 10 small kernels per render).  Same values; the views' gradients are summed before instead of after
        # the activation's chain rule.  Off on the CPU, where the loops replay the reference bit for bit.
        self.share_activations = self.device.type == "cuda"
        self._act = {}

    # ---- activations (gaussian_model.py:53-61,76-101)
    def _activated(self, name, fn, *params):
        """Shared activation of one parameter set: every render between two parameter updates sees the same tensor (and one
        autograd node).  Consequence: ONE backward per set of renders, like the reference's iteration (mapper.py:426-490);
        `render A, render B, lossA.backward(), lossB.backward()` needs share_activations = False (or retain_graph).
        FusedAdam bumps `_version` of every parameter it steps (like torch.optim.Adam's in-place ops); other kernels that write
        parameters through raw pointers (the fused loops' sgr_map_step / sgr_map_run, sgr_deform_points) call
        invalidate_activations()."""
        if not self.share_activations or not torch.is_grad_enabled():
            return fn(*params)
        vers = tuple(p._version for p in params)
        hit = self._act.get(name)
        if hit is not None and hit[1] == vers and len(hit[0]) == len(params) and all(a is b for a, b in zip(hit[0], params)):
            return hit[2]
        out = fn(*params)
        if out.requires_grad:
            # a backward frees this node's graph: whatever is rendered afterwards needs a fresh one
            # (the hook holds a weak reference: out -> hook -> closure -> out would leave every iteration's [N, .] activations
            #  to the cyclic collector instead of the reference count)
            def drop(grad, n=name, o=weakref.ref(out)):
                h = self._act.get(n)
                if h is not None and h[2] is o():
                    del self._act[n]
            out.register_hook(drop)
            self._act[name] = (params, vers, out)      # (the Parameters are held: identity, not id(), decides a hit)
        return out

    def invalidate_activations(self):
        self._act.clear()

    @property
    def get_scaling(self):
        return self._activated("scaling", torch.exp, self._scaling)

    @property
    def get_rotation(self):
        return self._activated("rotation", torch.nn.functional.normalize, self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return self._activated("features", lambda a, b: torch.cat((a, b), dim=1), self._features_dc, self._features_rest)

    @property
    def get_opacity(self):
        return self._activated("opacity", torch.sigmoid, self._opacity)

    def _knn(self, pts):
        if self._knn_fn is not None:
            return self._knn_fn(pts)
        from simple_knn._C import distCUDA2
        return distCUDA2(pts)

    # ---- seeding (gaussian_model.py:134-219)
    def create_pcd_from_image(self, cam, init=False, scale=2.0, depthmap=None):
        image_ab = torch.clamp(torch.exp(cam.exposure_a) * cam.original_image + cam.exposure_b, 0.0, 1.0)
        rgb_u8 = (image_ab * 255).byte().permute(1, 2, 0).contiguous()
        depth = depthmap if depthmap is not None else cam.depth
        depth = torch.as_tensor(depth, dtype=torch.float32, device=self.device)
        return self.create_pcd_from_image_and_depth(cam, rgb_u8, depth, init)

    def create_pcd_from_image_and_depth(self, cam, rgb_u8, depth, init=False):
        cfg = self.config["mapping"]
        downsample_factor = cfg["pcd_downsample_init"] if init else cfg["pcd_downsample"]
        point_size = cfg["point_size"]
        if cfg.get("adaptive_pointsize", False):
            point_size = min(0.05, point_size * float(torch.median(depth)))
        H, W = depth.shape
        valid = (depth > 0) & (depth < 100.0)              # depth_trunc=100, project_valid_depth_only
        v, u = torch.nonzero(valid, as_tuple=True)
        d = depth[v, u]
        pc = torch.stack([(u.float() - cam.cx) * d / cam.fx, (v.float() - cam.cy) * d / cam.fy, d], dim=1)
        Rw2c, tw2c = cam.R.float(), cam.T.float()
        pw = (pc - tw2c[None, :]) @ Rw2c                     # R^T (p - t)
        cols = rgb_u8[v, u].float() / 255.0
        n_keep = int(pw.shape[0] * (1.0 / downsample_factor))
        keep = torch.randperm(pw.shape[0], device=self.device)[:n_keep]
        keep, _ = torch.sort(keep)
        new_xyz, new_rgb = pw[keep].contiguous(), cols[keep]
        fused_color = RGB2SH(new_rgb)
        features = torch.zeros((fused_color.shape[0], 3, (self.max_sh_degree + 1) ** 2), device=self.device)
        features[:, :3, 0] = fused_color
        dist2 = torch.clamp_min(self._knn(new_xyz), 0.0000001) * point_size
        scales = torch.log(torch.sqrt(dist2))[..., None]
        if not self.isotropic:
            scales = scales.repeat(1, 3)
        rots = torch.zeros((new_xyz.shape[0], 4), device=self.device)
        rots[:, 0] = 1
        opacities = inverse_sigmoid(0.5 * torch.ones((new_xyz.shape[0], 1), dtype=torch.float, device=self.device))
        return new_xyz, features, scales, rots, opacities

    def init_lr(self, spatial_lr_scale):
        self.spatial_lr_scale = spatial_lr_scale

    def extend_from_pcd(self, fused_point_cloud, features, scales, rots, opacities, kf_id):
        new_xyz = nn.Parameter(fused_point_cloud.requires_grad_(True))
        new_features_dc = nn.Parameter(features[:, :, 0:1].transpose(1, 2).contiguous().requires_grad_(True))
        new_features_rest = nn.Parameter(features[:, :, 1:].transpose(1, 2).contiguous().requires_grad_(True))
        new_scaling = nn.Parameter(scales.requires_grad_(True))
        new_rotation = nn.Parameter(rots.requires_grad_(True))
        new_opacity = nn.Parameter(opacities.requires_grad_(True))
        new_unique_kfIDs = (torch.ones((new_xyz.shape[0]), device=self.device) * kf_id).int()
        new_n_obs = torch.zeros((new_xyz.shape[0]), device=self.device).int()
        self.densification_postfix(new_xyz, new_features_dc, new_features_rest, new_opacity, new_scaling, new_rotation,
                                   new_kf_ids=new_unique_kfIDs, new_n_obs=new_n_obs)

    def extend_from_pcd_seq(self, cam_info, kf_id=-1, init=False, scale=2.0, depthmap=None):
        self.extend_from_pcd(*self.create_pcd_from_image(cam_info, init, scale=scale, depthmap=depthmap), kf_id)

    # ---- optimiser (gaussian_model.py:264-329)
    def training_setup(self, training_args):
        self.percent_dense = training_args.percent_dense
        n = self.get_xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros((n, 1), device=self.device)
        self.denom = torch.zeros((n, 1), device=self.device)
        l = [
            {"params": [self._xyz], "lr": training_args.position_lr_init * self.spatial_lr_scale, "name": "xyz"},
            {"params": [self._features_dc], "lr": training_args.feature_lr, "name": "f_dc"},
            {"params": [self._features_rest], "lr": training_args.feature_lr / 20.0, "name": "f_rest"},
            {"params": [self._opacity], "lr": training_args.opacity_lr, "name": "opacity"},
            {"params": [self._scaling], "lr": training_args.scaling_lr * self.spatial_lr_scale, "name": "scaling"},
            {"params": [self._rotation], "lr": training_args.rotation_lr, "name": "rotation"},
        ]
        if self.device.type == "cuda":        # same groups, same state dict, one launch per tensor (splat_slam_amd/optim.py)
            from splat_slam_amd.optim import FusedAdam
            self.optimizer = FusedAdam(l, lr=0.0, eps=1e-15)
        else:
            self.optimizer = torch.optim.Adam(l, lr=0.0, eps=1e-15)
        self.lr_init = training_args.position_lr_init * self.spatial_lr_scale
        self.lr_final = training_args.position_lr_final * self.spatial_lr_scale
        self.lr_delay_mult = training_args.position_lr_delay_mult
        self.max_steps = training_args.position_lr_max_steps

    def lr_at(self, iteration):
        """xyz learning rate update_learning_rate(iteration) would set (no side effect)."""
        return float(helper(iteration, lr_init=self.lr_init, lr_final=self.lr_final, lr_delay_mult=self.lr_delay_mult,
                            max_steps=self.max_steps))

    def update_learning_rate(self, iteration):
        for param_group in self.optimizer.param_groups:
            if param_group["name"] == "xyz":
                lr = helper(iteration, lr_init=self.lr_init, lr_final=self.lr_final, lr_delay_mult=self.lr_delay_mult,
                            max_steps=self.max_steps)
                param_group["lr"] = lr
                return lr

    # ---- PLY checkpoint (gaussian_model.py:331-380, 397-486): one all-float32 `vertex` element, binary little endian,
    # property order x y z nx ny nz f_dc_* f_rest_* opacity scale_* rot_*  (what plyfile writes for that dtype list)
    def construct_list_of_attributes(self):
        blocks = (("f_dc", self._features_dc.shape[1] * self._features_dc.shape[2]),
                  ("f_rest", self._features_rest.shape[1] * self._features_rest.shape[2]),
                  ("opacity", None), ("scale", self._scaling.shape[1]), ("rot", self._rotation.shape[1]))
        names = ["x", "y", "z", "nx", "ny", "nz"]
        for prefix, count in blocks:
            names += [prefix] if count is None else [f"{prefix}_{i}" for i in range(count)]
        return names

    def save_ply(self, path):
        import os
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        xyz = self._xyz.detach().cpu().numpy()
        f_dc = self._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
        f_rest = self._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
        attributes = np.concatenate((xyz, np.zeros_like(xyz), f_dc, f_rest, self._opacity.detach().cpu().numpy(),
                                     self._scaling.detach().cpu().numpy(), self._rotation.detach().cpu().numpy()), axis=1)
        names = self.construct_list_of_attributes()
        assert attributes.shape[1] == len(names)
        header = "ply\nformat binary_little_endian 1.0\nelement vertex {}\n".format(xyz.shape[0])
        header += "".join("property float {}\n".format(n) for n in names) + "end_header\n"
        with open(path, "wb") as f:
            f.write(header.encode("ascii"))
            f.write(np.ascontiguousarray(attributes, dtype="<f4").tobytes())

    def load_ply(self, path):
        with open(path, "rb") as f:
            names, n, fmt = [], 0, None
            line = f.readline().decode("ascii").strip()
            assert line == "ply", "not a PLY file"
            while True:
                line = f.readline().decode("ascii").strip()
                if line.startswith("format"):
                    fmt = line.split()[1]
                elif line.startswith("element vertex"):
                    n = int(line.split()[-1])
                elif line.startswith("property"):
                    assert line.split()[1] in ("float", "float32"), "only all-float vertex elements are supported"
                    names.append(line.split()[-1])
                elif line == "end_header":
                    break
            assert fmt == "binary_little_endian", "only binary_little_endian PLY is supported"
            data = np.frombuffer(f.read(n * len(names) * 4), dtype="<f4").reshape(n, len(names))
        col = {name: data[:, i] for i, name in enumerate(names)}
        xyz = np.stack((col["x"], col["y"], col["z"]), axis=1)
        features_dc = np.stack((col["f_dc_0"], col["f_dc_1"], col["f_dc_2"]), axis=1)[:, :, None]
        extra = sorted([k for k in names if k.startswith("f_rest_")], key=lambda x: int(x.split("_")[-1]))
        assert len(extra) == 3 * (self.max_sh_degree + 1) ** 2 - 3
        features_extra = np.stack([col[k] for k in extra], axis=1).reshape(n, 3, (self.max_sh_degree + 1) ** 2 - 1) \
            if extra else np.zeros((n, 3, 0), dtype=np.float32)
        scale_names = sorted([k for k in names if k.startswith("scale_")], key=lambda x: int(x.split("_")[-1]))
        rot_names = sorted([k for k in names if k.startswith("rot")], key=lambda x: int(x.split("_")[-1]))
        T = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float, device=self.device)
        self._xyz = nn.Parameter(T(xyz).requires_grad_(True))
        self._features_dc = nn.Parameter(T(features_dc).transpose(1, 2).contiguous().requires_grad_(True))
        self._features_rest = nn.Parameter(T(features_extra).transpose(1, 2).contiguous().requires_grad_(True))
        self._opacity = nn.Parameter(T(col["opacity"][:, None]).requires_grad_(True))
        self._scaling = nn.Parameter(T(np.stack([col[k] for k in scale_names], axis=1)).requires_grad_(True))
        self._rotation = nn.Parameter(T(np.stack([col[k] for k in rot_names], axis=1)).requires_grad_(True))
        self.active_sh_degree = self.max_sh_degree
        self.max_radii2D = torch.zeros((n), device=self.device)
        self.unique_kfIDs = torch.zeros((n), device=self.device).int()
        self.n_obs = torch.zeros((n), device=self.device).int()

    # ---- opacity resets (gaussian_model.py:382-395)
    def reset_opacity(self):
        opacities_new = inverse_sigmoid(torch.ones_like(self.get_opacity) * 0.01)
        self._opacity = self.replace_tensor_to_optimizer(opacities_new, "opacity")["opacity"]

    def reset_opacity_nonvisible(self, visibility_filters):
        opacities_new = inverse_sigmoid(torch.ones_like(self.get_opacity) * 0.4)
        for filter in visibility_filters:
            opacities_new[filter] = self.get_opacity[filter]
        self._opacity = self.replace_tensor_to_optimizer(opacities_new, "opacity")["opacity"]

    # ---- optimiser surgery (gaussian_model.py:488-593).  Every map edit is the same operation on the Adam groups: the
    # group's tensor becomes a fresh leaf Parameter, its two moments are transformed alongside, and the state dict (with
    # its `step`) moves to the new Parameter -- which is also why the new tensor has no .grad until the next backward.
    def _remap_groups(self, new_tensor, new_moment, only=None):
        out = {}
        for group in self.optimizer.param_groups:
            name = group["name"]
            if only is not None and name != only:
                continue
            old = group["params"][0]
            state = self.optimizer.state.pop(old, None)
            fresh = nn.Parameter(new_tensor(name, old).requires_grad_(True))
            if state is not None:
                state["exp_avg"] = new_moment(name, state["exp_avg"])
                state["exp_avg_sq"] = new_moment(name, state["exp_avg_sq"])
                self.optimizer.state[fresh] = state
            group["params"][0] = fresh
            out[name] = fresh
        return out

    def replace_tensor_to_optimizer(self, tensor, name):
        """New values for one group, moments reset to zero (opacity resets, map deformation)."""
        return self._remap_groups(lambda n, old: tensor, lambda n, m: torch.zeros_like(tensor), only=name)

    def _prune_optimizer(self, mask):
        """Rows where `mask` is True survive, in every group."""
        return self._remap_groups(lambda n, old: old[mask], lambda n, m: m[mask])

    def _adopt(self, t):
        self._xyz, self._features_dc, self._features_rest = t["xyz"], t["f_dc"], t["f_rest"]
        self._opacity, self._scaling, self._rotation = t["opacity"], t["scaling"], t["rotation"]

    def _select_rows_hip(self, mask, tensors):
        """[t[mask] for t in tensors] with ONE keep-list scan + ONE gather launch (bit-identical to boolean indexing)."""
        from splat_slam_amd import _native as nat
        lib, dev, n = nat.lib(), self._xyz.device, int(mask.shape[0])
        stream = torch.cuda.current_stream(dev).cuda_stream
        keep = mask.to(device=dev, dtype=torch.uint8).contiguous()
        src = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        scratch = torch.empty(lib.sgr_compact_scratch_bytes(n), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            nat.check(lib.sgr_keep_list(n, keep.data_ptr(), src.data_ptr(), cnt.data_ptr(), scratch.data_ptr(), scratch.numel(),
                                        stream), "sgr_keep_list")
        m = int(cnt.item())                       # (boolean indexing synchronises here as well)
        table, hold, outs = [], [], []
        for t in tensors:
            t = t.detach().to(dev).contiguous()
            out = torch.empty((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
            rb = (t.numel() // max(n, 1)) * t.element_size()
            if rb and m:
                table.append(nat.SgrRowTensor(t.data_ptr(), out.data_ptr(), rb))
                hold.append(t)
            outs.append(out)
        if table:
            arr = (nat.SgrRowTensor * len(table))(*table)
            with torch.cuda.device(dev):
                nat.check(lib.sgr_gather_rows(m, src.data_ptr(), len(table), arr, stream), "sgr_gather_rows")
        return outs

    def _prune_hip(self, valid):
        """prune_points on the device: ONE keep-list scan + ONE gather launch over every per-Gaussian tensor (parameters,
        both Adam moments, densification statistics, keyframe ids) instead of ~25 boolean-index kernels."""
        import ctypes as C
        from splat_slam_amd import _native as nat
        lib, dev, n = nat.lib(), self._xyz.device, int(valid.shape[0])
        stream = torch.cuda.current_stream(dev).cuda_stream
        keep = valid.to(device=dev, dtype=torch.uint8).contiguous()
        src = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        scratch = torch.empty(lib.sgr_compact_scratch_bytes(n), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            nat.check(lib.sgr_keep_list(n, keep.data_ptr(), src.data_ptr(), cnt.data_ptr(), scratch.data_ptr(), scratch.numel(),
                                        stream), "sgr_keep_list")
        m = int(cnt.item())                       # (the reference's boolean indexing synchronises here as well)
        table, hold = [], []

        def compact(t):
            t = t.contiguous()
            out = torch.empty((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
            rb = (t.numel() // max(n, 1)) * t.element_size()
            if rb and m:
                table.append(nat.SgrRowTensor(t.data_ptr(), out.data_ptr(), rb))
                hold.append(t)
            return out

        new_params = {}
        for group in self.optimizer.param_groups:
            p = group["params"][0]
            st = self.optimizer.state.get(p, None)
            np_ = compact(p.detach())
            if st is not None:
                st["exp_avg"], st["exp_avg_sq"] = compact(st["exp_avg"]), compact(st["exp_avg_sq"])
                del self.optimizer.state[p]
            group["params"][0] = nn.Parameter(np_.requires_grad_(True))
            if st is not None:
                self.optimizer.state[group["params"][0]] = st
            new_params[group["name"]] = group["params"][0]
        aux = [compact(getattr(self, a).to(dev)) for a in ("xyz_gradient_accum", "denom", "max_radii2D", "unique_kfIDs", "n_obs")]
        if table:
            arr = (nat.SgrRowTensor * len(table))(*table)
            with torch.cuda.device(dev):
                nat.check(lib.sgr_gather_rows(m, src.data_ptr(), len(table), arr, stream), "sgr_gather_rows")
        self._adopt(new_params)
        self.xyz_gradient_accum, self.denom, self.max_radii2D, self.unique_kfIDs, self.n_obs = aux

    def prune_points(self, mask):
        valid = ~mask
        if self._xyz.is_cuda and self.optimizer is not None and self.use_hip_compaction:
            return self._prune_hip(valid)
        self._adopt(self._prune_optimizer(valid))
        self.xyz_gradient_accum = self.xyz_gradient_accum[valid]
        self.denom = self.denom[valid]
        self.max_radii2D = self.max_radii2D[valid]
        self.unique_kfIDs = self.unique_kfIDs[valid]
        self.n_obs = self.n_obs[valid]

    def cat_tensors_to_optimizer(self, tensors_dict):
        """Appends rows to every group; the new rows start with zero moments."""
        assert all(len(g["params"]) == 1 for g in self.optimizer.param_groups)
        return self._remap_groups(lambda n, old: torch.cat((old, tensors_dict[n]), dim=0),
                                  lambda n, m: torch.cat((m, torch.zeros_like(tensors_dict[n])), dim=0))

    def densification_postfix(self, new_xyz, new_features_dc, new_features_rest, new_opacities, new_scaling,
                              new_rotation, new_kf_ids=None, new_n_obs=None):
        d = {"xyz": new_xyz, "f_dc": new_features_dc, "f_rest": new_features_rest, "opacity": new_opacities,
             "scaling": new_scaling, "rotation": new_rotation}
        if self.optimizer is None:
            # first extension happens before training_setup in the reference too (mapper.py:959-962): the tensors
            # simply become the parameters
            t = {k: (nn.Parameter(torch.cat((getattr(self, a), v), dim=0).requires_grad_(True)) if getattr(self, a).numel()
                     else nn.Parameter(v.detach().clone().requires_grad_(True)))
                 for (k, v), a in zip(d.items(), ["_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"])}
            self._adopt(t)
        else:
            self._adopt(self.cat_tensors_to_optimizer(d))
        n = self.get_xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros((n, 1), device=self.device)
        self.denom = torch.zeros((n, 1), device=self.device)
        self.max_radii2D = torch.zeros((n), device=self.device)
        if new_kf_ids is not None:
            self.unique_kfIDs = torch.cat((self.unique_kfIDs, new_kf_ids.to(self.device))).int()
        if new_n_obs is not None:
            self.n_obs = torch.cat((self.n_obs, new_n_obs.to(self.device))).int()

    # ---- densify / prune (gaussian_model.py:639-742)
    def densify_and_split(self, grads, grad_threshold, scene_extent, N=2):
        n_init_points = self.get_xyz.shape[0]
        padded_grad = torch.zeros((n_init_points), device=self.device)
        padded_grad[: grads.shape[0]] = grads.squeeze()
        selected = padded_grad >= grad_threshold
        selected = torch.logical_and(selected, torch.max(self.get_scaling, dim=1).values > self.percent_dense * scene_extent)
        stds = self.get_scaling[selected].repeat(N, 1)
        means = torch.zeros((stds.size(0), 3), device=self.device)
        samples = torch.normal(mean=means, std=stds)
        rots = build_rotation(self._rotation[selected]).repeat(N, 1, 1)
        new_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self.get_xyz[selected].repeat(N, 1)
        new_scaling = torch.log(self.get_scaling[selected].repeat(N, 1) / (0.8 * N))
        new_rotation = self._rotation[selected].repeat(N, 1)
        new_features_dc = self._features_dc[selected].repeat(N, 1, 1)
        new_features_rest = self._features_rest[selected].repeat(N, 1, 1)
        new_opacity = self._opacity[selected].repeat(N, 1)
        new_kf_id = self.unique_kfIDs[selected].repeat(N)
        new_n_obs = self.n_obs[selected].repeat(N)
        self.densification_postfix(new_xyz, new_features_dc, new_features_rest, new_opacity, new_scaling, new_rotation,
                                   new_kf_ids=new_kf_id, new_n_obs=new_n_obs)
        prune_filter = torch.cat((selected, torch.zeros(N * int(selected.sum()), device=self.device, dtype=bool)))
        self.prune_points(prune_filter)

    def densify_and_clone(self, grads, grad_threshold, scene_extent):
        selected = torch.norm(grads, dim=-1) >= grad_threshold
        selected = torch.logical_and(selected, torch.max(self.get_scaling, dim=1).values <= self.percent_dense * scene_extent)
        if self._xyz.is_cuda and self.use_hip_compaction:
            rows = self._select_rows_hip(selected, [self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling,
                                                    self._rotation, self.unique_kfIDs, self.n_obs])
            self.densification_postfix(*rows[:6], new_kf_ids=rows[6], new_n_obs=rows[7])
            return
        self.densification_postfix(self._xyz[selected], self._features_dc[selected], self._features_rest[selected],
                                   self._opacity[selected], self._scaling[selected], self._rotation[selected],
                                   new_kf_ids=self.unique_kfIDs[selected], new_n_obs=self.n_obs[selected])

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size):
        """Clone small / split large Gaussians whose mean screen-space gradient reached `max_grad`, then drop the
        near-transparent ones and -- when a screen-size limit is given -- those that got too large on screen (radius) or in
        the world (largest axis > 10 % of the extent); gaussian_model.py:721-736."""
        mean_grad = self.xyz_gradient_accum / self.denom
        mean_grad = torch.where(mean_grad.isnan(), torch.zeros_like(mean_grad), mean_grad)      # never observed: 0 / 0
        self.densify_and_clone(mean_grad, max_grad, extent)
        self.densify_and_split(mean_grad, max_grad, extent)
        doomed = (self.get_opacity < min_opacity).squeeze()
        if max_screen_size:
            too_wide_on_screen = self.max_radii2D > max_screen_size
            too_wide_in_world = self.get_scaling.max(dim=1).values > 0.1 * extent
            doomed = doomed | too_wide_on_screen | too_wide_in_world
        self.prune_points(doomed)

    def add_view_stats(self, viewspace_point_tensor, radii):
        """add_densification_stats + the max_radii2D update of one view (gaussian_model.py:738-742, mapper.py:522-529) as ONE
        launch on the GPU (sgr_densify_stats); returns False when the caller has to use the torch formulation."""
        g = viewspace_point_tensor.grad
        if (g is None or not g.is_cuda or g.dtype is not torch.float32 or not g.is_contiguous() or radii.dtype is not torch.int32
                or self.max_radii2D.dtype is not torch.float32 or not radii.is_contiguous()):
            return False
        from splat_slam_amd import _native as nat
        n = g.shape[0]
        nat.check(nat.lib().sgr_densify_stats(n, g.data_ptr(), radii.data_ptr(), self.xyz_gradient_accum.data_ptr(),
                                              self.denom.data_ptr(), self.max_radii2D.data_ptr(),
                                              torch.cuda.current_stream(g.device).cuda_stream), "sgr_densify_stats")
        return True

    def add_views_stats(self, views):
        """The same for every view of an iteration [(viewspace_points, radii), ...] in ONE call into the drop-in package's C++ half
        (a launch per view, no interpreter round trip per view); False: the caller loops over add_view_stats / the torch form."""
        import diff_gaussian_rasterization as drg
        ext = getattr(drg, "native_extension", lambda: None)()       # (tests inject the oracle under this module name)
        if ext is None or not views or self.max_radii2D.dtype is not torch.float32:
            return False
        grads, radii = [], []
        for vsp, r in views:
            g = vsp.grad
            if (g is None or not g.is_cuda or g.dtype is not torch.float32 or not g.is_contiguous() or r.dtype is not torch.int32
                    or not r.is_contiguous() or r.shape[0] != self.xyz_gradient_accum.shape[0] or g.shape[0] != r.shape[0]):
                return False
            grads.append(g)
            radii.append(r)
        ext.densify_stats_views(grads, radii, self.xyz_gradient_accum, self.denom, self.max_radii2D)
        return True

    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        """accum[visible] += |dL/d(mean2D)|, denom[visible] += 1 (gaussian_model.py:738-742), mask-free (no nonzero() sync)."""
        seen = update_filter.unsqueeze(-1)
        norm = torch.norm(viewspace_point_tensor.grad[:, :2], dim=-1, keepdim=True)
        self.xyz_gradient_accum += torch.where(seen, norm, torch.zeros_like(norm))
        self.denom += seen.to(self.denom.dtype)
