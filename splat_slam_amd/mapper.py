"""Mapping optimisation loops -- mirror of the three loops of /root/reference/src/mapper.py:
  initialize_map :303-398,  map :400-614,  final_refine :656-708, and the keyframe-optimiser set-up :1067-1111.

The loop bodies keep the reference's order of operations (and its quirks, SURVEY.md 3.7: the prune pass that returns
before optimizer.step, no lr update during initialisation, numpy RNG in final_refine vs torch RNG in map).
Everything that the reference's Mapper does around these loops -- the tracker pipe, DepthVideo, map deformation,
keyframe selection -- is outside this file; `splat_slam_amd.synthetic` feeds the loops with a mapping-only stream.
"""
import numpy as np
import torch

from splat_slam_amd.camera import getProjectionMatrix2
from splat_slam_amd.gaussian_model import GaussianModel, OptParams
from splat_slam_amd.losses import get_loss_mapping, get_loss_mapping_fused
from splat_slam_amd.pose import update_pose
from splat_slam_amd.renderer import render


class PipelineParams:
    convert_SHs_python = False
    compute_cov3D_python = False


class MappingLoop:
    def __init__(self, config, device="cuda:0", fused_loss=True, knn_fn=None):
        self.config = config
        self.device = torch.device(device)
        tr = config["mapping"]["Training"]
        self.opt_params = OptParams(config["mapping"].get("opt_params"))
        self.pipeline_params = PipelineParams()
        self.iteration_count = 0
        self.occ_aware_visibility = {}
        self.viewpoints = {}
        self.current_window = []
        self.keyframe_optimizers = None
        sh_degree = 3 if tr.get("spherical_harmonics", False) else 0        # mapper.py:85
        self.gaussians = GaussianModel(sh_degree, config=config, device=device, knn_fn=knn_fn)
        self.gaussians.init_lr(6.0)                                          # mapper.py:87
        self.gaussians.training_setup(self.opt_params)
        self.background = torch.tensor([0, 0, 0], dtype=torch.float32, device=self.device)
        self.cameras_extent = 6.0                                            # mapper.py:93
        self.init_itr_num = tr["init_itr_num"]
        self.init_gaussian_update = tr["init_gaussian_update"]
        self.init_gaussian_reset = tr["init_gaussian_reset"]
        self.init_gaussian_th = tr["init_gaussian_th"]
        self.init_gaussian_extent = self.cameras_extent * tr["init_gaussian_extent"]
        self.mapping_itr_num = tr["mapping_itr_num"]
        self.gaussian_update_every = tr["gaussian_update_every"]
        self.gaussian_update_offset = tr["gaussian_update_offset"]
        self.gaussian_th = tr["gaussian_th"]
        self.gaussian_extent = self.cameras_extent * tr["gaussian_extent"]
        self.gaussian_reset = tr["gaussian_reset"]
        self.size_threshold = tr["size_threshold"]
        self.window_size = tr["window_size"]
        self.loss_fn = get_loss_mapping_fused if fused_loss else get_loss_mapping
        self.grad_sync = None

    # mapper.py:841-850
    def projection_matrix(self, intr):
        return getProjectionMatrix2(znear=0.01, zfar=100.0, fx=intr["fx"], fy=intr["fy"], cx=intr["cx"], cy=intr["cy"],
                                    W=intr["W"], H=intr["H"]).transpose(0, 1).to(device=self.device)

    def add_next_kf(self, frame_idx, viewpoint, init=False, scale=2.0, depth_map=None):
        self.gaussians.extend_from_pcd_seq(viewpoint, kf_id=frame_idx, init=init, scale=scale, depthmap=depth_map)

    def reset(self):
        self.iteration_count = 0
        self.occ_aware_visibility = {}
        self.viewpoints = {}
        self.current_window = []
        self.keyframe_optimizers = None
        self.gaussians.prune_points(self.gaussians.unique_kfIDs >= 0)

    @torch.no_grad()
    def render_forward(self, viewpoint):
        """Forward-only render used for keyframe selection (mapper.py:972-978)."""
        return render(viewpoint, self.gaussians, self.pipeline_params, self.background)

    def build_keyframe_optimizers(self):
        """A fresh keyframe optimiser for the current window (mapper.py:1067-1111): exposure a, b of every window keyframe
        but the first frame at lr 0.01, pose deltas of the first `pose_window` ones (only with mapping.BA) at half their
        configured rates.  The reference makes one param group per tensor; groups with identical hyper-parameters are merged
        here -- the same update per parameter (Adam state is per parameter), but torch's foreach Adam then needs ~12 launches
        for all exposures instead of ~12 per tensor (240 tiny launches per iteration for a 10-keyframe window)."""
        tr = self.config["mapping"]["Training"]
        pose_opt = bool(self.config["mapping"]["BA"]) and not tr.get("gt_camera", False)
        exposures, groups = [], []
        for cam_idx, kf in enumerate(self.current_window):
            if kf == 0:
                continue
            viewpoint = self.viewpoints[kf]
            if pose_opt and cam_idx < tr["pose_window"]:
                groups.append({"params": [viewpoint.cam_rot_delta], "lr": tr["lr"]["cam_rot_delta"] * 0.5, "name": f"rot_{viewpoint.uid}"})
                groups.append({"params": [viewpoint.cam_trans_delta], "lr": tr["lr"]["cam_trans_delta"] * 0.5, "name": f"trans_{viewpoint.uid}"})
            exposures += [viewpoint.exposure_a, viewpoint.exposure_b]
        if exposures:
            groups.append({"params": exposures, "lr": 0.01, "name": "exposure"})
        if groups and self.device.type == "cuda":      # same groups and state; all exposure tensors step in ONE launch
            from splat_slam_amd.optim import FusedAdam
            self.keyframe_optimizers = FusedAdam(groups)
        else:
            self.keyframe_optimizers = torch.optim.Adam(groups) if groups else None

    def _backward(self, make_loss):
        """loss = make_loss(); loss.backward() -- once more from scratch if the rasterizer reports that a forward of this iteration
        exceeded its pair capacity.  (Only the Python-node path of the drop-in package raises: inputs outside the reference's call
        shape.  It has produced no gradient for the batch that raised and has already grown the capacity; batches of one that ran
        before it may have accumulated into `.grad`.  The second attempt must start from the gradients the iteration STARTED with --
        a prune pass deliberately leaves its gradients behind for the next step, mapper.py:490-520 -- so those are kept aside first
        and put back, and only what the failed attempt produced is dropped: ADVICE r3 / r4.  The C++ nodes re-run a truncated
        forward inside backward() themselves.)"""
        import diff_gaussian_rasterization as drg
        params = [p for g in self.gaussians.optimizer.param_groups for p in g["params"]]
        if self.keyframe_optimizers is not None:
            params += [p for g in self.keyframe_optimizers.param_groups for p in g["params"]]
        before = {id(p): p.grad.detach().clone() for p in params if p.grad is not None}     # (empty in a regular iteration: free)
        for attempt in (0, 1):
            out = make_loss()
            try:
                out[0].backward()
                return out
            except RuntimeError as e:
                if attempt or not drg.is_capacity_overflow(e):
                    raise
                for p in params:
                    p.grad = before.get(id(p))

    def _visible_stats(self, viewspace_points, vis, radii):
        """max_radii2D[vis] = max(max_radii2D[vis], radii[vis]) and add_densification_stats (mapper.py:332-335,523-529,
        gaussian_model.py:738-742) written without boolean-mask indexing: the same values element for element, but a mask
        index is a nonzero() -- a host synchronisation and three extra kernels -- per statement, 36 of them per 12-view
        iteration on the GPU."""
        gm = self.gaussians
        if gm.add_view_stats(viewspace_points, radii):          # GPU: one launch for the three updates
            return
        gm.max_radii2D = torch.where(vis, torch.max(gm.max_radii2D, radii), gm.max_radii2D)
        gm.add_densification_stats(viewspace_points, vis)

    # ---------------------------------------------------------------------------------- mapper.py:303-353
    def initialize_map(self, cur_frame_idx, viewpoint, iters=None):
        n_touched = None
        for mapping_iteration in range(self.init_itr_num if iters is None else iters):
            self.iteration_count += 1
            def init_pass():
                pkg = render(viewpoint, self.gaussians, self.pipeline_params, self.background)
                return self.loss_fn(self.config["mapping"], pkg["render"], pkg["depth"], viewpoint, pkg["opacity"], initialization=True), pkg
            loss_init, pkg = self._backward(init_pass)
            vsp, vis, radii, n_touched = pkg["viewspace_points"], pkg["visibility_filter"], pkg["radii"], pkg["n_touched"]
            with torch.no_grad():
                self._visible_stats(vsp, vis, radii)
                if mapping_iteration % self.init_gaussian_update == 0:
                    self.gaussians.densify_and_prune(self.opt_params.densify_grad_threshold, self.init_gaussian_th,
                                                     self.init_gaussian_extent, None)
                if self.iteration_count == self.init_gaussian_reset or (
                        self.iteration_count == self.opt_params.densify_from_iter):
                    self.gaussians.reset_opacity()
                self.gaussians.optimizer.step()
                self.gaussians.optimizer.zero_grad(set_to_none=True)
        self.occ_aware_visibility[cur_frame_idx] = (n_touched > 0).long()
        return pkg

    def _count_observations(self, current_window):
        """mapper.py:503-508: with a full window the prune pass leaves in `n_obs` how many window keyframes see each Gaussian
        (the mask it then derives from it is dropped, the counts stay in the model)."""
        if len(current_window) == self.window_size:
            n_obs = self.gaussians.n_obs                 # (one stack + sum + cast instead of two launches per window keyframe)
            vis = [v.to(n_obs.device) for v in self.occ_aware_visibility.values()]
            if vis:
                n_obs.copy_(torch.stack(vis).sum(dim=0).to(n_obs.dtype))
            else:
                n_obs.fill_(0)

    # ---------------------------------------------------------------------------------- mapper.py:400-568
    def map(self, current_window, prune=False, iters=1):
        if len(current_window) == 0:
            return
        viewpoint_stack = [self.viewpoints[kf_idx] for kf_idx in current_window]
        frames_to_optimize = self.config["mapping"]["Training"]["pose_window"]
        current_window_set = set(current_window)
        random_viewpoint_stack = [v for idx, v in self.viewpoints.items() if idx not in current_window_set]
        pose_opt = bool(self.config["mapping"]["BA"]) and not self.config["mapping"]["Training"].get("gt_camera", False)
        gaussian_split = False
        mapping_cfg = self.config["mapping"]

        def view_pass(cam):
            """render + mapping loss of one keyframe (mapper.py:426-456 / :458-485): the loss term and what the statistics need."""
            pkg = render(cam, self.gaussians, self.pipeline_params, self.background)
            term = self.loss_fn(mapping_cfg, pkg["render"], pkg["depth"], cam, pkg["opacity"])
            return term, (pkg["viewspace_points"], pkg["visibility_filter"], pkg["radii"]), pkg["n_touched"]

        for it in range(iters):
            gaussian_split = False                      # (per iteration, mapper.py:491: the LAST iteration decides the result)
            self.iteration_count += 1
            # the window keyframes, then two keyframes drawn from the rest (torch RNG, mapper.py:470)
            extra = [random_viewpoint_stack[k] for k in torch.randperm(len(random_viewpoint_stack))[:2]]

            def iteration_loss():
                passes = [view_pass(cam) for cam in viewpoint_stack + extra]
                loss_mapping = 0
                for term, _, _ in passes:
                    loss_mapping += term
                scaling = self.gaussians.get_scaling
                isotropic_loss = torch.abs(scaling - scaling.mean(dim=1).view(-1, 1))
                # multi-GPU (grad_sync set): the view losses are summed over the ranks, the isotropy term must enter that sum ONCE
                iso_weight = 10.0 if self.grad_sync is None else 10.0 / self.grad_sync.world
                loss_mapping += iso_weight * isotropic_loss.mean()
                return loss_mapping, passes
            _, passes = self._backward(iteration_loss)
            per_view = [p[1] for p in passes]
            n_touched_acm = [p[2] for p in passes[:len(current_window)]]
            if self.grad_sync is not None and not prune:      # (the prune pass never steps: nothing to exchange)
                self.grad_sync.reduce()
            with torch.no_grad():
                # the reference rebuilds this dict every iteration (mapper.py:494-498); it is only ever READ by the prune branch
                # below and by the caller after map() returns: build it when it can be observed (20 launches per iteration saved)
                if prune or it == iters - 1:
                    self.occ_aware_visibility = {}
                    for idx in range(len(current_window)):
                        self.occ_aware_visibility[current_window[idx]] = (n_touched_acm[idx] > 0).long()
                if prune:
                    # the reference computes `to_prune` here and drops it (mapper.py:502-520): a pass that refreshes
                    # occ_aware_visibility and n_obs and returns before optimizer.step()/zero_grad()
                    self._count_observations(current_window)
                    return False
                if not self.gaussians.add_views_stats([(vsp, radii) for vsp, _, radii in per_view]):    # (GPU: one call for all views)
                    for vsp, vis, radii in per_view:
                        self._visible_stats(vsp, vis, radii)
                update_gaussian = self.iteration_count % self.gaussian_update_every == self.gaussian_update_offset
                if update_gaussian:
                    self.gaussians.densify_and_prune(self.opt_params.densify_grad_threshold, self.gaussian_th,
                                                     self.gaussian_extent, self.size_threshold)
                    gaussian_split = True
                if (self.iteration_count % self.gaussian_reset) == 0 and (not update_gaussian):
                    self.gaussians.reset_opacity_nonvisible([v[1] for v in per_view])
                    gaussian_split = True
                self.gaussians.optimizer.step()
                self.gaussians.optimizer.zero_grad(set_to_none=True)
                self.gaussians.update_learning_rate(self.iteration_count)
                if self.keyframe_optimizers is not None:
                    self.keyframe_optimizers.step()
                    self.keyframe_optimizers.zero_grad(set_to_none=True)
                if pose_opt:       # with mapping.BA False the deltas are never stepped: update_pose is the identity
                    for cam_idx in range(min(frames_to_optimize, len(current_window))):
                        viewpoint = viewpoint_stack[cam_idx]
                        if viewpoint.uid == 0:
                            continue
                        update_pose(viewpoint)
        return gaussian_split

    # ---------------------------------------------------------------------------------- mapper.py:649-708
    def final_refine(self, iters=26000):
        random_viewpoint_stack = list(self.viewpoints.values())
        for _ in range(iters):
            self.iteration_count += 1
            rand_idx = np.random.randint(0, len(random_viewpoint_stack))
            viewpoint = random_viewpoint_stack[rand_idx]
            def refine_pass(viewpoint=viewpoint):
                pkg = render(viewpoint, self.gaussians, self.pipeline_params, self.background)
                return (self.loss_fn(self.config["mapping"], pkg["render"], pkg["depth"], viewpoint, pkg["opacity"]),)
            self._backward(refine_pass)
            with torch.no_grad():
                self.gaussians.optimizer.step()
                self.gaussians.optimizer.zero_grad(set_to_none=True)
                self.gaussians.update_learning_rate(self.iteration_count)
                if self.keyframe_optimizers is not None:
                    self.keyframe_optimizers.step()
                    self.keyframe_optimizers.zero_grad(set_to_none=True)
