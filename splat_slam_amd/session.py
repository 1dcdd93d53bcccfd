"""Per-keyframe mapping logic -- mirror of the body of Mapper.run, /root/reference/src/mapper.py:876-1116, with the
tracker pipe / DepthVideo / dataset replaced by a pull-style call: `process(video_idx, idx, color, depth, w2c)` is what
one `{"video_idx", "timestamp"}` pipe message plus `get_w2c_and_depth` and `frame_reader[idx]` deliver.

Includes the keyframe management that consumes the rasterizer's `n_touched` / `opacity` / `depth` outputs:
`is_keyframe` (:744-772), `add_to_window` (:774-831), median depth (slam_utils.py:108-119), and the map deformation
call for past keyframes when the pose source moved them (:1021-1055).
"""
import numpy as np
import torch

from splat_slam_amd.camera import Camera, focal2fov, getProjectionMatrix2, getWorld2View2
from splat_slam_amd.deform import update_mapping_points
from splat_slam_amd.losses import get_median_depth


class MappingSession:
    def __init__(self, loop, intr, pose_source=None):
        """loop: MappingLoop or FusedMappingLoop.  intr: dict W,H,fx,fy,cx,cy.
        pose_source(video_idx) -> (w2c[4,4], depth[H,W]) or None: the tracker's refined estimate for a past keyframe."""
        self.loop = loop
        self.config = loop.config
        self.device = loop.device
        self.intr = intr
        self.pose_source = pose_source
        self.projection_matrix = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=intr["fx"], fy=intr["fy"], cx=intr["cx"],
                                                      cy=intr["cy"], W=intr["W"], H=intr["H"]).transpose(0, 1).to(self.device)
        self.intrinsics = torch.tensor([[intr["fx"], 0, intr["cx"]], [0, intr["fy"], intr["cy"]], [0, 0, 1.0]], device=self.device)
        self.cameras, self.is_kf, self.depth_dict = {}, {}, {}
        self.keyframe_idxs, self.video_idxs = [], []
        self.init = True
        self.median_depth = 1.0
        self.move_points = self.config["mapping"].get("move_points", True)

    # ---- mapper.py:744-772
    def is_keyframe(self, cur_frame_idx, last_keyframe_idx, cur_vis, occ_aware_visibility):
        tr = self.config["mapping"]["Training"]
        curr_frame, last_kf = self.cameras[cur_frame_idx], self.cameras[last_keyframe_idx]
        pose_CW = getWorld2View2(curr_frame.R, curr_frame.T)
        last_kf_WC = torch.linalg.inv(getWorld2View2(last_kf.R, last_kf.T))
        dist = torch.norm((pose_CW @ last_kf_WC)[0:3, 3])
        dist_check = dist > tr["kf_translation"] * self.median_depth
        dist_check2 = dist > tr["kf_min_translation"] * self.median_depth
        union = torch.logical_or(cur_vis, occ_aware_visibility[last_keyframe_idx]).count_nonzero()
        intersection = torch.logical_and(cur_vis, occ_aware_visibility[last_keyframe_idx]).count_nonzero()
        point_ratio_2 = intersection / union
        return bool((point_ratio_2 < tr["kf_overlap"] and dist_check2) or dist_check)

    # ---- mapper.py:774-831
    def add_to_window(self, cur_frame_idx, cur_vis, occ_aware_visibility, window):
        N_dont_touch = 2
        window = [cur_frame_idx] + window
        curr_frame = self.cameras[cur_frame_idx]
        to_remove, removed_frame = [], None
        cut_off = self.config["mapping"]["Training"].get("kf_cutoff", 0.4)
        for i in range(N_dont_touch, len(window)):
            kf_idx = window[i]
            intersection = torch.logical_and(cur_vis, occ_aware_visibility[kf_idx]).count_nonzero()
            denom = min(cur_vis.count_nonzero(), occ_aware_visibility[kf_idx].count_nonzero())
            if intersection / denom <= cut_off:
                to_remove.append(kf_idx)
        if to_remove:
            window.remove(to_remove[-1])
            removed_frame = to_remove[-1]
        kf_0_WC = torch.linalg.inv(getWorld2View2(curr_frame.R, curr_frame.T))
        if len(window) > self.loop.window_size:
            inv_dist = []
            for i in range(N_dont_touch, len(window)):
                inv_dists = []
                kf_i = self.cameras[window[i]]
                kf_i_CW = getWorld2View2(kf_i.R, kf_i.T)
                for j in range(N_dont_touch, len(window)):
                    if i == j:
                        continue
                    kf_j = self.cameras[window[j]]
                    kf_j_WC = torch.linalg.inv(getWorld2View2(kf_j.R, kf_j.T))
                    T_CiCj = kf_i_CW @ kf_j_WC
                    inv_dists.append(1.0 / (torch.norm(T_CiCj[0:3, 3]) + 1e-6).item())
                T_CiC0 = kf_i_CW @ kf_0_WC
                k = torch.sqrt(torch.norm(T_CiC0[0:3, 3])).item()
                inv_dist.append(k * sum(inv_dists))
            idx = int(np.argmax(inv_dist))
            removed_frame = window[N_dont_touch + idx]
            window.remove(removed_frame)
        return window, removed_frame

    def _camera(self, video_idx, color, depth, w2c):
        intr = self.intr
        cam = Camera(video_idx, color.to(self.device).float().contiguous(), depth, w2c.to(self.device), self.projection_matrix,
                     intr["fx"], intr["fy"], intr["cx"], intr["cy"], focal2fov(intr["fx"], intr["W"]),
                     focal2fov(intr["fy"], intr["H"]), intr["H"], intr["W"], device=self.device)
        cam.update_RT(cam.R_gt, cam.T_gt)                 # mapper.py:939: the tracked pose becomes the camera pose
        return cam

    def process(self, video_idx, idx, color, depth, w2c):
        """One tracker message.  Returns "init", "mapped" or "skipped" (not a keyframe)."""
        loop = self.loop
        self.keyframe_idxs.append(idx)
        self.video_idxs.append(video_idx)
        depth = torch.as_tensor(depth, dtype=torch.float32, device=self.device)
        viewpoint = self._camera(video_idx, color, depth, w2c)
        self.cameras[video_idx] = viewpoint
        if self.init:
            loop.reset()
            loop.current_window.append(video_idx)
            self.depth_dict[video_idx] = depth
            self.is_kf[video_idx] = True
            loop.viewpoints[video_idx] = viewpoint
            loop.add_next_kf(video_idx, viewpoint, depth_map=depth, init=True)
            loop.initialize_map(video_idx, viewpoint)
            self.init = False
            return "init"
        pkg = loop.render_forward(viewpoint)
        self.median_depth = get_median_depth(pkg["depth"], pkg["opacity"])
        last_keyframe_idx = loop.current_window[0]
        curr_visibility = (pkg["n_touched"] > 0).long()
        create_kf = self.is_keyframe(video_idx, last_keyframe_idx, curr_visibility, loop.occ_aware_visibility)
        if len(loop.current_window) < loop.window_size:
            occ = loop.occ_aware_visibility[last_keyframe_idx]
            union = torch.logical_or(curr_visibility, occ).count_nonzero()
            intersection = torch.logical_and(curr_visibility, occ).count_nonzero()
            create_kf = bool(intersection / union < self.config["mapping"]["Training"]["kf_overlap"])
        if not create_kf:
            self.is_kf[video_idx] = False
            return "skipped"
        loop.current_window, _ = self.add_to_window(video_idx, curr_visibility, loop.occ_aware_visibility, loop.current_window)
        self.is_kf[video_idx] = True
        # past keyframes the tracker moved since they were mapped: deform the Gaussians anchored to them
        last_idx = self.keyframe_idxs[-1]
        if self.pose_source is not None:
            for keyframe_idx, frame_idx in zip(self.video_idxs, self.keyframe_idxs):
                upd = self.pose_source(keyframe_idx)
                if upd is None:
                    continue
                w2c_temp, depth_temp = upd
                w2c_temp = w2c_temp.to(self.device)
                depth_temp = torch.as_tensor(depth_temp, dtype=torch.float32, device=self.device)
                if keyframe_idx not in self.depth_dict and self.is_kf.get(keyframe_idx, False):
                    self.depth_dict[keyframe_idx] = depth_temp
                if frame_idx == last_idx:
                    continue
                cam = self.cameras[keyframe_idx]
                w2c_old = torch.eye(4, device=self.device)
                w2c_old[:3, :3], w2c_old[:3, 3] = cam.R, cam.T
                cam.update_RT(w2c_temp[:3, :3], w2c_temp[:3, 3])
                cam.depth = depth_temp
                if self.move_points and self.is_kf.get(keyframe_idx, False):
                    update_mapping_points(loop.gaussians, keyframe_idx, w2c_temp, w2c_old, depth_temp,
                                          self.depth_dict[keyframe_idx], self.intrinsics)
                    self.depth_dict[keyframe_idx] = depth_temp
        loop.viewpoints[video_idx] = viewpoint
        loop.add_next_kf(video_idx, viewpoint, depth_map=depth, init=False)
        loop.build_keyframe_optimizers()
        loop.map(loop.current_window, iters=loop.mapping_itr_num)
        loop.map(loop.current_window, prune=True)
        return "mapped"

    def finish(self, refine_iters=None):
        """SLAM.terminate's mapping part (slam.py:179-181): final refinement, then the per-keyframe PSNR."""
        from splat_slam_amd.eval import eval_rendering_psnr
        from splat_slam_amd.mapper import PipelineParams
        iters = self.config["mapping"]["final_refine_iters"] if refine_iters is None else refine_iters
        if iters:
            self.loop.final_refine(iters=iters)
        frames = [self.loop.viewpoints[k] for k in sorted(self.loop.viewpoints)]
        return eval_rendering_psnr(frames, self.loop.gaussians, PipelineParams(), self.loop.background)
