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

    # ---- keyframe management on the consumers of n_touched (mapper.py:744-831), batched over the window
    def _w2c(self, idxs):
        # the cameras cache getWorld2View2(R, T) (transposed, camera.py _matrices): the same bits without two linalg.inv launches per
        # camera and decision (16 small launches per keyframe with the GPU idle behind the keyframe-selection read-back)
        return torch.stack([self.cameras[k].world_view_transform.transpose(0, 1) for k in idxs])

    @staticmethod
    def _shared(vis_a, vis_b):
        return torch.logical_and(vis_a, vis_b).count_nonzero()

    def is_keyframe(self, cur_frame_idx, last_keyframe_idx, cur_vis, occ_aware_visibility):
        """New keyframe if the camera moved far (relative to the median depth), or moved a little AND sees a set of
        Gaussians whose IoU with the last keyframe's dropped below kf_overlap (mapper.py:744-772)."""
        tr = self.config["mapping"]["Training"]
        cur, last = self._w2c([cur_frame_idx, last_keyframe_idx])
        baseline = torch.norm((cur @ torch.linalg.inv(last))[:3, 3])
        last_vis = occ_aware_visibility[last_keyframe_idx]
        iou = self._shared(cur_vis, last_vis) / torch.logical_or(cur_vis, last_vis).count_nonzero()
        far = baseline > tr["kf_translation"] * self.median_depth
        moved = baseline > tr["kf_min_translation"] * self.median_depth
        return bool(far | (moved & (iou < tr["kf_overlap"])))          # ONE read-back (the three tests as one device expression)

    def add_to_window(self, cur_frame_idx, cur_vis, occ_aware_visibility, window):
        """Pushes the new keyframe in front; the two newest entries are never evicted.  First the OLDEST window entry whose
        overlap coefficient (Szymkiewicz-Simpson) with the new frame is <= kf_cutoff goes; if the window is still too long,
        the entry that is most redundant -- sqrt(distance to the new frame) * sum of inverse distances to the others --
        goes (mapper.py:774-831)."""
        keep_newest = 2
        window = [cur_frame_idx] + window
        cut_off = self.config["mapping"]["Training"].get("kf_cutoff", 0.4)
        removed_frame = None
        n_cur = cur_vis.count_nonzero()
        older = window[keep_newest:]
        low_overlap = []
        if older:                                   # every window entry's overlap coefficient in one pass, ONE read-back
            occ = torch.stack([occ_aware_visibility[k] for k in older]) != 0
            shared = torch.logical_and(occ, (cur_vis != 0)[None]).count_nonzero(dim=1)
            ratio = shared / torch.minimum(n_cur, occ.count_nonzero(dim=1))
            low_overlap = [k for k, low in zip(older, (ratio <= cut_off).tolist()) if low]
        if low_overlap:
            removed_frame = low_overlap[-1]
            window.remove(removed_frame)
        if len(window) > self.loop.window_size:
            cand = window[keep_newest:]
            w2c = self._w2c(cand)
            c2w = torch.linalg.inv(w2c)
            pair = (w2c[:, None] @ c2w[None, :])[..., :3, 3].norm(dim=-1)            # |t| of T_CiCj, [n, n]
            inv = (1.0 / (pair + 1e-6)).double()
            inv.fill_diagonal_(0.0)
            cur_c2w = torch.linalg.inv(self._w2c([cur_frame_idx])[0])
            to_cur = (w2c @ cur_c2w)[:, :3, 3].norm(dim=-1).sqrt().double()
            removed_frame = cand[int(torch.argmax(to_cur * inv.sum(dim=1)))]
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
