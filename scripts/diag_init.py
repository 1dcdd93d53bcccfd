import copy, os, sys
import numpy as np, torch
ROOT = sys.argv[1]
sys.path.insert(0, ROOT)
from splat_slam_amd import synthetic as syn
from splat_slam_amd.fused import FusedMappingLoop
from splat_slam_amd.session import MappingSession
DEV = "cuda:0"
cfg = copy.deepcopy(syn.DEFAULT_CONFIG)
tr = cfg["mapping"]["Training"]
tr["init_itr_num"], tr["mapping_itr_num"], tr["window_size"] = int(sys.argv[3]), 12, 4
tr["init_gaussian_update"], tr["init_gaussian_reset"] = 40, 10 ** 9
cfg["mapping"]["opt_params"]["densify_from_iter"] = 10 ** 9
intr = syn.INTRINSICS["tiny"]
frames = syn.keyframe_stream(7, intr, DEV, n_world=20000, seed=5, sweep_deg=70.0)
torch.manual_seed(43); np.random.seed(43)
loop = FusedMappingLoop(cfg, device=DEV)
sess = MappingSession(loop, intr)
from splat_slam_amd.gaussian_model import GaussianModel
_orig = GaussianModel.densify_and_prune
def _dump(self, *a, **k):
    if not os.path.exists(sys.argv[2] + ".pre"):
        torch.cuda.synchronize()
        vb = list(loop._views.values())[0]
        torch.save({"accum": self.xyz_gradient_accum.cpu(), "denom": self.denom.cpu(), "maxr": self.max_radii2D.cpu(),
                    "flat": loop._acc["flat"].cpu(), "radii": vb.radii.cpu(), "nt": vb.n_touched.cpu(), "color": vb.color.cpu(),
                    "xyz": self._xyz.detach().cpu(), "act_scale": loop._acc["act_scale"].cpu(), "d_color": vb.d_color.cpu(), "d_depth": vb.d_depth.cpu(), "scratch_head": vb.scratch[:384*48*4].clone().view(torch.float32).cpu(), "saved_head": vb.saved[:200000].clone().cpu()}, sys.argv[2] + ".pre")
        from splat_slam_amd.renderer import render
        from splat_slam_amd.mapper import PipelineParams
        from splat_slam_amd.losses import get_loss_mapping
        cam = list(loop.viewpoints.values())[0]
        with torch.enable_grad():
            pkg = render(cam, self, PipelineParams(), loop.background)
            loss = get_loss_mapping(loop.config["mapping"], pkg["render"], pkg["depth"], cam, pkg["opacity"], initialization=True)
            loss.backward()
        N = self._xyz.shape[0]
        fx = loop._acc["flat"][:3 * N].view(N, 3)
        ff = loop._acc["flat"][3 * N:6 * N].view(N, 1, 3)
        print("autograd vs fused: xyz", float((self._xyz.grad - fx).abs().max()), "of", float(self._xyz.grad.abs().max()),
              " f_dc", float((self._features_dc.grad - ff).abs().max()), "of", float(self._features_dc.grad.abs().max()),
              " color", float((pkg["render"] - vb.color).abs().max()), flush=True)
        self.optimizer.zero_grad(set_to_none=True)
    return _orig(self, *a, **k)
GaussianModel.densify_and_prune = _dump
sess.process(*frames[0])
gm = loop.gaussians
torch.save({"xyz": gm._xyz.detach().cpu(), "f_dc": gm._features_dc.detach().cpu(), "op": gm._opacity.detach().cpu(),
            "sc": gm._scaling.detach().cpu(), "rot": gm._rotation.detach().cpu(), "accum": gm.xyz_gradient_accum.cpu(),
            "denom": gm.denom.cpu(), "maxr": gm.max_radii2D.cpu()}, sys.argv[2])
