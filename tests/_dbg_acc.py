import sys, torch, ctypes as C
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from test_gpu_fused import _scene, _loop, DEV
from splat_slam_amd.fused import FusedMappingLoop
from splat_slam_amd.mapper import MappingLoop, PipelineParams
from splat_slam_amd.renderer import render
from splat_slam_amd import _native as nat
syn, params, cams = _scene()
a = _loop(MappingLoop, syn, params, cams, range(4))
f = _loop(FusedMappingLoop, syn, params, cams, range(4))
pkg = render(cams[0], a.gaussians, PipelineParams(), a.background)
pkg['depth'].retain_grad(); pkg['render'].retain_grad()
loss = a.loss_fn(a.config["mapping"], pkg["render"], pkg["depth"], cams[0], pkg["opacity"])
loss.backward()
g=a.gaussians._xyz.grad
f._ensure_state(); f._activate()
for trial in range(2):
    for k in f._acc: 
        if not k.startswith('act'): f._acc[k].zero_()
    vb=f._view_step(cams[0], stats=False)
    torch.cuda.synchronize()
    h=f._acc['xyz']
    print('trial',trial,'xyz diff', (g-h).abs().max(0).values.tolist())
    print('  d_depth diff', (vb.d_depth-pkg['depth'].grad).abs().max().item(), 'd_color diff', (vb.d_color-pkg['render'].grad).abs().max().item(), 'depth out diff', (vb.depth-pkg['depth']).abs().max().item())
    print('  loss', vb.loss.item(), loss.item())
# direct: same saved block, accumulate vs write
import math
vb=f._views[cams[0].uid]; gm=f.gaussians; acc=f._acc
N=gm._xyz.shape[0]; H,W=64,96
s=f._settings(cams[0],N)
inp=nat.SgrInputs(gm._xyz.data_ptr(), acc["act_opac"].data_ptr(), gm._features_dc.data_ptr(), None, acc["act_scale"].data_ptr(), acc["act_rot"].data_ptr(), None)
ws=f._workspace(vb,N,H,W,vb.capacity)
go=nat.SgrGradOutputs(vb.d_color.data_ptr(), vb.d_depth.data_ptr())
res={}
for mode in (0,1):
    z=lambda *sh: torch.zeros(sh,device=DEV)
    b=dict(xyz=z(N,3),m2=z(N,3),op=z(N,1),sh=z(N,1,3),sc=z(N,3),rot=z(N,4),tau=z(6))
    gi=nat.SgrGradInputs(b['xyz'].data_ptr(), b['m2'].data_ptr() if mode==0 else None, b['op'].data_ptr(), b['sh'].data_ptr(), None, b['sc'].data_ptr(), b['rot'].data_ptr(), None, b['tau'].data_ptr(), mode, None,None,None)
    nat.check(f.lib.sgr_backward(C.byref(s),C.byref(inp),vb.radii.data_ptr(),C.byref(go),C.byref(gi),C.byref(ws),f._stream()),'bwd')
    torch.cuda.synchronize(); res[mode]=b
print('write vs acc xyz', (res[0]['xyz']-res[1]['xyz']).abs().max(0).values.tolist())
print('write vs autograd xyz', (res[0]['xyz']-g).abs().max(0).values.tolist())
print('acc vs autograd xyz', (res[1]['xyz']-g).abs().max(0).values.tolist())
print('tau', res[0]['tau'].tolist(), res[1]['tau'].tolist(), cams[0].cam_trans_delta.grad, cams[0].cam_rot_delta.grad)
# oracle
from oracle import raster_oracle as O
cam=cams[0]
x={k:v.detach().cpu().double().requires_grad_(True) for k,v in dict(means3D=gm._xyz, opacities=torch.sigmoid(gm._opacity), shs=gm._features_dc, scales=torch.exp(gm._scaling), rotations=torch.nn.functional.normalize(gm._rotation)).items()}
dd=lambda t: t.detach().cpu().double()
so=O.OracleSettings(64,96,math.tan(cam.FoVx*0.5),math.tan(cam.FoVy*0.5),torch.zeros(3).double(),1.0,dd(cam.world_view_transform),dd(cam.full_proj_transform),dd(cam.projection_matrix),0,dd(cam.camera_center),False,False)
th=torch.zeros(3,dtype=torch.float64,requires_grad=True); rh=torch.zeros(3,dtype=torch.float64,requires_grad=True)
col,radii,dep,opa,nt=O.rasterize(x['means3D'],torch.zeros_like(x['means3D']),x['opacities'],shs=x['shs'],scales=x['scales'],rotations=x['rotations'],theta=th,rho=rh,settings=so)
L=(col*dd(vb.d_color)).sum()+(dep*dd(vb.d_depth)).sum()
L.backward()
go_=x['means3D'].grad
print('oracle vs autograd', (go_-g.cpu().double()).abs().max(0).values.tolist())
print('oracle vs fused', (go_-res[1]['xyz'].cpu().double()).abs().max(0).values.tolist())
print('oracle tau', rh.grad.tolist(), th.grad.tolist())
