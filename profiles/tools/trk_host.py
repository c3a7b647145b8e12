import math, sys, time, torch
sys.path.insert(0, ".")
from monogs_amd import _cabi, synthetic as S
from monogs_amd.gaussian_renderer import render
from monogs_amd.pose import SE3_exp
from monogs_amd.slam_loops import GaussianParams, Pipe, ViewCamera
from monogs_amd.tracking_native import NativeTracker
dev = torch.device("cuda:0")
for N in (8000, 8000, 50000):
    sc = S.make_scene(N, 640, 480, seed=0); cam = sc.cam; H, W = cam.H, cam.W
    gauss = GaussianParams(sc.means3D.to(dev), sc.log_scales.to(dev), sc.rot.to(dev), sc.opacity_logit.to(dev), sc.features_dc.to(dev))
    fovx, fovy = 2 * math.atan(cam.tanfovx), 2 * math.atan(cam.tanfovy)
    bg = torch.zeros(3, device=dev)
    def view(T):
        return ViewCamera(1, torch.zeros(3, H, W), T, cam.projmatrix_raw, fovx, fovy, H, W, dev)
    with torch.no_grad():
        target = render(view(torch.eye(4)), gauss, Pipe, bg)["render"].clone()
    vp = view(SE3_exp(torch.tensor([0.01, -0.008, 0.006, 0.002, -0.003, 0.002])))
    vp.original_image = target
    vp.rgb_pixel_mask_mapping = (target.sum(0) > 0.01).view(1, H, W)
    trk = NativeTracker(vp, gauss, bg)
    trk.args.adam.sticky_converged = 0
    for _ in range(20): trk.step()
    torch.cuda.synchronize()
    host = []
    t0 = time.perf_counter()
    for _ in range(400):
        a = time.perf_counter(); trk.step(); host.append(time.perf_counter() - a)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    host.sort()
    print(N, "host us per step: median %.1f p90 %.1f max %.1f; enqueue loop %.1f us/step, with final sync %.1f us/step; D %d" % (
        host[200] * 1e6, host[360] * 1e6, host[-1] * 1e6, (t1 - t0) / 400 * 1e6, (t2 - t0) / 400 * 1e6, trk.pairs()), flush=True)
