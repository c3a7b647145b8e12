import sys, torch
sys.path.insert(0, ".")
from monogs_amd import rasterizer as R, synthetic as S
dev = torch.device("cuda:0")
for (N, W, H) in ((300000, 640, 480), (100000, 640, 480), (8000, 640, 480)):
    sc = S.make_scene(N, W, H, seed=0)
    cam = sc.cam
    m, s, r, o, sh = S.activated(sc)
    params = [t.to(dev).requires_grad_(True) for t in (m, s, r, o, sh)]
    st = R.GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, sc.bg.to(dev), 1.0, cam.viewmatrix.to(dev), cam.projmatrix.to(dev), cam.projmatrix_raw.to(dev), 0, cam.viewmatrix.to(dev), False, False)
    ras = R.GaussianRasterizer(st)
    m2d = torch.zeros(N, 3, device=dev, requires_grad=True)
    out = ras(means3D=params[0], means2D=m2d, shs=params[4], opacities=params[3], scales=params[1], rotations=params[2])
    gt = torch.rand(3, H, W, device=dev)
    loss = (out[0] - gt).abs().mean() + out[2].abs().mean() * 0.1
    loss.backward()
    radii = out[1]
    vis = radii > 0
    g_op = params[3].grad.reshape(N)
    g_m2 = m2d.grad.abs().sum(1)
    g_sh = params[4].grad.abs().reshape(N, -1).sum(1)
    zero = vis & (g_op == 0) & (g_m2 == 0) & (g_sh == 0)
    print(N, "visible", int(vis.sum()), "visible with all-zero gradients", int(zero.sum()), "= %.1f %%" % (100.0 * zero.sum().item() / max(1, vis.sum().item())), "pairs", R.last_stats.get("pairs") if hasattr(R, "last_stats") else None, flush=True)
