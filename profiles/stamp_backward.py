"""In-kernel timeline of k_blend_bwd (diagnostic; not part of the product or the test suite).

    hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -DMGS_STAMP \
          monogs_amd/csrc/*.hip -o scratch/libstamp.so
    python profiles/stamp_backward.py

Every work item records s_memrealtime at entry / exit, the cycles it spent up to (0) the per-pixel
state, (1) the staged records + reach masks, (2) the end of the walk, and its quadrant-visit count."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, "/root/repo")
import monogs_amd._cabi as cabi
cabi.LIB_PATH = os.environ.get("MGS_STAMP_LIB", "/root/repo/scratch/libstamp.so")
import torch
from monogs_amd import rasterizer as R, synthetic as S
dev = torch.device("cuda:0")
N, W, H = 300000, 640, 480
sc = S.make_scene(N, W, H, seed=0)
cam = sc.cam
m, s, r, o, sh = S.activated(sc)
params = [t.to(dev).requires_grad_(True) for t in (m, s, r, o, sh)]
st = R.GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, sc.bg.to(dev), 1.0, cam.viewmatrix.to(dev), cam.projmatrix.to(dev), cam.projmatrix_raw.to(dev), 0, cam.viewmatrix.to(dev), False, False)
ras = R.GaussianRasterizer(st)
gt = torch.rand(3, H, W, device=dev)
for _ in range(10):
    out = ras(means3D=params[0], means2D=torch.zeros(N, 3, device=dev, requires_grad=True), shs=params[4], opacities=params[3], scales=params[1], rotations=params[2])
    loss = (out[0] - gt).abs().mean() + out[2].abs().mean() * 0.1
    loss.backward()
torch.cuda.synchronize()
n = 65536
sb = (C.c_longlong * (4 * n))(); pb = (C.c_longlong * (4 * n))()
rc = cabi.lib().mgs_debug_read_bwd_stamps(sb, pb, 4 * n)
a = np.array(sb[:], dtype=np.int64).reshape(n, 4); ph = np.array(pb[:], dtype=np.int64).reshape(n, 4)
ok = a[:, 0] > 0
a, ph = a[ok], ph[ok]
# keep the last launch only (stamps of earlier launches are overwritten item by item; drop stragglers)
t0 = a[:, 0].max() - 30000
keep = a[:, 0] >= t0
a, ph = a[keep], ph[keep]
T = len(a)
PH0 = os.environ.get("MGS_STAMP_PH0", "0") == "1"     # library built with -DMGS_STAMP_PH0: phase 0 instead of lane statistics
if PH0:
    nmiss = nlanes = np.zeros(T, dtype=np.int64)
else:
    nmiss = ph[:, 0] >> 32; nlanes = ph[:, 0] & 0xffffffff; ph[:, 0] = 0
print("quadrant visits %d, of which no pixel passes %d (%.1f%%); passing lanes per remaining visit %.1f of 64" % (
    (a[:, 2] & 0xffffffff).sum(), nmiss.sum(), 100.0 * nmiss.sum() / max(1, (a[:, 2] & 0xffffffff).sum()),
    nlanes.sum() / max(1, (a[:, 2] & 0xffffffff).sum() - nmiss.sum())))
t0 = a[:, 0].min()
start = (a[:, 0] - t0) / 100.0; end = (a[:, 1] - t0) / 100.0; dur = end - start
nvisit = a[:, 2] & 0xffffffff; nany = a[:, 2] >> 32
print("rc", rc, "items", T, "kernel span us %.1f" % end.max())
print("item dur us: mean %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (dur.mean(), *np.percentile(dur, [10, 50, 90]), dur.max()))
clk = 2000.0
print("phase us per item (mean, at %.0f MHz): per-pixel state %.2f  staging %.2f  walk %.2f ; quadrant visits %.1f, splats with work %.1f" % (
    clk, ph[:, 0].mean() / clk, ph[:, 1].mean() / clk, ph[:, 2].mean() / clk, nvisit.mean(), nany.mean()))
live = nvisit > 0
print("items with work %d: walk us %.2f for %.1f quadrant visits (%.0f cycles per visit)" % (live.sum(), ph[live, 2].mean() / clk, nvisit[live].mean(), ph[live, 2].mean() / max(1.0, nvisit[live].mean())))
FINE = os.environ.get("MGS_STAMP_FINE", "0") == "1"   # library built with -DMGS_STAMP_PH0 -DMGS_STAMP_FINE: the prologue in four steps
if FINE:
    names = ("item record", "quadrant ends", "per-pixel loads", "rest of the state phase")
    for sel, what in ((live, "items with work"), (~live, "items without work")):
        print(what + ": cycles " + ", ".join("%s %.0f" % (n, ph[sel, k].mean()) for k, n in enumerate(names)) +
              "; item dur us %.2f" % dur[sel].mean())
elif PH0:
    cyc = ph[live, 0] + ph[live, 1] + ph[live, 2]
    print("items with work: dur us %.2f; phase cycles: state %.0f staging %.0f walk %.0f (sum %.0f) -> %.0f MHz if the phases cover the item" % (
        dur[live].mean(), ph[live, 0].mean(), ph[live, 1].mean(), ph[live, 2].mean(), cyc.mean(), cyc.mean() / dur[live].mean()))
    print("items without work: phase cycles: state %.0f staging %.0f walk %.0f" % (ph[~live, 0].mean(), ph[~live, 1].mean(), ph[~live, 2].mean()))
print("items without work %d: dur us %.2f" % ((~live).sum(), dur[~live].mean() if (~live).any() else 0))
ts = np.linspace(0, end.max(), 14)
print("resident items at t:", [(round(float(t), 1), int(((start <= t) & (end > t)).sum())) for t in ts])
print("starts histogram (10 us bins)", np.histogram(start, bins=np.arange(0, end.max() + 10, 10))[0])
hw = a[:, 3] & 0xffffffff; xcc = a[:, 3] >> 32
cu = (hw >> 8) & 0xf; sh_ = (hw >> 12) & 1; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 0x3
skey = ((xcc * 8 + se) * 2 + sh_) * 64 + cu * 4 + simd
us, cnt = np.unique(skey, return_counts=True)
print("SIMDs", len(us), "items per SIMD: min %d mean %.1f max %d" % (cnt.min(), cnt.mean(), cnt.max()))
le = np.array([end[skey == k].max() for k in us])
print("per-SIMD last end: mean %.1f min %.1f max %.1f" % (le.mean(), le.min(), le.max()))
