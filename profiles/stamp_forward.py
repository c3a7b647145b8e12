"""In-kernel timeline of k_blend_fwd (diagnostic; not part of the product or the test suite).

Build a stamped copy of the library and run this on the GPU box:

    hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -DMGS_STAMP \
          monogs_amd/csrc/*.hip -o scratch/libstamp.so
    python profiles/stamp_forward.py

Every workgroup records s_memrealtime at entry / exit, its cycle count and its HW_ID; the script
prints the distribution of wave lifetimes, the resident-wave count over time and the per-CU /
per-SIMD placement.  This is how the 64 KB-per-CU LDS admission limit and the tail of the forward
blend were found."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, "/root/repo")
import monogs_amd._cabi as cabi
cabi.LIB_PATH = "/root/repo/scratch/libstamp.so"
import torch
from monogs_amd import rasterizer as R, synthetic as S
dev = torch.device("cuda:0")
N, W, H = 300000, 640, 480
sc = S.make_scene(N, W, H, seed=0)
cam = sc.cam
m, s, r, o, sh = S.activated(sc)
params = [t.to(dev) for t in (m, s, r, o, sh)]
st = R.GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, sc.bg.to(dev), 1.0, cam.viewmatrix.to(dev), cam.projmatrix.to(dev), cam.projmatrix_raw.to(dev), 0, cam.viewmatrix.to(dev), False, False)
ras = R.GaussianRasterizer(st)
for _ in range(20):
    with torch.no_grad():
        out = ras(means3D=params[0], means2D=torch.zeros(N, 3, device=dev), shs=params[4], opacities=params[3], scales=params[1], rotations=params[2])
torch.cuda.synchronize()
T = 4800
buf = (C.c_longlong * (4 * T))()
rc = cabi.lib().mgs_debug_read_stamps(buf, 4 * T)
a = np.array(buf[:], dtype=np.int64).reshape(T, 4)
t0 = a[:, 0].min()
start = (a[:, 0] - t0) / 100.0   # us (100 MHz)
end = (a[:, 1] - t0) / 100.0
dur = end - start
nseg = (a[:, 2] >> 24) & 0xFFFF
nvisit = (a[:, 2] >> 40) & 0xFFFFFF
a[:, 2] = a[:, 2] & 0xFFFFFF
clk = a[:, 2] / np.maximum(1, (a[:, 1] - a[:, 0])) * 100.0  # MHz
hw = a[:, 3] & 0xffffffff
xcc = a[:, 3] >> 32
cu = (hw >> 8) & 0xf; sh_ = (hw >> 12) & 1; se = (hw >> 13) & 0x7
key = xcc * 1000 + se * 100 + sh_ * 16 + cu
tl = np.zeros(T)
print("rc", rc, "kernel span us", end.max(), "start spread", start.max())
print("dur us: mean %.1f min %.1f p50 %.1f p90 %.1f max %.1f" % (dur.mean(), dur.min(), np.median(dur), np.percentile(dur, 90), dur.max()))
print("clock MHz mean %.0f min %.0f max %.0f" % (clk.mean(), clk.min(), clk.max()))
u, cnt = np.unique(key, return_counts=True)
print("distinct CUs", len(u), "WGs per CU hist", np.bincount(cnt))
# end time vs WGs per CU
for c in sorted(set(cnt)):
    ks = u[cnt == c]
    e = [end[key == k].max() for k in ks]
    print("  CUs with", c, "WGs: n", len(ks), "mean last-end us %.1f max %.1f" % (np.mean(e), np.max(e)))
late = np.argsort(-end)[:10]
print("latest tiles", [(int(i), round(float(start[i]),1), round(float(end[i]),1)) for i in late])
# tile list lengths
shown = 0
for k in u[cnt == 5][:6]:
    idx = np.where(key == k)[0]
    idx = idx[np.argsort(start[idx])]
    print("CU", int(k), [(int(i), round(float(start[i]), 1), round(float(end[i]), 1)) for i in idx])
# max concurrency per CU
mc = []
for k in u:
    idx = np.where(key == k)[0]
    ev = sorted([(start[i], 1) for i in idx] + [(end[i], -1) for i in idx])
    c = m_ = 0
    for _, d in ev:
        c += d; m_ = max(m_, c)
    mc.append(m_)
print("max concurrency per CU hist", np.bincount(mc))
print("starts histogram (10us bins)", np.histogram(start, bins=[0,1,5,10,20,30,40,50,60,70])[0])

simd = (hw >> 4) & 0x3
skey = key * 4 + simd
us, scnt = np.unique(skey, return_counts=True)
print("distinct SIMDs", len(us), "waves per SIMD hist", np.bincount(scnt))
busy = np.array([dur[skey == k].sum() for k in us])
lastend = np.array([end[skey == k].max() for k in us])
print("per-SIMD sum of wave durations: mean %.1f max %.1f ; last end mean %.1f max %.1f" % (busy.mean(), busy.max(), lastend.mean(), lastend.max()))
print("wave dur percentiles", np.percentile(dur, [1, 10, 50, 90, 99, 100]).round(1))
# occupancy over time
ts = np.linspace(0, end.max(), 12)
print("resident waves at t:", [(round(float(t),1), int(((start <= t) & (end > t)).sum())) for t in ts])
# per-wave instruction-time proxy: cycles (a[:,2]) and which tiles the slowest waves belong to
cyc = a[:, 2]
order = np.argsort(-dur)[:24]
print("slowest waves (wg, tile, quad, dur us, Mcycles):", [(int(i), int(i) // 4, int(i) % 4, round(float(dur[i]), 1), round(float(cyc[i]) / 1e6, 3)) for i in order])
print("sum of wave durations us %.0f  (= %.1f us if spread over 1024 SIMDs x 1)" % (dur.sum(), dur.sum() / 1024))
print("dur histogram (10 us bins)", np.histogram(dur, bins=np.arange(0, 110, 10))[0])

print("segments per wave: mean %.1f max %d ; visits per wave: mean %.0f max %d" % (nseg.mean(), nseg.max(), nvisit.mean(), nvisit.max()))
print("slowest waves (nseg, nvisit):", [(int(nseg[i]), int(nvisit[i])) for i in order])
A = np.stack([np.ones(T), nseg, nvisit], 1).astype(np.float64)
coef, *_ = np.linalg.lstsq(A, dur, rcond=None)
print("fit dur us = %.2f + %.3f * nseg + %.4f * nvisit ; residual std %.2f" % (coef[0], coef[1], coef[2], (A @ coef - dur).std()))
work = A @ np.array([0.0, coef[1], coef[2]])
wsum = np.array([work[skey == k].sum() for k in us])
print("per-SIMD modelled work: mean %.1f max %.1f min %.1f ; corr(work sum, last end) %.2f" % (wsum.mean(), wsum.max(), wsum.min(), np.corrcoef(wsum, lastend)[0, 1]))
wmax = np.array([work[skey == k].max() for k in us])
print("per-SIMD longest-wave work: mean %.1f max %.1f ; corr(longest, last end) %.2f" % (wmax.mean(), wmax.max(), np.corrcoef(wmax, lastend)[0, 1]))

pb = (C.c_longlong * (4 * T))()
if hasattr(cabi.lib(), "mgs_debug_read_phases"):
    cabi.lib().mgs_debug_read_phases(pb, 4 * T)
    ph = np.array(pb[:], dtype=np.int64).reshape(T, 4) / (clk.mean())   # us
    print("phase us per wave (mean): wait+reach %.1f  stage %.1f  walk %.1f  loop/ckpt/prefetch-issue %.1f ; per segment: %s" % (
        ph[:, 0].mean(), ph[:, 1].mean(), ph[:, 2].mean(), ph[:, 3].mean(), (ph.sum(0) / nseg.sum()).round(2)))
