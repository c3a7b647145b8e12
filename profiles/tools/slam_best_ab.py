import json, sys, torch
sys.path.insert(0, ".")
import __graft_entry__ as g
from monogs_amd import bench_legs as BL
dev = torch.device("cuda:0")
out = {}
for name, kw in [("best_on", dict(use_first_order_best=True, use_best_loss=True)),
                 ("best_off", dict(use_first_order_best=False, use_best_loss=False)),
                 ("best_on_2", dict(use_first_order_best=True, use_best_loss=True)),
                 ("best_off_2", dict(use_first_order_best=False, use_best_loss=False))]:
    r = BL.bench_slam_surrogate(dev, 41, **kw)
    out[name] = {k: r[k] for k in ("ate_rmse_m", "ate_rmse_keyframes_m", "psnr_db", "fps_total", "tracking_iters_per_s", "gaussians")}
    print(name, out[name], flush=True)
json.dump(out, open("gpurun_out/r03_slam_best_ab.json", "w"), indent=1)
