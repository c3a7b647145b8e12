"""Offline evaluation for the `--eval` path: ATE after Umeyama
alignment, PSNR / SSIM of held-out renders, and a TUM sequence reader - without `evo`,
`cv2`, `trimesh`, `wandb` or `torchmetrics` (none of them is installable offline).

Mirrors /root/reference/utils/eval_utils.py:26-178 (evaluate_evo, eval_ate, eval_rendering)
and the TUM parser of utils/dataset.py:50-124; PSNR is gaussian_splatting/utils/
image_utils.py:19-21, SSIM gaussian_splatting/utils/loss_utils.py:44-96 (11-tap Gaussian
window, sigma 1.5, zero padding).  LPIPS needs downloaded AlexNet weights (eval_utils.py:128)
and is reported as None.  ATE parity against `evo` itself is unpinned (the package is absent):
the alignment is the closed form of Umeyama 1991 that evo's `PosePath3D.align` documents,
checked against synthetic similarity transforms.
"""
from __future__ import annotations

import json
import os
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------ trajectory error
def umeyama_alignment(x: np.ndarray, y: np.ndarray, with_scale: bool):
    """Least-squares similarity (R, t, c) with y ~ c R x + t for 3xN point sets
    (Umeyama, PAMI 1991, eq. 34-42); c = 1 when `with_scale` is False."""
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    assert x.shape == y.shape and x.shape[0] == 3
    n = x.shape[1]
    mx, my = x.mean(1, keepdims=True), y.mean(1, keepdims=True)
    xc, yc = x - mx, y - my
    var_x = (xc * xc).sum() / n
    cov = yc @ xc.T / n
    U, d, Vt = np.linalg.svd(cov)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1.0
    R = U @ S @ Vt
    c = float(np.trace(np.diag(d) @ S) / var_x) if with_scale else 1.0
    t = my - c * R @ mx
    return R, t.reshape(3), c


def ate_statistics(poses_gt: Sequence[np.ndarray], poses_est: Sequence[np.ndarray],
                   monocular: bool = False) -> Dict[str, float]:
    """evaluate_evo (eval_utils.py:26-44): align the estimated camera-to-world trajectory to
    the ground truth (scale corrected for monocular runs), translation APE statistics."""
    P = np.stack([np.asarray(p, np.float64)[:3, 3] for p in poses_gt], 1)
    Q = np.stack([np.asarray(p, np.float64)[:3, 3] for p in poses_est], 1)
    R, t, c = umeyama_alignment(Q, P, with_scale=monocular)
    err = np.linalg.norm(c * R @ Q + t[:, None] - P, axis=0)
    return {"rmse": float(np.sqrt((err ** 2).mean())), "mean": float(err.mean()),
            "median": float(np.median(err)), "std": float(err.std()), "min": float(err.min()),
            "max": float(err.max()), "sse": float((err ** 2).sum())}


def eval_ate(frames, kf_ids, save_dir: Optional[str] = None, iterations=0, final=False,
             monocular=False) -> float:
    """eval_ate (eval_utils.py:72-111): keyframe poses (`frame.T`, `frame.T_gt`: world-to-camera)
    -> ATE RMSE [m]; trajectory + statistics are written as JSON when `save_dir` is given."""
    est, gt, ids = [], [], []
    for k in kf_ids:
        f = frames[k]
        est.append(np.linalg.inv(torch.as_tensor(f.T).double().cpu().numpy()))
        gt.append(np.linalg.inv(torch.as_tensor(f.T_gt).double().cpu().numpy()))
        ids.append(int(getattr(f, "uid", k)))
    stats = ate_statistics(gt, est, monocular)
    if save_dir is not None:
        plot_dir = os.path.join(save_dir, "plot")
        os.makedirs(plot_dir, exist_ok=True)
        label = "final" if final else "{:04}".format(iterations)
        with open(os.path.join(plot_dir, f"trj_{label}.json"), "w", encoding="utf-8") as fh:
            json.dump({"trj_id": ids, "trj_est": [p.tolist() for p in est],
                       "trj_gt": [p.tolist() for p in gt]}, fh, indent=4)
        with open(os.path.join(plot_dir, f"stats_{label}.json"), "w", encoding="utf-8") as fh:
            json.dump(stats, fh, indent=4)
    return stats["rmse"]


# ------------------------------------------------------------------ image metrics
def psnr(img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
    """image_utils.py:19-21: per-batch-row PSNR for images in [0, 1]."""
    mse = ((img1 - img2) ** 2).reshape(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


def _gauss_window(size: int, sigma: float, channels: int, like: torch.Tensor) -> torch.Tensor:
    x = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-(x ** 2) / (2 * sigma ** 2))
    g = (g / g.sum()).unsqueeze(1)
    w = (g @ g.t()).unsqueeze(0).unsqueeze(0)
    return w.expand(channels, 1, size, size).contiguous().to(like)


def ssim_map(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11) -> torch.Tensor:
    """Per-pixel SSIM index [B,C,H,W] (loss_utils.py:61-101, `_ssim` before its mean): 11x11
    Gaussian window (sigma 1.5) applied per channel with ZERO padding, C1 = 0.01^2, C2 = 0.03^2."""
    if img1.dim() == 3:
        img1, img2 = img1.unsqueeze(0), img2.unsqueeze(0)
    ch = img1.shape[-3]
    w = _gauss_window(window_size, 1.5, ch, img1)
    pad = window_size // 2
    mu1 = F.conv2d(img1, w, padding=pad, groups=ch)
    mu2 = F.conv2d(img2, w, padding=pad, groups=ch)
    s11 = F.conv2d(img1 * img1, w, padding=pad, groups=ch) - mu1 * mu1
    s22 = F.conv2d(img2 * img2, w, padding=pad, groups=ch) - mu2 * mu2
    s12 = F.conv2d(img1 * img2, w, padding=pad, groups=ch) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))


def ssim(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11, size_average: bool = True):
    """loss_utils.py:63-96: mean of `ssim_map` over everything, or per batch row; inputs
    [B,C,H,W] (or [C,H,W]) in [0, 1]."""
    m = ssim_map(img1, img2, window_size)
    return m.mean() if size_average else m.mean(1).mean(1).mean(1)


def eval_rendering(frames, gaussians, dataset, render_fn: Callable, pipe, background,
                   kf_indices, save_dir: Optional[str] = None, iteration="final", interval: int = 5):
    """eval_rendering (eval_utils.py:114-178): every `interval`-th non-keyframe view is
    rendered and scored against its ground-truth image (`dataset[idx][0]`, [3,H,W] in [0,1]);
    PSNR over pixels with gt > 0, SSIM over the full image.  `render_fn` is
    monogs_amd.gaussian_renderer.render (or the reference's own render on the drop-in)."""
    ps, ss = [], []
    end_idx = len(frames) - 1
    for idx in range(0, end_idx, interval):
        if idx in kf_indices:
            continue
        gt_image = dataset[idx][0]
        with torch.no_grad():
            image = torch.clamp(render_fn(frames[idx], gaussians, pipe, background)["render"], 0.0, 1.0)
        gt_image = gt_image.to(image.device)
        mask = gt_image > 0
        ps.append(psnr(image[mask].unsqueeze(0), gt_image[mask].unsqueeze(0)).item())
        ss.append(ssim(image.unsqueeze(0), gt_image.unsqueeze(0)).item())
    out = {"mean_psnr": float(np.mean(ps)) if ps else float("nan"),
           "mean_ssim": float(np.mean(ss)) if ss else float("nan"), "mean_lpips": None}
    if save_dir is not None:
        d = os.path.join(save_dir, "psnr", str(iteration))
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "final_result.json"), "w", encoding="utf-8") as fh:
            json.dump(out, fh, indent=4)
    return out


# ------------------------------------------------------------------ TUM sequence reader
def _quat_xyzw_to_matrix(q: np.ndarray) -> np.ndarray:
    x, y, z, w = (q / np.linalg.norm(q)).tolist()
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _read_list(path: str, skip_comments: bool = True) -> List[List[str]]:
    rows = []
    with open(path, "r", encoding="utf-8") as fh:
        for line in fh:
            line = line.strip()
            if not line or (skip_comments and line.startswith("#")):
                continue
            rows.append(line.split())
    return rows


class TUMSequence:
    """utils/dataset.py:50-124 (TUMParser): rgb / depth / ground-truth association within
    0.08 s, sub-sampled to at most 32 frames per second; `poses[i]` is the world-to-camera
    matrix inv(T_wc) of frame i.  Images are decoded with PIL on access (`image(i)` ->
    float [3,H,W] in [0,1], `depth(i)` -> float [H,W] metres at `depth_scale`)."""

    def __init__(self, folder: str, frame_rate: float = 32, max_dt: float = 0.08, depth_scale: float = 5000.0):
        self.folder, self.depth_scale = folder, depth_scale
        pose_file = "groundtruth.txt" if os.path.isfile(os.path.join(folder, "groundtruth.txt")) else "pose.txt"
        rgb = _read_list(os.path.join(folder, "rgb.txt"))
        dep = _read_list(os.path.join(folder, "depth.txt"))
        gt = _read_list(os.path.join(folder, pose_file))
        t_rgb = np.array([float(r[0]) for r in rgb])
        t_dep = np.array([float(r[0]) for r in dep])
        t_gt = np.array([float(r[0]) for r in gt])
        vec = np.array([[float(v) for v in r[1:8]] for r in gt])
        assoc = []
        for i, t in enumerate(t_rgb):
            j = int(np.argmin(np.abs(t_dep - t)))
            k = int(np.argmin(np.abs(t_gt - t)))
            if abs(t_dep[j] - t) < max_dt and abs(t_gt[k] - t) < max_dt:
                assoc.append((i, j, k))
        keep = [0] if assoc else []
        for a in range(1, len(assoc)):
            if t_rgb[assoc[a][0]] - t_rgb[assoc[keep[-1]][0]] > 1.0 / frame_rate:
                keep.append(a)
        self.color_paths, self.depth_paths, self.poses, self.timestamps = [], [], [], []
        for a in keep:
            i, j, k = assoc[a]
            T = np.eye(4)
            T[:3, :3] = _quat_xyzw_to_matrix(vec[k, 3:7])
            T[:3, 3] = vec[k, 0:3]
            self.color_paths.append(os.path.join(folder, rgb[i][1]))
            self.depth_paths.append(os.path.join(folder, dep[j][1]))
            self.poses.append(np.linalg.inv(T))
            self.timestamps.append(float(t_rgb[i]))

    def __len__(self):
        return len(self.color_paths)

    def image(self, i: int) -> torch.Tensor:
        from PIL import Image
        a = np.asarray(Image.open(self.color_paths[i]).convert("RGB"), dtype=np.float32) / 255.0
        return torch.from_numpy(a).permute(2, 0, 1).contiguous()

    def depth(self, i: int) -> torch.Tensor:
        from PIL import Image
        a = np.asarray(Image.open(self.depth_paths[i]), dtype=np.float32) / self.depth_scale
        return torch.from_numpy(a)

    def __getitem__(self, i: int):
        return self.image(i), self.depth(i), torch.from_numpy(self.poses[i]).float()
