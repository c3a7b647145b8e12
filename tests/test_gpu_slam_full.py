"""BASELINE config 4's shape at full image size (640x480, fr3_office intrinsics) on a synthetic
sequence (monogs_amd/slam_surrogate.py): initialisation, native tracking (first order + sketched
LM), keyframe insertion on the device, native mapping over the keyframe window with the prune pass,
ATE and PSNR.  Budgets are cut to keep the test short (300 initialisation / 60 mapping iterations
instead of 1050 / 150; bench.py's `slam` leg runs the reference's budgets).  The TUM sequence itself
is not available offline; MONOGS_TUM_DIR switches the loader to a mounted copy."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sensor_depth", [False, True])
def test_slam_surrogate_640x480(built, sensor_depth):
    from monogs_amd import slam_surrogate as SS
    dev = torch.device("cuda:0")
    n = 21
    frames, cam, source = SS.load_sequence(n, 640, 480, dev)
    res = SS.run_sequence(frames, cam, dev, sensor_depth=sensor_depth, init_iters=300, mapping_iters=60)
    torch.cuda.synchronize()
    assert res["capacity_ok"] and res["kf_ids"] == [0, 5, 10, 15, 20]
    ev = SS.evaluate(res, frames, dev, monocular=not sensor_depth)
    path = ev["path_length_m"]
    print(source, "sensor_depth" if sensor_depth else "monocular", ev,
          {k: round(v, 3) for k, v in res.items() if k.startswith("t_")})
    assert all(torch.isfinite(c.T).all() for c in res["cameras"].values())
    assert 1000 < ev["gaussians"] < 200_000
    if sensor_depth:
        # metric map (keyframes inserted from the sensor depth): rigid alignment, no scale
        assert ev["ate_rmse_m"] < 0.15 * path and ev["psnr_db"] > 17.0
        last = res["cameras"][n - 1]
        err = (torch.linalg.inv(last.T.cpu().double())[:3, 3] - torch.linalg.inv(frames[n - 1].T_gt.double())[:3, 3]).norm()
        assert float(err) < 0.6 * path          # tracking removed most of the accumulated motion
    else:
        # the reference's own configuration (monocular): scale is free, Sim(3) alignment (eval_utils.py:26-44)
        assert ev["ate_rmse_m"] < 0.03 * path and ev["ate_rmse_keyframes_m"] < 0.02 * path
        assert ev["psnr_db"] > 18.0
    # native tracking cost per frame stays near 50 iterations of GPU work (no host round trips inside)
    assert 0.8 * (n - 1) * 50 <= res["n_track_iters"] <= (n - 1) * 50


def _rot_to_quat_xyzw(R):
    import numpy as np
    w = 0.5 * np.sqrt(max(0.0, 1.0 + R[0, 0] + R[1, 1] + R[2, 2]))
    x, y, z = (R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w)
    return x, y, z, w


def test_slam_surrogate_through_a_tum_format_folder(built, tmp_path, monkeypatch):
    """The real-data branch of the surrogate (slam_surrogate.load_sequence -> eval_metrics.TUMSequence,
    reference utils/dataset.py:50-124): 11 frames of the synthetic world are written as a TUM-format
    folder (8-bit rgb/*.png, 16-bit depth/*.png at 5000 per metre, rgb.txt / depth.txt /
    groundtruth.txt with camera-to-world `tx ty tz qx qy qz qw`), MONOGS_TUM_DIR points at it, and the
    sequence read back from disk runs through tracking + mapping + evaluation."""
    import numpy as np
    from PIL import Image
    from monogs_amd import slam_surrogate as SS
    dev = torch.device("cuda:0")
    n, W, H = 11, 320, 240
    monkeypatch.delenv("MONOGS_TUM_DIR", raising=False)
    frames, cam, _ = SS.load_sequence(n, W, H, dev, world_gaussians=40_000)
    d = tmp_path / "rgbd_dataset_synthetic"
    (d / "rgb").mkdir(parents=True)
    (d / "depth").mkdir()
    rgb_l, dep_l, gt_l = ["# color images"], ["# depth maps"], ["# ground truth trajectory", "# timestamp tx ty tz qx qy qz qw"]
    for k, fr in enumerate(frames):
        t = 1341847980.0 + k / 30.0                        # 30 Hz: kept by the 32 fps sub-sampling
        img = (fr.image.clamp(0, 1) * 255.0).round().byte().permute(1, 2, 0).cpu().numpy()
        Image.fromarray(img).save(d / "rgb" / f"{t:.6f}.png")
        dep = (fr.depth.clamp(0, 13.0) * 5000.0).round().cpu().numpy().astype(np.uint16)
        Image.fromarray(dep).save(d / "depth" / f"{t:.6f}.png")
        Twc = np.linalg.inv(fr.T_gt.double().numpy())
        q = _rot_to_quat_xyzw(Twc[:3, :3])
        rgb_l.append(f"{t:.6f} rgb/{t:.6f}.png")
        dep_l.append(f"{t + 0.004:.6f} depth/{t:.6f}.png")
        gt_l.append(f"{t + 0.001:.6f} " + " ".join(f"{v:.9f}" for v in (*Twc[:3, 3], *q)))
    (d / "rgb.txt").write_text("\n".join(rgb_l) + "\n")
    (d / "depth.txt").write_text("\n".join(dep_l) + "\n")
    (d / "groundtruth.txt").write_text("\n".join(gt_l) + "\n")

    monkeypatch.setenv("MONOGS_TUM_DIR", str(d))
    frames2, cam2, source = SS.load_sequence(n, W, H, dev)
    assert source.startswith("TUM sequence at") and len(frames2) == n
    for a, b in zip(frames, frames2):
        assert (a.image - b.image).abs().max().item() <= 0.5 / 255 + 1e-6            # 8-bit quantisation
        assert (a.depth.clamp(0, 13.0) - b.depth).abs().max().item() <= 1.01e-4      # 1 / 5000 m steps
        assert torch.allclose(a.T_gt, b.T_gt, atol=1e-5)
    res = SS.run_sequence(frames2, cam2, dev, init_iters=200, mapping_iters=40, kf_interval=5)
    torch.cuda.synchronize()
    assert res["capacity_ok"] and res["kf_ids"] == [0, 5, 10]
    ev = SS.evaluate(res, frames2, dev, monocular=True)
    print(source, ev)
    assert all(torch.isfinite(c.T).all() for c in res["cameras"].values())
    assert ev["ate_rmse_m"] < 0.05 * ev["path_length_m"] and ev["psnr_db"] > 15.0
