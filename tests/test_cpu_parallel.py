"""world_size-2 gloo test of the keyframe-parallel gradient exchange."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from monogs_amd.parallel import FlatGradBucket, view_pose
    N = 257
    g = torch.Generator().manual_seed(0)
    params = [torch.zeros(N, 3), torch.zeros(N, 1, 3), torch.zeros(N, 1), torch.zeros(N, 3), torch.zeros(N, 4)]
    grads = []
    for r in range(world):
        grads.append([torch.randn(p.shape, generator=g) for p in params]
                     + [torch.randn(N, 3, generator=g), torch.randint(0, 30, (N,), generator=g, dtype=torch.int32)])
    for p, gr in zip(params, grads[rank][:5]):
        p.grad = gr.clone()
    bucket = FlatGradBucket(params)
    stat, denom, radii = bucket.all_reduce(grads[rank][5], grads[rank][6])
    ok = True
    for i, p in enumerate(params):
        want = sum(grads[r][i] for r in range(world))
        ok &= torch.allclose(p.grad, want, atol=1e-6) and p.grad.is_contiguous()
    # add_densification_stats (gaussian_model.py:693-697) only touches visible Gaussians
    want_stat = sum(torch.linalg.norm(grads[r][5][:, :2], dim=-1) * (grads[r][6] > 0) for r in range(world))
    want_den = sum((grads[r][6] > 0).float() for r in range(world))
    want_rad = torch.stack([grads[r][6] for r in range(world)]).max(0).values
    ok &= torch.allclose(stat, want_stat, atol=1e-5) and torch.equal(denom, want_den)
    ok &= torch.equal(radii, want_rad)
    ok &= not torch.equal(view_pose(0), view_pose(1))
    # prune iterations: every rank learns every view's occlusion-aware visibility -> n_obs
    from monogs_amd.parallel import all_gather_visibility, broadcast_split_noise, observation_counts
    nts = [torch.randint(0, 3, (N,), generator=g, dtype=torch.int32) for _ in range(world)]
    vis = all_gather_visibility(nts[rank])
    ok &= vis.shape == (world, N) and all(torch.equal(vis[r], nts[r] > 0) for r in range(world))
    ok &= torch.equal(observation_counts(vis), sum((t > 0).int() for t in nts))
    # densify_and_split draws identical offsets on every rank
    noise = broadcast_split_noise(37, "cpu", generator=torch.Generator().manual_seed(100 + rank))
    want = torch.randn(74, 3, generator=torch.Generator().manual_seed(100))
    ok &= torch.equal(noise, want)
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_flat_gradient_bucket_allreduce_world2():
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 500)
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret[0] and ret[1]


def test_flat_gradient_bucket_single_process_roundtrip():
    sys.path.insert(0, ROOT)
    from monogs_amd.parallel import FlatGradBucket
    params = [torch.zeros(10, 3), torch.zeros(10, 4)]
    for p in params:
        p.grad = torch.randn(p.shape)
    ref = [p.grad.clone() for p in params]
    b = FlatGradBucket(params)
    stat, denom, radii = b.all_reduce(torch.ones(10, 3), torch.arange(10, dtype=torch.int32))
    assert all(torch.equal(p.grad, r) for p, r in zip(params, ref))
    assert torch.allclose(stat[1:], torch.full((9,), 2 ** 0.5)) and stat[0] == 0 and denom.sum() == 9 and radii.max() == 9
