"""GPU: the whole iterative multiview loop (uncond sample -> mesh -> warp -> conditional inpainting) at mini scale,
and the sharding invariant behind the sample-parallel multi-GPU mode: a sample's result does not depend on which
batch / rank it was generated in."""
import numpy as np
import pytest
import torch

import common as C

pytestmark = pytest.mark.gpu


def _frameworks():
    from ivid_amd.diffusion import frameworks
    from ivid_amd.diffusion.backbones import AdmUnet2d
    mu = AdmUnet2d(**C.MINI, precision="fp32"); mu.load_state_dict(C.synth_weights(C.MINI, 0)); mu = mu.cuda()
    mc = AdmUnet2d(**C.MINI_COND, precision="fp32"); mc.load_state_dict(C.synth_weights(C.MINI_COND, 2)); mc = mc.cuda()
    return (frameworks.ClassifierFreeGuidance(mu, timesteps=1000, beta_schedule="linear", p_uncond=0.1),
            frameworks.InpaintCFG(mc, timesteps=1000, beta_schedule="linear", p_uncond=0.1, p_uncond_img=0.0))


def _run(fu, fc, seeds, classes, batchsize, views):
    from ivid_amd.inference.sample import sample_all
    out = list(sample_all(fu, fc, seeds, 4, 3, views, classes=classes, guidance=0.5, batchsize=batchsize, erode_rgb=1))
    return [o[0].cpu() for o in out], [o[1] for o in out]


def test_multiview_loop_runs_and_is_partition_invariant():
    from ivid_amd import parallel
    from ivid_amd.rgbd_3d import camera
    fu, fc = _frameworks()
    views = camera.viewset("3x9")[:3]
    seeds, classes = [0, 1, 2, 3, 4], [0, 1, 2, 3, 4]
    full, conds = _run(fu, fc, seeds, classes, 5, views)
    assert len(full) == 5 and full[0].shape == (3, 4, 32, 32) and all(torch.isfinite(s).all() for s in full)
    assert conds[0]["color"].shape == (2, 3, 32, 32) and conds[0]["depth"].shape == (2, 1, 32, 32)
    # two "ranks" with the reference's strided partition, different batch sizes
    for rank in range(2):
        part, _ = _run(fu, fc, parallel.shard(seeds, rank, 2), parallel.shard(classes, rank, 2), 2, views)
        for k, s in zip(parallel.shard(list(range(5)), rank, 2), part):
            assert torch.allclose(s, full[k], rtol=1e-4, atol=1e-4), (rank, k, float((s - full[k]).abs().max()))


def test_random_viewset_and_uncond_only():
    from ivid_amd.rgbd_3d import camera
    fu, fc = _frameworks()
    views = camera.viewset("random", 2, np.random.default_rng(0))
    out, conds = _run(fu, fc, [7, 8], [1, 2], 2, views)
    assert out[0].shape == (2, 4, 32, 32) and conds[0]["color"].shape[0] == 1
    out, conds = _run(fu, None, [7], [1], 1, camera.viewset("uncond"))
    assert out[0].shape == (1, 4, 32, 32) and conds[0] is None
