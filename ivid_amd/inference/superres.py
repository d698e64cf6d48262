"""128 -> 256 super-resolution of generated views (BASELINE config 5).

The reference ships the SR model (`rgbd_imagenet_adm_256_128_small_sr.json`, `SuperResCFG`, sr_cfg.py) but no inference
driver for it; its only sampling call is the trainer's preview (`SuperResTrainer.sample`, trainers/superres.py:120-124:
`sampler.sample(batch, y=..., classes=..., steps=50, strength=3.0)`).  This is that call as a function over the views
produced by `sample_all`, so the chain  uncond -> warp/inpaint views -> SR  runs on the GPU end to end."""
import os

import torch

from ..diffusion import samplers


@torch.no_grad()
def super_resolve(framework_sr, views, classes=None, steps=50, strength=3.0, batchsize=8, noise_fn=None):
    """views: [V,4,s,s] low-resolution RGBD in [-1,1] (device) -> [V,4,S,S] with S = framework_sr.backbone.image_size.
    classes: None, an int (one class for every view of the sample) or a [V] tensor."""
    sampler = samplers.DdimSampler(framework_sr)
    dev = views.device
    out = []
    extra = {"noise_fn": noise_fn} if noise_fn is not None else {}
    if os.environ.get("IVID_DEVICE_LOOP", "0") == "1":   # every chain as ONE C call (ivid_sample); bit-identical samples
        extra["device_loop"] = True
    for i in range(0, views.shape[0], batchsize):
        y = views[i:i + batchsize].float().contiguous()
        b = y.shape[0]
        if classes is None:
            cls = None
        elif isinstance(classes, int):
            cls = torch.full((b,), classes, dtype=torch.long, device=dev)
        else:
            cls = classes[i:i + b].to(dev)
        kw = dict(strength=strength) if cls is not None else {}
        res = sampler.sample(b, classes=cls, steps=steps, verbose=False, keep_intermediates=False, y=y, **kw, **extra)
        out.append(res.samples)
    return torch.cat(out, dim=0)
