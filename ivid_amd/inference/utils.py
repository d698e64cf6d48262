"""inference/utils.py helpers of the reference that the sampling path needs (parse_int_list :13-22, reorder :44-55)
plus PIL-based image writers (the reference uses imageio / torchvision, which are not dependencies here)."""
import os
import re

import numpy as np
import torch


def parse_int_list(s):
    """'1,2,5-10' -> [1, 2, 5, 6, 7, 8, 9, 10] (inference/utils.py:13-22)."""
    if isinstance(s, list):
        return s
    out = []
    rng = re.compile(r"^(\d+)-(\d+)$")
    for part in s.split(","):
        m = rng.match(part)
        out.extend(range(int(m.group(1)), int(m.group(2)) + 1) if m else [int(part)])
    return out


def reorder(x, viewset="3x9"):
    """Grid order of the 3x9 viewset (inference/utils.py:44-55).  Views are generated yaw-major / pitch-minor from the
    centre outwards (index = 3*yaw_slot + pitch_slot with yaw slots [0,+.15,-.15,+.3,-.3,...] and pitch slots [0,+.15,-.15]);
    the grid shows rows pitch -.15 / 0 / +.15 and columns yaw +.6 ... -.6.  A 26-entry list (the conditioning images,
    which have no entry for view 0) gets a blank (-1) image prepended first."""
    if viewset != "3x9":
        raise NotImplementedError(viewset)
    x = list(x)
    if len(x) == 26:
        x.insert(0, -torch.ones_like(x[0]))
    yaw_slots = [7, 5, 3, 1, 0, 2, 4, 6, 8]     # +.6, +.45, +.3, +.15, 0, -.15, -.3, -.45, -.6
    pitch_slots = [2, 0, 1]                     # -.15, 0, +.15
    return torch.stack([x[3 * y + p] for p in pitch_slots for y in yaw_slots], dim=0)


def to_uint8_image(chw):
    """[-1,1] CHW tensor -> HWC uint8 (np.clip(x*0.5+0.5,0,1)*255, sample.py:155)."""
    a = chw.detach().float().cpu().numpy().transpose(1, 2, 0) * 0.5 + 0.5
    return (np.clip(a, 0, 1) * 255).astype(np.uint8)


def save_png(path, chw):
    from PIL import Image
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(to_uint8_image(chw)).save(path)


def save_grid(path, nchw, nrow):
    from PIL import Image
    n, c, h, w = nchw.shape
    rows = (n + nrow - 1) // nrow
    canvas = np.zeros((rows * h, nrow * w, 3), dtype=np.uint8)
    for i in range(n):
        canvas[(i // nrow) * h:(i // nrow + 1) * h, (i % nrow) * w:(i % nrow + 1) * w] = to_uint8_image(nchw[i, :3])
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(canvas).save(path)
