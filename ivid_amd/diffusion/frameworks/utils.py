"""Beta schedules (reference: /root/reference/diffusion/frameworks/utils.py:7-61), float64 host tables."""
import numpy as np


def get_betas_by_name(schedule_name, num_diffusion_timesteps):
    n = int(num_diffusion_timesteps)
    if schedule_name == "linear":
        s = 1000 / n  # Ho et al. schedule rescaled to n steps (utils.py:22-28)
        return np.linspace(s * 0.0001, s * 0.02, n, dtype=np.float64)
    if schedule_name == "cosine":
        abar = lambda u: np.cos((u + 0.008) / 1.008 * np.pi / 2) ** 2  # utils.py:30-33
        i = np.arange(n, dtype=np.float64)
        return np.minimum(1 - abar((i + 1) / n) / abar(i / n), 0.999)
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")
