"""Oracle (test infrastructure): noise schedule and EDM scalar formulas."""
import math

import torch


def cosine_schedule(timesteps=1000):
    """model1.py:674-684 (betas) and model1.py:658-665 (derived buffers).

    The cosine alpha-bar curve is only used to derive betas; after the clamp to
    [1e-4, 0.02] the alphas and alpha-bars are recomputed from the clamped betas.
    """
    s = 0.008
    x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float32)
    ac = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = torch.clamp(1 - (ac[1:] / ac[:-1]), 0.0001, 0.02)
    alphas = 1.0 - betas
    alphas_cumprod = torch.cumprod(alphas, dim=0)
    return {
        "betas": betas,
        "alphas": alphas,
        "alphas_cumprod": alphas_cumprod,
        "sqrt_alphas_cumprod": torch.sqrt(alphas_cumprod),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - alphas_cumprod),
    }


def edm_sigma_from_u(u, sigma_min=0.002, sigma_max=80.0, rho=7.0):
    """model1.py:36-41 with the uniform draw u supplied by the caller."""
    return (sigma_max ** (1 / rho) + u * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho


def edm_scalings(sigma, sigma_data=0.5):
    """model1.py:27-34.  sigma is [B] with B >= 2 (the reference's .squeeze() makes B == 1
    zero-dimensional, which its own callers cannot index; that case is outside the
    parity domain).  Returns c_skip, c_out, c_in, c_noise, each [B]."""
    s = sigma.view(-1, 1)
    c_skip = sigma_data ** 2 / (s ** 2 + sigma_data ** 2)
    c_out = s * sigma_data / (s ** 2 + sigma_data ** 2).sqrt()
    c_in = 1 / (sigma_data ** 2 + s ** 2).sqrt()
    c_noise = s.log() / 4
    return c_skip.squeeze(-1), c_out.squeeze(-1), c_in.squeeze(-1), c_noise.squeeze(-1)


def edm_loss_weight(sigma, sigma_data=0.5):
    """model1.py:43-45."""
    return (sigma ** 2 + sigma_data ** 2) / (sigma * sigma_data) ** 2
