"""Pixel-space operations of the PGD loop, restated with stock torch-CPU ops
(oracle side; test infrastructure only).  All functions are differentiable through
torch autograd exactly as the reference's are.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------ a2  tanh reparam
def tanh_reparam(p, epsilon):
    """attack_model.py:300 / crossattack_models.py:329 : x = eps * tanh(p)."""
    return epsilon * torch.tanh(p)


# ------------------------------------------------------------------ a3  Gaussian blur
def gaussian_kernel1d(kernel_size, sigma):
    """torchvision 0.21 `_get_gaussian_kernel1d` (published algorithm; torchvision is not
    installed here -> torchvision-unverified): linspace(-(k-1)/2, (k-1)/2, k),
    exp(-0.5 (t/sigma)^2), normalised to sum 1, float32."""
    half = (kernel_size - 1) * 0.5
    t = torch.linspace(-half, half, steps=kernel_size, dtype=torch.float32)
    pdf = torch.exp(-0.5 * (t / sigma).pow(2))
    return pdf / pdf.sum()


def gaussian_blur(x, kernel_size, sigma):
    """torchvision `GaussianBlur(kernel_size, sigma)(x)` on a CHW float tensor, as used at
    attack_model.py:191-194,303-304 : reflect-pad k//2, depthwise conv with the 2-D
    product kernel outer(g, g)."""
    g = gaussian_kernel1d(kernel_size, sigma)
    k2d = torch.mm(g[:, None], g[None, :])
    C = x.shape[0]
    w = k2d.to(x.device).expand(C, 1, kernel_size, kernel_size)      # the CPU kernel, wherever x lives
    r = kernel_size // 2
    xp = F.pad(x.unsqueeze(0), [r, r, r, r], mode="reflect")
    return F.conv2d(xp, w, groups=C).squeeze(0)


# --------------------------------------------------------- a4  random resized crop
def random_resized_crop_params(height, width, scale, ratio, generator=None):
    """torchvision 0.21 `RandomResizedCrop.get_params` (published algorithm;
    torchvision-unverified): up to 10 tries, then centre-crop fallback.  Draws come from
    torch's CPU generator like the original's `torch.empty(1).uniform_` / `randint`."""
    area = height * width
    log_ratio = torch.log(torch.tensor(ratio))
    for _ in range(10):
        target_area = area * torch.empty(1).uniform_(scale[0], scale[1], generator=generator).item()
        aspect = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1], generator=generator)).item()
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if 0 < w <= width and 0 < h <= height:
            i = torch.randint(0, height - h + 1, size=(1,), generator=generator).item()
            j = torch.randint(0, width - w + 1, size=(1,), generator=generator).item()
            return i, j, h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def resized_crop(img, i, j, h, w, size):
    """torchvision `resized_crop` on a float CHW tensor: crop the window then
    AA-bilinear back to `size` (attack_model.py:309-310)."""
    win = img[:, i:i + h, j:j + w]
    return F.interpolate(win.unsqueeze(0), size=list(size), mode="bilinear",
                         align_corners=False, antialias=True).squeeze(0)


# ------------------------------------------------------------------ a5  image-fit loss
def image_fit_loss(x0, x, center_force=0.9):
    """attack_model.py:86-106.  The bounds passed by the caller are ignored there (Q2):
    effective penalty is relu(-s)^2 + relu(s - 0.9)^2 averaged over all elements."""
    s = x0 + x
    lower = torch.relu(center_force * torch.zeros_like(s) - s)
    upper = torch.relu(s - center_force * torch.ones_like(s))
    return torch.mean(lower ** 2 + upper ** 2)


# ------------------------------------------------------------------ a13 mask
def create_mask(mask_type, mask_size, shape):
    """attack_model.py:66-84 for the usable types (corner / bottom_lines / none)."""
    C, H, W = shape
    if mask_type == "corner":
        m = torch.zeros(shape)
        m[:, :mask_size, :mask_size] = 1.0
        return m
    if mask_type == "bottom_lines":
        m = torch.zeros(shape)
        m[:, -mask_size:, :] = 1.0
        return m
    return torch.ones(shape)


# ------------------------------------------------------------------ a15 quantise error
def quantise(s):
    """tensor2pil + pil_to_tensor round trip (llavaprocessor.py:151-161): PNG is lossless
    so the round trip equals uint8 TRUNCATION of clamp(s,0,1)*255, divided by 255 (Q1)."""
    q = (s.detach().clamp(0, 1) * 255).cpu().numpy().astype(np.uint8)     # tensor2pil moves to the CPU too (:153)
    return torch.tensor(q.astype(np.float32) / 255).to(s.device)


def quantise_error_stats(s):
    """attack_model.py:366-373,389-391: (std_unbiased, mean, l1) of |q - s|."""
    d = (quantise(s) - s.detach()).abs()
    return d.std(), d.mean(), d.sum()


# ------------------------------------------------------------------ a10 broadcast+noise
def broadcast_with_noise(pixel_values, batch, unit_noise, sigma):
    """attack_model.py:316-321: repeat along dim 0, add randn * sigma.  `unit_noise` is the
    N(0,1) draw (shape of the repeated tensor) supplied as an input for parity."""
    rep = [1] * pixel_values.dim()
    rep[0] = batch
    pv = pixel_values.repeat(rep)
    if unit_noise is None:
        return pv
    return pv + unit_noise * sigma
