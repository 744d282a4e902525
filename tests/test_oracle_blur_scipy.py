"""CPU tier: an independent second opinion on the oracle's Gaussian blur.

torchvision is not installed here, so `oracle.pixel_ops.gaussian_blur` restates
`GaussianBlur` from its published algorithm and DESIGN.md marks the row "parity unpinned".
This does not pin it to torchvision; it checks the restated arithmetic (reflect padding that
does not repeat the edge, separable product kernel, unit gain) against scipy.ndimage, which
shares no code with torch."""
import numpy as np
import pytest
import torch
from scipy import ndimage

from oracle import pixel_ops as P


@pytest.mark.parametrize("k,sigma", [(5, 7.0), (9, 10.0), (5, 0.15), (3, 1.9)])
def test_blur_matches_scipy_mirror_correlation(k, sigma):
    x = torch.rand(3, 37, 29, generator=torch.Generator().manual_seed(k))
    got = P.gaussian_blur(x, k, sigma).numpy()
    g = P.gaussian_kernel1d(k, sigma).numpy().astype(np.float64)
    assert abs(g.sum() - 1.0) < 1e-6 and np.allclose(g, g[::-1])
    want = x.numpy().astype(np.float64)
    for axis in (1, 2):
        want = ndimage.correlate1d(want, g, axis=axis, mode="mirror")      # 'mirror' = torch 'reflect'
    assert np.abs(got - want).max() < 2e-6


def test_blur_preserves_constants_and_is_self_adjoint_inside():
    x = torch.full((3, 20, 20), 0.37)
    assert torch.allclose(P.gaussian_blur(x, 5, 7.0), x, atol=1e-6)
