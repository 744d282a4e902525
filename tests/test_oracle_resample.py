"""CPU tier: the explicit tap-table restatement (oracle/resample.py) against
F.interpolate - the third-party arithmetic the reference calls."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import resample as R


def _ref(img, oh, ow, mode):
    t = torch.from_numpy(img)[None]
    if mode == "aa":
        return F.interpolate(t, size=[oh, ow], mode="bilinear", align_corners=False, antialias=True)[0].numpy()
    if mode == "bilinear":
        return F.interpolate(t, size=[oh, ow], mode="bilinear")[0].numpy()
    return F.interpolate(t, size=[oh, ow], mode="bicubic")[0].numpy()


TAPS = {"aa": R.aa_bilinear_taps, "bilinear": R.bilinear_taps, "bicubic": R.bicubic_taps}


@pytest.mark.parametrize("mode,H,W,oh,ow", [
    ("aa", 512, 512, 336, 336), ("aa", 336, 336, 560, 560), ("aa", 50, 37, 33, 36), ("aa", 37, 91, 224, 112),
    ("aa", 48, 48, 48, 48), ("aa", 135, 199, 76, 112), ("aa", 300, 20, 28, 28),
    ("bilinear", 336, 336, 672, 672), ("bilinear", 100, 150, 448, 672), ("bilinear", 150, 100, 1008, 672),
    ("bicubic", 672, 672, 336, 336), ("bicubic", 672, 1008, 336, 336), ("bicubic", 336, 336, 336, 336),
])
def test_taps_reproduce_interpolate(mode, H, W, oh, ow):
    rng = np.random.default_rng(1)
    img = rng.random((3, H, W), dtype=np.float32)
    out = R.resize_separable(img, TAPS[mode](H, oh), TAPS[mode](W, ow))
    assert np.abs(out - _ref(img, oh, ow, mode)).max() <= 1e-6


def test_identity_is_exact_passthrough():
    rng = np.random.default_rng(2)
    img = rng.random((3, 31, 17), dtype=np.float32)
    for mode in TAPS:
        out = R.resize_separable(img, TAPS[mode](31, 31), TAPS[mode](17, 17))
        assert np.array_equal(out, img), mode


@pytest.mark.parametrize("mode,n,m", [("aa", 91, 32), ("aa", 20, 50), ("bilinear", 30, 77), ("bicubic", 96, 24)])
def test_transposed_taps_are_the_matrix_transpose(mode, n, m):
    taps = TAPS[mode](n, m)
    M = R.taps_to_matrix(taps, n)
    Mt = R.taps_to_matrix(R.transpose_taps(taps, n), m)
    assert np.array_equal(Mt, M.T)


@pytest.mark.parametrize("H,W,oh", [(594, 24, 16), (538, 21, 28), (40, 1, 90), (64, 5, 64)])
def test_width_one_result_follows_the_tap_tables(H, W, oh):
    """ATen's CPU antialias kernel mis-evaluates the height pass of a one-pixel-wide image (see
    oracle/processors.py::_aa); the oracle's `_aa` must give what the explicit tables give, with a
    gradient that is the transposed tables applied to the upstream gradient."""
    from oracle.processors import _aa
    rng = np.random.default_rng(3)
    img = rng.random((3, H, W), dtype=np.float32)
    x = torch.from_numpy(img).requires_grad_(True)
    out = _aa(x, oh, 1)
    assert tuple(out.shape) == (3, oh, 1)
    th, tw = R.aa_bilinear_taps(H, oh), R.aa_bilinear_taps(W, 1)
    assert np.abs(out.detach().numpy() - R.resize_separable(img, th, tw)).max() <= 1e-6
    up = rng.random((3, oh, 1), dtype=np.float32)
    out.backward(torch.from_numpy(up))
    Mh, Mw = R.taps_to_matrix(th, H).astype(np.float64), R.taps_to_matrix(tw, W).astype(np.float64)
    want = np.einsum("yh,cyx,xw->chw", Mh, up.astype(np.float64), Mw)
    assert np.abs(x.grad.numpy() - want).max() <= 1e-6
