"""Known-answer tests for the correlation lookups that do not use the restatement's own arithmetic (CPU).

The reference has no CPU lookup and no test vectors for it (correlation_kernels.cu / altcorr_kernel.cu are CUDA only),
so the C oracle's lookup is pinned here against an INDEPENDENT statement of what those kernels compute: output
channel a*(2r+1)+b of a pixel is the bilinear sample, with zero padding, of that pixel's (h2, w2) correlation
plane at (x - r + a, y - r + b)  --  derived from the four read-modify-writes of correlation_kernels.cu:55-65
 and altcorr_kernel.cu:102-125.  The float32 oracle must agree to float32 rounding; the float16
oracle within the bound that its seven roundings allow."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import oracle as orc  # noqa: E402


def bilinear_window_f64(plane, x, y, r):
    """plane [..., h2, w2] float64, x / y [...] float64 -> [..., 2r+1 (a: x offset), 2r+1 (b: y offset)]"""
    h2, w2 = plane.shape[-2:]
    rd = 2 * r + 1
    pad = np.zeros(plane.shape[:-2] + (h2 + 2 * (rd + 2), w2 + 2 * (rd + 2)))
    o = rd + 2
    pad[..., o:o + h2, o:o + w2] = plane
    fx, fy = np.floor(x), np.floor(y)
    dx, dy = x - fx, y - fy
    ix = fx.astype(np.int64) - r
    iy = fy.astype(np.int64) - r
    out = np.zeros(x.shape + (rd, rd))
    lead = np.indices(x.shape)
    for a in range(rd):
        for b in range(rd):
            xx = np.clip(ix + a + o, 0, pad.shape[-1] - 2)
            yy = np.clip(iy + b + o, 0, pad.shape[-2] - 2)
            inside = (ix + a + o >= 0) & (ix + a + o <= pad.shape[-1] - 2) & (iy + b + o >= 0) & (
                iy + b + o <= pad.shape[-2] - 2)
            idx = tuple(lead)
            s00 = pad[idx + (yy, xx)]
            s01 = pad[idx + (yy + 1, xx)]      # one row down (y + 1)
            s10 = pad[idx + (yy, xx + 1)]      # one column right (x + 1)
            s11 = pad[idx + (yy + 1, xx + 1)]
            v = (1 - dx) * (1 - dy) * s00 + (1 - dx) * dy * s01 + dx * (1 - dy) * s10 + dx * dy * s11
            out[..., a, b] = np.where(inside, v, 0.0)
    return out


def _case(seed, n, h1, w1, h2, w2, spread=3.0, oob=0.1):
    rng = np.random.default_rng(seed)
    vol = rng.standard_normal((n, h1, w1, h2, w2))
    yy, xx = np.meshgrid(np.arange(h1), np.arange(w1), indexing="ij")
    coords = np.stack([xx * (w2 / w1) + spread * rng.standard_normal((n, h1, w1)),
                       yy * (h2 / h1) + spread * rng.standard_normal((n, h1, w1))], 1)  # [n, 2 (x, y), h1, w1]
    push = rng.uniform(size=(n, h1, w1)) < oob
    coords[:, 0][push] += rng.choice([-1.0, 1.0], size=int(push.sum())) * rng.uniform(5, 40, size=int(push.sum()))
    coords[0, :, 0, 0] = (2.0, 3.0)            # integer coordinates: dx = dy = 0
    coords[0, :, 0, 1] = (-0.5, h2 - 0.5)      # straddles two borders
    return vol, coords.astype(np.float32)


@pytest.mark.parametrize("shape", [(2, 6, 7, 12, 16), (1, 5, 5, 9, 11)])
@pytest.mark.parametrize("radius", [3, 2])
def test_f32_lookup_is_the_zero_padded_bilinear_window(shape, radius):
    vol, coords = _case(1, *shape)
    got = orc.corr_index_forward(vol.astype(np.float32), coords, radius)       # [n, rd (a), rd (b), h1, w1]
    ref = bilinear_window_f64(vol.astype(np.float32).astype(np.float64), coords[:, 0].astype(np.float64),
                              coords[:, 1].astype(np.float64), radius)           # [n, h1, w1, a, b]
    ref = ref.transpose(0, 3, 4, 1, 2)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)                     # |taps| ~ 1: a few f32 ulps (measured 2e-7)
    assert np.abs(ref).max() > 1.0


def test_lookup_channel_order_is_x_offset_major():
    """a single non-zero volume entry lands in exactly the channels the reference's index arithmetic gives it"""
    n, h1, w1, h2, w2, r = 1, 1, 1, 12, 12, 3
    vol = np.zeros((n, h1, w1, h2, w2), np.float32)
    vol[0, 0, 0, 7, 4] = 1.0                                     # target row (y) 7, column (x) 4
    coords = np.array([[[[5.0]], [[6.0]]]], np.float32)          # x = 5, y = 6 exactly
    got = orc.corr_index_forward(vol, coords, r)
    a, b = 4 - (5 - r), 7 - (6 - r)                              # x offset 2, y offset 4
    expect = np.zeros((2 * r + 1, 2 * r + 1), np.float32)
    expect[a, b] = 1.0
    assert np.array_equal(got[0, :, :, 0, 0], expect)
    flat = got.reshape(n, -1, h1, w1)[0, :, 0, 0]                # CorrBlock.__call__ views [n, 49, h, w]
    assert flat[a * (2 * r + 1) + b] == 1.0 and flat.sum() == 1.0


def test_f16_lookup_is_within_its_rounding_bound_of_the_exact_window():
    """half path: weights rounded to half, four products and three sums each rounded to half (u = 2^-11):
    |err| <= (2u + 3u + O(u^2)) * sum|w s| + subnormal slack"""
    radius = 3
    vol, coords = _case(2, 2, 6, 7, 12, 16)
    vol16 = (4.0 * vol).astype(np.float16)
    got = orc.corr_index_forward(vol16, coords, radius).astype(np.float64)
    exact = bilinear_window_f64(vol16.astype(np.float64), coords[:, 0].astype(np.float64),
                                coords[:, 1].astype(np.float64), radius).transpose(0, 3, 4, 1, 2)
    mass = bilinear_window_f64(np.abs(vol16.astype(np.float64)), coords[:, 0].astype(np.float64),
                               coords[:, 1].astype(np.float64), radius).transpose(0, 3, 4, 1, 2)
    u = 2.0 ** -11
    bound = 5.1 * u * mass + 8 * 2.0 ** -25
    assert (np.abs(got - exact) <= bound).all(), float((np.abs(got - exact) / bound).max())
    assert np.abs(got - exact).max() > 1e-4   # the rounding is really there: f16 is not silently computed in f32


@pytest.mark.parametrize("radius", [1, 2, 3, 4])
def test_altcorr_is_the_bilinear_window_of_the_feature_dot_products(radius):
    """altcorr_forward (altcorr_kernel.cu:27-149): the same window, of s(h2, w2) = <fmap1[pixel], fmap2[h2, w2]>,
    channel = y_offset + (2r+1) * x_offset"""
    rng = np.random.default_rng(3)
    B, S, H1, W1, H2, W2, C = 2, 2, 5, 6, 8, 9, 64
    f1 = rng.standard_normal((B, H1, W1, C)).astype(np.float32)
    f2 = rng.standard_normal((B, H2, W2, C)).astype(np.float32)
    yy, xx = np.meshgrid(np.arange(H1), np.arange(W1), indexing="ij")
    coords = np.stack([xx * (W2 / W1) + 2.0 * rng.standard_normal((B, S, H1, W1)),
                       yy * (H2 / H1) + 2.0 * rng.standard_normal((B, S, H1, W1))], -1).astype(np.float32)
    got = orc.altcorr_forward(f1, f2, coords, radius)            # [B, S, rd*rd, H1, W1]
    plane = np.einsum("byxc,bhwc->byxhw", f1.astype(np.float64), f2.astype(np.float64))   # [B, H1, W1, H2, W2]
    rd = 2 * radius + 1
    for s in range(S):
        ref = bilinear_window_f64(plane, coords[:, s, :, :, 0].astype(np.float64),
                                  coords[:, s, :, :, 1].astype(np.float64), radius)        # [B, H1, W1, a (x), b (y)]
        ref = ref.reshape(B, H1, W1, rd * rd).transpose(0, 3, 1, 2)                         # channel = a * rd + b
        np.testing.assert_allclose(got[:, s], ref, rtol=0, atol=2e-4 * np.sqrt(C / 64))
