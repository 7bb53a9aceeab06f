"""Synthetic image pairs for the benchmark configurations.

Image 1: band-limited colour noise (6 octaves) scaled to the bundled Middlebury pair's statistics
(mean ~90, sigma ~47); image 2: image 1 warped by a smooth two-region affine flow plus N(0,2) noise.
Ground-truth flow is returned for EPE sanity checks.  numpy only; deterministic in the seed.
"""
import numpy as np


def _octave_noise(rng, h, w, octaves=6):
    img = np.zeros((h, w), np.float64)
    for o in range(octaves):
        gh, gw = max(2, h >> (octaves - o)), max(2, w >> (octaves - o))
        g = rng.standard_normal((gh + 1, gw + 1))
        ys = np.linspace(0, gh, h, endpoint=False)
        xs = np.linspace(0, gw, w, endpoint=False)
        y0, x0 = ys.astype(int), xs.astype(int)
        fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
        a = g[y0][:, x0] * (1 - fx) + g[y0][:, x0 + 1] * fx
        b = g[y0 + 1][:, x0] * (1 - fx) + g[y0 + 1][:, x0 + 1] * fx
        img += (a * (1 - fy) + b * fy) * (0.5 ** (0.6 * o))
    return img


def _bilinear(img, x, y):
    h, w = img.shape[:2]
    x = np.clip(x, 0, w - 1.001)
    y = np.clip(y, 0, h - 1.001)
    x0, y0 = x.astype(int), y.astype(int)
    fx, fy = (x - x0)[..., None], (y - y0)[..., None]
    return (img[y0, x0] * (1 - fx) * (1 - fy) + img[y0, x0 + 1] * fx * (1 - fy) +
            img[y0 + 1, x0] * (1 - fx) * fy + img[y0 + 1, x0 + 1] * fx * fy)


def make_pair(h, w, seed=1234, max_flow=20.0):
    """Returns (img1, img2, u, v): uint8 (h,w,3) images and the float32 ground-truth flow 1->2."""
    rng = np.random.default_rng(seed)
    chans = []
    base = _octave_noise(rng, h, w)
    for _ in range(3):
        c = 0.7 * base + 0.3 * _octave_noise(rng, h, w)
        c = (c - c.mean()) / (c.std() + 1e-9)
        chans.append(c * 47.0 + 90.0)
    img1 = np.clip(np.stack(chans, -1), 0, 255)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    cx, cy = w / 2.0, h / 2.0

    def affine():
        t = rng.uniform(-0.5, 0.5, 2) * max_flow
        a = rng.uniform(-0.25, 0.25, 4) * max_flow / max(h, w) * 2
        return t[0] + a[0] * (xx - cx) + a[1] * (yy - cy), t[1] + a[2] * (xx - cx) + a[3] * (yy - cy)

    u1, v1 = affine()
    u2, v2 = affine()
    mask = _octave_noise(rng, h, w, 3) > 0.3          # second motion region
    u = np.where(mask, u2, u1)
    v = np.where(mask, v2, v1)
    u = np.clip(u, -max_flow, max_flow)
    v = np.clip(v, -max_flow, max_flow)
    # backward warp: img2(p + flow(p)) = img1(p) approximated by sampling img1 at p - flow(p)
    img2 = _bilinear(img1, xx - u, yy - v) + rng.normal(0, 2.0, (h, w, 3))
    return (img1.astype(np.uint8), np.clip(img2, 0, 255).astype(np.uint8), u.astype(np.float32), v.astype(np.float32))
