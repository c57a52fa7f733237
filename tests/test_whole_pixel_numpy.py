"""The oracle against an independent numpy-float32 re-derivation of whole pixels (tests/whole_pixel_numpy.py).

The reference holds no vectors to pin the oracle with, so a second restatement of the per-pixel function, written from
the Rust sources in another language and style, stands in: camera ray, closest hit, both shadow rays, Blinn-Phong, gamma,
the reflection chain (scene3), and march, normal, occlusion, tri-planar texture, fade (scene4).  Agreement is to the
accuracy of the transcendental functions (numpy's powf here, a specified algorithm in the oracle): 2e-5 absolute on the
linear value, one step on the bytes.  PARITY UNPINNED against reference outputs all the same.
"""
import numpy as np
import pytest

import whole_pixel_numpy as wp


def _pick(mask, n, rng):
    ys, xs = np.nonzero(mask)
    if len(ys) == 0:
        return []
    sel = rng.choice(len(ys), size=min(n, len(ys)), replace=False)
    return [(int(xs[i]), int(ys[i])) for i in sel]


@pytest.mark.parametrize("bounces", [1, 4])
def test_whitted_pixels_of_the_default_scene(pkg, orc, bounces):
    s = pkg.Scene("scene3")
    s.desc.max_ray_bounces = bounces
    w, h = 160, 120
    cam = s.camera(w, h)
    ref = orc.render(s.desc, cam, s.textures(), n_threads=4, want_linear=True)
    lin = ref["linear"]   # Scene::pixel before render_chunk clamps it
    rng = np.random.default_rng(7)
    # pixels of every kind: sky, the sphere, the lit plane, the plane in shadow (darker than the lit plane next to it)
    sky = lin[..., :3].sum(axis=2) == 0
    blue = (lin[..., 2] > lin[..., 0] * 1.5) & ~sky
    plane = ~sky & ~blue
    dark = plane & (lin[..., :3].sum(axis=2) < np.median(lin[..., :3].sum(axis=2)[plane]) * 0.7)
    pixels = _pick(sky, 2, rng) + _pick(blue, 8, rng) + _pick(plane & ~dark, 6, rng) + _pick(dark, 6, rng)
    assert len(pixels) >= 16
    tracer = wp.Whitted(pkg, s.desc)
    for x, y in pixels:
        p = tracer.pixel(cam, x, y)
        assert np.allclose(p, lin[y, x, :3], rtol=2e-5, atol=2e-5), f"pixel ({x}, {y}): numpy {p}, oracle {lin[y, x, :3]}"
        assert np.abs(wp.to_rgba(p).astype(int) - ref["rgba"][y, x].astype(int)).max() <= 1
    assert tracer.rays > len(pixels) * 2   # shadow and reflection rays were traced


def test_ray_marched_pixels_of_the_sdf_scene(pkg, orc):
    s = pkg.Scene("scene4")
    w, h = 96, 72
    cam = s.camera(w, h)
    tex = s.textures()
    ref = orc.render(s.desc, cam, tex, n_threads=4, want_linear=True)
    lin = ref["linear"]
    rng = np.random.default_rng(11)
    amb = np.array(s.desc.ambient_color[:], dtype=np.float32)
    miss = np.all(lin[..., :3] == amb, axis=2)
    pixels = _pick(miss, 2, rng) + _pick(~miss, 22, rng)
    m = wp.Marcher(pkg, s.desc, tex)
    for x, y in pixels:
        p = m.pixel(cam, x, y)
        # a march that ends within an ulp of the 0.001 hit threshold or the 40.0 limit can take one step more or less
        # under a different powf / sqrt rounding; none of the sampled pixels does
        assert np.allclose(p, lin[y, x, :3], rtol=5e-5, atol=2e-5), f"pixel ({x}, {y}): numpy {p}, oracle {lin[y, x, :3]}"
        assert np.abs(wp.to_rgba(p).astype(int) - ref["rgba"][y, x].astype(int)).max() <= 1
    # textured objects (tri-planar) and the untextured union are both among the samples
    assert 0 < m.textured_hits < len(pixels) - 2
    ro, rd = wp.camera_ray(cam, 0, 0)
    assert ro.dtype == np.float32 and rd.dtype == np.float32
