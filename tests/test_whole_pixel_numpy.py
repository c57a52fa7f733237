"""The oracle against an independent numpy-float32 re-derivation of whole pixels (tests/whole_pixel_numpy.py).

The reference holds no vectors to pin the oracle with, so a second restatement of the per-pixel function, written from
the Rust sources in another language and style, stands in: camera ray, closest hit, both shadow rays, Blinn-Phong, gamma,
the reflection chain (scene3), all six primitives' intersect() with their UVs and textures (scene1), the Refractive arm with Fresnel weight, refraction ray and total internal reflection (a scene
with two glass spheres), and march, normal, occlusion, tri-planar texture, fade (scene4).  Agreement is to the
accuracy of the transcendental functions (numpy's powf here, a specified algorithm in the oracle): 2e-5 absolute on the
linear value, one step on the bytes.  PARITY UNPINNED against reference outputs all the same.
"""
import numpy as np
import pytest

import whole_pixel_numpy as wp
from helpers import SceneBuilder, make_light, make_material, make_object


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


def test_whitted_pixels_through_glass(pkg, orc):
    """The Refractive arm of RayTracing::color (src/ray_tracing/ray_tracing.rs:197-236), Material::fresnel and
    Ray::refraction_ray re-derived in numpy: a plane, two opaque spheres and two glass spheres (refraction index above and
    below 1: entering, leaving, the n1 > n2 branch of fresnel and total internal reflection), two lights with shadows."""
    mats = [make_material(pkg, albedo=(0.5, 0.5, 0.5), roughness=0.7, ambience=0.5, shininess=8.0, specular=0.6),
            make_material(pkg, albedo=(0.9, 0.3, 0.2), roughness=0.4, ambience=0.4, shininess=40.0, specular=1.0),
            make_material(pkg, albedo=(0.95, 0.97, 1.0), kind=pkg.RT_REFRACTIVE, transparency=0.9, refraction_index=1.5, reflectivity=0.1, ambience=0.3),
            make_material(pkg, albedo=(0.8, 1.0, 0.8), kind=pkg.RT_REFRACTIVE, transparency=0.8, refraction_index=0.8, reflectivity=0.2, ambience=0.3)]
    objs = [make_object(pkg, pkg.RT_PLANE, 0, normal=(0, 1, 0), point=(0, -0.5, 0), max_dist=(8.0, 8.0), has_max_dist=1),
            make_object(pkg, pkg.RT_SPHERE, 1, position=(-1.4, 0.1, -1.0), radius=0.6),
            make_object(pkg, pkg.RT_SPHERE, 2, position=(0.0, 0.2, 0.8), radius=0.7),
            make_object(pkg, pkg.RT_SPHERE, 3, position=(1.5, 0.1, 0.2), radius=0.6),
            make_object(pkg, pkg.RT_SPHERE, 1, position=(0.6, 0.0, -2.2), rotation_axis=(20.0, 40.0, 0.0), radius=0.5)]
    lights = [make_light(pkg, pkg.RT_POSITIONAL, (1, 1, 1), (2.0, 5.0, 4.0), 25.0), make_light(pkg, pkg.RT_DIRECTIONAL, (0.8, 0.9, 1.0), (-0.3, -1.0, -0.4), 0.7)]
    sb = SceneBuilder(pkg, objs, mats, lights, max_ray_bounces=4, shadow_casting=1, ambient_color=(0.1, 0.15, 0.3))
    w, h = 128, 96
    cam = pkg.make_camera((0.2, 1.0, 4.5), (0.0, -0.15, -1.0), w, h)
    ref = orc.render(sb.desc, cam, (), n_threads=4, want_linear=True, want_counters=True)
    lin = ref["linear"]
    counters = dict(zip(pkg.COUNTER_NAMES, ref["counters"]))
    assert counters["refraction"] > 500
    tracer = wp.Whitted(pkg, sb.desc)
    # pixels whose camera ray meets glass first, found with the numpy tracer's own closest hit; and some that do not
    glass, other = [], []
    rng = np.random.default_rng(3)
    for _ in range(4000):
        x, y = int(rng.integers(0, w)), int(rng.integers(0, h))
        hit = tracer.trace_ray(*wp.camera_ray(cam, x, y))
        first = None if hit is None else sb.desc.objects[hit[1]].material_index
        (glass if first in (2, 3) else other).append((x, y, first))
        if len(glass) >= 40 and len(other) >= 8:
            break
    n2 = [g for g in glass if g[2] == 2][:14]
    n3 = [g for g in glass if g[2] == 3][:14]
    assert len(n2) >= 8 and len(n3) >= 8
    for x, y, _ in n2 + n3 + other[:8]:
        p = tracer.pixel(cam, x, y)
        assert np.allclose(p, lin[y, x, :3], rtol=5e-5, atol=3e-5), f"pixel ({x}, {y}): numpy {p}, oracle {lin[y, x, :3]}"
        assert np.abs(wp.to_rgba(p).astype(int) - ref["rgba"][y, x].astype(int)).max() <= 1
    assert tracer.refractive_nodes >= len(n2) + len(n3) and tracer.total_internal > 0


def test_whitted_pixels_of_the_all_primitives_scene(pkg, orc):
    """src/bin/scene1.rs: every primitive kind (sphere, bounded plane, cuboid, cone, cylinder, triangle), textured materials
    with the primitives' own UVs (atan-derived on the sphere, cylinder and cone), a glass sphere: every intersect() of the
    reference re-derived in numpy and checked against the oracle on camera pixels of every object, reflections included."""
    s = pkg.Scene("scene1")
    s.desc.max_ray_bounces = 3
    w, h = 160, 120
    cam = s.camera(w, h)
    tex = s.textures()
    ref = orc.render(s.desc, cam, tex, n_threads=4, want_linear=True)
    lin = ref["linear"]
    tracer = wp.Whitted(pkg, s.desc, tex)
    rng = np.random.default_rng(5)
    # sample until the camera rays have met every object of the scene at least twice
    seen, pixels = {}, []
    for _ in range(6000):
        x, y = int(rng.integers(0, w)), int(rng.integers(0, h))
        hit = tracer.trace_ray(*wp.camera_ray(cam, x, y))
        key = -1 if hit is None else hit[1]
        if seen.get(key, 0) < 4:
            seen[key] = seen.get(key, 0) + 1
            pixels.append((x, y))
    assert len([k for k in seen if k >= 0]) == s.desc.n_objects, f"objects seen: {sorted(seen)}"
    flips = 0
    for x, y in pixels:
        p = tracer.pixel(cam, x, y)
        if not np.allclose(p, lin[y, x, :3], rtol=5e-5, atol=3e-5):
            # numpy's atan against the specified atanf: a UV within an ulp of a texel border may pick the neighbouring texel
            flips += 1
            assert flips <= 2, f"pixel ({x}, {y}): numpy {p}, oracle {lin[y, x, :3]}"
    assert tracer.kinds_hit >= {pkg.RT_SPHERE, pkg.RT_PLANE, pkg.RT_CUBOID, pkg.RT_CONE, pkg.RT_CYLINDER, pkg.RT_TRIANGLE}
    assert tracer.textured_hits > 10 and tracer.refractive_nodes > 0


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
