"""Known-answer tests of the CPU oracle, hand-derived from the reference's formulas.

The reference ships no tests or golden vectors, so these anchors (closed-form geometry, the documented quirks of
each intersect(), the texture wrap rule, to_rgba's saturating casts) are what pins the restatement.  Each case
cites the reference lines whose behaviour it checks.  PARITY UNPINNED against reference *outputs*.
"""
import ctypes as C
import math

import numpy as np
import pytest

from helpers import SceneBuilder, intersect, make_light, make_material, make_object


def test_camera_default_anchor(pkg, orc):
    """camera.rs:22-53,128-157: pos (0,0,3), forward (0,0,-1), fov 45 passed as radians, near -1, far -100."""
    cam = pkg.make_camera((0, 0, 3), (0, 0, -1), 800, 600)
    d = (C.c_float * 3)()
    orc.load().orc_camera_ray(C.addressof(cam), 0, 0, C.addressof(d))
    h = math.cos(22.5) / math.sin(22.5)
    w = h / (800 / 600)
    expect = np.array([-100 / w - 0, 100 / h - 0, -100 - 3.0])  # sx = sy = -1: (100 sx / w, -100 sy / h, -100) - position
    expect = expect / np.linalg.norm(expect)
    assert np.allclose(np.array(d[:]), expect, atol=2e-5)
    # forward is -z, image row 0 looks up (+y), column 0 looks left (-x)
    assert d[2] < 0 and d[1] > 0 and d[0] < 0
    orc.load().orc_camera_ray(C.addressof(cam), 400, 300, C.addressof(d))
    assert np.allclose(np.array(d[:]), [0, 0, -1], atol=1e-6)


def test_sphere_near_root_only_and_uv_swap(pkg, orc):
    """sphere.rs:98-147: unit sphere under T*R*S(r); only the near root; u/v swapped and divided by pi."""
    s = make_object(pkg, pkg.RT_SPHERE, position=(0, 0, 0), radius=2.0)
    ok, h = intersect(orc, s, (0, 0, 5), (0, 0, -1))
    assert ok and h[0] == pytest.approx(3.0) and np.allclose(h[1:4], [0, 0, 2]) and np.allclose(h[4:7], [0, 0, 1])
    # a ray starting inside returns the negative near root (trace_ray then rejects it: no fallback to the far root)
    ok, h = intersect(orc, s, (0, 0, 0), (0, 0, -1))
    assert ok and h[0] == pytest.approx(-2.0)
    # grazing miss
    ok, _ = intersect(orc, s, (2.5, 0, 5), (0, 0, -1))
    assert not ok
    # uv: l = object-space hit; hit.u = atan(ly/lx)/pi, hit.v = atan((lx^2+ly^2)/lz)/pi
    ok, h = intersect(orc, s, (1.0, 0.5, 5), (0, 0, -1))
    lx, ly = 0.5, 0.25
    lz = math.sqrt(1 - lx * lx - ly * ly)
    assert h[7] == pytest.approx(math.atan(ly / lx) / math.pi, abs=1e-6)
    assert h[8] == pytest.approx(math.atan((lx * lx + ly * ly) / lz) / math.pi, abs=1e-6)


def test_plane_bounds_and_two_sided_normal(pkg, orc):
    """plane.rs:41-74: |d.n| < 1e-6 and t < 0 reject; bounds on world |x|, |z|; normal faces the ray; uv = 0.1 (x, z)."""
    p = make_object(pkg, pkg.RT_PLANE, normal=(0, 1, 0), point=(0, 0, 0), max_dist=(5, 5), has_max_dist=1)
    ok, h = intersect(orc, p, (1, 2, 3), (0, -1, 0))
    assert ok and h[0] == pytest.approx(2.0) and np.allclose(h[4:7], [0, 1, 0]) and h[7] == pytest.approx(0.1) and h[8] == pytest.approx(0.3)
    ok, h = intersect(orc, p, (1, -2, 3), (0, 1, 0))
    assert ok and np.allclose(h[4:7], [0, -1, 0])
    assert not intersect(orc, p, (6, 2, 0), (0, -1, 0))[0]      # outside |x| bound
    assert not intersect(orc, p, (0, 2, 0), (1, 0, 0))[0]       # parallel
    assert not intersect(orc, p, (0, 2, 0), (0, 1, 0))[0]       # behind
    p.has_max_dist = 0
    assert intersect(orc, p, (600, 2, 0), (0, -1, 0))[0]


def test_cuboid_slab_and_face_normal(pkg, orc):
    """cuboid.rs:84-127: b_min/b_max = -/+dimension; normal = -signum(d) * step(t_near, t_enter)."""
    c = make_object(pkg, pkg.RT_CUBOID, position=(0, 0, 0), dimension=(1, 2, 3))
    ok, h = intersect(orc, c, (0.25, 0.5, 10), (0, 0, -1))
    assert ok and h[0] == pytest.approx(7.0) and np.allclose(h[4:7], [0, 0, 1]) and np.allclose(h[7:9], [0.25, 0.5])
    ok, h = intersect(orc, c, (-10, 0.5, 0.25), (1, 0, 0))
    assert ok and h[0] == pytest.approx(9.0) and np.allclose(h[4:7], [-1, 0, 0])
    assert not intersect(orc, c, (5, 5, 10), (0, 0, -1))[0]
    # origin inside: t_near negative is returned as is (trace_ray rejects it)
    ok, h = intersect(orc, c, (0, 0, 0), (0, 0, -1))
    assert ok and h[0] == pytest.approx(-3.0)


def test_cylinder_is_unit_and_ignores_radius_height(pkg, orc):
    """cylinder.rs:45-54,90-174: the transform omits the scale, so intersect() is always r = 1, |z| < 0.5."""
    c = make_object(pkg, pkg.RT_CYLINDER, position=(0, 0, 0), radius=0.4, height=3.0)
    ok, h = intersect(orc, c, (5, 0, 0.2), (-1, 0, 0))
    assert ok and h[0] == pytest.approx(4.0) and np.allclose(h[4:7], [1, 0, 0]) and h[8] == pytest.approx(0.4)
    ok, h = intersect(orc, c, (0.3, 0.2, 5), (0, 0, -1))     # cap at z = +0.5
    assert ok and h[0] == pytest.approx(4.5) and np.allclose(h[4:7], [0, 0, 1]) and np.allclose(h[7:9], [0.3, 0.2])
    assert not intersect(orc, c, (5, 0, 2.0), (-1, 0, 0))[0]  # above the unit height even though height = 3
    # side root valid but negative: returned immediately, caps are never tried
    ok, h = intersect(orc, c, (3, 0, 0), (1, 0, 0))
    assert ok and h[0] == pytest.approx(-4.0)


def test_cone_side_and_cap(pkg, orc):
    """cone.rs:73-152: x^2 + y^2 = z^2 for 0 < z < 1, cap disc at z = 1; the nearer of side and cap wins."""
    c = make_object(pkg, pkg.RT_CONE, position=(0, 0, 0), radius=1.0, height=1.0)
    ok, h = intersect(orc, c, (0.25, 0, 5), (0, 0, -1))
    assert ok and h[0] == pytest.approx(4.0) and np.allclose(h[4:7], [0, 0, 1])  # cap first
    ok, h = intersect(orc, c, (5, 0, 0.5), (-1, 0, 0))
    assert ok and h[0] == pytest.approx(4.5)                                      # lateral surface at x = 0.5
    n = np.array([0.5, 0, -0.5]); n /= np.linalg.norm(n)
    assert np.allclose(h[4:7], n, atol=1e-6) and h[8] == pytest.approx(2.0)
    assert not intersect(orc, c, (5, 0, 1.5), (-1, 0, 0))[0]


def test_triangle_edge_tests_and_unnormalised_uv(pkg, orc):
    """triangle.rs:35-89: no t < 0 rejection inside intersect(); strict edge tests; normal faces the ray."""
    t = make_object(pkg, pkg.RT_TRIANGLE, v1=(1.5, 1, 0), v2=(1.5, 0, 0), v3=(2.5, 1, 0))
    ok, h = intersect(orc, t, (1.8, 0.8, 4), (0, 0, -1))
    assert ok and h[0] == pytest.approx(4.0) and np.allclose(h[4:7], [0, 0, 1])
    ok, h = intersect(orc, t, (1.8, 0.8, -4), (0, 0, 1))
    assert ok and np.allclose(h[4:7], [0, 0, -1])
    assert not intersect(orc, t, (2.4, 0.2, 4), (0, 0, -1))[0]
    ok, h = intersect(orc, t, (1.8, 0.8, 4), (0, 0, 1))   # plane behind the origin: still Some, t negative
    assert ok and h[0] == pytest.approx(-4.0)


def test_texture_wrap_rule(pkg, orc):
    """objects.rs:121-137: p < 0 -> 1 - |ceil(p) - p|; p > 1 -> fract; texel = ((dim-1) * uu) as u32 (truncation)."""
    tex = np.zeros((4, 8, 3), dtype=np.uint8)
    for y in range(4):
        for x in range(8):
            tex[y, x] = (x * 10, y * 10, x + y)
    t = orc.orc_texture()
    t.width, t.height, t.rgb8 = 8, 4, tex.ctypes.data
    out = (C.c_float * 3)()

    def sample(u, v):
        orc.load().orc_texture_from_uv(C.addressof(t), u, v, C.addressof(out))
        return round(out[0] * 255 / 10), round(out[1] * 255 / 10)

    assert sample(0.0, 0.0) == (0, 0) and sample(1.0, 1.0) == (7, 3)
    assert sample(0.999, 0.0) == (6, 0)           # (7 * 0.999) truncates to 6
    assert sample(1.25, 0.0) == (1, 0)            # fract(1.25) = 0.25 -> 1.75 -> 1
    assert sample(-0.25, 0.0) == (5, 0)           # 1 - |0 - (-0.25)| = 0.75 -> 5.25 -> 5
    assert sample(float("nan"), 0.5) == (0, 1)    # NaN -> texel 0
    assert sample(float("inf"), float("-inf")) == (0, 0)


def test_to_rgba_saturates_and_nan_is_zero(pkg, orc):
    """renderer.rs:29-36,64-68: clamp then (c*255) as u8 truncates; NaN lanes become 0; alpha is 255."""
    mats = [make_material(pkg, albedo=(0.5, 0.5, 0.5))]
    sb = SceneBuilder(pkg, [], mats, ambient_color=(2.0, 0.2509, float("nan")), max_ray_bounces=4)
    cam = pkg.make_camera((0, 0, 3), (0, 0, -1), 16, 16)
    r = orc.render(sb.desc, cam)
    assert (r["rgba"][..., 0] == 255).all() and (r["rgba"][..., 1] == 63).all() and (r["rgba"][..., 2] == 0).all()
    assert (r["rgba"][..., 3] == 255).all()
    assert (r["accum"][..., 2] == 0).all() and (r["accum"][..., 0] == 1).all()


def test_trace_ray_lowest_index_wins_ties_and_inside_rays_miss(pkg, orc):
    """ray_tracing.rs:59-78: strict `<` keeps the first of two coincident hits; t > 0 drops rays starting inside."""
    mats = [make_material(pkg, albedo=(1, 0, 0), emission_power=1.0), make_material(pkg, albedo=(0, 1, 0), emission_power=1.0)]
    objs = [make_object(pkg, pkg.RT_SPHERE, 0, position=(0, 0, 0), radius=1.0), make_object(pkg, pkg.RT_SPHERE, 1, position=(0, 0, 0), radius=1.0)]
    sb = SceneBuilder(pkg, objs, mats, diffuse=1, max_ray_bounces=1)
    cam = pkg.make_camera((0, 0, 3), (0, 0, -1), 32, 32)
    r = orc.render(sb.desc, cam, want_linear=True)
    centre = r["linear"][16, 16]
    assert centre[0] == 1.0 and centre[1] == 0.0           # object 0's emission, not object 1's
    # camera inside a big sphere: every primary ray misses it
    objs = [make_object(pkg, pkg.RT_SPHERE, 0, position=(0, 0, 3), radius=10.0)]
    sb = SceneBuilder(pkg, objs, mats, diffuse=1, max_ray_bounces=1, ambient_color=(0.25, 0.5, 0.75))
    r = orc.render(sb.desc, cam, want_linear=True)
    assert np.allclose(r["linear"][..., :3], [0.25, 0.5, 0.75])


def test_depth_limit_returns_carried_light_and_shadow_halves(pkg, orc):
    """ray_tracing.rs:42-54,168-170,194: light_color is returned as is at the depth limit; an occluded light halves l_acc."""
    m = make_material(pkg, ambience=1.0, diffuse=0.0, specular=0.0, shininess=1.0, albedo=(0.5, 0.5, 0.5), roughness=1.0)
    plane = make_object(pkg, pkg.RT_PLANE, 0, normal=(0, 1, 0), point=(0, 0, 0))
    light = make_light(pkg, pkg.RT_DIRECTIONAL, (1, 1, 1), (0, -1, 0), 1.0)
    cam = pkg.make_camera((0, 2, 0), (0, -1, 0.001), 8, 8)
    sb = SceneBuilder(pkg, [plane], [m], [light], max_ray_bounces=1)
    r = orc.render(sb.desc, cam, want_linear=True)
    # p = (ambience * albedo / 1^2 * 1 * 1) ^ 0.4166 ; result = p*1 + p*p*(1-1) = p (reflection child at the depth limit returns p)
    p = 0.5 ** 0.4166
    assert np.allclose(r["linear"][..., :3], p, atol=1e-6)
    # an occluder above the plane (between it and the light): every plane hit is shadowed -> (0.5 * 0.5) ^ 0.4166
    blocker = make_object(pkg, pkg.RT_PLANE, 0, normal=(0, 1, 0), point=(0, 5, 0))
    sb = SceneBuilder(pkg, [plane, blocker], [m], [light], max_ray_bounces=1, shadow_casting=1)
    r = orc.render(sb.desc, cam, want_linear=True)
    assert np.allclose(r["linear"][..., :3], 0.25 ** 0.4166, atol=1e-6)


def test_sdf_leaf_values_and_union(pkg, orc):
    """sphere.rs:72-82 (|p_obj| - radius in already scaled object space), plane.rs:31-37, union.rs:26-42, ray_marching.rs:74-83."""
    mats = [make_material(pkg, albedo=(1, 0, 0)), make_material(pkg, albedo=(0, 0, 1))]
    objs = [pkg.rt_object(), make_object(pkg, pkg.RT_PLANE, 0, normal=(0, 1, 0), point=(0, 0, 0)),
            make_object(pkg, pkg.RT_SPHERE, 1, position=(0, 3, 0), radius=1.0)]
    objs[0] = make_object(pkg, pkg.RT_UNION, 0, first=1, second=2)
    out = (C.c_float * 11)()
    o = (C.c_float * 3)(0, 10, 0)
    d = (C.c_float * 3)(0, -1, 0)
    sb = SceneBuilder(pkg, objs, mats, sdfs=[1, 2], ray_marching=1)
    assert orc.load().orc_sdfs(C.addressof(sb.desc), C.addressof(o), C.addressof(d), 2.0, C.addressof(out)) == 0
    assert out[0] == pytest.approx(4.0) and int(out[10]) == 2        # sphere: |(0,8,0)-(0,3,0)| - 1 ; plane would be 8
    sb = SceneBuilder(pkg, objs, mats, sdfs=[0], ray_marching=1)
    orc.load().orc_sdfs(C.addressof(sb.desc), C.addressof(o), C.addressof(d), 2.0, C.addressof(out))
    # smooth union, k = 0.7: h = clamp(0.5 + 0.5 (d2 - d1)/k) with d1 = 8 (plane), d2 = 4 (sphere) -> h = 0 -> d = d2
    assert out[0] == pytest.approx(4.0) and int(out[10]) == 2         # reported index is Union.second
    assert np.allclose(out[1:4], [0, 0, 1])                           # albedo mix at i = 0 -> second's albedo
    # sphere of radius 2: the reference subtracts the radius in object space that is already divided by it
    objs2 = [make_object(pkg, pkg.RT_SPHERE, 0, position=(0, 0, 0), radius=2.0)]
    sb = SceneBuilder(pkg, objs2, mats, sdfs=[0], ray_marching=1)
    orc.load().orc_sdfs(C.addressof(sb.desc), C.addressof(o), C.addressof(d), 0.0, C.addressof(out))
    assert out[0] == pytest.approx(10.0 / 2.0 - 2.0)


def test_oracle_rejects_out_of_range_indices(pkg, orc):
    """ray_tracing.rs:92,98: the reference panics on a bad material / texture index; the oracle refuses the scene."""
    mats = [make_material(pkg)]
    cam = pkg.make_camera((0, 0, 3), (0, 0, -1), 8, 8)
    sb = SceneBuilder(pkg, [make_object(pkg, pkg.RT_SPHERE, 3, position=(0, 0, 0), radius=1.0)], mats)
    with pytest.raises(ValueError):
        orc.render(sb.desc, cam)
    sb = SceneBuilder(pkg, [make_object(pkg, pkg.RT_SPHERE, 0, position=(0, 0, 0), radius=1.0)], [make_material(pkg, texture=2)], n_textures=4)
    with pytest.raises(ValueError):
        orc.render(sb.desc, cam)


def test_reference_chunk_loop_and_its_short_last_chunk(pkg, orc):
    """renderer.rs:104-143 as written.  With width*height a multiple of the chunk count the chunk loop is the plain frame.
    Otherwise `chunks_mut` yields a short last chunk whose pixel offset is computed from its OWN length (:115-117): its s
    pixels -- the last s of the frame -- show pixels [n*s, n*s + s) of the frame instead (n = its chunk index), bytes and
    accumulator alike; every other pixel is unaffected.  Frames with fewer pixels than chunks panic (chunks_mut(0))."""
    s = pkg.Scene("scene3")
    tex = s.textures()
    # 40 x 25 = 1000 pixels, 200 chunks of 5: no short chunk
    cam = s.camera(40, 25)
    ref = orc.render(s.desc, cam, tex)
    quirk = orc.render_reference_chunks(s.desc, cam, tex)
    assert np.array_equal(ref["rgba"], quirk["rgba"]) and np.array_equal(ref["accum"].view(np.uint32), quirk["accum"].view(np.uint32))
    # (the offset n * s never exceeds the short chunk's true start n * chunk, so the reference's slices stay in bounds)
    # 10 chunks over 7 x 15 = 105 pixels: chunk = 10 pixels, ten full chunks, then s = 5, n = 10 -> offset 50
    cam = s.camera(7, 15)
    ref = orc.render(s.desc, cam, tex)
    quirk = orc.render_reference_chunks(s.desc, cam, tex, num_chunks=10)
    a, b = ref["rgba"].reshape(-1, 4), quirk["rgba"].reshape(-1, 4)
    fa, fb = ref["accum"].reshape(-1, 4).view(np.uint32), quirk["accum"].reshape(-1, 4).view(np.uint32)
    assert np.array_equal(a[:100], b[:100]) and np.array_equal(fa[:100], fb[:100])          # ten full chunks
    assert np.array_equal(b[100:105], a[50:55]) and np.array_equal(fb[100:105], fa[50:55])  # the short chunk shows pixels 50..54
    assert not np.array_equal(a[100:105], b[100:105]), "the test frame must make the difference visible"
    # accumulation: the short chunk adds this frame's value of pixel 50 + i to the accumulator entry of pixel 50 + i and stores
    # the sum at 100 + i, so over two frames the tail keeps following pixels 50..54
    d = pkg.Scene("scene2")
    cam = d.camera(7, 15)
    acc_ref = np.zeros((15, 7, 4), dtype=np.float32)
    acc_q = np.zeros((15, 7, 4), dtype=np.float32)
    for f in (1, 2):
        ref = orc.render(d.desc, cam, (), frame_index=f, accumulate=True, seed=5, accum=acc_ref)
        quirk = orc.render_reference_chunks(d.desc, cam, (), frame_index=f, accumulate=True, seed=5, num_chunks=10, accum=acc_q)
        fa, fb = acc_ref.reshape(-1, 4).view(np.uint32), acc_q.reshape(-1, 4).view(np.uint32)
        assert np.array_equal(fa[:100], fb[:100]) and np.array_equal(fb[100:105], fa[50:55])
        assert np.array_equal(quirk["rgba"].reshape(-1, 4)[100:105], ref["rgba"].reshape(-1, 4)[50:55])
    # fewer pixels than chunks: the reference panics, the restatement says so; orc_render renders the frame
    cam = s.camera(13, 15)   # 195 pixels < 200 chunks
    with pytest.raises(OverflowError):
        orc.render_reference_chunks(s.desc, cam, tex)
    assert orc.render(s.desc, cam, tex)["rgba"].shape == (15, 13, 4)
