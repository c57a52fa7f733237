"""Helpers shared by the test modules: building rt_object records from Python and parity metrics."""
import ctypes as C

import numpy as np


def identity16():
    return (C.c_float * 16)(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1)


def make_object(pkg, kind, material_index=0, **kw):
    """An rt_object with the matrices its update() would build (through the C++ host mirror)."""
    o = pkg.rt_object()
    o.kind = kind
    o.material_index = material_index
    o.transform = identity16()
    o.inv_transform = identity16()
    for k, v in kw.items():
        if isinstance(v, (list, tuple)):
            setattr(o, k, (C.c_float * len(v))(*v))
        else:
            setattr(o, k, v)
    pkg.host().rth_object_update(C.byref(o))
    return o


def make_material(pkg, **kw):
    m = pkg.rt_material()
    m.ambience, m.diffuse, m.specular, m.shininess = 0.2, 0.7, 0.5, 5.0
    m.texture = -1
    m.kind = pkg.RT_REFLECTIVE
    m.roughness = 1.0
    for k, v in kw.items():
        if isinstance(v, (list, tuple)):
            setattr(m, k, (C.c_float * len(v))(*v))
        else:
            setattr(m, k, v)
    return m


def make_light(pkg, kind, albedo, vec, intensity, radius=0.0):
    l = pkg.rt_light()
    l.kind = kind
    l.albedo = (C.c_float * 3)(*albedo)
    l.vec = (C.c_float * 3)(*vec)
    l.intensity = intensity
    l.radius = radius
    return l


class SceneBuilder:
    """Keeps the ctypes arrays of a hand-built rt_scene_desc alive."""

    def __init__(self, pkg, objects, materials, lights=(), sdfs=(), n_textures=0, **flags):
        self.objects = (pkg.rt_object * max(len(objects), 1))(*objects)
        self.materials = (pkg.rt_material * max(len(materials), 1))(*materials)
        self.lights = (pkg.rt_light * max(len(lights), 1))(*lights)
        self.sdfs = (C.c_uint32 * max(len(sdfs), 1))(*sdfs)
        d = pkg.rt_scene_desc()
        d.objects = C.cast(self.objects, C.POINTER(pkg.rt_object))
        d.materials = C.cast(self.materials, C.POINTER(pkg.rt_material))
        d.lights = C.cast(self.lights, C.POINTER(pkg.rt_light))
        d.sdfs = C.cast(self.sdfs, C.POINTER(C.c_uint32))
        d.n_objects, d.n_materials, d.n_lights, d.n_sdfs = len(objects), len(materials), len(lights), len(sdfs)
        d.n_textures = n_textures
        d.max_ray_bounces = 4
        for k, v in flags.items():
            if k == "ambient_color":
                d.ambient_color = (C.c_float * 3)(*v)
            else:
                setattr(d, k, v)
        self.desc = d


def intersect(orc, obj, origin, direction):
    out = (C.c_float * 9)()
    o = (C.c_float * 3)(*origin)
    d = (C.c_float * 3)(*direction)
    ok = orc.load().orc_intersect(C.addressof(obj), C.addressof(o), C.addressof(d), C.addressof(out))
    return (bool(ok), np.array(out[:], dtype=np.float32))


# The parity bar of the north star: at least 99.9 % of 8-bit channel values within 1 LSB and linear-RGB max
# absolute error <= 1e-3 against the CPU reference algorithm in deterministic mode.
LSB_FRACTION = 0.999
LINEAR_TOL = 1e-3


def parity(gpu_rgba, ref_rgba, gpu_lin, ref_lin):
    d = np.abs(gpu_rgba.astype(np.int32) - ref_rgba.astype(np.int32))
    within = float((d <= 1).mean())
    both_nan = np.isnan(gpu_lin) & np.isnan(ref_lin)
    lin = np.abs(np.where(both_nan, 0.0, gpu_lin - ref_lin))
    lin_max = float(np.max(lin)) if lin.size else 0.0
    words = int((gpu_lin.view(np.uint32) != ref_lin.view(np.uint32)).sum())
    return within, lin_max, words


def assert_parity(gpu_rgba, ref_rgba, gpu_lin, ref_lin, exact=True, what=""):
    within, lin_max, words = parity(gpu_rgba, ref_rgba, gpu_lin, ref_lin)
    assert within >= LSB_FRACTION, f"{what}: only {within * 100:.4f}% of bytes within 1 LSB"
    assert lin_max <= LINEAR_TOL or np.isnan(lin_max) is False and lin_max <= LINEAR_TOL, f"{what}: linear max abs error {lin_max:.3e}"
    if exact:
        # stronger than the bar: the kernels keep the reference's operation order, so the f32 frame is identical
        assert words == 0, f"{what}: {words} f32 words differ from the oracle"
        assert np.array_equal(gpu_rgba, ref_rgba), f"{what}: RGBA8 bytes differ"
