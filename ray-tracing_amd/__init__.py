"""ctypes bindings of the MI355X rendering backend for mariusdanciu/ray-tracing.

The directory name carries a hyphen, so the package is loaded under the importable name
``ray_tracing_amd`` by ``__graft_entry__.load_package()`` (tests and bench.py do the same).

Two shared libraries, both built in-tree by ``csrc/Makefile``:

* ``librt_mi355x.so`` — HIP kernels + the C ABI declared in ``include/rt_mi355x.h``.
* ``librt_host.so``   — C++ mirror of the reference's ``Scene`` / ``Camera`` / ``Renderer`` API
  (``host/rt_host.h``) with a small C view used here to build the reference's demo scenes.

Nothing in this package falls back to a CPU path: if the HIP library is missing, or no GPU is
present, the calls fail loudly.  The CPU oracle under ``oracle/`` is test infrastructure and is
never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)

# --------------------------------------------------------------------------------------------
# plain-data mirrors of include/rt_mi355x.h
# --------------------------------------------------------------------------------------------
RT_SPHERE, RT_TRIANGLE, RT_CUBOID, RT_PLANE, RT_CYLINDER, RT_CONE, RT_UNION, RT_SUBSTRACTION = range(8)
RT_REFLECTIVE, RT_REFRACTIVE = 0, 1
RT_DIRECTIONAL, RT_POSITIONAL, RT_SPHERICAL_POSITIONAL = 0, 1, 2
RT_MATH_POWF, RT_MATH_ATANF, RT_MATH_SQRT = 0, 1, 2
RT_RENDER_ONE_LANE_PER_PIXEL, RT_RENDER_SKIP_ACCUM_STORE, RT_RENDER_NO_SKY_TILES, RT_RENDER_MARCH_ALL_ENTRIES = 1, 2, 8, 16
RT_RENDER_NO_FIRST_HIT_REUSE = 32

COUNTER_NAMES = [
    "rays_primary", "rays_reflection", "rays_refraction", "rays_shadow", "rays_march", "raygen",
    "test_sphere", "test_triangle", "test_cuboid", "test_plane", "test_cylinder", "test_cone",
    "hit_sphere", "hit_triangle", "hit_cuboid", "hit_plane", "hit_cylinder", "hit_cone",
    "light_call", "light_shade", "reflection_ray", "diffuse_ray", "refraction", "combine_refl", "combine_refr",
    "tex_fetch", "triplanar", "sdf_sphere", "sdf_cuboid", "sdf_cylinder", "sdf_plane", "sdf_union", "sdf_substraction",
    "march_step", "sdfs_call", "occlusion_tap", "march_shade", "pixel_pack", "bvh_node",
]
RT_CNT_COUNT = len(COUNTER_NAMES)

# Algorithmic FLOPs per counted event: the reference's own arithmetic, 1 per add/mul/div/sqrt/min/max/
# transcendental, minimal 3x3(+t) form for matrix-vector products (direction 15, point 18).
EVENT_FLOPS = {
    # "raygen" counts the pixels whose primary direction a frame reads from Camera.ray_directions (src/renderer.rs:49).
    # The 65 FLOPs of computing one direction (src/camera.rs:137-148) are spent once per camera change, by the reference
    # and by rt_set_camera alike, so they are not part of a frame's algorithmic work.
    "raygen": 0,
    "test_sphere": 54, "hit_sphere": 51,
    "test_plane": 14, "hit_plane": 16,
    "test_cuboid": 58, "hit_cuboid": 61,
    "test_cylinder": 82, "hit_cylinder": 24,
    "test_cone": 77, "hit_cone": 27,
    "test_triangle": 92, "hit_triangle": 47,
    "light_call": 3, "light_shade": 90,
    "reflection_ray": 31, "diffuse_ray": 19, "refraction": 50,
    "combine_refl": 16, "combine_refr": 13,
    "tex_fetch": 14, "triplanar": 62,
    "sdf_sphere": 47, "sdf_cuboid": 61, "sdf_cylinder": 57, "sdf_plane": 14, "sdf_union": 22, "sdf_substraction": 2,
    "march_step": 3, "occlusion_tap": 6, "march_shade": 9, "pixel_pack": 15,
    "bvh_node": 21,  # slab test of one hierarchy node: executed work of the large-scene path, not reference arithmetic
}


def algorithmic_flops(counters: dict) -> int:
    """Sum of event counts times the per-event FLOP constants."""
    return int(sum(int(counters.get(k, 0)) * v for k, v in EVENT_FLOPS.items()))


class rt_object(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("material_index", C.c_int32),
        ("transform", C.c_float * 16), ("inv_transform", C.c_float * 16),
        ("position", C.c_float * 3), ("rotation_axis", C.c_float * 3),
        ("radius", C.c_float), ("height", C.c_float), ("dimension", C.c_float * 3),
        ("normal", C.c_float * 3), ("point", C.c_float * 3), ("max_dist", C.c_float * 2), ("has_max_dist", C.c_int32),
        ("v1", C.c_float * 3), ("v2", C.c_float * 3), ("v3", C.c_float * 3),
        ("first", C.c_int32), ("second", C.c_int32),
    ]


class rt_material(C.Structure):
    _fields_ = [
        ("ambience", C.c_float), ("diffuse", C.c_float), ("specular", C.c_float), ("shininess", C.c_float),
        ("albedo", C.c_float * 3), ("texture", C.c_int32), ("kind", C.c_int32), ("roughness", C.c_float),
        ("transparency", C.c_float), ("refraction_index", C.c_float), ("reflectivity", C.c_float),
        ("emission_power", C.c_float),
    ]


class rt_light(C.Structure):
    _fields_ = [("kind", C.c_int32), ("albedo", C.c_float * 3), ("vec", C.c_float * 3), ("radius", C.c_float),
                ("intensity", C.c_float)]


class rt_scene_desc(C.Structure):
    _fields_ = [
        ("objects", C.POINTER(rt_object)), ("materials", C.POINTER(rt_material)), ("lights", C.POINTER(rt_light)),
        ("sdfs", C.POINTER(C.c_uint32)),
        ("n_objects", C.c_uint32), ("n_materials", C.c_uint32), ("n_lights", C.c_uint32), ("n_sdfs", C.c_uint32),
        ("n_textures", C.c_uint32), ("ambient_color", C.c_float * 3),
        ("max_ray_bounces", C.c_uint8), ("shadow_casting", C.c_uint8), ("ray_marching", C.c_uint8), ("diffuse", C.c_uint8),
        ("enable_accumulation", C.c_uint8), ("reserved", C.c_uint8 * 3),
    ]


class rt_camera_desc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("position", C.c_float * 3),
                ("inverse_perspective", C.c_float * 16), ("inverse_view", C.c_float * 16)]


class rt_stats(C.Structure):
    _fields_ = [("rays_total", C.c_uint64), ("counters", C.c_uint64 * RT_CNT_COUNT), ("last_kernel_ms", C.c_double),
                ("last_copy_ms", C.c_double), ("frames_rendered", C.c_uint64), ("rays_sky_tiles", C.c_uint64),
                ("rays_reused_camera", C.c_uint64)]


class rt_hierarchy_info(C.Structure):
    _fields_ = [("stackless_nodes", C.c_uint32), ("stackless_linear", C.c_uint32), ("two_child_nodes", C.c_uint32),
                ("two_child_linear", C.c_uint32), ("regrouping_kernel", C.c_uint32), ("sky_workgroups", C.c_uint32),
                ("classified_workgroups", C.c_uint32), ("reserved", C.c_uint32 * 1)]


class rt_render_params(C.Structure):
    _fields_ = [("frame_index", C.c_uint32), ("row_begin", C.c_uint32), ("row_end", C.c_uint32), ("accumulate", C.c_int32),
                ("collect_events", C.c_int32), ("no_timing", C.c_int32), ("seed", C.c_uint64),
                ("tile_row_stride", C.c_uint32), ("flags", C.c_uint32)]


# every symbol include/rt_mi355x.h declares
ABI_SYMBOLS = [
    "rt_create", "rt_destroy", "rt_set_scene", "rt_set_texture", "rt_set_camera", "rt_render", "rt_render_device",
    "rt_resolve_device", "rt_reset_accum", "rt_read_accum", "rt_get_stats", "rt_context_stream", "rt_device_rgba8", "rt_device_accum", "rt_last_error",
    "rt_abi_version", "rt_alloc_host_frame", "rt_free_host_frame", "rt_update_objects", "rt_read_ray_directions",
    "rt_render_device_noalloc", "rt_get_hierarchy_info", "rt_eval_math", "rt_render_samples_device",
]

# RT_MI355X_LIB overrides the library file (used to A/B kernel builds); it is still a HIP build of this backend
LIB_PATH = os.environ.get("RT_MI355X_LIB") or os.path.join(_HERE, "librt_mi355x.so")
HOST_LIB_PATH = os.path.join(_HERE, "librt_host.so")

_lib = None
_host = None


class BackendError(RuntimeError):
    """A C-ABI call returned a negative status."""


def lib() -> C.CDLL:
    """Loads librt_mi355x.so (raises if it was not built: there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} is missing: build it with `make -C ray-tracing_amd/csrc` "
                                    "(or __graft_entry__.build()); this backend has no CPU fallback")
        l = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        vp = C.c_void_p
        l.rt_create.argtypes = [C.c_int32, C.POINTER(vp)]; l.rt_create.restype = C.c_int32
        l.rt_destroy.argtypes = [vp]; l.rt_destroy.restype = None
        l.rt_set_scene.argtypes = [vp, C.POINTER(rt_scene_desc)]; l.rt_set_scene.restype = C.c_int32
        l.rt_set_texture.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]; l.rt_set_texture.restype = C.c_int32
        l.rt_set_camera.argtypes = [vp, C.POINTER(rt_camera_desc)]; l.rt_set_camera.restype = C.c_int32
        l.rt_render.argtypes = [vp, C.c_uint32, C.c_int32, C.c_int32, C.c_uint64, C.c_void_p, C.c_size_t]; l.rt_render.restype = C.c_int32
        l.rt_render_device.argtypes = [vp, C.POINTER(rt_render_params), vp, vp, vp]; l.rt_render_device.restype = C.c_int32
        l.rt_render_device_noalloc.argtypes = [vp, C.POINTER(rt_render_params), vp, vp, vp]; l.rt_render_device_noalloc.restype = C.c_int32
        l.rt_resolve_device.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp]; l.rt_resolve_device.restype = C.c_int32
        l.rt_reset_accum.argtypes = [vp]; l.rt_reset_accum.restype = C.c_int32
        l.rt_read_accum.argtypes = [vp, C.c_void_p, C.c_size_t]; l.rt_read_accum.restype = C.c_int32
        l.rt_update_objects.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(rt_object)]; l.rt_update_objects.restype = C.c_int32
        l.rt_read_ray_directions.argtypes = [vp, C.c_void_p, C.c_size_t]; l.rt_read_ray_directions.restype = C.c_int32
        l.rt_alloc_host_frame.argtypes = [vp, C.c_size_t, C.POINTER(C.c_void_p)]; l.rt_alloc_host_frame.restype = C.c_int32
        l.rt_free_host_frame.argtypes = [vp, C.c_void_p]; l.rt_free_host_frame.restype = C.c_int32
        l.rt_get_stats.argtypes = [vp, C.POINTER(rt_stats)]; l.rt_get_stats.restype = C.c_int32
        l.rt_get_hierarchy_info.argtypes = [vp, C.POINTER(rt_hierarchy_info)]; l.rt_get_hierarchy_info.restype = C.c_int32
        l.rt_render_samples_device.argtypes = [vp, C.POINTER(rt_render_params), C.c_uint32, vp, vp, vp]; l.rt_render_samples_device.restype = C.c_int32
        l.rt_eval_math.argtypes = [vp, C.c_int32, C.c_size_t, vp, vp, vp]; l.rt_eval_math.restype = C.c_int32
        l.rt_context_stream.argtypes = [vp]; l.rt_context_stream.restype = vp
        l.rt_device_rgba8.argtypes = [vp]; l.rt_device_rgba8.restype = vp
        l.rt_device_accum.argtypes = [vp]; l.rt_device_accum.restype = vp
        l.rt_last_error.argtypes = [vp]; l.rt_last_error.restype = C.c_char_p
        l.rt_abi_version.argtypes = []; l.rt_abi_version.restype = C.c_int32
        _lib = l
    return _lib


def host() -> C.CDLL:
    """Loads librt_host.so, the C++ mirror of the reference's Scene / Camera / Renderer API."""
    global _host
    if _host is None:
        lib()
        if not os.path.exists(HOST_LIB_PATH):
            raise FileNotFoundError(f"{HOST_LIB_PATH} is missing: build it with `make -C ray-tracing_amd/csrc`")
        h = C.CDLL(HOST_LIB_PATH)
        vp = C.c_void_p
        h.rth_scene_build.argtypes = [C.c_char_p]; h.rth_scene_build.restype = vp
        h.rth_scene_free.argtypes = [vp]; h.rth_scene_free.restype = None
        h.rth_scene_desc.argtypes = [vp]; h.rth_scene_desc.restype = C.POINTER(rt_scene_desc)
        h.rth_scene_camera.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(rt_camera_desc)]; h.rth_scene_camera.restype = None
        h.rth_scene_renderer_accumulation.argtypes = [vp]; h.rth_scene_renderer_accumulation.restype = C.c_int32
        h.rth_camera.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.POINTER(rt_camera_desc)]
        h.rth_camera.restype = None
        h.rth_object_update.argtypes = [C.POINTER(rt_object)]; h.rth_object_update.restype = None
        h.rth_render_frames.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_uint32, C.POINTER(C.c_uint32),
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_char_p, C.c_size_t]
        h.rth_render_frames.restype = C.c_int32
        h.rth_render_animation.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, C.POINTER(C.c_uint32),
                                           C.POINTER(C.c_uint32), C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.POINTER(rt_object),
                                           C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]
        h.rth_render_animation.restype = C.c_int32
        h.rth_load_image.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_uint8)), C.c_char_p, C.c_size_t]
        h.rth_load_image.restype = C.c_int32
        h.rth_free_image.argtypes = [C.POINTER(C.c_uint8)]; h.rth_free_image.restype = None
        _host = h
    return _host


# --------------------------------------------------------------------------------------------
# scenes (built by the C++ host mirror) and textures
# --------------------------------------------------------------------------------------------
TEXTURE_FILES = ["chess.png", "wood.png", "stone3.jpg", "earth_clouds.jpg"]  # slots 0..3 of scenes 1, 3, 4
TEXTURE_DIR = os.path.join(REPO_ROOT, "tests", "golden", "textures")

_tex_cache: dict = {}


def load_textures(directory: str = TEXTURE_DIR) -> list:
    """Decodes the four images the reference's binaries load into tightly packed RGB8 arrays (h, w, 3).

    The reference decodes with the `image` crate (src/utils/image.rs:13-29); here PIL does it on the
    caller's side of the C ABI.  PNG decoding is lossless; JPEG decoders may differ by an LSB or two,
    which is why decoding stays outside the boundary.
    """
    if directory not in _tex_cache:
        from PIL import Image
        out = []
        for name in TEXTURE_FILES:
            img = Image.open(os.path.join(directory, name)).convert("RGB")
            out.append(np.ascontiguousarray(np.asarray(img, dtype=np.uint8)))
        _tex_cache[directory] = out
    return _tex_cache[directory]


def load_image(path: str) -> np.ndarray:
    """The host mirror's load_image (ImageUtils::load_image, src/utils/image.rs:13-29): PNG or binary PPM file ->
    (h, w, 3) uint8.  Raises ValueError with the loader's message otherwise (JPEG files are refused: see load_textures)."""
    w, h, ptr = C.c_uint32(), C.c_uint32(), C.POINTER(C.c_uint8)()
    err = C.create_string_buffer(512)
    if host().rth_load_image(os.fsencode(path), C.byref(w), C.byref(h), C.byref(ptr), err, len(err)) != 0:
        raise ValueError(err.value.decode(errors="replace"))
    try:
        return np.ctypeslib.as_array(ptr, shape=(h.value, w.value, 3)).copy()
    finally:
        host().rth_free_image(ptr)


class Scene:
    """A scene built by the host mirror: owns the flat arrays `desc` points into."""

    def __init__(self, name: str):
        self._h = host()
        self._p = self._h.rth_scene_build(name.encode())
        if not self._p:
            raise ValueError(f"unknown scene {name!r}")
        self.name = name
        # private copy of the descriptor so that flags can be overridden per test
        self.desc = rt_scene_desc()
        C.memmove(C.byref(self.desc), self._h.rth_scene_desc(self._p), C.sizeof(rt_scene_desc))
        self.renderer_accumulation = bool(self._h.rth_scene_renderer_accumulation(self._p))

    def camera(self, width: int, height: int) -> rt_camera_desc:
        cam = rt_camera_desc()
        self._h.rth_scene_camera(self._p, width, height, C.byref(cam))
        return cam

    def textures(self) -> list:
        return load_textures()[: self.desc.n_textures]

    def handle(self):
        return self._p

    def __del__(self):
        try:
            if self._p:
                self._h.rth_scene_free(self._p)
                self._p = None
        except Exception:
            pass


def make_camera(position: Sequence[float], look_at: Sequence[float], width: int, height: int) -> rt_camera_desc:
    """Camera::new_with_pos(position, look_at) followed by a Resize event (src/camera.rs:70-95)."""
    cam = rt_camera_desc()
    p = (C.c_float * 3)(*position)
    d = (C.c_float * 3)(*look_at)
    host().rth_camera(p, d, width, height, C.byref(cam))
    return cam


# --------------------------------------------------------------------------------------------
# context wrapper
# --------------------------------------------------------------------------------------------
class Context:
    """One rt_ctx: one GPU, one thread at a time."""

    def __init__(self, device_id: int = 0):
        self._l = lib()
        self._ctx = C.c_void_p()
        st = self._l.rt_create(device_id, C.byref(self._ctx))
        if st != 0:
            self._ctx = C.c_void_p()
            raise BackendError(f"rt_create(device {device_id}) failed with status {st}: no usable MI355X device?")
        self.width = 0
        self.height = 0
        self._keep = []

    def _check(self, st: int, what: str):
        if st != 0:
            raise BackendError(f"{what} failed ({st}): {self._l.rt_last_error(self._ctx).decode()}")

    def set_scene(self, desc: rt_scene_desc):
        self._check(self._l.rt_set_scene(self._ctx, C.byref(desc)), "rt_set_scene")

    def set_texture(self, index: int, rgb: np.ndarray):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        assert rgb.ndim == 3 and rgb.shape[2] == 3
        self._check(self._l.rt_set_texture(self._ctx, index, rgb.shape[1], rgb.shape[0], rgb.ctypes.data), "rt_set_texture")

    def set_camera(self, cam: rt_camera_desc):
        self._check(self._l.rt_set_camera(self._ctx, C.byref(cam)), "rt_set_camera")
        self.width, self.height = cam.width, cam.height

    def load(self, scene: Scene, width: int, height: int):
        """Uploads a host-mirror scene, its textures and its camera."""
        for i, t in enumerate(scene.textures()):
            self.set_texture(i, t)
        self.set_scene(scene.desc)
        self.set_camera(scene.camera(width, height))

    def render(self, frame_index: int = 1, accumulate: bool = False, reset_accum: bool = False, seed: int = 0x5EED,
               out: Optional[np.ndarray] = None) -> np.ndarray:
        """rt_render: one frame into a host array (h, w, 4) uint8 (`out` is reused when given, like the reference's `img`)."""
        if out is None:
            out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        if out.dtype != np.uint8 or out.shape != (self.height, self.width, 4) or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous (height, width, 4) uint8 array")
        self._check(self._l.rt_render(self._ctx, frame_index, int(accumulate), int(reset_accum), seed, out.ctypes.data, out.nbytes), "rt_render")
        return out

    def update_objects(self, first: int, objects) -> None:
        """rt_update_objects: re-upload the given rt_object records at indices first, first + 1, ..."""
        arr = (rt_object * len(objects))(*objects)
        self._check(self._l.rt_update_objects(self._ctx, first, len(objects), arr), "rt_update_objects")

    def update_objects_array(self, first: int, array) -> None:
        """rt_update_objects with a ready ctypes array of rt_object (no per-call conversion: animation loops)."""
        self._check(self._l.rt_update_objects(self._ctx, first, len(array), array), "rt_update_objects")

    def read_ray_directions(self) -> np.ndarray:
        """rt_read_ray_directions: Camera.ray_directions as an (h, w, 3) float32 array."""
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        self._check(self._l.rt_read_ray_directions(self._ctx, out.ctypes.data, self.width * self.height), "rt_read_ray_directions")
        return out

    def alloc_host_frame(self) -> np.ndarray:
        """rt_alloc_host_frame: a page-locked (height, width, 4) uint8 array for render(out=...); valid until free_host_frame / close."""
        p = C.c_void_p()
        n = self.width * self.height * 4
        self._check(self._l.rt_alloc_host_frame(self._ctx, n, C.byref(p)), "rt_alloc_host_frame")
        return np.ctypeslib.as_array((C.c_uint8 * n).from_address(p.value)).reshape(self.height, self.width, 4)

    def free_host_frame(self, frame: np.ndarray) -> None:
        self._check(self._l.rt_free_host_frame(self._ctx, frame.ctypes.data), "rt_free_host_frame")

    def render_device(self, frame_index: int = 1, accumulate: bool = False, seed: int = 0x5EED, rows: Optional[tuple] = None,
                      collect_events: bool = False, d_rgba8: int = 0, d_accum: int = 0, stream: Optional[int] = None,
                      no_timing: bool = False, noalloc: bool = False, tile_row_stride: int = 0, flags: int = 0):
        """rt_render_device: asynchronous launch into device buffers (0 = the context's own).

        `stream` is a hipStream_t handle used as is (0 is HIP's default stream); None selects the context's own stream.
        `noalloc` calls rt_render_device_noalloc instead (same frame, counter rows in the library's fixed device array)."""
        p = rt_render_params()
        p.frame_index = frame_index
        p.accumulate = int(accumulate)
        p.collect_events = int(collect_events)
        p.no_timing = int(no_timing)
        p.seed = seed
        if rows is not None:
            p.row_begin, p.row_end = rows
        p.tile_row_stride = tile_row_stride
        p.flags = flags
        if stream is None:
            stream = int(self._l.rt_context_stream(self._ctx) or 0)
        fn = self._l.rt_render_device_noalloc if noalloc else self._l.rt_render_device
        self._check(fn(self._ctx, C.byref(p), C.c_void_p(d_rgba8 or None), C.c_void_p(d_accum or None), C.c_void_p(stream or None)),
                    "rt_render_device_noalloc" if noalloc else "rt_render_device")

    def render_samples_device(self, first_frame_index: int, sample_count: int, seed: int = 0x5EED, rows: Optional[tuple] = None,
                              d_rgba8: int = 0, d_accum: int = 0, stream: Optional[int] = None, tile_row_stride: int = 0, flags: int = 0):
        """rt_render_samples_device: `sample_count` accumulating frames (frame_index = first_frame_index ...) in one call."""
        p = rt_render_params()
        p.frame_index = first_frame_index
        p.accumulate = 1
        p.no_timing = 1
        p.seed = seed
        if rows is not None:
            p.row_begin, p.row_end = rows
        p.tile_row_stride = tile_row_stride
        p.flags = flags
        if stream is None:
            stream = int(self._l.rt_context_stream(self._ctx) or 0)
        self._check(self._l.rt_render_samples_device(self._ctx, C.byref(p), sample_count, C.c_void_p(d_rgba8 or None), C.c_void_p(d_accum or None),
                                                     C.c_void_p(stream or None)), "rt_render_samples_device")

    def resolve_device(self, sample_count: int, rows: Optional[tuple] = None, d_accum: int = 0, d_rgba8: int = 0, stream: Optional[int] = None):
        """rt_resolve_device: accumulated / sample_count -> clamp -> RGBA8 (after an external all-reduce of accumulators)."""
        rb, re = rows if rows is not None else (0, 0)
        if stream is None:
            stream = int(self._l.rt_context_stream(self._ctx) or 0)
        self._check(self._l.rt_resolve_device(self._ctx, sample_count, rb, re, C.c_void_p(d_accum or None), C.c_void_p(d_rgba8 or None),
                                              C.c_void_p(stream or None)), "rt_resolve_device")

    def reset_accum(self):
        self._check(self._l.rt_reset_accum(self._ctx), "rt_reset_accum")

    def read_accum(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._check(self._l.rt_read_accum(self._ctx, out.ctypes.data, self.width * self.height), "rt_read_accum")
        return out

    def stats(self) -> dict:
        s = rt_stats()
        self._check(self._l.rt_get_stats(self._ctx, C.byref(s)), "rt_get_stats")
        d = {name: int(s.counters[i]) for i, name in enumerate(COUNTER_NAMES)}
        d["rays_total"] = int(s.rays_total)
        d["last_kernel_ms"] = float(s.last_kernel_ms)
        d["last_copy_ms"] = float(s.last_copy_ms)
        d["frames_rendered"] = int(s.frames_rendered)
        d["rays_sky_tiles"] = int(s.rays_sky_tiles)
        d["rays_reused_camera"] = int(s.rays_reused_camera)
        return d

    def hierarchy_info(self) -> dict:
        """rt_get_hierarchy_info: node and linear-object counts of the hierarchies built for the current scene."""
        h = rt_hierarchy_info()
        self._check(self._l.rt_get_hierarchy_info(self._ctx, C.byref(h)), "rt_get_hierarchy_info")
        return {k: int(getattr(h, k)) for k, _ in rt_hierarchy_info._fields_ if k != "reserved"}

    def eval_math(self, op: int, x: np.ndarray, y: np.ndarray = None) -> np.ndarray:
        """rt_eval_math: the kernels' powf(x, y) (op RT_MATH_POWF), atanf(x) (op RT_MATH_ATANF) or sqrt(x) (op RT_MATH_SQRT), evaluated on the device."""
        x = np.ascontiguousarray(x, dtype=np.float32).ravel()
        if y is not None:
            y = np.ascontiguousarray(y, dtype=np.float32).ravel()
            if y.size != x.size:
                raise ValueError("x and y must have the same number of elements")
        out = np.empty_like(x)
        self._check(self._l.rt_eval_math(self._ctx, op, x.size, x.ctypes.data, y.ctypes.data if y is not None else None, out.ctypes.data), "rt_eval_math")
        return out

    def device_rgba8(self) -> int:
        return int(self._l.rt_device_rgba8(self._ctx) or 0)

    def device_accum(self) -> int:
        return int(self._l.rt_device_accum(self._ctx) or 0)

    def close(self):
        if self._ctx:
            self._l.rt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


from .sharding import (FrameExchange, equal_row_bands, gather_row_bands, interleaved_tile_rows,  # noqa: E402,F401
                       render_sample_sharded, shard_rows, shard_samples)
