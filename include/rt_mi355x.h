/*
 * rt_mi355x.h — C ABI of librt_mi355x.so, the MI355X (gfx950) rendering backend that
 * replaces the per-pixel hot path of mariusdanciu/ray-tracing.
 *
 * The reference has no FFI of its own: the seam is Renderer::render (src/renderer.rs:81-151),
 * whose rayon block (src/renderer.rs:104-143) is what these entry points replace.  Every
 * struct below is the plain-data mirror of a reference type, every function cites the
 * reference code whose behaviour it takes over.
 *
 * Conventions
 *   - every function returns 0 on success and a negative rt_status on failure; the text of
 *     the last failure of a context is returned by rt_last_error().  Nothing throws or
 *     aborts across this boundary (reference: Result<(), String>, src/renderer.rs:89).
 *   - matrices are 16 floats, column-major, exactly glam's Mat4::to_cols_array().
 *   - a context belongs to one GPU and is driven from one thread at a time
 *     (reference: render is called from the single SDL main thread, src/app.rs:248).
 */
#ifndef RT_MI355X_H
#define RT_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 3

typedef struct rt_ctx rt_ctx;

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARGUMENT = -1, /* null pointer, bad size, index out of range          */
    RT_ERR_UNSUPPORTED = -2,      /* legal in the reference but beyond a stated limit    */
    RT_ERR_DEVICE = -3,           /* a HIP call failed; rt_last_error has the HIP string */
    RT_ERR_STATE = -4             /* call order: scene/camera not set yet                */
} rt_status;

/* Object3D variants in declaration order (src/objects.rs:21-30). */
typedef enum rt_object_kind {
    RT_SPHERE = 0,
    RT_TRIANGLE = 1,
    RT_CUBOID = 2,
    RT_PLANE = 3,
    RT_CYLINDER = 4,
    RT_CONE = 5,
    RT_UNION = 6,
    RT_SUBSTRACTION = 7
} rt_object_kind;

/*
 * One Object3D.  Only the fields of the object's own kind are read.
 *   Sphere       src/utils/sphere.rs:15-23     transform, inv_transform, radius
 *   Triangle     src/utils/triangle.rs:12-17   v1, v2, v3
 *   Cuboid       src/utils/cuboid.rs:12-21     transform, inv_transform, dimension (b_min/b_max = -/+dimension)
 *   Plane        src/utils/plane.rs:10-15      normal, point, max_dist (+has_max_dist for Option)
 *   Cylinder     src/utils/cylinder.rs:14-22   transform, inv_transform, radius, height
 *   Cone         src/utils/cone.rs:14-22       transform, inv_transform
 *   Union        src/utils/union.rs:12-15      first, second
 *   Substraction src/utils/substraction.rs:12-15 first, second
 * transform/inv_transform must be the identity for Triangle, Plane, Union and Substraction
 * (Object3D::transform, src/objects.rs:71-82).  material_index is 0 for Union and
 * Substraction (src/objects.rs:54-55).  position/rotation_axis are informative: they are
 * already folded into the matrices by the object's update() on the caller's side.
 */
typedef struct rt_object {
    int32_t kind;
    int32_t material_index;
    float transform[16];
    float inv_transform[16];
    float position[3];
    float rotation_axis[3];
    float radius;
    float height;
    float dimension[3];
    float normal[3];
    float point[3];
    float max_dist[2];
    int32_t has_max_dist;
    float v1[3];
    float v2[3];
    float v3[3];
    int32_t first;
    int32_t second;
} rt_object;

/* MaterialType (src/objects.rs:86-95). */
typedef enum rt_material_kind { RT_REFLECTIVE = 0, RT_REFRACTIVE = 1 } rt_material_kind;

/* Material (src/objects.rs:151-160); texture = -1 encodes Option::None. */
typedef struct rt_material {
    float ambience;
    float diffuse;
    float specular;
    float shininess;
    float albedo[3];
    int32_t texture;
    int32_t kind;
    float roughness;        /* Reflective */
    float transparency;     /* Refractive */
    float refraction_index; /* Refractive */
    float reflectivity;     /* Refractive */
    float emission_power;
} rt_material;

/* Light variants (src/light.rs:29-33). */
typedef enum rt_light_kind { RT_DIRECTIONAL = 0, RT_POSITIONAL = 1, RT_SPHERICAL_POSITIONAL = 2 } rt_light_kind;

/* Light (src/light.rs:3-20); vec is `direction` for Directional, `position` otherwise. */
typedef struct rt_light {
    int32_t kind;
    float albedo[3];
    float vec[3];
    float radius; /* SphericalPositional only; never read by the integrators (src/light.rs:60-75) */
    float intensity;
} rt_light;

/* Scene (src/scene.rs:13-29) without textures (rt_set_texture) and update_func (stays host-side). */
typedef struct rt_scene_desc {
    const rt_object *objects;
    const rt_material *materials;
    const rt_light *lights;
    const uint32_t *sdfs; /* Scene.sdfs: indices into objects, ray-marching list */
    uint32_t n_objects;
    uint32_t n_materials;
    uint32_t n_lights;
    uint32_t n_sdfs;
    uint32_t n_textures; /* number of texture slots Scene.textures holds */
    float ambient_color[3];
    uint8_t max_ray_bounces;
    uint8_t shadow_casting;
    uint8_t ray_marching;
    uint8_t diffuse;
    uint8_t enable_accumulation; /* Scene.enable_accumulation: reflection jitter (src/ray.rs:111-120) */
    uint8_t reserved[3];
} rt_scene_desc;

/* The parts of Camera (src/camera.rs:6-20) that calculate_ray_directions (src/camera.rs:128-157) reads. */
typedef struct rt_camera_desc {
    uint32_t width;
    uint32_t height;
    float position[3];
    float inverse_perspective[16];
    float inverse_view[16];
} rt_camera_desc;

/* Counter slots of rt_stats.counters.  A "ray" is one trace_ray (src/ray_tracing/ray_tracing.rs:59)
 * or one march_ray (src/ray_marching/ray_marching.rs:181) invocation.  The remaining slots count the
 * events the algorithmic-FLOP model prices; they are only filled by a render with collect_events=1. */
typedef enum rt_counter {
    RT_CNT_RAYS_PRIMARY = 0,
    RT_CNT_RAYS_REFLECTION,
    RT_CNT_RAYS_REFRACTION,
    RT_CNT_RAYS_SHADOW,
    RT_CNT_RAYS_MARCH,
    RT_CNT_RAYGEN,          /* camera ray generated (one per pixel)                   */
    RT_CNT_TEST_SPHERE,     /* candidate tests inside trace_ray, by primitive kind    */
    RT_CNT_TEST_TRIANGLE,
    RT_CNT_TEST_CUBOID,
    RT_CNT_TEST_PLANE,
    RT_CNT_TEST_CYLINDER,
    RT_CNT_TEST_CONE,
    RT_CNT_HIT_SPHERE,      /* closest hits finalised (shaded or shadow-tested), by kind */
    RT_CNT_HIT_TRIANGLE,
    RT_CNT_HIT_CUBOID,
    RT_CNT_HIT_PLANE,
    RT_CNT_HIT_CYLINDER,
    RT_CNT_HIT_CONE,
    RT_CNT_LIGHT_CALL,      /* light() invocations (one gamma powf triple each)        */
    RT_CNT_LIGHT_SHADE,     /* per light inside light(): blinn_phong + falloff         */
    RT_CNT_REFLECTION_RAY,  /* Ray::reflection_ray, mirror or jittered                 */
    RT_CNT_DIFFUSE_RAY,     /* Ray::reflection_ray, diffuse branch                     */
    RT_CNT_REFRACTION,      /* fresnel + refraction_ray                                */
    RT_CNT_COMBINE_REFL,    /* reflective combine (src/ray_tracing/ray_tracing.rs:194) */
    RT_CNT_COMBINE_REFR,    /* refractive combine (:232-235)                           */
    RT_CNT_TEX_FETCH,       /* Texture::from_uv outside tri-planar                     */
    RT_CNT_TRIPLANAR,       /* geometry::tri_planar_mapping (3 fetches included)       */
    RT_CNT_SDF_SPHERE,      /* leaf SDF evaluations by kind                            */
    RT_CNT_SDF_CUBOID,
    RT_CNT_SDF_CYLINDER,
    RT_CNT_SDF_PLANE,
    RT_CNT_SDF_UNION,
    RT_CNT_SDF_SUBSTRACTION,
    RT_CNT_MARCH_STEP,      /* iterations of march_ray's loop that evaluate sdfs()     */
    RT_CNT_SDFS_CALL,       /* sdfs() invocations (march + normal + occlusion)         */
    RT_CNT_OCCLUSION_TAP,
    RT_CNT_MARCH_SHADE,     /* ray-marching hits shaded (fade + occlusion scale)       */
    RT_CNT_PIXEL_PACK,      /* clamp / accumulate / to_rgba                            */
    RT_CNT_BVH_NODE,        /* box tests of the bounding-volume hierarchy (large scenes only; not in the reference) */
    RT_CNT_COUNT
} rt_counter;

typedef struct rt_stats {
    uint64_t rays_total;     /* all trace_ray + march_ray invocations of the last render (always filled) */
    uint64_t counters[RT_CNT_COUNT]; /* per-kind rays and events; zero unless collect_events was set    */
    double last_kernel_ms;   /* HIP-event time of the last render's kernel(s) on the launch stream (0 if no_timing) */
    double last_copy_ms;     /* D2H time of the last rt_render (0 for rt_render_device)            */
    uint64_t frames_rendered;
    uint64_t rays_sky_tiles; /* part of rays_total: camera rays of workgroups that the tile classification (see
                              * RT_RENDER_NO_SKY_TILES) gave the value of a miss without tracing them; the reference traces them */
    uint64_t rays_reused_camera; /* part of rays_total: camera rays that were not traced because the same ray --
                              * Camera.ray_directions[pos] does not change between frames -- was traced for the pixel's first
                              * sample of an rt_render_samples_device call, or by an earlier frame of the same camera and
                              * scene (see RT_RENDER_NO_FIRST_HIT_REUSE); the reference traces them in every frame */
} rt_stats;

/* Arguments of one frame.  Mirrors the per-frame state of Renderer (src/renderer.rs:13-18) as
 * render_chunk consumes it (src/renderer.rs:38-79). */
typedef struct rt_render_params {
    uint32_t frame_index;    /* Renderer.frame_index: divisor in accumulation mode, >= 1             */
    uint32_t row_begin;      /* rows [row_begin, row_end) are rendered; 0,0 means the whole frame     */
    uint32_t row_end;
    int32_t accumulate;      /* Renderer.enable_accumulation                                         */
    int32_t collect_events;  /* 1: also fill the event slots of rt_stats (slower kernel variant)     */
    int32_t no_timing;       /* 1: do not bracket the launch with timing events (about 6 us per launch on the
                              * stream); rt_stats.last_kernel_ms then reads 0 for it                   */
    uint64_t seed;           /* keys the counter-based generator that stands in for ThreadRng        */
    uint32_t tile_row_stride; /* 0 or 1: every row of [row_begin, row_end).  N > 1: only the 8-row bands that start at
                              * row_begin, row_begin + 8 N, row_begin + 16 N, ... (row_begin a multiple of 8): the
                              * interleaved share of one of N GPUs, the counterpart of the reference's chunk list
                              * dealt round-robin to workers (src/renderer.rs:104-110)                    */
    uint32_t flags;          /* rt_render_flag bits, 0 for the default behaviour                       */
} rt_render_params;

/* rt_render_params.flags.  No flag changes a pixel. */
typedef enum rt_render_flag {
    /* Diffuse scenes (Scene.diffuse set, no refractive material) are normally traced by kernels that do not keep one
     * pixel per lane for a whole path: large scenes (hierarchy in use) by the kernel that re-groups rays between bounces,
     * small ones by the kernel whose lanes take the next pixel when their path ends.  This flag selects the
     * one-lane-per-pixel kernel instead (A/B runs and tests that compare the two). */
    RT_RENDER_ONE_LANE_PER_PIXEL = 1,
    /* With accumulate == 0 the frame's linear value is still written to the accumulator (Renderer.accumulated,
     * src/renderer.rs:67-68) unless this flag is set: a caller that only displays `img` and never reads the accumulator
     * back saves 16 of the 20 bytes written per pixel.  Ignored when accumulate != 0. */
    RT_RENDER_SKIP_ACCUM_STORE = 2,
    /* Value 4 belonged to a flag of ABI version 2 (RT_RENDER_SAMPLE_SHARES) that no longer has a meaning: accepted and ignored. */
    RT_RENDER_RESERVED_4 = 4,
    /* Small scenes (linear scan) are rendered with a per-tile classification in front: 8x8-pixel tiles whose camera rays cannot
     * reach any object (by the objects' bounding spheres against the tile's pyramid of rays) get the value of a miss without
     * being traced.  The classification is recomputed whenever the scene, the camera or the rows of the launch change.  This flag
     * renders without it (A/B runs and tests that compare the two). */
    RT_RENDER_NO_SKY_TILES = 8,
    /* Ray-marching scenes: a march step (src/ray_marching/ray_marching.rs:186-199) normally evaluates only the entries of
     * Scene.sdfs that can hold the smallest distance -- an entry whose distance function cannot have fallen below the running
     * minimum since its last evaluation is left out.  This flag evaluates every entry in every step, as RayMarching::sdfs does
     * (A/B runs and tests that compare the two). */
    RT_RENDER_MARCH_ALL_ENTRIES = 16,
    /* Large diffuse scenes: consecutive frames of one camera and scene (what Renderer accumulates, src/renderer.rs:59-66) trace
     * the same camera rays -- Camera.ray_directions[pos] is not jittered (src/renderer.rs:49) -- so from the third frame on the
     * closest hit of every pixel's camera ray (distance and object: 8 bytes per pixel, written by the second frame) is read
     * back instead of traced again (single frames: rt_render / rt_render_device); paths continue from there with the frame's
     * own random numbers.  rt_stats counts such
     * camera rays in rays_total and in rays_reused_camera.  Any rt_set_scene / rt_update_objects / rt_set_camera, or a launch
     * over other rows, starts over.  This flag traces every camera ray (A/B runs and tests that compare the two). */
    RT_RENDER_NO_FIRST_HIT_REUSE = 32,
    RT_RENDER_ALL_FLAGS = 63   /* any other bit in params.flags is refused with RT_ERR_INVALID_ARGUMENT */
} rt_render_flag;

/* Creates a context bound to HIP device `device_id` (one context per GPU; one process per GPU
 * in multi-GPU runs).  Replaces nothing in the reference: it owns what Renderer::render keeps on
 * the host (accumulated, src/renderer.rs:14) plus the device copy of Scene/Camera. */
int32_t rt_create(int32_t device_id, rt_ctx **out);
void rt_destroy(rt_ctx *ctx);

/* Uploads Scene (src/scene.rs:13-25) as flat device arrays.  Validates what the reference
 * would panic on (material / texture / object indices out of range, src/ray_tracing/ray_tracing.rs:92,98)
 * and the limits of this backend: max_ray_bounces <= 32; Union/Substraction nesting depth <= 2.
 * Scenes with 32 or more bounded objects are traced through a bounding-volume hierarchy (same hits as the linear scan). */
int32_t rt_set_scene(rt_ctx *ctx, const rt_scene_desc *scene);

/* Re-uploads objects [first, first + count) after the caller changed them, as the binaries' update functions do every
 * frame (src/bin/scene1.rs:13-24, src/bin/scene4.rs:14-26: mutate position / rotation_axis, then Object3D::update()).
 * Kinds and Union/Substraction children must be the ones given to rt_set_scene; everything else may change.  Costs
 * one small copy instead of a scene rebuild (plus a hierarchy rebuild for scenes that use one). */
int32_t rt_update_objects(rt_ctx *ctx, uint32_t first, uint32_t count, const rt_object *objects);

/* Uploads Scene.textures[index]: tightly packed RGB8, row-major, exactly Texture.bytes
 * (src/objects.rs:105-111) as produced by ImageUtils::load_image (src/utils/image.rs:13-29). */
int32_t rt_set_texture(rt_ctx *ctx, uint32_t index, uint32_t width, uint32_t height, const uint8_t *rgb8);

/* Camera::update's tail (src/camera.rs:117-124): stores the matrices and runs Camera::calculate_ray_directions
 * (src/camera.rs:128-157) on the device, once per camera change as in the reference. */
int32_t rt_set_camera(rt_ctx *ctx, const rt_camera_desc *camera);

/* Renderer::render's pixel work (src/renderer.rs:104-143) for one frame into a host buffer:
 * every pixel p = Scene::pixel (src/scene.rs:80-90); accumulate / clamp / to_rgba as
 * render_chunk (src/renderer.rs:59-75).  out_rgba8 receives width*height*4 bytes (R,G,B,A).
 * reset_accum=1 zeroes the device accumulator first (src/renderer.rs:93-96).  Synchronous. */
int32_t rt_render(rt_ctx *ctx, uint32_t frame_index, int32_t accumulate, int32_t reset_accum, uint64_t seed,
                  uint8_t *out_rgba8, size_t out_len);

/* Optional: page-locked host memory for the frame (the reference's `img`, allocated once by App3D::run,
 * src/app.rs:77).  When out_rgba8 of rt_render lies in such a block the copy is one DMA transfer instead of a staged
 * pageable copy (measured about 3x faster delivery of a 4K frame).  Freed by rt_free_host_frame or rt_destroy.
 * No counterpart in the reference. */
int32_t rt_alloc_host_frame(rt_ctx *ctx, size_t len, void **out_ptr);
int32_t rt_free_host_frame(rt_ctx *ctx, void *ptr);

/* Same pixel work, device-resident: writes RGBA8 (width*height*4 bytes) to d_rgba8 and the
 * accumulator (width*height float4, Renderer.accumulated) to d_accum, both device pointers owned
 * by the caller; either may be NULL to use the context's own buffers.  `stream` is a hipStream_t used
 * as is (NULL is HIP's default stream; pass rt_context_stream() for the context's own).  Asynchronous:
 * returns after the launch; the context's other calls wait for it before touching its buffers. */
int32_t rt_render_device(rt_ctx *ctx, const rt_render_params *params, void *d_rgba8, void *d_accum, void *stream);

/* rt_render_device for callers that cannot allocate device memory while they render (tight serving loops, harnesses that
 * forbid hipMalloc, code under stream capture -- there only with no_timing set and on the stream of the context's previous
 * launch: a call on another stream first waits on the host for that launch, like every entry point, and capture forbids
 * that).  No counterpart in the reference.  Same frame, same arguments; the
 * difference is where the launch's per-workgroup counter rows live: in a fixed 4 MiB device array inside the library
 * instead of a block that rt_render_device grows on demand.  Calls no hipMalloc / hipFree.  Limits: workgroups x counters
 * per workgroup <= 2^20 (every frame up to 7680x4320 without collect_events; with collect_events about 26 000 workgroups,
 * e.g. 1280x720), otherwise RT_ERR_UNSUPPORTED.  d_rgba8 / d_accum NULL select the context's own buffers, which
 * rt_set_camera allocated earlier.  All calls on one GPU share the array (the kernels overwrite every row they own, so
 * nothing needs clearing between calls): launches that overlap on one GPU -- two contexts, or two streams -- race on it,
 * and rt_get_stats must be read before the next such launch.  The other device buffers a launch may want -- the pools of the
 * path tracer that re-groups rays, the sky words of the tile classification -- are used when an earlier rt_render_device
 * launch of this shape has left them in the context, and done without otherwise (one lane per pixel, no classification:
 * slower, the same pixels). */
int32_t rt_render_device_noalloc(rt_ctx *ctx, const rt_render_params *params, void *d_rgba8, void *d_accum, void *stream);

/* sample_count accumulating frames in one call: frame_index = params->frame_index, +1, ..., the pixels, accumulator and frame
 * of that many rt_render_device calls (the loop of App3D::run around Renderer::render, src/app.rs:248 with
 * src/renderer.rs:59-66, when a caller accumulates a fixed number of samples).  params->accumulate must be set;
 * collect_events is not supported.  Large diffuse scenes are rendered by ONE launch in which each pixel's samples are traced
 * one after the other and added in frame order (the order of `accumulated[pos] += p` over consecutive frames), so the
 * accumulator is read and written once per call; other scenes by one launch per sample.  Asynchronous like rt_render_device.
 * Afterwards rt_get_stats reports the rays of the whole call. */
int32_t rt_render_samples_device(rt_ctx *ctx, const rt_render_params *params, uint32_t sample_count, void *d_rgba8, void *d_accum,
                                 void *stream);

/* render_chunk's accumulation tail (src/renderer.rs:62-66) on its own: out = to_rgba(clamp(accum / sample_count, 0, 1))
 * for rows [row_begin, row_end) (0,0 = all).  Used when the samples of one frame were rendered by several GPUs and
 * their accumulators summed (all-reduce) outside this library; d_accum/d_rgba8 NULL = the context's own buffers. */
int32_t rt_resolve_device(rt_ctx *ctx, uint32_t sample_count, uint32_t row_begin, uint32_t row_end, void *d_accum,
                          void *d_rgba8, void *stream);

/* Zeroes the context's accumulator (Renderer::render on `updated`, src/renderer.rs:93-96). */
int32_t rt_reset_accum(rt_ctx *ctx);

/* Copies Renderer.accumulated (src/renderer.rs:14) back: n = width*height float4 values. */
int32_t rt_read_accum(rt_ctx *ctx, float *out_vec4, size_t n_vec4);

/* Copies Camera.ray_directions (src/camera.rs:16) back: n = width*height directions of 3 floats, row-major.  rt_set_camera
 * fills the device table (Camera::calculate_ray_directions, src/camera.rs:128-157); frames only read it. */
int32_t rt_read_ray_directions(rt_ctx *ctx, float *out_vec3, size_t n_vec3);

/* What rt_set_scene built for the current scene (no counterpart in the reference, which scans all objects linearly).
 * Scenes with fewer than 32 bounded objects have no hierarchy: every count is 0. */
typedef struct rt_hierarchy_info {
    uint32_t stackless_nodes;    /* nodes of one layout of the skip-link hierarchy (one-lane-per-pixel kernels)              */
    uint32_t stackless_linear;   /* objects outside it, tested linearly (planes, and cones: their intersect() has no usable bounds) */
    uint32_t two_child_nodes;    /* nodes of the quantised two-child hierarchy; 0 unless the scene is diffuse without glass */
    uint32_t two_child_linear;   /* objects outside it: planes, cones and objects too large for its 16-bit grid           */
    uint32_t regrouping_kernel;  /* 1: rt_render_device traces this scene with the kernel that re-groups rays between bounces */
    uint32_t sky_workgroups;     /* last tile classification (see RT_RENDER_NO_SKY_TILES): workgroups whose tiles are all sky ... */
    uint32_t classified_workgroups; /* ... of this many workgroups of that launch; both 0 when no launch has used one          */
    uint32_t reserved[1];
} rt_hierarchy_info;
int32_t rt_get_hierarchy_info(rt_ctx *ctx, rt_hierarchy_info *out);

/* Self-check of the transcendental functions the kernels share with every caller that wants bit-equal frames, and of
 * their square root (op RT_MATH_SQRT: out[i] = f32::sqrt(x[i]) as the kernels evaluate it, correctly rounded; y ignored):
 * op RT_MATH_POWF: out[i] = powf(x[i], y[i]) as Ray::blinn_phong and the gamma steps evaluate it (f32::powf at src/ray.rs:81,
 * src/ray_tracing/ray_tracing.rs:55, src/ray_marching/ray_marching.rs:178); op RT_MATH_ATANF: out[i] = atanf(x[i]) (f32::atan at
 * src/utils/sphere.rs:136-137; y is ignored and may be NULL).  Evaluated on the device by the same inlined code the
 * render kernels use; host pointers, n elements each.  Synchronous. */
enum rt_math_op { RT_MATH_POWF = 0, RT_MATH_ATANF = 1, RT_MATH_SQRT = 2 };
int32_t rt_eval_math(rt_ctx *ctx, int32_t op, size_t n, const float *x, const float *y, float *out);

/* Waits for the context's stream and sums the per-workgroup counters of the last render. */
int32_t rt_get_stats(rt_ctx *ctx, rt_stats *out);

/* The hipStream_t the context created for itself (rt_render runs on it).  No counterpart in the reference. */
void *rt_context_stream(rt_ctx *ctx);

/* Device pointers of the context's own frame buffers (valid until the next rt_set_camera). */
void *rt_device_rgba8(rt_ctx *ctx);
void *rt_device_accum(rt_ctx *ctx);

const char *rt_last_error(rt_ctx *ctx);
int32_t rt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355X_H */
