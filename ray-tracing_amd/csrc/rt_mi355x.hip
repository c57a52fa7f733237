// rt_mi355x.hip — host runtime behind the C ABI of include/rt_mi355x.h: owns the device copy of
// Scene / Camera / frame buffers and launches the gfx950 kernels of rt_kernels.h.
//
// What it replaces in the reference: the body of Renderer::render between the reset-on-update and the
// texture upload (src/renderer.rs:104-143), plus Camera::calculate_ray_directions (src/camera.rs:128-157),
// which the kernels fold into ray generation.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "rt_bvh.h"
#include "rt_dev_scene.h"
#include "rt_kernels.h"
#include "rt_wavefront.h"

#pragma clang fp contract(off)

namespace {

constexpr uint32_t kMaxTextures = 64;
constexpr uint32_t kMaxBounces = 32;  // capacity of the per-lane frame storage
constexpr uint32_t kMaxCompositeDepth = 2;

// Counter rows of rt_render_device_noalloc: one row per workgroup of a launch.  A file-scope device array instead of a
// hipMalloc'd block, so that the entry point never allocates; every GPU the library runs on holds its own 4 MiB copy until
// the process ends.  Host code never uses the array's name as an address: the per-device address comes from
// hipGetSymbolAddress (scratch_for_device) and travels to the kernels as an ordinary pointer argument.
constexpr size_t kNoAllocCounterWords = (size_t)1 << 20;
static __device__ __align__(256) uint32_t scratch[kNoAllocCounterWords];

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
uint32_t *g_scratch_ptr[kMaxDevices] = {};

// Address of `scratch` on `device`, which must be the current device.  Looked up once per device.
hipError_t scratch_for_device(int device, uint32_t **out) {
    *out = nullptr;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_scratch_ptr[device] = (uint32_t *)p;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

template <typename T>
struct DevBuf {
    T *ptr = nullptr;
    size_t cap = 0;  // elements
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (ptr) { (void)hipFree(ptr); ptr = nullptr; cap = 0; }
        hipError_t e = hipMalloc((void **)&ptr, n * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release() { if (ptr) (void)hipFree(ptr); ptr = nullptr; cap = 0; }
};

}  // namespace

struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, evc0 = nullptr, evc1 = nullptr;
    std::string err;

    DevBuf<DevObject> objects;
    DevBuf<DevMaterial> materials;
    DevBuf<DevLight> lights;
    DevBuf<uint32_t> sdfs;
    DevBuf<DevTexture> textures;
    DevBuf<DevBvhNode> bvh_nodes;
    DevBuf<uint32_t> unbounded;
    DevBuf<DevBvh2Node> bvh2_nodes;   // quantised two-child hierarchy of the re-grouping path tracer
    DevBuf<uint32_t> bvh2_linear;
    DevBuf<uint32_t> tex_data[kMaxTextures];
    DevTexture h_textures[kMaxTextures] = {};
    std::vector<int32_t> mat_texture;  // texture index of every material, for the render-time check
    std::vector<rt_object> h_objects;  // host copy of the scene's objects (rt_update_objects, hierarchy rebuilds)
    std::vector<uint32_t> h_sdfs;      // host copy of Scene.sdfs

    DevScene scene = {};
    DevCamera camera = {};
    bool has_scene = false, has_camera = false, has_refractive = false;
    bool spheres_planes_only = false;   // every object of the scene is a sphere or a plane: kernels without the other primitives
    uint32_t leaf_slots = 0;  // ray-marching scenes: leaves per sdfs() call with an object-space ray

    DevBuf<uint32_t> rgba;
    DevBuf<float> accum;
    DevBuf<float> ray_dirs;   // Camera.ray_directions: 3 floats per pixel
    // The table is filled lazily.  The first frame after a camera change computes its directions in the render kernel
    // (a moving camera never pays for writing and re-reading 12 bytes per pixel); the second frame of an unchanged camera
    // fills the table on its stream, and frames read it from then on, as the reference's frames do (src/renderer.rs:49).
    bool dirs_valid = false;          // ray_dirs holds the directions of `camera`
    uint32_t frames_this_camera = 0;  // renders since the last rt_set_camera
    std::vector<std::pair<uint32_t, DevObject>> pending_objects;   // rt_update_objects records not yet on the device
    DevBuf<float4> wf_pool;       // ray pools of the re-grouping path tracer: wf_groups x kWfPool entries of 3.5 float4
    DevBuf<uint32_t> wf_ctrl;     // its tile counter
    uint32_t wf_groups = 0;       // persistent single-wave workgroups that fit the device (0: not queried yet)
    uint32_t wf_chain_groups = 0; // the same for the kernel with per-pixel sample chains
    size_t n_bvh2_nodes = 0;      // nodes of the quantised two-child hierarchy of the current scene
    uint64_t wf_order_key[2] = {~0ull, ~0ull};   // scene / camera / row range the tile-row costs in wf_ctrl belong to
    DevBuf<float2> first_hit;                    // closest hit of every pixel's camera ray (see WfArgs::first_hit)
    uint64_t first_hit_key[2] = {~0ull, ~0ull};  // scene / camera / row range of the launches counted in first_hit_launches
    uint32_t first_hit_launches = 0;             // consecutive launches of the re-grouping kernel with that key
    bool first_hit_filled = false;               // one of them has written the table
    uint64_t host_counted_rays = 0;              // camera rays of the last render(s) that were read from the table: counted here, not in the kernel
    uint32_t scene_serial = 0;    // bumped by rt_set_scene, rt_update_objects and rt_set_camera
    DevBuf<uint32_t> tile_sky;    // rt_tile_sky_kernel's word per workgroup of the launch described by tile_sky_key
    uint64_t tile_sky_key[2] = {~0ull, ~0ull};
    uint32_t tile_sky_words = 0, tile_sky_waves = 4;
    hipEvent_t tile_sky_done = nullptr;      // recorded behind the last classification kernel
    DevBuf<uint32_t> counters;
    uint32_t *static_counters = nullptr;       // this device's copy of the file-scope `scratch` array (rt_render_device_noalloc)
    const uint32_t *last_counters = nullptr;   // counter table the last render wrote
    std::vector<uint32_t> h_counters;
    uint32_t last_blocks = 0, last_ncnt = 0;
    uint32_t last_reused_rows = 0;   // > 0: the table has this many more rows behind the first last_blocks: rays counted there but not traced
    bool kernel_timed = false;           // the last launch was bracketed by ev0 / ev1
    bool last_events = false, timed = false;
    hipStream_t last_stream = nullptr;
    double last_copy_ms = 0.0;
    uint64_t frames = 0;
    std::vector<std::pair<void *, size_t>> pinned;  // page-locked blocks handed out by rt_alloc_host_frame
};

namespace {

int32_t fail(rt_ctx *c, rt_status st, const std::string &msg) {
    if (c) c->err = msg;
    return (int32_t)st;
}
// Nothing may be thrown across the C boundary: every entry point that can allocate host memory runs inside
// RT_GUARD_BEGIN / RT_GUARD_END, which turn an exception (std::bad_alloc from a vector or string) into RT_ERR_DEVICE.
int32_t guard_fail(rt_ctx *c) {
    if (c) { try { c->err = "out of host memory (or another C++ exception) inside the library"; } catch (...) { } }
    return (int32_t)RT_ERR_DEVICE;
}
#define RT_GUARD_BEGIN try {
#define RT_GUARD_END(ctx) } catch (...) { return guard_fail(ctx); }

#define RT_HIP(ctx, call)                                                                                   \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess) return fail(ctx, RT_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// columns of a glam Mat4 (xyz rows) + the `column 3 times 0.0` term of a direction transform
DevXform make_xform(const float *m) {
    DevXform x;
    for (int r = 0; r < 3; r++) {
        x.c0[r] = m[r]; x.c1[r] = m[4 + r]; x.c2[r] = m[8 + r]; x.c3[r] = m[12 + r];
        x.z[r] = m[12 + r] * 0.0f;
    }
    return x;
}

// Device record of one object: matrices as 3-row columns, kind-specific parameters in p[].
DevObject make_dev_object(const rt_object &o) {
    DevObject g;
    std::memset(&g, 0, sizeof(g));
    g.kind = o.kind;
    g.material = (o.kind == RT_UNION || o.kind == RT_SUBSTRACTION) ? 0 : o.material_index;
    g.first = o.first; g.second = o.second;
    g.inv = make_xform(o.inv_transform);
    g.fwd = make_xform(o.transform);
    switch (o.kind) {
        case RT_SPHERE: g.p[0] = o.radius; g.diag = rt_object_is_diag_sphere(o) ? 1 : 0; break;
        case RT_CUBOID:
            for (int k = 0; k < 3; k++) {
                g.p[k] = o.dimension[k];
                // b_min / b_max = (Mat4::from_scale(dimension) * vec4(-+1, -+1, -+1, 0)).xyz (src/utils/cuboid.rs:48-49)
                float col[3] = {0.0f, 0.0f, 0.0f};
                col[k] = o.dimension[k];
                g.p[3 + k] = ((col[0] * -1.0f + col[1] * -1.0f) + col[2] * -1.0f) + 0.0f * 0.0f;
                g.p[6 + k] = ((col[0] * 1.0f + col[1] * 1.0f) + col[2] * 1.0f) + 0.0f * 0.0f;
            }
            break;
        case RT_PLANE:
            for (int k = 0; k < 3; k++) { g.p[k] = o.normal[k]; g.p[3 + k] = o.point[k]; }
            g.p[6] = o.max_dist[0]; g.p[7] = o.max_dist[1]; g.p[8] = o.has_max_dist ? 1.0f : 0.0f;
            break;
        case RT_CYLINDER: g.p[0] = o.radius; g.p[1] = o.height; break;
        case RT_TRIANGLE:
            for (int k = 0; k < 3; k++) { g.p[k] = o.v1[k]; g.p[3 + k] = o.v2[k]; g.p[6 + k] = o.v3[k]; }
            break;
        default: break;
    }
    // bounding sphere around the (already padded) world box of rt_object_bounds; radius rounded up generously
    RtAabb box;
    g.bounded = 0;
    g.bs[0] = g.bs[1] = g.bs[2] = 0.0f; g.bs[3] = 3.0e38f;
    if (rt_object_bounds(o, box)) {
        double r2 = 0.0;
        for (int k = 0; k < 3; k++) {
            const double c = 0.5 * ((double)box.lo[k] + (double)box.hi[k]), h = 0.5 * ((double)box.hi[k] - (double)box.lo[k]);
            g.bs[k] = (float)c;
            const double hh = h + std::fabs(c - (double)g.bs[k]);   // the centre was rounded to binary32
            r2 += hh * hh;
        }
        const float r2f = (float)(r2 * 1.002 + 1e-12);
        if (std::isfinite(r2f)) { g.bs[3] = r2f; g.bounded = 1; }
    }
    return g;
}

int composite_depth(const rt_scene_desc *d, uint32_t i, int guard) {
    if (i >= d->n_objects || guard > 8) return -1;
    const rt_object &o = d->objects[i];
    if (o.kind != RT_UNION && o.kind != RT_SUBSTRACTION) return 0;
    if (o.first < 0 || o.second < 0) return -1;
    int a = composite_depth(d, (uint32_t)o.first, guard + 1), b = composite_depth(d, (uint32_t)o.second, guard + 1);
    if (a < 0 || b < 0) return -1;
    return 1 + (a > b ? a : b);
}

// Leaves whose object-space ray the march kernel can cache, counted the way Sdf::eval / rm_fill_leaf visit them
// (`level` = remaining composite nesting).
uint32_t march_leaf_slots(const rt_scene_desc *d, uint32_t i, int level) {
    const rt_object &o = d->objects[i];
    switch (o.kind) {
        case RT_SPHERE: case RT_CUBOID: case RT_CYLINDER: return 1u;
        case RT_UNION: case RT_SUBSTRACTION:
            return level > 0 ? march_leaf_slots(d, (uint32_t)o.first, level - 1) + march_leaf_slots(d, (uint32_t)o.second, level - 1) : 0u;
        default: return 0u;
    }
}

// Camera::update ends with calculate_ray_directions (src/camera.rs:124): fills Camera.ray_directions for the current camera on
// stream `s` (the caller has ordered `s` behind whatever still reads the table).
hipError_t fill_ray_directions(rt_ctx *c, hipStream_t s) {
    const DevCamera &C = c->camera;
    hipLaunchKernelGGL(rt_raygen_kernel, dim3((C.width + 63u) / 64u, (C.height + 3u) / 4u), dim3(256), 0, s, C, c->ray_dirs.ptr);
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) c->dirs_valid = true;
    return e;
}

// Largest factor by which the 3x3 part of a column-major 4x4 matrix stretches a vector (its spectral norm), rounded up: the
// square root of the largest eigenvalue of A^T A, by Jacobi rotations in binary64.
double stretch3(const float *m) {
    double a[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            a[i][j] = 0.0;
            for (int r = 0; r < 3; r++) a[i][j] += (double)m[4 * i + r] * (double)m[4 * j + r];   // columns i and j
        }
    for (int sweep = 0; sweep < 32; sweep++)
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                if (std::fabs(a[p][q]) < 1e-300) continue;
                const double theta = 0.5 * std::atan2(2.0 * a[p][q], a[q][q] - a[p][p]), cs = std::cos(theta), sn = std::sin(theta);
                double b[3][3];
                std::memcpy(b, a, sizeof(b));
                for (int k = 0; k < 3; k++) { b[k][p] = cs * a[k][p] - sn * a[k][q]; b[k][q] = sn * a[k][p] + cs * a[k][q]; }
                std::memcpy(a, b, sizeof(a));
                for (int k = 0; k < 3; k++) { b[p][k] = cs * a[p][k] - sn * a[q][k]; b[q][k] = sn * a[p][k] + cs * a[q][k]; }
                std::memcpy(a, b, sizeof(a));
            }
    const double ev = std::max(a[0][0], std::max(a[1][1], a[2][2]));
    // the rotations leave the trace alone and only shrink the off-diagonal part: whatever is left of it is added as a bound
    const double off = std::fabs(a[0][1]) + std::fabs(a[0][2]) + std::fabs(a[1][2]);
    return std::sqrt(std::max(ev + 2.0 * off, 0.0));
}

// How fast the distance function of object i (Object3D::sdf, src/objects.rs:33-44) can fall along a ray, per unit length:
// the leaf functions are distances (or, for the sphere's `length(p) - radius`, 1-Lipschitz) in the object's space, so the
// bound is the stretch of inv_transform; a plane's is |normal|; a smooth union or a subtraction falls no faster than the
// faster of its children.  Infinity where no bound is known (such an entry is evaluated in every step).
double march_fall_rate(const std::vector<rt_object> &objs, uint32_t i, int level) {
    const rt_object &o = objs[i];
    switch (o.kind) {
        case RT_SPHERE: case RT_CUBOID: case RT_CYLINDER: return stretch3(o.inv_transform);
        case RT_PLANE: return std::sqrt((double)o.normal[0] * o.normal[0] + (double)o.normal[1] * o.normal[1] + (double)o.normal[2] * o.normal[2]);
        case RT_UNION: case RT_SUBSTRACTION:
            if (level <= 0 || o.first < 0 || o.second < 0) return HUGE_VAL;
            return std::max(march_fall_rate(objs, (uint32_t)o.first, level - 1), march_fall_rate(objs, (uint32_t)o.second, level - 1));
        default: return 0.0;   // cone, triangle: never evaluated (src/ray_marching/ray_marching.rs:95)
    }
}

// A render may have been launched on a caller's stream; anything that touches the same buffers from another
// stream waits for it first.
hipError_t wait_external(rt_ctx *c, hipStream_t next) {
    if (c->timed && c->last_stream != next) return hipStreamSynchronize(c->last_stream);
    return hipSuccess;
}

#ifndef RT_MARCH_CACHE_MAX_SLOTS
#define RT_MARCH_CACHE_MAX_SLOTS 16
#endif

template <typename K>
void launch(K kernel, dim3 grid, uint32_t threads, size_t lds, hipStream_t s, const DevScene &S, const DevCamera &C, const DevFrame &F) {
    hipLaunchKernelGGL(kernel, grid, dim3(threads), lds, s, S, C, F);
}

// Scenes with many bounded objects get a bounding-volume hierarchy; small scenes keep the reference's linear scan.
// Built from the context's host copy of the objects (again after rt_update_objects moved some of them).
int32_t upload_hierarchy(rt_ctx *c) {
    RT_GUARD_BEGIN
    DevScene &S = c->scene;
    S.bvh_nodes = nullptr; S.unbounded = nullptr; S.n_bvh_nodes = 0; S.n_unbounded = 0;
    S.bvh2_nodes = nullptr; S.bvh2_linear = nullptr; S.n_bvh2_linear = 0;
    rt_scene_desc d;
    std::memset(&d, 0, sizeof(d));
    d.objects = c->h_objects.data(); d.n_objects = (uint32_t)c->h_objects.size();
    if (S.ray_marching || !rt_bvh_wanted(&d)) return RT_OK;
    std::vector<DevBvhNode> nodes; std::vector<uint32_t> unbounded;
    rt_bvh_build(&d, nodes, unbounded);
    RT_HIP(c, c->bvh_nodes.ensure(nodes.size()));
    RT_HIP(c, c->unbounded.ensure(unbounded.size() ? unbounded.size() : 1));
    RT_HIP(c, hipMemcpy(c->bvh_nodes.ptr, nodes.data(), nodes.size() * sizeof(DevBvhNode), hipMemcpyHostToDevice));
    if (!unbounded.empty()) RT_HIP(c, hipMemcpy(c->unbounded.ptr, unbounded.data(), unbounded.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    S.bvh_nodes = c->bvh_nodes.ptr; S.unbounded = c->unbounded.ptr;
    S.n_bvh_nodes = (uint32_t)(nodes.size() / kBvhLayouts); S.n_unbounded = (uint32_t)unbounded.size();
    // the second form of the hierarchy, for the kernel that re-groups rays: only scenes that kernel can render
    if (S.diffuse && !c->has_refractive) {
        RtBvh2 b2;
        rt_bvh2_build(&d, b2);
        if (b2.usable) {
            RT_HIP(c, c->bvh2_nodes.ensure(b2.nodes.size()));
            RT_HIP(c, c->bvh2_linear.ensure(b2.linear.size() ? b2.linear.size() : 1));
            RT_HIP(c, hipMemcpy(c->bvh2_nodes.ptr, b2.nodes.data(), b2.nodes.size() * sizeof(DevBvh2Node), hipMemcpyHostToDevice));
            if (!b2.linear.empty()) RT_HIP(c, hipMemcpy(c->bvh2_linear.ptr, b2.linear.data(), b2.linear.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            S.bvh2_nodes = c->bvh2_nodes.ptr; S.bvh2_linear = c->bvh2_linear.ptr; S.n_bvh2_linear = (uint32_t)b2.linear.size();
            c->n_bvh2_nodes = b2.nodes.size();
            for (int k = 0; k < 3; k++) { S.bvh2_origin[k] = b2.origin[k]; S.bvh2_scale[k] = b2.scale[k]; }
        }
    }
    return RT_OK;
    RT_GUARD_END(c)
}

}  // namespace

extern "C" {

int32_t rt_abi_version(void) { return RT_ABI_VERSION; }

const char *rt_last_error(rt_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int32_t rt_create(int32_t device_id, rt_ctx **out) {
    RT_GUARD_BEGIN
    if (!out) return RT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return RT_ERR_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return RT_ERR_DEVICE;
    rt_ctx *c = new rt_ctx();
    c->device = device_id;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&c->ev0) != hipSuccess ||
        hipEventCreate(&c->ev1) != hipSuccess || hipEventCreate(&c->evc0) != hipSuccess || hipEventCreate(&c->evc1) != hipSuccess ||
        c->textures.ensure(kMaxTextures) != hipSuccess) {
        rt_destroy(c);
        return RT_ERR_DEVICE;
    }
    if (hipMemset(c->textures.ptr, 0, kMaxTextures * sizeof(DevTexture)) != hipSuccess || scratch_for_device(device_id, &c->static_counters) != hipSuccess) {
        rt_destroy(c);
        return RT_ERR_DEVICE;
    }
    *out = c;
    return RT_OK;
    RT_GUARD_END(nullptr)
}

void rt_destroy(rt_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    // every launch of this context -- on its own stream, on a caller's stream (which may no longer exist: it is never touched
    // here) or on the share streams of a sample batch -- has drained before the first buffer goes
    (void)hipDeviceSynchronize();
    for (const auto &r : c->pinned) (void)hipHostFree(r.first);
    c->objects.release(); c->materials.release(); c->lights.release(); c->sdfs.release(); c->textures.release();
    c->bvh_nodes.release(); c->unbounded.release(); c->bvh2_nodes.release(); c->bvh2_linear.release();
    for (auto &t : c->tex_data) t.release();
    c->rgba.release(); c->accum.release(); c->ray_dirs.release(); c->counters.release(); c->wf_pool.release(); c->wf_ctrl.release();
    c->tile_sky.release();
    c->first_hit.release();
    if (c->tile_sky_done) (void)hipEventDestroy(c->tile_sky_done);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->evc0) (void)hipEventDestroy(c->evc0);
    if (c->evc1) (void)hipEventDestroy(c->evc1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int32_t rt_set_scene(rt_ctx *c, const rt_scene_desc *d) {
    RT_GUARD_BEGIN
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (!d) return fail(c, RT_ERR_INVALID_ARGUMENT, "scene is null");
    if ((d->n_objects && !d->objects) || (d->n_materials && !d->materials) || (d->n_lights && !d->lights) || (d->n_sdfs && !d->sdfs))
        return fail(c, RT_ERR_INVALID_ARGUMENT, "scene array pointer is null");
    if (d->n_textures > kMaxTextures) return fail(c, RT_ERR_UNSUPPORTED, "more than 64 texture slots");
    if (d->max_ray_bounces > kMaxBounces) return fail(c, RT_ERR_UNSUPPORTED, "max_ray_bounces above 32");
    if (d->n_objects >= (1u << 28)) return fail(c, RT_ERR_UNSUPPORTED, "2^28 or more objects");  // hierarchy leaves pack kind and index in 32 bits
    // what the reference would panic on (index out of bounds), checked up front
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const rt_material &m = d->materials[i];
        if (m.kind != RT_REFLECTIVE && m.kind != RT_REFRACTIVE) return fail(c, RT_ERR_INVALID_ARGUMENT, "material " + std::to_string(i) + ": unknown kind");
        if (m.texture < -1 || (m.texture >= 0 && (uint32_t)m.texture >= d->n_textures))
            return fail(c, RT_ERR_INVALID_ARGUMENT, "material " + std::to_string(i) + ": texture index out of range");
    }
    for (uint32_t i = 0; i < d->n_objects; i++) {
        const rt_object &o = d->objects[i];
        if (o.kind < RT_SPHERE || o.kind > RT_SUBSTRACTION) return fail(c, RT_ERR_INVALID_ARGUMENT, "object " + std::to_string(i) + ": unknown kind");
        if (o.kind == RT_UNION || o.kind == RT_SUBSTRACTION) {
            int depth = composite_depth(d, i, 0);
            if (depth < 0) return fail(c, RT_ERR_INVALID_ARGUMENT, "object " + std::to_string(i) + ": child index out of range or cyclic");
            if ((uint32_t)depth > kMaxCompositeDepth) return fail(c, RT_ERR_UNSUPPORTED, "object " + std::to_string(i) + ": Union/Substraction nested deeper than 2");
        } else if (o.material_index < 0 || (uint32_t)o.material_index >= d->n_materials) {
            return fail(c, RT_ERR_INVALID_ARGUMENT, "object " + std::to_string(i) + ": material index out of range");
        }
    }
    for (uint32_t i = 0; i < d->n_sdfs; i++)
        if (d->sdfs[i] >= d->n_objects) return fail(c, RT_ERR_INVALID_ARGUMENT, "sdfs[" + std::to_string(i) + "] out of range");
    for (uint32_t i = 0; i < d->n_lights; i++)
        if (d->lights[i].kind < RT_DIRECTIONAL || d->lights[i].kind > RT_SPHERICAL_POSITIONAL) return fail(c, RT_ERR_INVALID_ARGUMENT, "light " + std::to_string(i) + ": unknown kind");
    if (d->ray_marching && d->n_sdfs > 0 && d->n_materials == 0) return fail(c, RT_ERR_INVALID_ARGUMENT, "ray marching scene without materials");

    RT_HIP(c, hipSetDevice(c->device));
    // from here on the context's scene is being replaced: it counts as "not set" until every step below has succeeded, so a
    // failure in between can never leave a render launching on freed or half-updated buffers
    c->has_scene = false;
    c->pending_objects.clear();
    std::vector<DevObject> objs(d->n_objects);
    for (uint32_t i = 0; i < d->n_objects; i++) objs[i] = make_dev_object(d->objects[i]);
    c->spheres_planes_only = d->n_objects > 0;
    for (uint32_t i = 0; i < d->n_objects; i++)
        if (d->objects[i].kind != RT_SPHERE && d->objects[i].kind != RT_PLANE) c->spheres_planes_only = false;
    std::vector<DevMaterial> mats(d->n_materials);
    c->mat_texture.assign(d->n_materials, -1);
    c->has_refractive = false;
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const rt_material &m = d->materials[i];
        DevMaterial &g = mats[i];
        std::memset(&g, 0, sizeof(g));
        g.ambience = m.ambience; g.diffuse = m.diffuse; g.specular = m.specular; g.shininess = m.shininess;
        for (int k = 0; k < 3; k++) g.albedo[k] = m.albedo[k];
        g.texture = m.texture; g.kind = m.kind; g.roughness = m.roughness; g.transparency = m.transparency;
        g.refraction_index = m.refraction_index; g.reflectivity = m.reflectivity; g.emission_power = m.emission_power;
        c->mat_texture[i] = m.texture;
        if (m.kind == RT_REFRACTIVE) c->has_refractive = true;
    }
    std::vector<DevLight> lights(d->n_lights);
    for (uint32_t i = 0; i < d->n_lights; i++) {
        const rt_light &l = d->lights[i];
        DevLight &g = lights[i];
        g.kind = l.kind; g.intensity = l.intensity;
        for (int k = 0; k < 3; k++) { g.albedo[k] = l.albedo[k]; g.vec[k] = l.vec[k]; }
    }
    RT_HIP(c, wait_external(c, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    RT_HIP(c, c->objects.ensure(objs.size() ? objs.size() : 1));
    RT_HIP(c, c->materials.ensure(mats.size() ? mats.size() : 1));
    RT_HIP(c, c->lights.ensure(lights.size() ? lights.size() : 1));
    RT_HIP(c, c->sdfs.ensure(d->n_sdfs ? d->n_sdfs : 1));
    if (!objs.empty()) RT_HIP(c, hipMemcpy(c->objects.ptr, objs.data(), objs.size() * sizeof(DevObject), hipMemcpyHostToDevice));
    if (!mats.empty()) RT_HIP(c, hipMemcpy(c->materials.ptr, mats.data(), mats.size() * sizeof(DevMaterial), hipMemcpyHostToDevice));
    if (!lights.empty()) RT_HIP(c, hipMemcpy(c->lights.ptr, lights.data(), lights.size() * sizeof(DevLight), hipMemcpyHostToDevice));
    if (d->n_sdfs) RT_HIP(c, hipMemcpy(c->sdfs.ptr, d->sdfs, d->n_sdfs * sizeof(uint32_t), hipMemcpyHostToDevice));

    DevScene &S = c->scene;
    std::memset(&S, 0, sizeof(S));
    S.objects = c->objects.ptr; S.materials = c->materials.ptr; S.lights = c->lights.ptr; S.sdfs = c->sdfs.ptr;
    S.textures = c->textures.ptr;
    S.n_objects = d->n_objects; S.n_materials = d->n_materials; S.n_lights = d->n_lights; S.n_sdfs = d->n_sdfs;
    for (int k = 0; k < 3; k++) S.ambient[k] = d->ambient_color[k];
    S.max_ray_bounces = d->max_ray_bounces; S.shadow_casting = d->shadow_casting; S.ray_marching = d->ray_marching;
    S.diffuse = d->diffuse; S.enable_accumulation = d->enable_accumulation;

    c->h_objects.assign(d->objects, d->objects + d->n_objects);
    c->h_sdfs.assign(d->sdfs, d->sdfs + d->n_sdfs);
    { int32_t r = upload_hierarchy(c); if (r != RT_OK) return r; }
    c->leaf_slots = 0;
    if (d->ray_marching) {
        for (uint32_t k = 0; k < d->n_sdfs; k++) {
            const int kind = d->objects[d->sdfs[k]].kind;
            if (kind == RT_TRIANGLE || kind == RT_CONE) continue;
            c->leaf_slots += march_leaf_slots(d, d->sdfs[k], 2);
        }
    }
    c->scene_serial += 1u;
    c->has_scene = true;
    return RT_OK;
    RT_GUARD_END(c)
}

int32_t rt_update_objects(rt_ctx *c, uint32_t first, uint32_t count, const rt_object *objects) {
    RT_GUARD_BEGIN
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (!c->has_scene) return fail(c, RT_ERR_STATE, "scene not set");
    if (count == 0) return RT_OK;
    if (!objects) return fail(c, RT_ERR_INVALID_ARGUMENT, "objects is null");
    if ((uint64_t)first + count > c->h_objects.size()) return fail(c, RT_ERR_INVALID_ARGUMENT, "object range out of bounds");
    for (uint32_t i = 0; i < count; i++) {
        const rt_object &o = objects[i], &old = c->h_objects[first + i];
        const std::string who = "object " + std::to_string(first + i);
        if (o.kind != old.kind) return fail(c, RT_ERR_INVALID_ARGUMENT, who + ": kind differs from the scene's (use rt_set_scene)");
        if (o.kind == RT_UNION || o.kind == RT_SUBSTRACTION) {
            if (o.first != old.first || o.second != old.second) return fail(c, RT_ERR_INVALID_ARGUMENT, who + ": children differ from the scene's (use rt_set_scene)");
        } else if (o.material_index < 0 || (uint32_t)o.material_index >= c->scene.n_materials) {
            return fail(c, RT_ERR_INVALID_ARGUMENT, who + ": material index out of range");
        }
    }
    std::vector<DevObject> objs(count);
    for (uint32_t i = 0; i < count; i++) objs[i] = make_dev_object(objects[i]);
    RT_HIP(c, hipSetDevice(c->device));
    if (!c->scene.bvh_nodes && count <= 8u * kPatchObjects) {
        // A few objects of a scene without hierarchy (what the reference's update functions do every frame): no waiting here.
        // The records are handed to the next render launch, which patches the device array on its own stream first.
        for (uint32_t i = 0; i < count; i++) {
            bool found = false;
            for (auto &pend : c->pending_objects) if (pend.first == first + i) { pend.second = objs[i]; found = true; }
            if (!found) c->pending_objects.emplace_back(first + i, objs[i]);
        }
        std::copy(objects, objects + count, c->h_objects.begin() + first);
        c->scene_serial += 1u;
        return RT_OK;
    }
    RT_HIP(c, wait_external(c, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    c->pending_objects.clear();   // (none: scenes with a hierarchy never queue)
    c->has_scene = false;   // set again once the device copy and the hierarchies are consistent
    RT_HIP(c, hipMemcpy(c->objects.ptr + first, objs.data(), (size_t)count * sizeof(DevObject), hipMemcpyHostToDevice));
    std::copy(objects, objects + count, c->h_objects.begin() + first);
    c->scene_serial += 1u;
    if (c->scene.bvh_nodes) { int32_t r = upload_hierarchy(c); if (r != RT_OK) return r; }  // moved objects: new boxes
    c->has_scene = true;
    return RT_OK;
    RT_GUARD_END(c)
}

int32_t rt_set_texture(rt_ctx *c, uint32_t index, uint32_t width, uint32_t height, const uint8_t *rgb8) {
    RT_GUARD_BEGIN
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (index >= kMaxTextures) return fail(c, RT_ERR_UNSUPPORTED, "texture slot above 63");
    if (!rgb8 || width == 0 || height == 0 || width > 32768 || height > 32768) return fail(c, RT_ERR_INVALID_ARGUMENT, "bad texture arguments");
    RT_HIP(c, hipSetDevice(c->device));
    const size_t n = (size_t)width * height;
    std::vector<uint32_t> packed(n);
    for (size_t i = 0; i < n; i++) packed[i] = (uint32_t)rgb8[3 * i] | ((uint32_t)rgb8[3 * i + 1] << 8) | ((uint32_t)rgb8[3 * i + 2] << 16);
    RT_HIP(c, wait_external(c, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    RT_HIP(c, c->tex_data[index].ensure(n));
    RT_HIP(c, hipMemcpy(c->tex_data[index].ptr, packed.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->h_textures[index].texels = c->tex_data[index].ptr;
    c->h_textures[index].width = width;
    c->h_textures[index].height = height;
    RT_HIP(c, hipMemcpy(c->textures.ptr + index, &c->h_textures[index], sizeof(DevTexture), hipMemcpyHostToDevice));
    return RT_OK;
    RT_GUARD_END(c)
}

int32_t rt_set_camera(rt_ctx *c, const rt_camera_desc *cam) {
    RT_GUARD_BEGIN
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (!cam || cam->width == 0 || cam->height == 0) return fail(c, RT_ERR_INVALID_ARGUMENT, "bad camera");
    if ((uint64_t)cam->width * cam->height > (1ull << 31)) return fail(c, RT_ERR_UNSUPPORTED, "frame above 2^31 pixels");
    if (cam->width > 65535u || cam->height > 65535u) return fail(c, RT_ERR_UNSUPPORTED, "frame wider or higher than 65535 pixels");
    RT_HIP(c, hipSetDevice(c->device));
    c->has_camera = false;   // until the buffers and the direction table below are in place
    DevCamera &C = c->camera;
    for (int r = 0; r < 4; r++) {
        C.ip_c0[r] = cam->inverse_perspective[r]; C.ip_c1[r] = cam->inverse_perspective[4 + r];
        C.ip_c2[r] = cam->inverse_perspective[8 + r]; C.ip_c3[r] = cam->inverse_perspective[12 + r];
    }
    C.iv = make_xform(cam->inverse_view);
    for (int k = 0; k < 3; k++) C.position[k] = cam->position[k];
    const size_t px = (size_t)cam->width * cam->height;
    if (px != (size_t)C.width * C.height || !c->rgba.ptr) {
        // a new frame size: the buffers are replaced, so everything in flight has to drain first
        RT_HIP(c, hipDeviceSynchronize());
        RT_HIP(c, c->rgba.ensure(px));
        RT_HIP(c, c->accum.ensure(px * 4));
        RT_HIP(c, c->ray_dirs.ensure(px * 3));
        RT_HIP(c, hipMemsetAsync(c->accum.ptr, 0, px * 4 * sizeof(float), c->stream));
        RT_HIP(c, hipStreamSynchronize(c->stream));
    }
    // Same size: nothing to wait for and nothing to launch.  Every render launch carries its own copy of the camera, and the
    // direction table is rebuilt by a later frame on that frame's stream (see dirs_valid), ordered behind the frames that
    // still read the old one.
    C.width = cam->width; C.height = cam->height;
    C.inv_w = 1.0f / (float)cam->width;
    C.inv_h = 1.0f / (float)cam->height;
    C.ray_dirs = c->ray_dirs.ptr;
    c->dirs_valid = false;
    c->frames_this_camera = 0;
    c->scene_serial += 1u;
    c->has_camera = true;
    return RT_OK;
    RT_GUARD_END(c)
}

int32_t rt_reset_accum(rt_ctx *c) {
    RT_GUARD_BEGIN
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (!c->has_camera) return fail(c, RT_ERR_STATE, "camera not set");
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, wait_external(c, c->stream));
    RT_HIP(c, hipMemsetAsync(c->accum.ptr, 0, (size_t)c->camera.width * c->camera.height * 4 * sizeof(float), c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
    RT_GUARD_END(c)
}

}  // extern "C"

namespace {

// One frame into device buffers.  `noalloc`: the launch's counter rows go to the file-scope `scratch` array instead of a
// block grown with hipMalloc, and launches that would need more than it holds are refused.
// Persistent single-wave workgroups of the path tracing kernel that are resident at once on this device.
int32_t query_wf_groups(rt_ctx *c) {
    if (c->wf_groups != 0u) return RT_OK;
    int per_cu = 0, cus = 0;
    RT_HIP(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rt_wf_diffuse_kernel, 64, 0));
    RT_HIP(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
    if (per_cu <= 0 || cus <= 0) return fail(c, RT_ERR_DEVICE, "occupancy query of the path tracing kernel failed");
    c->wf_groups = (uint32_t)per_cu * (uint32_t)cus;
    RT_HIP(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rt_wf_chain_kernel, 64, 0));
    if (per_cu <= 0) return fail(c, RT_ERR_DEVICE, "occupancy query of the path tracing kernel failed");
    c->wf_chain_groups = (uint32_t)per_cu * (uint32_t)cus;
#ifdef RT_CH_GROUPS_PER_CU   // experiment: fewer resident wavefronts than fit
    c->wf_chain_groups = std::min<uint32_t>(c->wf_chain_groups, (uint32_t)(RT_CH_GROUPS_PER_CU) * (uint32_t)cus);
#endif
    return RT_OK;
}

#ifndef RT_WF_CHAIN_SINGLE
#define RT_WF_CHAIN_SINGLE 0   // 1: single frames of large diffuse scenes also go through the kernel with sample chains (A/B builds)
#endif
// `samples` > 1: that many accumulating frames in one launch (large diffuse scenes only, see rt_render_samples_device).
// `slot` of `slots`: the launch is one of several whose counter rows rt_get_stats sums afterwards (a sample batch rendered as
// consecutive launches); every launch then writes its own block of rows.
int32_t render_impl(rt_ctx *c, const rt_render_params *p, void *d_rgba8, void *d_accum, void *stream, bool noalloc, uint32_t samples = 1u,
                    uint32_t slot = 0u, uint32_t slots = 1u) {
    RT_GUARD_BEGIN
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (!p) return fail(c, RT_ERR_INVALID_ARGUMENT, "params is null");
    if (!c->has_scene || !c->has_camera) return fail(c, RT_ERR_STATE, "scene or camera not set");
    if (p->frame_index == 0) return fail(c, RT_ERR_INVALID_ARGUMENT, "frame_index must be >= 1");
    if (p->flags & ~(uint32_t)RT_RENDER_ALL_FLAGS) return fail(c, RT_ERR_INVALID_ARGUMENT, "unknown bits in params.flags");
    uint32_t rb = p->row_begin, re = p->row_end;
    if (rb == 0 && re == 0) re = c->camera.height;
    if (re > c->camera.height || rb >= re) return fail(c, RT_ERR_INVALID_ARGUMENT, "bad row range");
    const uint32_t stride = p->tile_row_stride ? p->tile_row_stride : 1u;
    if (stride > 1u && (rb % 8u) != 0u) return fail(c, RT_ERR_INVALID_ARGUMENT, "row_begin must be a multiple of 8 when tile_row_stride > 1");
    if (stride > 4096u) return fail(c, RT_ERR_INVALID_ARGUMENT, "tile_row_stride above 4096");
    for (size_t i = 0; i < c->mat_texture.size(); i++) {
        int32_t t = c->mat_texture[i];
        if (t >= 0 && !c->h_textures[t].texels) return fail(c, RT_ERR_STATE, "material " + std::to_string(i) + " uses texture " + std::to_string(t) + " which was never set");
    }
    RT_HIP(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    RT_HIP(c, wait_external(c, s));
    // objects changed by rt_update_objects since the last launch: patched on this stream, behind the frames that read the old ones
    for (size_t i = 0; i < c->pending_objects.size(); i += kPatchObjects) {
        PatchArgs P;
        P.n = (uint32_t)std::min<size_t>(kPatchObjects, c->pending_objects.size() - i);
        for (uint32_t k = 0; k < P.n; k++) { P.index[k] = c->pending_objects[i + k].first; P.object[k] = c->pending_objects[i + k].second; }
        hipLaunchKernelGGL(rt_patch_objects_kernel, dim3(1), dim3(64), 0, s, c->objects.ptr, P);
        RT_HIP(c, hipGetLastError());
    }
    c->pending_objects.clear();

    const DevScene &S = c->scene;
    const bool events = p->collect_events != 0;
    const uint32_t maxd = S.max_ray_bounces;
    // 256-thread workgroups keep max_ray_bounces * 4 KiB of reflective frames in LDS; deeper recursions use one wave per workgroup
    // (hierarchy scenes: traversal lengths differ widely between wavefronts, and single-wave workgroups release their slots one by one)
    const uint32_t threads = S.ray_marching ? kMarchThreads : ((maxd <= 8u && !S.bvh_nodes) ? 256u : 64u);
    const uint32_t waves_per_block = threads / 64u;
    // tile rows of this launch: every stride-th 8-row band of [rb, re)
    const uint32_t tiles_x = (c->camera.width + 7u) / 8u, tiles_y = ((re - rb + 7u) / 8u + stride - 1u) / stride;
    // 2-D grid: y = tile row, x = group of waves_per_block tile columns
    dim3 grid((tiles_x + waves_per_block - 1u) / waves_per_block, tiles_y);
    uint32_t blocks = grid.x * grid.y;
    const uint64_t n_waves = (uint64_t)blocks * waves_per_block;  // diagnostic builds size their table by it
    (void)n_waves;
    // Large diffuse scenes without refractive materials: persistent wavefronts that re-group their rays between bounces
    // (rt_wavefront.h).  The counting variant and the A/B flag keep the one-lane-per-pixel kernel, and so does the no-alloc
    // entry point unless the pools an earlier rt_render_device launch of this shape allocated are there to be used.
    const bool wf_buffers_fit = c->wf_groups != 0u && c->wf_pool.cap >= (size_t)c->wf_groups * kWfPool * 4u &&
                                c->wf_ctrl.cap >= (size_t)kWfCtrlRows + tiles_y;
    const bool regroup = !S.ray_marching && S.bvh2_nodes && S.diffuse && !c->has_refractive && maxd >= 1u && !events &&
                         (!noalloc || wf_buffers_fit) && !(p->flags & RT_RENDER_ONE_LANE_PER_PIXEL);
    // Small diffuse scenes (linear scan) without refractive materials: lanes take the strip's next pixel when their path ends
    const bool refill = !regroup && !S.ray_marching && !S.bvh_nodes && S.diffuse && !c->has_refractive && maxd >= 1u && !events &&
                        !(p->flags & RT_RENDER_ONE_LANE_PER_PIXEL);
    if (refill) {
        grid = dim3((tiles_x + kRfTiles - 1u) / kRfTiles, tiles_y);
        blocks = grid.x * grid.y;
    }
    if (samples > 1u && !regroup) return fail(c, RT_ERR_STATE, "several samples per launch need the re-grouping path tracer");
    if (samples > kChMaxSamples) return fail(c, RT_ERR_UNSUPPORTED, "more than 2^24 samples in one launch");
    // Several samples per launch: the kernel with pixel slots.  (Single frames stay with the other kernel also when their
    // camera rays' hits are known, below: measured at 4K, 1.83 ms in pixel slots against 1.20 ms.)
    const bool chain = regroup && (samples > 1u || RT_WF_CHAIN_SINGLE != 0);
    WfArgs W = {};
    uint64_t reused_here = 0;   // camera rays of this launch that rt_wf_diffuse_kernel reads from WfArgs::first_hit (it does not count them)
    if (regroup && !chain) {
        // camera rays' closest hits from one frame to the next: nothing on the first launch of a (scene, camera, rows), the
        // second writes the table, later ones read it (a camera that moves every frame never pays for the table)
        const uint64_t k0 = ((uint64_t)c->scene_serial << 32) | ((uint64_t)c->camera.width << 16) | (uint64_t)c->camera.height;
        const uint64_t k1 = ((uint64_t)rb << 44) | ((uint64_t)re << 20) | (uint64_t)stride;
        if (k0 != c->first_hit_key[0] || k1 != c->first_hit_key[1]) {
            c->first_hit_key[0] = k0; c->first_hit_key[1] = k1; c->first_hit_launches = 0u; c->first_hit_filled = false;
        }
        const size_t px = (size_t)c->camera.width * c->camera.height;
        const bool wanted = !(p->flags & RT_RENDER_NO_FIRST_HIT_REUSE) && c->first_hit_launches >= 1u;
        if (wanted && (noalloc ? c->first_hit.cap >= px : c->first_hit.ensure(px) == hipSuccess)) {
            W.first_hit = c->first_hit.ptr;
            W.first_hit_mode = c->first_hit_filled ? kFirstHitRead : kFirstHitWrite;   // (marked as filled once the launch is under way, below)
        }
        if (c->first_hit_launches < 0xFFFFFFFFu) c->first_hit_launches += 1u;
    }
    if (chain) {
        { const int32_t r = query_wf_groups(c); if (r != RT_OK) return r; }
        W.tiles_x = (c->camera.width + kChTileW - 1u) / kChTileW; W.tiles_y = tiles_y;
        W.samples = samples;
        // one wavefront per pool of pixel slots the launch can fill, at most as many as are resident at once
        const uint64_t pools = ((uint64_t)W.tiles_x * W.tiles_y * kChTilePixels + kChPool - 1u) / kChPool;
        grid = dim3((uint32_t)std::min<uint64_t>(c->wf_chain_groups, pools));
        blocks = grid.x;
        const size_t plane = (size_t)c->wf_chain_groups * kChPool;
        // per pixel slot: a, b, c, e, k0, k1, k2 (16 bytes each), h, g (8 bytes each), d1 (4 bytes): 132 bytes, 8.25 planes of 16
        RT_HIP(c, c->wf_pool.ensure(plane * 9u));
        RT_HIP(c, c->wf_ctrl.ensure((size_t)kWfCtrlRows + tiles_y));
        c->wf_order_key[0] = c->wf_order_key[1] = ~0ull;   // the other kernel's per-row history in this buffer is gone
        RT_HIP(c, hipMemsetAsync(c->wf_ctrl.ptr, 0, sizeof(uint32_t), s));   // the work counter
        W.a = c->wf_pool.ptr; W.b = W.a + plane; W.c = W.b + plane; W.e = W.c + plane;
        W.k0 = W.e + plane; W.k1 = W.k0 + plane; W.k2 = W.k1 + plane;
        W.h = reinterpret_cast<float2 *>(W.k2 + plane);
        W.g = reinterpret_cast<uint2 *>(W.h + plane);
        W.d1 = reinterpret_cast<float *>(W.g + plane);
        W.ctrl = c->wf_ctrl.ptr;
    } else if (regroup) {
        { const int32_t r = query_wf_groups(c); if (r != RT_OK) return r; }
        W.tiles_x = (c->camera.width + kWfTileW - 1u) / kWfTileW; W.tiles_y = tiles_y;
        grid = dim3(std::min<uint32_t>(c->wf_groups, W.tiles_x * W.tiles_y * 8u));   // at most eight items per tile
        blocks = grid.x;
        DevBuf<float4> &pool = c->wf_pool;
        DevBuf<uint32_t> &ctrl = c->wf_ctrl;
        uint64_t *order_key = c->wf_order_key;
        RT_HIP(c, pool.ensure((size_t)c->wf_groups * kWfPool * 4u));   // four float4 planes per pool entry
        {
            const uint32_t *before = ctrl.ptr;
            RT_HIP(c, ctrl.ensure((size_t)kWfCtrlRows + tiles_y));
            if (ctrl.ptr != before) order_key[0] = order_key[1] = ~0ull;   // a new block holds no history
        }
        {   // the per-row costs only describe a launch over the same rows of the same frame (non-overlapping fields)
            const uint64_t k0 = ((uint64_t)c->scene_serial << 32) | ((uint64_t)c->camera.width << 16) | (uint64_t)c->camera.height;
            const uint64_t k1 = ((uint64_t)rb << 44) | ((uint64_t)re << 20) | (uint64_t)stride;   // stride <= 4096
            if (k0 != order_key[0] || k1 != order_key[1]) {
                RT_HIP(c, hipMemsetAsync(ctrl.ptr, 0, ((size_t)kWfCtrlRows + tiles_y) * sizeof(uint32_t), s));
                order_key[0] = k0; order_key[1] = k1;
            }
        }
        const size_t plane = (size_t)c->wf_groups * kWfPool;
        W.a = pool.ptr; W.b = W.a + plane; W.c = W.b + plane; W.d = W.c + plane;
        W.ctrl = ctrl.ptr;
        if (W.first_hit_mode == kFirstHitRead) {   // every pixel of the launch has one camera ray that this kernel does not count
            uint64_t rows_in = 0;
            for (uint32_t y0 = rb; y0 < re; y0 += 8u * stride) rows_in += std::min<uint32_t>(8u, re - y0);
            reused_here = rows_in * c->camera.width;
        }
    }
    const uint32_t ncnt = (events ? (uint32_t)RT_CNT_COUNT : 0u) + 1u;
    uint32_t *counters = nullptr;
    if (noalloc) {
        if ((uint64_t)blocks * ncnt > kNoAllocCounterWords)
            return fail(c, RT_ERR_UNSUPPORTED, "rt_render_device_noalloc: the launch needs more than 2^20 counter words (workgroups x counters per workgroup)");
        counters = c->static_counters;
    } else {
        // (the first launch of a batch sizes the table; the kernel with sample chains writes a second half, see rt_get_stats)
        if (slot == 0u) RT_HIP(c, c->counters.ensure((size_t)blocks * ncnt * slots * (chain ? 2u : 1u)));
        counters = c->counters.ptr + (size_t)slot * blocks * ncnt;
    }

    // primary directions: computed in the kernel by the first frame after a camera change (trace kernels on the linear scan
    // and the march kernel can), read from the table otherwise
    DevCamera cam = c->camera;
    if (!c->dirs_valid) {
        const bool can_inline = !regroup && !refill && !S.bvh_nodes;
        if (can_inline && c->frames_this_camera == 0u) cam.ray_dirs = nullptr;
        else RT_HIP(c, fill_ray_directions(c, s));
    }
    c->frames_this_camera += 1u;

    DevFrame F;
    F.frame_index = p->frame_index; F.row_begin = rb; F.row_end = re; F.tile_row_step = 8u * stride; F.leaf_slots = 0; F.accumulate = p->accumulate; F.seed = p->seed;
    F.march_skip = 0u; F.march_pair = 0u;
    for (int k = 0; k < 8; k++) { F.march_lip[k] = 0.0f; F.march_slot[k] = 0u; F.march_lip2[k] = 0.0f; F.march_slot2[k] = 0u; }
    F.skip_accum_store = (!p->accumulate && (p->flags & RT_RENDER_SKIP_ACCUM_STORE)) ? 1 : 0;
    F.rgba = d_rgba8 ? (uint32_t *)d_rgba8 : c->rgba.ptr;
    F.accum = d_accum ? (float *)d_accum : c->accum.ptr;
    F.counters = counters;
    F.diag = nullptr;
    F.tile_sky = nullptr;
    // Sky tiles for the linear-scan trace kernels: classified once per (scene, camera, rows of the launch), on the launch's
    // stream ahead of the first render that needs them; every later launch over the same rows reuses the words.
    // (the no-alloc entry point: only with the words and the event of an earlier launch of this shape in place)
    const bool sky_tiles = !regroup && !refill && !S.ray_marching && !S.bvh_nodes && !c->has_refractive && !events && maxd >= 1u &&
                           (!noalloc || (c->tile_sky.cap >= blocks && c->tile_sky_done != nullptr)) && S.n_objects >= 1u &&
                           !(p->flags & RT_RENDER_NO_SKY_TILES);
    if (sky_tiles) {
        const uint64_t k0 = ((uint64_t)c->scene_serial << 32) | ((uint64_t)c->camera.width << 16) | (uint64_t)c->camera.height;
        const uint64_t k1 = ((uint64_t)rb << 44) | ((uint64_t)re << 20) | ((uint64_t)stride << 4) | (uint64_t)waves_per_block;   // stride <= 4096: 13 bits
        if (k0 != c->tile_sky_key[0] || k1 != c->tile_sky_key[1] || c->tile_sky_words != blocks) {
            RT_HIP(c, c->tile_sky.ensure(blocks));
            hipLaunchKernelGGL(rt_tile_sky_kernel, dim3((4u * blocks + 255u) / 256u), dim3(256), 0, s, S, cam, F, grid.x, grid.y, waves_per_block,
                               reinterpret_cast<uint8_t *>(c->tile_sky.ptr));
            RT_HIP(c, hipGetLastError());
            if (!c->tile_sky_done) RT_HIP(c, hipEventCreateWithFlags(&c->tile_sky_done, hipEventDisableTiming));
            RT_HIP(c, hipEventRecord(c->tile_sky_done, s));
            c->tile_sky_key[0] = k0; c->tile_sky_key[1] = k1; c->tile_sky_words = blocks; c->tile_sky_waves = waves_per_block;
        }
        F.tile_sky = c->tile_sky.ptr;
    }
#ifdef RT_DIAG
    static unsigned long long *s_diag = nullptr;
    static size_t s_diag_cap = 0;
    const size_t diag_n = (size_t)n_waves * DG_N;
    if (diag_n > s_diag_cap) { if (s_diag) (void)hipFree(s_diag); (void)hipMalloc((void **)&s_diag, diag_n * sizeof(unsigned long long)); s_diag_cap = diag_n; }
    (void)hipMemsetAsync(s_diag, 0, diag_n * sizeof(unsigned long long), s);
    F.diag = s_diag;
#endif

    // the two timing events cost about 6 us per launch on the stream; callers that time their own stream opt out
    const bool time_kernel = p->no_timing == 0 && slots == 1u;
    if (time_kernel) RT_HIP(c, hipEventRecord(c->ev0, s));
    if (chain) {
#ifdef RT_WF_STATS
        static unsigned long long *s_chstats = nullptr;
        if (!s_chstats) (void)hipMalloc((void **)&s_chstats, 2 * WS_N * sizeof(unsigned long long));
        (void)hipMemsetAsync(s_chstats, 0, 2 * WS_N * sizeof(unsigned long long), s);
        W.stats = s_chstats;
#endif
        hipLaunchKernelGGL(rt_wf_chain_kernel, grid, dim3(64), 0, s, S, cam, F, W);
#ifdef RT_WF_STATS
        if (getenv("RT_WF_STATS_PRINT")) {
            unsigned long long h[2 * WS_N];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(h, s_chstats, sizeof(h), hipMemcpyDeviceToHost);
            static const char *names[WS_N] = {"refill", "walk step", "leaf test", "finish", "shade batch", "prepare batch"};
            for (int i = 0; i < WS_N; i++) fprintf(stderr, "WFSTAT %-14s runs %12llu lanes %14llu  mean lanes %5.1f\n", names[i], h[2 * i], h[2 * i + 1], h[2 * i] ? (double)h[2 * i + 1] / (double)h[2 * i] : 0.0);
        }
#endif
    } else if (regroup) {
        hipLaunchKernelGGL(rt_wf_order_kernel, dim3(1), dim3(kWfOrderThreads), 0, s, W.ctrl, W.tiles_x, W.tiles_y, grid.x);
#ifdef RT_WF_STATS
        static unsigned long long *s_wfstats = nullptr;
        if (!s_wfstats) (void)hipMalloc((void **)&s_wfstats, 2 * WS_N * sizeof(unsigned long long));
        (void)hipMemsetAsync(s_wfstats, 0, 2 * WS_N * sizeof(unsigned long long), s);
        W.stats = s_wfstats;
#endif
        hipLaunchKernelGGL(rt_wf_diffuse_kernel, grid, dim3(64), 0, s, S, cam, F, W);
#ifdef RT_WF_STATS
        if (getenv("RT_WF_STATS_PRINT")) {
            unsigned long long h[2 * WS_N];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(h, s_wfstats, sizeof(h), hipMemcpyDeviceToHost);
            static const char *names[WS_N] = {"refill", "walk step", "leaf test", "finish", "shade batch", "prepare batch"};
            {   // start / end of every wavefront: how much of the launch is a tail of few busy wavefronts
                std::vector<float4> all((size_t)grid.x * kWfPool);
                (void)hipMemcpy(all.data(), W.a, all.size() * sizeof(float4), hipMemcpyDeviceToHost);
                std::vector<double> t0(grid.x), t1(grid.x);
                auto u = [](float lo, float hi) { uint32_t a, b; std::memcpy(&a, &lo, 4); std::memcpy(&b, &hi, 4); return (double)(((unsigned long long)b << 32) | a); };
                double first = 1e300, last = 0.0, busy = 0.0;
                for (uint32_t g = 0; g < grid.x; g++) { const float4 v = all[(size_t)g * kWfPool]; t0[g] = u(v.x, v.y); t1[g] = u(v.z, v.w); first = std::min(first, t0[g]); last = std::max(last, t1[g]); busy += t1[g] - t0[g]; }
                const double span = last - first;
                std::vector<double> ends(t1); std::sort(ends.begin(), ends.end());
                fprintf(stderr, "WFSTAT span %.1f us, mean wavefront busy %.1f %% of it; wavefronts still running at 70/80/90/95 %% of the span: ", span / 100.0, 100.0 * busy / (span * grid.x));
                for (double f : {0.7, 0.8, 0.9, 0.95}) { const double t = first + f * span; fprintf(stderr, "%ld ", (long)(ends.end() - std::lower_bound(ends.begin(), ends.end(), t))); }
                fprintf(stderr, "of %u\n", grid.x);
                std::vector<double> starts(t0); std::sort(starts.begin(), starts.end());
                fprintf(stderr, "WFSTAT wavefronts started within 20 us of the first: %ld of %u\n", (long)(std::upper_bound(starts.begin(), starts.end(), first + 2000.0) - starts.begin()), grid.x);
                fprintf(stderr, "WFSTAT starts after the first one (us): median %.1f, 90 %% %.1f, 99 %% %.1f, last %.1f; ends before the last one (us): median %.1f, 10 %% %.1f, 1 %% %.1f\n",
                        (starts[grid.x / 2] - first) / 100.0, (starts[(size_t)(grid.x * 0.9)] - first) / 100.0, (starts[(size_t)(grid.x * 0.99)] - first) / 100.0, (starts.back() - first) / 100.0,
                        (last - ends[grid.x / 2]) / 100.0, (last - ends[(size_t)(grid.x * 0.1)]) / 100.0, (last - ends[(size_t)(grid.x * 0.01)]) / 100.0);
            }
            for (int i = 0; i < WS_N; i++) fprintf(stderr, "WFSTAT %-14s runs %12llu lanes %14llu  mean lanes %5.1f\n", names[i], h[2 * i], h[2 * i + 1], h[2 * i] ? (double)h[2 * i + 1] / (double)h[2 * i] : 0.0);
        }
#endif
    } else if (refill) {
        if (c->spheres_planes_only) launch(rt_refill_diffuse_kernel<true>, grid, 64u, 0, s, S, cam, F);
        else launch(rt_refill_diffuse_kernel<false>, grid, 64u, 0, s, S, cam, F);
    } else if (S.ray_marching) {
        // up to 16 leaves (12.5 KiB of LDS per 64-thread workgroup) keep their object-space rays in LDS during the march loop
        const bool cache = c->leaf_slots >= 1u && c->leaf_slots <= (uint32_t)RT_MARCH_CACHE_MAX_SLOTS;
        F.leaf_slots = cache ? c->leaf_slots : 0u;
        size_t lds = cache ? (size_t)c->leaf_slots * 3u * (threads + 1u) * sizeof(float) : 0u;
        // march steps that leave out the entries of Scene.sdfs which cannot hold the minimum (rm_sdfs_march): 2 to 8 entries
        // with known bounds, timed variant only (the counting variant evaluates what the reference evaluates)
        bool skip = !events && S.n_sdfs >= 2u && S.n_sdfs <= 8u && !(p->flags & RT_RENDER_MARCH_ALL_ENTRIES);
        if (skip) {
            uint32_t slot = 0u;
            for (uint32_t k = 0; k < S.n_sdfs && skip; k++) {
                const uint32_t idx = c->h_sdfs[k];
                const double rate = march_fall_rate(c->h_objects, idx, 2) * 1.001;
                if (!(rate < 1e30)) skip = false;
                F.march_lip[k] = (float)rate;
                if ((double)F.march_lip[k] < rate) F.march_lip[k] = std::nextafter(F.march_lip[k], HUGE_VALF);
                F.march_slot[k] = slot;
                const int kind = c->h_objects[idx].kind;
                if (kind == RT_UNION || kind == RT_SUBSTRACTION) {   // a pair of leaves: the second child gets a bound of its own
                    const rt_object &pair = c->h_objects[idx];
                    auto leaf = [&](int32_t i) { const int kk = c->h_objects[(size_t)i].kind; return kk == RT_SPHERE || kk == RT_CUBOID || kk == RT_CYLINDER || kk == RT_PLANE; };
                    if (leaf(pair.first) && leaf(pair.second)) {
                        const double r2 = march_fall_rate(c->h_objects, (uint32_t)pair.second, 0) * 1.001;
                        F.march_lip2[k] = (float)r2;
                        if ((double)F.march_lip2[k] < r2) F.march_lip2[k] = std::nextafter(F.march_lip2[k], HUGE_VALF);
                        F.march_slot2[k] = slot + ((cache && c->h_objects[(size_t)pair.first].kind != RT_PLANE) ? 1u : 0u);
                        F.march_pair |= 1u << k;
                    }
                }
                if (cache && kind != RT_TRIANGLE && kind != RT_CONE) {
                    rt_scene_desc d1;
                    std::memset(&d1, 0, sizeof(d1));
                    d1.objects = c->h_objects.data(); d1.n_objects = (uint32_t)c->h_objects.size();
                    slot += march_leaf_slots(&d1, idx, 2);
                }
            }
        }
        F.march_skip = skip ? 1u : 0u;
        if (skip) {
            lds += (size_t)2u * S.n_sdfs * threads * sizeof(float);
            if (cache) launch(rt_march_kernel<false, true, true>, grid, threads, lds, s, S, cam, F);
            else launch(rt_march_kernel<false, false, true>, grid, threads, lds, s, S, cam, F);
        } else if (cache) {
            if (events) launch(rt_march_kernel<true, true>, grid, threads, lds, s, S, cam, F);
            else launch(rt_march_kernel<false, true>, grid, threads, lds, s, S, cam, F);
        } else {
            if (events) launch(rt_march_kernel<true, false>, grid, threads, 0, s, S, cam, F);
            else launch(rt_march_kernel<false, false>, grid, threads, 0, s, S, cam, F);
        }
    } else {
        const size_t lds = (size_t)(maxd ? maxd : 1u) * 4u * threads * sizeof(float);
        // kernel variant: bit 0 events, bits 1-2 refractive frames (0 none, 1 up to 8, 2 up to 32), bit 3 hierarchy,
        // bit 4 spheres and planes only (timed variants without hierarchy)
#ifdef RT_FORCE_REFR   // experiment: the variant with refractive frames on scenes that have no refractive material
        const int rsel = maxd <= 8u ? 1 : 2;
#else
        const int rsel = !c->has_refractive ? 0 : (maxd <= 8u ? 1 : 2);
#endif
        const bool sp = c->spheres_planes_only && !events && !S.bvh_nodes;
        const int variant = (events ? 1 : 0) | (rsel << 1) | (S.bvh_nodes ? 8 : 0) | (sp ? 16 : 0);
        switch (variant) {
#define RT_CASE(V, E, R, D, B) case V: launch(rt_trace_kernel<E, R, D, B>, grid, threads, lds, s, S, cam, F); break;
            case 16: launch(rt_trace_kernel<false, false, 1, false, true>, grid, threads, lds, s, S, cam, F); break;
            case 18: launch(rt_trace_kernel<false, true, 8, false, true>, grid, threads, lds, s, S, cam, F); break;
            case 20: launch(rt_trace_kernel<false, true, 32, false, true>, grid, threads, lds, s, S, cam, F); break;
            RT_CASE(0, false, false, 1, false) RT_CASE(1, true, false, 1, false)
            RT_CASE(2, false, true, 8, false) RT_CASE(3, true, true, 8, false)
            RT_CASE(4, false, true, 32, false) RT_CASE(5, true, true, 32, false)
            RT_CASE(8, false, false, 1, true) RT_CASE(9, true, false, 1, true)
            RT_CASE(10, false, true, 8, true) RT_CASE(11, true, true, 8, true)
            RT_CASE(12, false, true, 32, true) RT_CASE(13, true, true, 32, true)
#undef RT_CASE
            default: return fail(c, RT_ERR_STATE, "no kernel variant");
        }
    }
    RT_HIP(c, hipGetLastError());
    if (W.first_hit_mode == kFirstHitWrite) c->first_hit_filled = true;
    if (time_kernel) RT_HIP(c, hipEventRecord(c->ev1, s));
    c->kernel_timed = time_kernel;
    c->last_blocks = blocks * (slot + 1u); c->last_ncnt = ncnt; c->last_events = events; c->timed = true; c->last_stream = s;
    c->last_counters = noalloc ? counters : c->counters.ptr;
    c->last_reused_rows = chain ? blocks : 0u;
    c->host_counted_rays = (slot == 0u ? 0u : c->host_counted_rays) + reused_here;
#ifdef RT_DIAG
    if (getenv("RT_DIAG_PRINT")) {
        std::vector<unsigned long long> all((size_t)n_waves * DG_N);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(all.data(), F.diag, all.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long h[DG_N] = {0};
        for (size_t w = 0; w < (size_t)n_waves; w++) for (int i = 0; i < DG_N; i++) h[i] += all[w * DG_N + i];
        static const char *names[DG_N] = {"total", "raygen", "trace", "finalize", "lights(all)", "  shadow_trace", "  pow(blinn+gamma)", "  blinn(incl its pow)", "reflection_ray", "unwind", "store", "texture"};
        for (int i = 0; i < DG_N; i++) fprintf(stderr, "DIAG %-24s %14llu wave-cycles %6.2f%%  (%.0f per wave)\n", names[i], h[i], 100.0 * (double)h[i] / (double)(h[0] ? h[0] : 1), (double)h[i] / (double)n_waves);
    }
#endif
    c->frames += samples;
    return RT_OK;
    RT_GUARD_END(c)
}

}  // namespace

extern "C" {

int32_t rt_render_device(rt_ctx *c, const rt_render_params *p, void *d_rgba8, void *d_accum, void *stream) {
    return render_impl(c, p, d_rgba8, d_accum, stream, false);
}

int32_t rt_render_device_noalloc(rt_ctx *c, const rt_render_params *p, void *d_rgba8, void *d_accum, void *stream) {
    return render_impl(c, p, d_rgba8, d_accum, stream, true);
}

// sample_count accumulating frames, frame_index = p->frame_index ... + sample_count - 1: the same pixels as that many
// rt_render_device calls.  Scenes that go through the re-grouping path tracer are rendered by ONE launch in which every
// pixel's samples are traced one after the other by the wavefront that holds the pixel (rt_wf_chain_kernel): the accumulator
// is read and written once per call instead of once per sample.  Other scenes: one launch per sample on the caller's stream.
int32_t rt_render_samples_device(rt_ctx *c, const rt_render_params *p, uint32_t sample_count, void *d_rgba8, void *d_accum, void *stream) {
    RT_GUARD_BEGIN
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (!p) return fail(c, RT_ERR_INVALID_ARGUMENT, "params is null");
    if (sample_count == 0) return fail(c, RT_ERR_INVALID_ARGUMENT, "sample_count must be >= 1");
    if (!p->accumulate) return fail(c, RT_ERR_INVALID_ARGUMENT, "rt_render_samples_device accumulates: params.accumulate must be set");
    if (p->collect_events) return fail(c, RT_ERR_INVALID_ARGUMENT, "rt_render_samples_device does not count events");
    if (p->frame_index == 0 || p->frame_index > 0xFFFFFFFFu - sample_count) return fail(c, RT_ERR_INVALID_ARGUMENT, "bad frame_index range");
    if (!c->has_scene || !c->has_camera) return fail(c, RT_ERR_STATE, "scene or camera not set");
    const DevScene &S = c->scene;
    const bool regroup = !S.ray_marching && S.bvh2_nodes && S.diffuse && !c->has_refractive && S.max_ray_bounces >= 1u &&
                         !(p->flags & RT_RENDER_ONE_LANE_PER_PIXEL);
    if (regroup) {
        rt_render_params q = *p;
        for (uint32_t done = 0; done < sample_count; done += kChMaxSamples) {   // (one launch unless the batch is longer than 2^24 samples)
            q.frame_index = p->frame_index + done;
            const int32_t r = render_impl(c, &q, d_rgba8, d_accum, stream, false, std::min(sample_count - done, kChMaxSamples));
            if (r != RT_OK) return r;
        }
        return RT_OK;
    }
    // one launch after the other on the caller's stream, each with its own block of counter rows
    rt_render_params q = *p;
    q.no_timing = 1;
    for (uint32_t i = 0; i < sample_count; i++) {
        q.frame_index = p->frame_index + i;
        const int32_t r = render_impl(c, &q, d_rgba8, d_accum, stream, false, 1u, i, sample_count);
        if (r != RT_OK) return r;
    }
    return RT_OK;
    RT_GUARD_END(c)
}

int32_t rt_resolve_device(rt_ctx *c, uint32_t sample_count, uint32_t row_begin, uint32_t row_end, void *d_accum, void *d_rgba8, void *stream) {
    RT_GUARD_BEGIN
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (!c->has_camera) return fail(c, RT_ERR_STATE, "camera not set");
    if (sample_count == 0) return fail(c, RT_ERR_INVALID_ARGUMENT, "sample_count must be >= 1");
    if (row_begin == 0 && row_end == 0) row_end = c->camera.height;
    if (row_end > c->camera.height || row_begin >= row_end) return fail(c, RT_ERR_INVALID_ARGUMENT, "bad row range");
    RT_HIP(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    RT_HIP(c, wait_external(c, s));
    const uint32_t first = row_begin * c->camera.width, count = (row_end - row_begin) * c->camera.width;
    const uint32_t blocks = std::min<uint32_t>((count + 255u) / 256u, 256u * 8u);
    hipLaunchKernelGGL(rt_resolve_kernel, dim3(blocks), dim3(256), 0, s, (const float4 *)(d_accum ? d_accum : (void *)c->accum.ptr),
                       (uint32_t *)(d_rgba8 ? d_rgba8 : (void *)c->rgba.ptr), first, count, (float)sample_count);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
    RT_GUARD_END(c)
}

int32_t rt_render(rt_ctx *c, uint32_t frame_index, int32_t accumulate, int32_t reset_accum, uint64_t seed, uint8_t *out, size_t out_len) {
    RT_GUARD_BEGIN
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (!c->has_scene || !c->has_camera) return fail(c, RT_ERR_STATE, "scene or camera not set");
    const size_t need = (size_t)c->camera.width * c->camera.height * 4;
    if (!out || out_len != need) return fail(c, RT_ERR_INVALID_ARGUMENT, "out_rgba8 must hold width*height*4 bytes");
    if (reset_accum) { int32_t r = rt_reset_accum(c); if (r != RT_OK) return r; }
    rt_render_params p;
    std::memset(&p, 0, sizeof(p));
    p.frame_index = frame_index; p.accumulate = accumulate; p.seed = seed;
    int32_t r = rt_render_device(c, &p, nullptr, nullptr, c->stream);
    if (r != RT_OK) return r;
    RT_HIP(c, hipEventRecord(c->evc0, c->stream));
    RT_HIP(c, hipMemcpyAsync(out, c->rgba.ptr, need, hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipEventRecord(c->evc1, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, c->evc0, c->evc1) == hipSuccess) c->last_copy_ms = ms;
    return RT_OK;
    RT_GUARD_END(c)
}

int32_t rt_alloc_host_frame(rt_ctx *c, size_t len, void **out_ptr) {
    RT_GUARD_BEGIN
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (!out_ptr || len == 0) return fail(c, RT_ERR_INVALID_ARGUMENT, "out_ptr is null or len is 0");
    *out_ptr = nullptr;
    RT_HIP(c, hipSetDevice(c->device));
    void *p = nullptr;
    RT_HIP(c, hipHostMalloc(&p, len, hipHostMallocDefault));
    c->pinned.emplace_back(p, len);
    *out_ptr = p;
    return RT_OK;
    RT_GUARD_END(c)
}

int32_t rt_free_host_frame(rt_ctx *c, void *ptr) {
    RT_GUARD_BEGIN
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < c->pinned.size(); i++) {
        if (c->pinned[i].first != ptr) continue;
        RT_HIP(c, hipSetDevice(c->device));
        RT_HIP(c, hipStreamSynchronize(c->stream));
        RT_HIP(c, hipHostFree(ptr));
        c->pinned.erase(c->pinned.begin() + (long)i);
        return RT_OK;
    }
    return fail(c, RT_ERR_INVALID_ARGUMENT, "pointer was not returned by rt_alloc_host_frame of this context");
    RT_GUARD_END(c)
}

int32_t rt_read_accum(rt_ctx *c, float *out, size_t n_vec4) {
    RT_GUARD_BEGIN
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (!c->has_camera) return fail(c, RT_ERR_STATE, "camera not set");
    if (!out || n_vec4 != (size_t)c->camera.width * c->camera.height) return fail(c, RT_ERR_INVALID_ARGUMENT, "out must hold width*height float4");
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, wait_external(c, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    RT_HIP(c, hipMemcpy(out, c->accum.ptr, n_vec4 * 4 * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
    RT_GUARD_END(c)
}

int32_t rt_read_ray_directions(rt_ctx *c, float *out, size_t n_vec3) {
    RT_GUARD_BEGIN
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (!c->has_camera) return fail(c, RT_ERR_STATE, "camera not set");
    if (!out || n_vec3 != (size_t)c->camera.width * c->camera.height) return fail(c, RT_ERR_INVALID_ARGUMENT, "out must hold width*height directions");
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, wait_external(c, c->stream));
    if (!c->dirs_valid) RT_HIP(c, fill_ray_directions(c, c->stream));   // no frame has needed the table since the camera changed
    RT_HIP(c, hipStreamSynchronize(c->stream));
    RT_HIP(c, hipMemcpy(out, c->ray_dirs.ptr, n_vec3 * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
    RT_GUARD_END(c)
}

// One lane per element; the same inlined rt_powf / rt_atanf / rt_sqrt as the render kernels, tables staged in LDS the same way.
__global__ void __launch_bounds__(256) rt_math_kernel(int op, size_t n, const float *x, const float *y, float *out) {
    __shared__ __attribute__((aligned(16))) double s_pow[RT_POW_LDS_DOUBLES];
    rt_pow_lds_init(s_pow);
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = op == RT_MATH_POWF ? rt_powf(x[i], y[i], s_pow) : (op == RT_MATH_ATANF ? rt_atanf(x[i]) : rt_sqrt(x[i]));
}

int32_t rt_eval_math(rt_ctx *c, int32_t op, size_t n, const float *x, const float *y, float *out) {
    RT_GUARD_BEGIN
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (op != RT_MATH_POWF && op != RT_MATH_ATANF && op != RT_MATH_SQRT) return fail(c, RT_ERR_INVALID_ARGUMENT, "unknown math op");
    if (n == 0) return RT_OK;
    if (!x || !out || (op == RT_MATH_POWF && !y)) return fail(c, RT_ERR_INVALID_ARGUMENT, "null operand array");
    RT_HIP(c, hipSetDevice(c->device));
    float *d = nullptr;   // x | y | out
    RT_HIP(c, hipMalloc(&d, 3 * n * sizeof(float)));
    hipError_t e = hipMemcpy(d, x, n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && op == RT_MATH_POWF) e = hipMemcpy(d + n, y, n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const size_t want = (n + 255) / 256;
        const unsigned blocks = (unsigned)(want < 65536 ? want : 65536);
        hipLaunchKernelGGL(rt_math_kernel, dim3(blocks), dim3(256), 0, c->stream, (int)op, n, (const float *)d, (const float *)(d + n), d + 2 * n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out, d + 2 * n, n * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    RT_HIP(c, e);
    return RT_OK;
    RT_GUARD_END(c)
}

int32_t rt_get_stats(rt_ctx *c, rt_stats *out) {
    RT_GUARD_BEGIN
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (!out) return fail(c, RT_ERR_INVALID_ARGUMENT, "out is null");
    std::memset(out, 0, sizeof(*out));
    RT_HIP(c, hipSetDevice(c->device));
    if (c->timed) {
        RT_HIP(c, hipStreamSynchronize(c->last_stream));
        float ms = 0.0f;
        if (c->kernel_timed && hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) out->last_kernel_ms = ms;
        const size_t n = (size_t)c->last_blocks * c->last_ncnt;
        c->h_counters.resize(n + c->last_reused_rows);
        RT_HIP(c, hipMemcpy(c->h_counters.data(), c->last_counters, (n + c->last_reused_rows) * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (uint32_t b = 0; b < c->last_reused_rows; b++) out->rays_reused_camera += c->h_counters[n + b];
        out->rays_reused_camera += c->host_counted_rays;
        out->rays_total += c->host_counted_rays;
        for (uint32_t b = 0; b < c->last_blocks; b++) {
            const uint32_t *row = c->h_counters.data() + (size_t)b * c->last_ncnt;
            out->rays_total += row[0] & 0x7FFFFFFFu;
            if (row[0] & 0x80000000u) out->rays_sky_tiles += row[0] & 0x7FFFFFFFu;   // a workgroup of nothing but sky tiles
            if (c->last_events) for (uint32_t i = 0; i < (uint32_t)RT_CNT_COUNT; i++) out->counters[i] += row[1 + i];
        }
    }
    out->last_copy_ms = c->last_copy_ms;
    out->frames_rendered = c->frames;
    return RT_OK;
    RT_GUARD_END(c)
}

int32_t rt_get_hierarchy_info(rt_ctx *c, rt_hierarchy_info *out) {
    RT_GUARD_BEGIN
    if (!c) return RT_ERR_INVALID_ARGUMENT;
    if (!out) return fail(c, RT_ERR_INVALID_ARGUMENT, "out is null");
    if (!c->has_scene) return fail(c, RT_ERR_STATE, "scene not set");
    std::memset(out, 0, sizeof(*out));
    const DevScene &S = c->scene;
    out->stackless_nodes = S.bvh_nodes ? S.n_bvh_nodes : 0u;
    out->stackless_linear = S.bvh_nodes ? S.n_unbounded : 0u;
    out->two_child_nodes = S.bvh2_nodes ? (uint32_t)c->n_bvh2_nodes : 0u;
    out->two_child_linear = S.bvh2_nodes ? S.n_bvh2_linear : 0u;
    out->regrouping_kernel = (!S.ray_marching && S.bvh2_nodes && S.diffuse && !c->has_refractive && S.max_ray_bounces >= 1u) ? 1u : 0u;
    if (c->tile_sky_words != 0u && c->tile_sky_key[0] != ~0ull) {
        RT_HIP(c, hipSetDevice(c->device));
        RT_HIP(c, hipEventSynchronize(c->tile_sky_done));
        std::vector<uint32_t> words(c->tile_sky_words);
        RT_HIP(c, hipMemcpy(words.data(), c->tile_sky.ptr, words.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        const uint32_t all = 0x01010101u >> (8u * (4u - c->tile_sky_waves));
        for (uint32_t w : words) out->sky_workgroups += (w == all) ? 1u : 0u;
        out->classified_workgroups = c->tile_sky_words;
    }
    return RT_OK;
    RT_GUARD_END(c)
}

void *rt_context_stream(rt_ctx *c) { return c ? (void *)c->stream : nullptr; }
void *rt_device_rgba8(rt_ctx *c) { return c ? (void *)c->rgba.ptr : nullptr; }
void *rt_device_accum(rt_ctx *c) { return c ? (void *)c->accum.ptr : nullptr; }

}  // extern "C"
