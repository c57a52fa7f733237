// rt_oracle.cpp — CPU parity oracle: TEST INFRASTRUCTURE, never part of the shipped path.
//
// A file-by-file restatement of the reference's per-pixel hot path in scalar C++ (f32, no FMA
// contraction, Rust evaluation order), each function citing the reference lines it follows.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load it.
//
// PARITY UNPINNED: the reference cannot be built or run here and holds no golden vectors; see
// orc_math.h for how the third-party arithmetic (glam, rand, libm) is restated.
//
// Deliberate substitution: rand's ThreadRng (OS-seeded, src/renderer.rs:112) is replaced by a
// counter-based generator keyed by (seed, pixel, frame, path node, draw) so that the random
// modes (Scene.diffuse, Scene.enable_accumulation) are reproducible on both sides.
#include "rt_oracle.h"

#include <atomic>
#include <thread>
#include <vector>

#include "orc_math.h"

namespace orc {

// ------------------------------------------------------------------------------------------------
// data types (src/ray.rs:11-55)
// ------------------------------------------------------------------------------------------------
struct Ray { Vec3 origin, direction; };
struct RayHit { float distance; Vec3 point, normal; uint32_t material_index; float u, v; };
struct RayMarchingHit { float distance; Vec3 albedo; Ray transformed_ray; };

struct Obj {
    int kind; uint32_t material_index;
    Mat4 transform, inv_transform;
    float radius, height;
    Vec3 dimension, normal, point; Vec2 max_dist; bool has_max_dist;
    Vec3 v1, v2, v3;
    uint32_t first, second;
};
struct Mat {
    float ambience, diffuse, specular, shininess; Vec3 albedo; int texture; int kind;
    float roughness, transparency, refraction_index, reflectivity, emission_power;
};
struct Lgt { int kind; Vec3 albedo, vec; float intensity; };

struct Counters { uint64_t c[RT_CNT_COUNT]; };

struct Scene {
    std::vector<Obj> objects; std::vector<Mat> materials; std::vector<Lgt> lights; std::vector<uint32_t> sdfs;
    const orc_texture *textures; uint32_t n_textures;
    Vec3 ambient_color; uint8_t max_ray_bounces; bool shadow_casting, ray_marching, diffuse, enable_accumulation;
};

static Vec3 ld3(const float *p) { return v3(p[0], p[1], p[2]); }

static Obj load_obj(const rt_object &o) {
    Obj r;
    r.kind = o.kind; r.material_index = (uint32_t)o.material_index;
    r.transform = mat4_from_cols_array(o.transform); r.inv_transform = mat4_from_cols_array(o.inv_transform);
    r.radius = o.radius; r.height = o.height; r.dimension = ld3(o.dimension);
    r.normal = ld3(o.normal); r.point = ld3(o.point); r.max_dist = v2(o.max_dist[0], o.max_dist[1]);
    r.has_max_dist = o.has_max_dist != 0;
    r.v1 = ld3(o.v1); r.v2 = ld3(o.v2); r.v3 = ld3(o.v3);
    r.first = (uint32_t)o.first; r.second = (uint32_t)o.second;
    return r;
}

// Depth of Union/Substraction nesting below object i; -1 on an index error or a cycle.
static int composite_depth(const rt_scene_desc *d, uint32_t i, int guard) {
    if (i >= d->n_objects || guard > 16) return -1;
    const rt_object &o = d->objects[i];
    if (o.kind != RT_UNION && o.kind != RT_SUBSTRACTION) return 0;
    if (o.first < 0 || o.second < 0) return -1;
    int a = composite_depth(d, (uint32_t)o.first, guard + 1), b = composite_depth(d, (uint32_t)o.second, guard + 1);
    if (a < 0 || b < 0) return -1;
    return 1 + (a > b ? a : b);
}

// Everything the reference would panic on with an index out of bounds.
static bool load_scene(const rt_scene_desc *d, const orc_texture *tex, Scene &s) {
    if (!d) return false;
    for (uint32_t i = 0; i < d->n_objects; i++) {
        const rt_object &o = d->objects[i];
        if (o.kind < RT_SPHERE || o.kind > RT_SUBSTRACTION) return false;
        if (o.kind == RT_UNION || o.kind == RT_SUBSTRACTION) {
            if (composite_depth(d, i, 0) < 0) return false;
        } else if (o.material_index < 0 || (uint32_t)o.material_index >= d->n_materials) {
            return false;
        }
        s.objects.push_back(load_obj(o));
    }
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const rt_material &m = d->materials[i];
        if (m.texture >= 0 && ((uint32_t)m.texture >= d->n_textures || !tex || !tex[m.texture].rgb8 ||
                               tex[m.texture].width == 0 || tex[m.texture].height == 0))
            return false;
        s.materials.push_back(Mat{m.ambience, m.diffuse, m.specular, m.shininess, ld3(m.albedo), m.texture, m.kind,
                                  m.roughness, m.transparency, m.refraction_index, m.reflectivity, m.emission_power});
    }
    for (uint32_t i = 0; i < d->n_lights; i++)
        s.lights.push_back(Lgt{d->lights[i].kind, ld3(d->lights[i].albedo), ld3(d->lights[i].vec), d->lights[i].intensity});
    for (uint32_t i = 0; i < d->n_sdfs; i++) {
        if (d->sdfs[i] >= d->n_objects) return false;
        s.sdfs.push_back(d->sdfs[i]);
    }
    if (d->ray_marching && d->n_materials == 0 && d->n_sdfs > 0) return false;
    s.textures = tex; s.n_textures = d->n_textures;
    s.ambient_color = ld3(d->ambient_color); s.max_ray_bounces = d->max_ray_bounces;
    s.shadow_casting = d->shadow_casting; s.ray_marching = d->ray_marching; s.diffuse = d->diffuse;
    s.enable_accumulation = d->enable_accumulation;
    return true;
}

// ------------------------------------------------------------------------------------------------
// counter-based generator (stands in for rand 0.8.5 ThreadRng; mapping per UniformFloat<f32>)
// ------------------------------------------------------------------------------------------------
static inline uint32_t pcg_hash(uint32_t v) {
    uint32_t state = v * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
static inline uint32_t rng_path_key(uint64_t seed, uint32_t pixel, uint32_t frame) {
    uint32_t k = pcg_hash((uint32_t)seed ^ pcg_hash((uint32_t)(seed >> 32)));
    k = pcg_hash(k + pixel);
    k = pcg_hash(k + frame);
    return k;
}
// branch 0: the reflection / continuation child, branch 1: the refraction child
static inline uint32_t rng_child_key(uint32_t key, uint32_t branch) { return pcg_hash(key + 0x9E3779B9u * (branch + 1u)); }
static inline float rng_uniform(uint32_t key, uint32_t draw, float low, float high) {
    uint32_t bits = pcg_hash(key ^ (0x85EBCA6Bu * (draw + 1u)));
    float value0_1 = (float)(bits >> 9) * 1.1920928955078125e-7f;  // 2^-23
    return value0_1 * (high - low) + low;
}

// ------------------------------------------------------------------------------------------------
// src/objects.rs:121-147  Texture::{textel, from_uv, pixel}
// ------------------------------------------------------------------------------------------------
static inline float textel(float p) {
    if (p < 0.0f) return 1.0f - std::fabs(std::ceil(p) - p);
    else if (p > 1.0f) return p - std::floor(p);
    return p;
}
static inline Vec3 texture_pixel(const orc_texture &t, uint32_t x, uint32_t y) {
    const float RGB_RATIO = 1.0f / 255.0f;  // objects.rs:14
    size_t pos = (size_t)(uint32_t)(y * 3u * t.width + x * 3u);
    return v3((float)t.rgb8[pos] * RGB_RATIO, (float)t.rgb8[pos + 1] * RGB_RATIO, (float)t.rgb8[pos + 2] * RGB_RATIO);
}
static inline Vec3 texture_from_uv(const orc_texture &t, float u, float v) {
    float uu = textel(u), vv = textel(v);
    uint32_t x = as_u32((float)(t.width - 1) * uu);
    uint32_t y = as_u32((float)(t.height - 1) * vv);
    return texture_pixel(t, x, y);
}

// ------------------------------------------------------------------------------------------------
// src/utils/geometry.rs
// ------------------------------------------------------------------------------------------------
static inline Vec3 reflect(Vec3 vec, Vec3 normal) { return vec - (2.0f * dot(vec, normal)) * normal; }  // :7-9
static inline Vec3 step(Vec3 a, Vec3 b) {                                                                // :21-27
    return v3(b.x < a.x ? 0.0f : 1.0f, b.y < a.y ? 0.0f : 1.0f, b.z < a.z ? 0.0f : 1.0f);
}
static inline float smooth_step(float e0, float e1, float x) {  // :29-32
    float t = fclamp((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
static inline float mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }          // :34-36
static inline Vec3 mix_vec3(Vec3 x, Vec3 y, float a) { return x * (1.0f - a) + y * a; }        // :38-40
static inline float interpolation(float d1, float d2, float k) {                               // :42-44
    return fclamp(0.5f + 0.5f * (d2 - d1) / k, 0.0f, 1.0f);
}
static inline float smooth_union(float d1, float d2, float k) {  // :46-50
    float h = fclamp(0.5f + 0.5f * (d2 - d1) / k, 0.0f, 1.0f);
    return mix(d2, d1, h) - k * h * (1.0f - h);
}
static inline Vec3 tri_planar_mapping(Vec3 p, Vec3 n, float blending, float scale, const orc_texture &tex) {  // :56-68
    Vec2 xy = v2(p.x, p.y) * scale, xz = v2(p.x, p.z) * scale, yz = v2(p.y, p.z) * scale;
    Vec3 x = texture_from_uv(tex, yz.x, yz.y);
    Vec3 y = texture_from_uv(tex, xz.x, xz.y);
    Vec3 z = texture_from_uv(tex, xy.x, xy.y);
    Vec3 bw = vpowf(vabs(n), blending);
    bw = bw / (bw.x + bw.y + bw.z);
    return x * bw.x + y * bw.y + z * bw.z;
}

// ------------------------------------------------------------------------------------------------
// src/light.rs:42-125  LightSource
// ------------------------------------------------------------------------------------------------
static inline Vec3 light_direction(const Lgt &l, Vec3 point) {
    if (l.kind == RT_DIRECTIONAL) return l.vec;
    return normalize(point - l.vec);
}
static inline float light_distance(const Lgt &l, Vec3 point) {
    if (l.kind == RT_DIRECTIONAL) return 1.0f;
    return length(point - l.vec);
}

// ------------------------------------------------------------------------------------------------
// Intersection impls
// ------------------------------------------------------------------------------------------------
static inline Ray transform_ray(const Obj &o, const Ray &n) {  // sphere.rs:88-94 and the identical copies
    Ray r;
    r.direction = xyz(o.inv_transform * v4(n.direction, 0.0f));
    r.origin = xyz(o.inv_transform * v4(n.origin, 1.0f));
    return r;
}

// src/utils/sphere.rs:98-147
static bool sphere_intersect(const Obj &s, const Ray &ray, RayHit &h) {
    Vec3 ray_dir = xyz(s.inv_transform * v4(ray.direction, 0.0f));
    Vec3 ray_origin = xyz(s.inv_transform * v4(ray.origin, 1.0f));
    float a = dot(ray_dir, ray_dir);
    float b = 2.0f * dot(ray_origin, ray_dir);
    float c = dot(ray_origin, ray_origin) - 1.0f;
    float disc = b * b - 4.0f * a * c;
    if (disc < 0.0f) return false;
    float t1 = (-b - std::sqrt(disc)) / (2.0f * a);
    Vec3 l = ray_origin + ray_dir * t1;
    Vec3 normal = normalize(xyz(s.transform * v4(l, 0.0f)));
    float u = orc_atanf((l.x * l.x + l.y * l.y) / l.z);
    float v = orc_atanf(l.y / l.x);
    h = RayHit{t1, ray.origin + ray.direction * t1, normal, s.material_index, v * INV_PI, u * INV_PI};
    return true;
}

// src/utils/plane.rs:41-74
static bool plane_intersect(const Obj &p, const Ray &ray, RayHit &h) {
    float denom = dot(ray.direction, p.normal);
    if (std::fabs(denom) < 1e-6f) return false;
    float t = dot(p.point - ray.origin, p.normal) / denom;
    if (t < 0.0f) return false;
    Vec3 hit_point = ray.origin + ray.direction * t;
    if (p.has_max_dist) {
        if (std::fabs(hit_point.z) > p.max_dist.y || std::fabs(hit_point.x) > p.max_dist.x) return false;
    }
    float sign = -1.0f;
    if (dot(ray.direction, p.normal) < 0.0f) sign = 1.0f;
    h = RayHit{t, hit_point, sign * p.normal, p.material_index, hit_point.x * 0.1f, hit_point.z * 0.1f};
    return true;
}

// src/utils/cuboid.rs:84-127 (b_min/b_max: :48-49)
static bool cuboid_intersect(const Obj &c, const Ray &ray, RayHit &h) {
    Vec3 ray_dir = xyz(c.inv_transform * v4(ray.direction, 0.0f));
    Vec3 ray_origin = xyz(c.inv_transform * v4(ray.origin, 1.0f));
    Vec3 b_min = xyz(mat4_from_scale(c.dimension) * v4(-1.0f, -1.0f, -1.0f, 0.0f));
    Vec3 b_max = xyz(mat4_from_scale(c.dimension) * v4(1.0f, 1.0f, 1.0f, 0.0f));
    Vec3 inv = 1.0f / ray_dir;
    Vec3 t_min = (b_min - ray_origin) * inv;
    Vec3 t_max = (b_max - ray_origin) * inv;
    Vec3 t_enter = vmin(t_min, t_max);
    Vec3 t_exit = vmax(t_min, t_max);
    float t_near = fmax_(fmax_(t_enter.x, t_enter.y), t_enter.z);
    float t_far = fmin_(fmin_(t_exit.x, t_exit.y), t_exit.z);
    if (t_near > t_far || t_far < 0.0f) return false;
    Vec3 a = -vsignum(ray_dir) * step(splat3(t_near), t_enter);
    Vec3 normal = xyz(c.transform * v4(a, 0.0f));
    Vec3 hit_point = ray.origin + ray.direction * t_near;
    Vec3 opos = xyz(c.inv_transform * v4(hit_point, 1.0f));
    Vec3 onor = a;
    Vec2 u_v = std::fabs(onor.x) * v2(opos.y, opos.z) + std::fabs(onor.y) * v2(opos.z, opos.x) +
               std::fabs(onor.z) * v2(opos.x, opos.y);
    h = RayHit{t_near, hit_point, normal, c.material_index, u_v.x, u_v.y};
    return true;
}

// src/utils/cylinder.rs:90-174
static bool cylinder_intersect(const Obj &c, const Ray &ray, RayHit &h) {
    Vec3 rd3 = xyz(c.inv_transform * v4(ray.direction, 0.0f));
    Vec3 ro3 = xyz(c.inv_transform * v4(ray.origin, 1.0f));
    Vec2 rd = v2(rd3.x, rd3.y), ro = v2(ro3.x, ro3.y);
    float a = dot(rd, rd);
    float b = 2.0f * dot(ro, rd);
    float cc = dot(ro, ro) - 1.0f;
    float disc = b * b - 4.0f * a * cc;
    if (disc > 0.0f) {
        float t1 = (-b - std::sqrt(disc)) / (2.0f * a);
        Vec3 h_t1 = ro3 + rd3 * t1;
        bool valid_t1 = std::fabs(h_t1.z) < 0.5f;
        if (valid_t1) {
            Vec3 n = v3(h_t1.x, h_t1.y, 0.0f);
            Vec3 normal = normalize(xyz(c.transform * v4(n, 0.0f)));
            Vec3 poi = ray.origin + ray.direction * t1;
            float u = orc_atanf(h_t1.y / h_t1.x);
            float v = h_t1.z * 2.0f;
            h = RayHit{t1, poi, normal, c.material_index, u, v};
            return true;
        }
    }
    float inv_rd3z = -1.0f / rd3.z;
    float t1 = (ro3.z - 0.5f) * inv_rd3z;
    float t2 = (ro3.z + 0.5f) * inv_rd3z;
    Vec3 h_t1 = ro3 + rd3 * t1;
    Vec3 h_t2 = ro3 + rd3 * t2;
    bool valid_t1 = dot(v2(h_t1.x, h_t1.y), v2(h_t1.x, h_t1.y)) < 1.0f;
    bool valid_t2 = dot(v2(h_t2.x, h_t2.y), v2(h_t2.x, h_t2.y)) < 1.0f;
    float t;
    if (valid_t1 && valid_t2) { t = (t1 < t2) ? t1 : t2; }
    else if (valid_t1) { t = t1; }
    else if (valid_t2) { t = t2; }
    else return false;
    Vec3 h_t = ro3 + rd3 * t;
    Vec3 n = v3(0.0f, 0.0f, h_t.z);
    Vec3 normal = normalize(xyz(c.transform * v4(n, 0.0f)));
    h = RayHit{t, ray.origin + ray.direction * t, normal, c.material_index, h_t.x, h_t.y};
    return true;
}

// src/utils/cone.rs:73-152
static bool cone_intersect(const Obj &c, const Ray &ray, RayHit &h) {
    Vec3 rd3 = xyz(c.inv_transform * v4(ray.direction, 0.0f));
    Vec3 ro3 = xyz(c.inv_transform * v4(ray.origin, 1.0f));
    Vec2 rd = v2(rd3.x, rd3.y), ro = v2(ro3.x, ro3.y);
    float a = dot(rd, rd) - rd3.z * rd3.z;
    float b = 2.0f * (dot(ro, rd) - ro3.z * rd3.z);
    float cc = dot(ro, ro) - ro3.z * ro3.z;
    float disc = b * b - 4.0f * a * cc;
    bool have_cone = false;
    RayHit cone{};
    if (disc > 0.0f) {
        float t1 = (-b - std::sqrt(disc)) / (2.0f * a);
        Vec3 h_t = ro3 + rd3 * t1;
        bool valid_t1 = h_t.z > 0.0f && h_t.z < 1.0f;
        if (valid_t1) {
            Vec3 poi = ray.origin + ray.direction * t1;
            Vec3 n = v3(h_t.x, h_t.y, -std::sqrt(h_t.x * h_t.x + h_t.y * h_t.y));
            Vec3 normal = normalize(xyz(c.transform * v4(n, 0.0f)));
            float u = orc_atanf(h_t.y / h_t.x);
            float v = h_t.z * 2.0f + 1.0f;
            cone = RayHit{t1, poi, normal, c.material_index, u, v};
            have_cone = true;
        }
    }
    float t1 = (ro3.z - 1.0f) / -rd3.z;
    Vec3 h_t1 = ro3 + rd3 * t1;
    bool valid_t1 = dot(v2(h_t1.x, h_t1.y), v2(h_t1.x, h_t1.y)) < 1.0f;
    if (valid_t1) {
        float t = t1;
        if (have_cone && cone.distance < t1) { h = cone; return true; }
        Vec3 h_t = ro3 + rd3 * t;
        Vec3 n = v3(0.0f, 0.0f, 1.0f);
        Vec3 normal = xyz(c.transform * v4(n, 0.0f));
        h = RayHit{t, ray.origin + ray.direction * t, normal, c.material_index, h_t.x, h_t.y};
        return true;
    }
    if (have_cone) { h = cone; return true; }
    return false;
}

// src/utils/triangle.rs:35-89
static bool triangle_intersect(const Obj &tr, const Ray &ray, RayHit &h) {
    Vec3 edge_ab = tr.v2 - tr.v1;
    Vec3 edge_ac = tr.v3 - tr.v1;
    Vec3 normal = cross(edge_ab, edge_ac);
    float denom = dot(ray.direction, normal);
    if (std::fabs(denom) < 1e-6f) return false;
    float t = dot(tr.v1 - ray.origin, normal) / denom;
    Vec3 hit_point = ray.origin + ray.direction * t;
    Vec3 edge_bc = tr.v3 - tr.v2;
    Vec3 edge_ca = tr.v1 - tr.v3;
    Vec3 a_to_hit = hit_point - tr.v1, b_to_hit = hit_point - tr.v2, c_to_hit = hit_point - tr.v3;
    Vec3 a_test = cross(edge_ab, a_to_hit), b_test = cross(edge_bc, b_to_hit), c_test = cross(edge_ca, c_to_hit);
    bool a_n = dot(a_test, normal) > 0.0f, b_n = dot(b_test, normal) > 0.0f, c_n = dot(c_test, normal) > 0.0f;
    if (!(a_n && b_n && c_n)) return false;
    Vec3 n = normal;
    float sign = -fsignum(dot(ray.direction, n));
    Vec3 abp_area = cross(edge_ab, a_to_hit);
    float u = dot(abp_area, n);
    Vec3 cap_area = cross(edge_ca, c_to_hit);
    float v = dot(cap_area, n);
    h = RayHit{t, hit_point, normalize(sign * n), tr.material_index, u, v};
    return true;
}

// src/ray.rs:167-177  Ray::hit
static bool ray_hit(const Ray &ray, const Obj &o, RayHit &h, Counters *cnt) {
    switch (o.kind) {
        case RT_SPHERE:   if (cnt) cnt->c[RT_CNT_TEST_SPHERE]++;   return sphere_intersect(o, ray, h);
        case RT_TRIANGLE: if (cnt) cnt->c[RT_CNT_TEST_TRIANGLE]++; return triangle_intersect(o, ray, h);
        case RT_CUBOID:   if (cnt) cnt->c[RT_CNT_TEST_CUBOID]++;   return cuboid_intersect(o, ray, h);
        case RT_PLANE:    if (cnt) cnt->c[RT_CNT_TEST_PLANE]++;    return plane_intersect(o, ray, h);
        case RT_CYLINDER: if (cnt) cnt->c[RT_CNT_TEST_CYLINDER]++; return cylinder_intersect(o, ray, h);
        case RT_CONE:     if (cnt) cnt->c[RT_CNT_TEST_CONE]++;     return cone_intersect(o, ray, h);
        default: return false;
    }
}
static int hit_counter_of(int kind) {
    switch (kind) {
        case RT_SPHERE: return RT_CNT_HIT_SPHERE; case RT_TRIANGLE: return RT_CNT_HIT_TRIANGLE;
        case RT_CUBOID: return RT_CNT_HIT_CUBOID; case RT_PLANE: return RT_CNT_HIT_PLANE;
        case RT_CYLINDER: return RT_CNT_HIT_CYLINDER; default: return RT_CNT_HIT_CONE;
    }
}

// ------------------------------------------------------------------------------------------------
// src/ray.rs:68-165 shading helpers, src/objects.rs:178-205 fresnel
// ------------------------------------------------------------------------------------------------
static Vec3 blinn_phong(const Ray &ray, const RayHit &hit, const Lgt &light, Vec3 color, const Mat &material) {
    float coeff = dot(hit.normal, -light_direction(light, hit.point));
    Vec3 ambience = material.ambience * color;
    Vec3 diffuse = material.diffuse * fmax_(coeff, 0.0f) * color;
    Vec3 half_angle = normalize(-ray.direction - light_direction(light, hit.point));
    float shininess = orc_powf(fmax_(dot(hit.normal, half_angle), 0.0f), material.shininess);
    Vec3 specular = material.specular * shininess * color;
    return ambience + diffuse + specular;
}

static Ray reflection_ray(const Ray &ray, const RayHit &hit, float roughness, uint32_t key, bool diffuse,
                          bool enable_accumulation) {
    Vec3 dir;
    if (!diffuse) {
        Vec3 factor = enable_accumulation
                          ? roughness * v3(rng_uniform(key, 0, -0.5f, 0.5f), rng_uniform(key, 1, -0.5f, 0.5f),
                                           rng_uniform(key, 2, -0.5f, 0.5f))
                          : splat3(0.0f);
        dir = normalize(reflect(ray.direction, hit.normal + factor));
    } else {
        Vec3 rnd = v3(rng_uniform(key, 0, -1.0f, 1.0f), rng_uniform(key, 1, -1.0f, 1.0f), rng_uniform(key, 2, -1.0f, 1.0f));
        dir = normalize(hit.normal + rnd);
    }
    return Ray{hit.point + hit.normal * EPSILON, dir};
}

static bool refraction_ray(const Ray &ray, const RayHit &hit, float refraction_index, Ray &out) {
    Vec3 normal = hit.normal;
    float eta_t = refraction_index;
    float eta_i = 1.0f;
    float c1 = dot(ray.direction, hit.normal);
    if (c1 < 0.0f) { c1 = -c1; }
    else { normal = -normal; eta_i = eta_t; eta_t = 1.0f; }
    float eta = eta_i / eta_t;
    float k = 1.0f - eta * eta * (1.0f - c1 * c1);
    if (k < 0.0f) return false;
    float c2 = std::sqrt(k);
    Vec3 direction = eta * ray.direction + normal * (eta * c1 - c2);
    out = Ray{hit.point - EPSILON * normal, direction};
    return true;
}

static float fresnel(Vec3 incident, Vec3 normal, float refraction_index, float reflectivity) {
    float n2 = refraction_index;
    float n1 = 1.0f;
    float r0 = (n1 - n2) / (n1 + n2);
    r0 *= r0;
    float cos_x = dot(normal, -incident);
    if (n1 > n2) {
        float n = n1 / n2;
        float sin_t2 = n * n * (1.0f - cos_x * cos_x);
        if (sin_t2 > 1.0f) return 1.0f;
        cos_x = std::sqrt(1.0f - sin_t2);
    }
    float x = 1.0f - cos_x;
    float ret = r0 + (1.0f - r0) * x * x * x * x * x;
    return reflectivity + (1.0f - reflectivity) * ret;
}

// ------------------------------------------------------------------------------------------------
// src/ray_tracing/ray_tracing.rs
// ------------------------------------------------------------------------------------------------
struct RayTracing {
    const Scene &scene; Counters *cnt;

    // :59-78
    bool trace_ray(const Ray &ray, RayHit &out, uint32_t &out_idx, int ray_counter) const {
        if (scene.objects.empty()) return false;
        if (cnt) cnt->c[ray_counter]++;
        float closest_t = 3.40282347e+38f;
        bool found = false;
        for (uint32_t idx = 0; idx < scene.objects.size(); idx++) {
            RayHit t;
            if (ray_hit(ray, scene.objects[idx], t, cnt)) {
                if (t.distance > 0.0f && t.distance < closest_t) {
                    out = t; out_idx = idx; found = true;
                    closest_t = t.distance;
                }
            }
        }
        if (found && cnt && ray_counter != RT_CNT_RAYS_SHADOW) cnt->c[hit_counter_of(scene.objects[out_idx].kind)]++;
        return found;
    }

    // :28-57
    Vec3 light(const Ray &ray, const RayHit &hit, Vec3 albedo, const Mat &material, uint32_t obj_index) const {
        if (cnt) cnt->c[RT_CNT_LIGHT_CALL]++;
        Vec3 l_acc = splat3(0.0f);
        for (const Lgt &l : scene.lights) {
            if (cnt) cnt->c[RT_CNT_LIGHT_SHADE]++;
            Vec3 k = blinn_phong(ray, hit, l, albedo, material);
            float light_dis = light_distance(l, hit.point);
            l_acc += (k / (light_dis * light_dis)) * l.albedo * l.intensity;
        }
        if (scene.shadow_casting) {
            for (const Lgt &l : scene.lights) {
                RayHit sh; uint32_t idx = 0;
                Ray sr{hit.point + EPSILON * hit.normal, -light_direction(l, hit.point)};
                if (trace_ray(sr, sh, idx, RT_CNT_RAYS_SHADOW)) {
                    if (idx != obj_index) l_acc *= 0.5f;
                }
            }
        }
        return vpowf(l_acc, 0.4166f);
    }

    Vec3 albedo_of(const Mat &material, const RayHit &hit) const {  // :177-179 / :97-99
        if (material.texture >= 0) {
            if (cnt) cnt->c[RT_CNT_TEX_FETCH]++;
            return texture_from_uv(scene.textures[material.texture], hit.u, hit.v);
        }
        return material.albedo;
    }

    // :160-241
    Vec3 color(const Ray &ray, uint32_t key, uint8_t depth, Vec3 light_color, Vec3 contribution, int ray_counter) const {
        if (depth >= scene.max_ray_bounces) return light_color;
        RayHit hit; uint32_t obj_index;
        if (trace_ray(ray, hit, obj_index, ray_counter)) {
            const Mat &material = scene.materials[hit.material_index];
            Vec3 albedo = material.albedo;
            if (material.kind == RT_REFLECTIVE) {
                float roughness = material.roughness;
                albedo = albedo_of(material, hit);
                Vec3 p_light = light(ray, hit, albedo, material, obj_index);
                if (cnt) cnt->c[scene.diffuse ? RT_CNT_DIFFUSE_RAY : RT_CNT_REFLECTION_RAY]++;
                Ray r = reflection_ray(ray, hit, roughness, key, scene.diffuse, scene.enable_accumulation);
                Vec3 reflected_col = color(r, rng_child_key(key, 0), depth + 1, p_light, contribution * albedo, RT_CNT_RAYS_REFLECTION);
                if (cnt) cnt->c[RT_CNT_COMBINE_REFL]++;
                return p_light * (roughness) + p_light * reflected_col * (1.0f - roughness);
            } else {
                Vec3 refraction_color = splat3(0.0f);
                if (cnt) cnt->c[RT_CNT_REFRACTION]++;
                float kr = fresnel(ray.direction, hit.normal, material.refraction_index, material.reflectivity);
                Ray rr;
                if (refraction_ray(ray, hit, material.refraction_index, rr)) {
                    refraction_color = color(rr, rng_child_key(key, 1), depth + 1, light_color, contribution * albedo, RT_CNT_RAYS_REFRACTION);
                }
                if (cnt) cnt->c[RT_CNT_REFLECTION_RAY]++;
                Ray refl{hit.point + EPSILON * hit.normal, reflect(ray.direction, hit.normal)};
                Vec3 p_light = light(ray, hit, albedo, material, obj_index);
                Vec3 reflection_color = color(refl, rng_child_key(key, 0), depth + 1, p_light, contribution * albedo, RT_CNT_RAYS_REFLECTION);
                if (cnt) cnt->c[RT_CNT_COMBINE_REFR]++;
                Vec3 c = reflection_color * kr + refraction_color * (1.0f - kr) * material.transparency;
                return c * albedo;
            }
        }
        return light_color + scene.ambient_color * contribution;
    }

    // :80-158
    Vec3 color_diffuse(const Ray &ray, uint32_t key, uint8_t depth, Vec3 light_color, Vec3 contribution, int ray_counter) const {
        if (depth >= scene.max_ray_bounces) return light_color;
        RayHit hit; uint32_t obj_index;
        if (trace_ray(ray, hit, obj_index, ray_counter)) {
            const Mat &material = scene.materials[hit.material_index];
            Vec3 albedo = material.albedo;
            if (material.kind == RT_REFLECTIVE) {
                albedo = albedo_of(material, hit);
                Vec3 p_light = light_color + material.emission_power * albedo;
                if (cnt) cnt->c[scene.diffuse ? RT_CNT_DIFFUSE_RAY : RT_CNT_REFLECTION_RAY]++;
                Ray r = reflection_ray(ray, hit, material.roughness, key, scene.diffuse, scene.enable_accumulation);
                if (cnt) cnt->c[RT_CNT_COMBINE_REFL]++;
                return color_diffuse(r, rng_child_key(key, 0), depth + 1, p_light, contribution * albedo, RT_CNT_RAYS_REFLECTION);
            } else {
                Vec3 refraction_color = splat3(0.0f);
                if (cnt) cnt->c[RT_CNT_REFRACTION]++;
                float kr = fresnel(ray.direction, hit.normal, material.refraction_index, material.reflectivity);
                Ray rr;
                if (refraction_ray(ray, hit, material.refraction_index, rr)) {
                    refraction_color = color(rr, rng_child_key(key, 1), depth + 1, light_color, contribution * albedo, RT_CNT_RAYS_REFRACTION);
                }
                if (cnt) cnt->c[RT_CNT_REFLECTION_RAY]++;
                Ray refl{hit.point + EPSILON * hit.normal, reflect(ray.direction, hit.normal)};
                Vec3 p_light = light_color + material.emission_power * albedo;
                Vec3 reflection_color = color_diffuse(refl, rng_child_key(key, 0), depth + 1, p_light, contribution * albedo, RT_CNT_RAYS_REFLECTION);
                if (cnt) cnt->c[RT_CNT_COMBINE_REFR]++;
                return reflection_color * kr + refraction_color * (1.0f - kr) * material.transparency;
            }
        }
        return light_color + scene.ambient_color * contribution;
    }

    // :17-26
    Vec3 albedo(const Ray &ray, uint32_t key) const {
        Vec3 light = splat3(0.0f), contribution = splat3(1.0f);
        if (scene.diffuse) return color_diffuse(ray, key, 0, light, contribution, RT_CNT_RAYS_PRIMARY);
        return color(ray, key, 0, light, contribution, RT_CNT_RAYS_PRIMARY);
    }
};

// ------------------------------------------------------------------------------------------------
// src/ray_marching/ray_marching.rs and the sdf() impls
// ------------------------------------------------------------------------------------------------
struct RayMarching {
    const Scene &scene; Counters *cnt;
    static constexpr int MAX_STEPS = 255;        // :13
    static constexpr float MAX_DISTANCE = 40.0f;  // :14
    static constexpr float HIT_PRECISION = 0.001f;  // :15

    // Object3D::sdf (src/objects.rs:33-44) with `object` = the object itself at every call site
    RayMarchingHit object_sdf(const Obj &o, const Ray &ray, float t) const {
        switch (o.kind) {
            case RT_SPHERE: {  // sphere.rs:72-82
                if (cnt) cnt->c[RT_CNT_SDF_SPHERE]++;
                Ray r = transform_ray(o, ray);
                Vec3 p = r.origin + r.direction * t;
                return RayMarchingHit{length(p) - o.radius, scene.materials[o.material_index].albedo, r};
            }
            case RT_CUBOID: {  // cuboid.rs:53-70
                if (cnt) cnt->c[RT_CNT_SDF_CUBOID]++;
                Ray r = transform_ray(o, ray);
                Vec3 p = r.origin + r.direction * t;
                float corner_radius = 0.1f;
                Vec3 q = vabs(p) - o.dimension + corner_radius;
                float d = length(vmax(q, splat3(0.0f))) + fmin_(fmax_(q.x, fmax_(q.y, q.z)), 0.0f) - corner_radius;
                return RayMarchingHit{d, scene.materials[o.material_index].albedo, r};
            }
            case RT_PLANE: {  // plane.rs:31-37
                if (cnt) cnt->c[RT_CNT_SDF_PLANE]++;
                Vec3 p = ray.origin + ray.direction * t;
                return RayMarchingHit{dot(p - o.point, o.normal), scene.materials[o.material_index].albedo, ray};
            }
            case RT_CYLINDER: {  // cylinder.rs:56-74
                if (cnt) cnt->c[RT_CNT_SDF_CYLINDER]++;
                Ray r = transform_ray(o, ray);
                Vec3 p = r.origin + r.direction * t;
                float corner_radius = 0.1f;
                Vec2 d = v2(length(v2(p.x, p.z)), std::fabs(p.y)) - v2(o.radius, o.height * 0.5f) + corner_radius;
                float dist = length(vmax(d, v2(0.0f, 0.0f))) + fmin_(fmax_(d.x, d.y), 0.0f) - corner_radius;
                return RayMarchingHit{dist, scene.materials[o.material_index].albedo, r};
            }
            case RT_UNION: {  // union.rs:26-42
                if (cnt) cnt->c[RT_CNT_SDF_UNION]++;
                const Obj &o1 = scene.objects[o.first], &o2 = scene.objects[o.second];
                RayMarchingHit h1 = object_sdf(o1, ray, t), h2 = object_sdf(o2, ray, t);
                float i = interpolation(h1.distance, h2.distance, 0.7f);
                Vec3 col = mix_vec3(h1.albedo, h2.albedo, 1.0f - i);
                float d = smooth_union(h1.distance, h2.distance, 0.7f);
                if (h1.distance < h2.distance) return RayMarchingHit{d, col, h1.transformed_ray};
                return RayMarchingHit{d, col, h2.transformed_ray};
            }
            case RT_SUBSTRACTION: {  // substraction.rs:26-37
                if (cnt) cnt->c[RT_CNT_SDF_SUBSTRACTION]++;
                const Obj &o1 = scene.objects[o.first], &o2 = scene.objects[o.second];
                RayMarchingHit h1 = object_sdf(o1, ray, t), h2 = object_sdf(o2, ray, t);
                float d = fmax_(-h2.distance, h1.distance);
                uint32_t m1 = (o1.kind == RT_UNION || o1.kind == RT_SUBSTRACTION) ? 0u : o1.material_index;
                return RayMarchingHit{d, scene.materials[m1].albedo, h2.transformed_ray};
            }
            default:  // cone.rs:55-57, triangle.rs:29-31
                return RayMarchingHit{3.40282347e+38f, splat3(0.0f), Ray{splat3(0.0f), splat3(0.0f)}};
        }
    }

    // :24-107
    void sdfs(const Ray &ray, float t, uint32_t &obj_idx, RayMarchingHit &out) const {
        if (cnt) cnt->c[RT_CNT_SDFS_CALL]++;
        float min_dist = 3.40282347e+38f;
        obj_idx = 0;
        Vec3 albedo = splat3(0.0f);
        Ray tray{splat3(0.0f), splat3(0.0f)};
        for (uint32_t idx : scene.sdfs) {
            const Obj &obj = scene.objects[idx];
            switch (obj.kind) {
                case RT_CUBOID: case RT_SPHERE: case RT_PLANE: case RT_CYLINDER: {
                    RayMarchingHit d = object_sdf(obj, ray, t);
                    if (d.distance < min_dist) { min_dist = d.distance; albedo = d.albedo; tray = d.transformed_ray; obj_idx = idx; }
                    break;
                }
                case RT_UNION: case RT_SUBSTRACTION: {
                    RayMarchingHit d = object_sdf(obj, ray, t);
                    if (d.distance < min_dist) { min_dist = d.distance; albedo = d.albedo; obj_idx = obj.second; tray = d.transformed_ray; }
                    break;
                }
                default: break;
            }
        }
        out = RayMarchingHit{min_dist, albedo, tray};
    }

    // :109-139
    Vec3 normal(Vec3 p) const {
        float k = 0.5773f * 0.0005f;
        Vec3 xyy = v3(1.0f, -1.0f, -1.0f), yyx = v3(-1.0f, -1.0f, 1.0f), yxy = v3(-1.0f, 1.0f, -1.0f), xxx = v3(1.0f, 1.0f, 1.0f);
        uint32_t i; RayMarchingHit a, b, c, d;
        sdfs(Ray{p, xyy}, k, i, a); sdfs(Ray{p, yyx}, k, i, b); sdfs(Ray{p, yxy}, k, i, c); sdfs(Ray{p, xxx}, k, i, d);
        return normalize(xyy * a.distance + yyx * b.distance + yxy * c.distance + xxx * d.distance);
    }

    // :141-158
    float occlusion(Vec3 pos, Vec3 nor) const {
        float occ = 0.0f, sca = 1.0f;
        for (int i = 0; i < 3; i++) {
            if (cnt) cnt->c[RT_CNT_OCCLUSION_TAP]++;
            float hr = 0.02f + 0.025f * (float)(i * i);
            uint32_t oi; RayMarchingHit dd;
            sdfs(Ray{pos, nor}, hr, oi, dd);
            occ += -(dd.distance - hr) * sca;
            sca *= 0.85f;
        }
        return 1.0f - fclamp(occ, 0.0f, 1.0f);
    }

    // :160-179
    Vec3 light(const Ray &ray, const RayHit &hit, Vec3 albedo) const {
        if (cnt) cnt->c[RT_CNT_LIGHT_CALL]++;
        Vec3 l_acc = splat3(0.0f);
        if (hit.material_index < scene.materials.size()) {
            const Mat &material = scene.materials[hit.material_index];
            for (const Lgt &l : scene.lights) {
                if (cnt) cnt->c[RT_CNT_LIGHT_SHADE]++;
                Vec3 phong = blinn_phong(ray, hit, l, albedo, material);
                float light_dis = light_distance(l, hit.point);
                l_acc += (phong / (light_dis * light_dis)) * l.albedo * l.intensity;
            }
        }
        return vpowf(l_acc, 0.4545f);
    }

    // :181-203
    bool march_ray(const Ray &ray, uint32_t &obj_idx, RayMarchingHit &out) const {
        if (cnt) cnt->c[RT_CNT_RAYS_MARCH]++;
        float t = 0.0f;
        int i = 0;
        while (i < MAX_STEPS) {
            if (t > MAX_DISTANCE) break;
            if (cnt) cnt->c[RT_CNT_MARCH_STEP]++;
            RayMarchingHit rh;
            sdfs(ray, t, obj_idx, rh);
            t += rh.distance;
            if (rh.distance < HIT_PRECISION) { out = RayMarchingHit{t, rh.albedo, rh.transformed_ray}; return true; }
            i++;
        }
        return false;
    }

    // :205-253
    Vec3 albedo(const Ray &ray) const {
        uint32_t obj_idx; RayMarchingHit rh;
        if (march_ray(ray, obj_idx, rh)) {
            if (cnt) cnt->c[RT_CNT_MARCH_SHADE]++;
            Vec3 hit = ray.origin + ray.direction * rh.distance;
            Vec3 n = normal(hit);
            const Obj &obj = scene.objects[obj_idx];
            uint32_t mat_idx = (obj.kind == RT_UNION || obj.kind == RT_SUBSTRACTION) ? 0u : obj.material_index;
            const Mat &mat = scene.materials[mat_idx];
            Vec3 albedo = rh.albedo;
            if (mat.texture >= 0) {
                if (cnt) cnt->c[RT_CNT_TRIPLANAR]++;
                Vec3 n1 = normalize(xyz(obj.inv_transform * v4(n, 0.0f)));
                Vec3 hit1 = rh.transformed_ray.origin + rh.transformed_ray.direction * rh.distance;
                albedo = tri_planar_mapping(hit1, n1, 0.8f, 0.5f, scene.textures[mat.texture]);
            }
            RayHit rayhit{rh.distance, hit, n, mat_idx, 0.0f, 0.0f};
            Vec3 color = light(ray, rayhit, albedo);
            float occ = occlusion(hit, n);
            color *= occ;
            color *= 1.0f - smooth_step(1.0f, 20.0f, rh.distance);
            return color;
        }
        return scene.ambient_color;
    }
};

// src/scene.rs:80-90  Scene::pixel
static Vec4 scene_pixel(const Scene &scene, const Ray &ray, uint32_t key, Counters *cnt) {
    Vec3 light;
    if (!scene.ray_marching) { RayTracing tracer{scene, cnt}; light = tracer.albedo(ray, key); }
    else { RayMarching tracer{scene, cnt}; light = tracer.albedo(ray); }
    return v4(light.x, light.y, light.z, 1.0f);
}

// src/camera.rs:128-157  Camera::calculate_ray_directions, one pixel
static Vec3 camera_ray_dir(const rt_camera_desc &cam, uint32_t x, uint32_t y) {
    Mat4 inverse_perspective = mat4_from_cols_array(cam.inverse_perspective);
    Mat4 inverse_view = mat4_from_cols_array(cam.inverse_view);
    Vec3 position = ld3(cam.position);
    float inv_w = 1.0f / (float)cam.width;
    float inv_h = 1.0f / (float)cam.height;
    float p_ndc_x = (float)x * inv_w;
    float p_ndc_y = (float)y * inv_h;
    float p_screen_x = 2.0f * p_ndc_x - 1.0f;
    float p_screen_y = 2.0f * p_ndc_y - 1.0f;
    Vec4 target = inverse_perspective * v4(p_screen_x, p_screen_y, 1.0f, 1.0f);
    Vec3 v = v3(target.x, target.y, target.z) / target.w;
    Vec4 world_coords = inverse_view * v4(v.x, v.y, v.z, 0.0f);
    Vec3 wc = v3(world_coords.x, world_coords.y, world_coords.z);
    return normalize(wc - position);
}

// src/renderer.rs:29-36
static void to_rgba(Vec4 c, uint8_t *out) {
    out[0] = as_u8(c.x * 255.0f); out[1] = as_u8(c.y * 255.0f); out[2] = as_u8(c.z * 255.0f); out[3] = as_u8(c.w + 255.0f);
}

// src/renderer.rs:38-79  render_chunk over pixels [p0, p1)
// ray_dirs: Camera.ray_directions (src/camera.rs:16), the table calculate_ray_directions fills on every camera change and
// render_chunk reads (src/renderer.rs:49); null = compute the same value on the spot.
static void render_chunk(const Scene &scene, const rt_camera_desc &cam, const float *ray_dirs, uint32_t frame_index, bool accumulate, uint64_t seed,
                         size_t p0, size_t p1, float *accum, uint8_t *rgba, float *linear, Counters *cnt) {
    Vec3 position = ld3(cam.position);
    for (size_t pos = p0; pos < p1; pos++) {
        uint32_t x = (uint32_t)(pos % cam.width), y = (uint32_t)(pos / cam.width);
        if (cnt) { cnt->c[RT_CNT_RAYGEN]++; cnt->c[RT_CNT_PIXEL_PACK]++; }
        Vec3 ray_dir = ray_dirs ? ld3(ray_dirs + 3 * pos) : camera_ray_dir(cam, x, y);
        uint32_t key = rng_path_key(seed, (uint32_t)pos, frame_index);
        Vec4 p = scene_pixel(scene, Ray{position, ray_dir}, key, cnt);
        if (linear) { linear[4 * pos] = p.x; linear[4 * pos + 1] = p.y; linear[4 * pos + 2] = p.z; linear[4 * pos + 3] = p.w; }
        Vec4 *acc = reinterpret_cast<Vec4 *>(accum + 4 * pos);
        if (accumulate) {
            *acc = *acc + p;
            Vec4 a = *acc;
            a = a / (float)frame_index;
            a = clamp01(a);
            to_rgba(a, rgba + 4 * pos);
        } else {
            *acc = clamp01(p);
            to_rgba(*acc, rgba + 4 * pos);
        }
    }
}

}  // namespace orc

using namespace orc;

extern "C" {

int32_t orc_render(const rt_scene_desc *sd, const orc_texture *textures, const rt_camera_desc *camera, uint32_t frame_index,
                   int32_t accumulate, uint64_t seed, uint32_t row_begin, uint32_t row_end, float *accum, uint8_t *rgba,
                   float *linear, uint64_t *counters, int32_t n_threads) {
    return orc_render_with_directions(sd, textures, camera, nullptr, frame_index, accumulate, seed, row_begin, row_end, accum, rgba,
                                      linear, counters, n_threads);
}

// src/camera.rs:128-157: the whole table, row-major, 3 floats per pixel
void orc_ray_directions(const rt_camera_desc *camera, float *out) {
    for (uint32_t y = 0; y < camera->height; y++)
        for (uint32_t x = 0; x < camera->width; x++) {
            Vec3 d = camera_ray_dir(*camera, x, y);
            float *o = out + 3 * ((size_t)x + (size_t)y * camera->width);
            o[0] = d.x; o[1] = d.y; o[2] = d.z;
        }
}

int32_t orc_render_with_directions(const rt_scene_desc *sd, const orc_texture *textures, const rt_camera_desc *camera,
                                   const float *ray_dirs, uint32_t frame_index, int32_t accumulate, uint64_t seed,
                                   uint32_t row_begin, uint32_t row_end, float *accum, uint8_t *rgba, float *linear,
                                   uint64_t *counters, int32_t n_threads) {
    Scene scene;
    if (!camera || !accum || !rgba || camera->width == 0 || camera->height == 0) return -1;
    if (!load_scene(sd, textures, scene)) return -1;
    if (row_begin == 0 && row_end == 0) row_end = camera->height;
    if (row_end > camera->height || row_begin > row_end) return -1;
    size_t first = (size_t)row_begin * camera->width, last = (size_t)row_end * camera->width;
    // src/renderer.rs:104-107: img.len()/(num_chunks*4) pixels per chunk, num_chunks = 200 (src/app.rs:80)
    size_t total = (size_t)camera->width * camera->height;
    size_t chunk = total / 200;
    if (chunk == 0) chunk = total;
    size_t n_chunks = (last - first + chunk - 1) / chunk;
    if (n_threads < 1) n_threads = 1;
    std::vector<Counters> cnts((size_t)n_threads);
    for (auto &c : cnts) std::memset(&c, 0, sizeof(c));
    std::atomic<size_t> next{0};
    auto worker = [&](int tid) {
        for (;;) {
            size_t ci = next.fetch_add(1);
            if (ci >= n_chunks) break;
            size_t p0 = first + ci * chunk, p1 = p0 + chunk;
            if (p1 > last) p1 = last;
            render_chunk(scene, *camera, ray_dirs, frame_index, accumulate != 0, seed, p0, p1, accum, rgba, linear,
                         counters ? &cnts[(size_t)tid] : nullptr);
        }
    };
    if (n_threads == 1) worker(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; t++) th.emplace_back(worker, t);
        for (auto &t : th) t.join();
    }
    if (counters) for (auto &c : cnts) for (int i = 0; i < RT_CNT_COUNT; i++) counters[i] += c.c[i];
    return 0;
}

// src/renderer.rs:104-143 AS WRITTEN, for whole frames: `img.chunks_mut(img_chunk_size)` yields a short last chunk whenever
// width * height is not a multiple of num_chunks, and the closure derives that chunk's pixel offset from ITS OWN length
// (`offset = e.0 * acc_size`, :115-117) instead of the length of the chunks before it.  The short chunk therefore reads the
// directions and accumulator entries of pixels [n * s, n * s + s) (n = its index, s = its length) while its bytes and, through
// the sequential copy-back (:138-143), its accumulator entries land at the end of the frame.  orc_render above gives every
// pixel its own direction (what the GPU backend does); this entry point exists to state the difference exactly.
// Returns -2 where the reference panics: `chunks_mut(0)` for frames of fewer than num_chunks pixels.
// The random-number key of a pixel is that of the pixel whose direction is used (the reference's ThreadRng has no such key).
int32_t orc_render_reference_chunks(const rt_scene_desc *sd, const orc_texture *textures, const rt_camera_desc *camera, uint32_t frame_index,
                                    int32_t accumulate, uint64_t seed, uint32_t num_chunks, float *accum, uint8_t *rgba) {
    Scene scene;
    if (!camera || !accum || !rgba || camera->width == 0 || camera->height == 0 || num_chunks == 0) return -1;
    if (!load_scene(sd, textures, scene)) return -1;
    const size_t total = (size_t)camera->width * camera->height;
    const size_t img_len = total * 4;
    const size_t img_chunk_size = (img_len / ((size_t)num_chunks * 4)) * 4;   // :105
    if (img_chunk_size == 0) return -2;                                       // :107 chunks_mut(0) panics
    const Vec3 position = ld3(camera->position);
    std::vector<std::vector<Vec4>> col;                                       // :109 one temporary Renderer per chunk
    size_t e0 = 0;
    for (size_t start = 0; start < img_len; start += img_chunk_size, e0++) {
        const size_t buf_len = std::min(img_chunk_size, img_len - start);     // :113
        const size_t acc_size = buf_len / 4;                                  // :115
        const size_t offset = e0 * acc_size;                                  // :117
        std::vector<Vec4> k(acc_size);                                        // :119-122 k.to_vec()
        for (size_t i = 0; i < acc_size; i++) k[i] = *reinterpret_cast<const Vec4 *>(accum + 4 * (offset + i));
        uint8_t *bytes = rgba + start;
        for (size_t pos = 0; pos < acc_size; pos++) {                         // render_chunk, :48-78
            const size_t src = pos + offset;                                  // :49 camera.ray_directions[pos + chunk.pixel_offset]
            const Vec3 ray_dir = camera_ray_dir(*camera, (uint32_t)(src % camera->width), (uint32_t)(src / camera->width));
            const uint32_t key = rng_path_key(seed, (uint32_t)src, frame_index);
            const Vec4 p = scene_pixel(scene, Ray{position, ray_dir}, key, nullptr);
            if (accumulate) {
                k[pos] = k[pos] + p;
                Vec4 a = k[pos];
                a = a / (float)frame_index;
                a = clamp01(a);
                to_rgba(a, bytes + 4 * pos);
            } else {
                k[pos] = clamp01(p);
                to_rgba(k[pos], bytes + 4 * pos);
            }
        }
        col.push_back(std::move(k));
    }
    size_t off = 0;                                                           // :138-143
    for (const auto &c : col) {
        for (size_t i = 0; i < c.size(); i++) *reinterpret_cast<Vec4 *>(accum + 4 * (off + i)) = c[i];
        off += c.size();
    }
    return 0;
}

void orc_camera_ray(const rt_camera_desc *camera, uint32_t x, uint32_t y, float dir[3]) {
    Vec3 d = camera_ray_dir(*camera, x, y);
    dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
}

// src/camera.rs:88-93 (Resize) and :117-123
void orc_camera_matrices(const float position[3], const float forward[3], const float up[3], float fov, float near_, float far_,
                         uint32_t width, uint32_t height, float inverse_perspective[16], float inverse_view[16]) {
    Mat4 perspective = mat4_perspective_rh(fov, (float)width / (float)height, near_, far_);
    mat4_to_cols_array(mat4_inverse(perspective), inverse_perspective);
    Vec3 pos = ld3(position), fwd = ld3(forward);
    Mat4 view = mat4_look_at_lh(pos, pos + fwd, ld3(up));
    mat4_to_cols_array(mat4_inverse(view), inverse_view);
}

void orc_object_transform(const rt_object *o, float transform[16], float inv_transform[16]) {
    Mat4 t = mat4_identity();
    Vec3 pos = ld3(o->position), rot = ld3(o->rotation_axis);
    bool has = true;
    switch (o->kind) {
        case RT_SPHERE:  // sphere.rs:61-70
            t = mat4_from_translation(pos) * mat4_from_rotation_x(rot.x * DEGREES) * mat4_from_rotation_y(rot.y * DEGREES) *
                mat4_from_rotation_z(rot.z * DEGREES) * mat4_from_scale(v3(o->radius, o->radius, o->radius));
            break;
        case RT_CUBOID:    // cuboid.rs:41-47
        case RT_CYLINDER:  // cylinder.rs:45-52 (no scale)
            t = mat4_from_translation(pos) * mat4_from_rotation_x(rot.x * DEGREES) * mat4_from_rotation_y(rot.y * DEGREES) *
                mat4_from_rotation_z(rot.z * DEGREES);
            break;
        case RT_CONE:  // cone.rs:44-51
            t = mat4_from_translation(pos) * mat4_from_rotation_x(rot.x * DEGREES) * mat4_from_rotation_y(rot.y * DEGREES) *
                mat4_from_rotation_z(rot.z * DEGREES) * mat4_from_scale(v3(o->radius, o->radius, o->height));
            break;
        default: has = false; break;
    }
    mat4_to_cols_array(t, transform);
    mat4_to_cols_array(has ? mat4_inverse(t) : mat4_identity(), inv_transform);
}

int32_t orc_intersect(const rt_object *obj, const float origin[3], const float dir[3], float out[9]) {
    Obj o = load_obj(*obj);
    RayHit h;
    if (!ray_hit(Ray{ld3(origin), ld3(dir)}, o, h, nullptr)) return 0;
    out[0] = h.distance; out[1] = h.point.x; out[2] = h.point.y; out[3] = h.point.z;
    out[4] = h.normal.x; out[5] = h.normal.y; out[6] = h.normal.z; out[7] = h.u; out[8] = h.v;
    return 1;
}

int32_t orc_sdfs(const rt_scene_desc *sd, const float origin[3], const float dir[3], float t, float out[11]) {
    Scene scene;
    if (!load_scene(sd, nullptr, scene)) return -1;
    RayMarching rm{scene, nullptr};
    uint32_t idx; RayMarchingHit h;
    rm.sdfs(Ray{ld3(origin), ld3(dir)}, t, idx, h);
    out[0] = h.distance; out[1] = h.albedo.x; out[2] = h.albedo.y; out[3] = h.albedo.z;
    out[4] = h.transformed_ray.origin.x; out[5] = h.transformed_ray.origin.y; out[6] = h.transformed_ray.origin.z;
    out[7] = h.transformed_ray.direction.x; out[8] = h.transformed_ray.direction.y; out[9] = h.transformed_ray.direction.z;
    out[10] = (float)idx;
    return 0;
}

void orc_texture_from_uv(const orc_texture *tex, float u, float v, float rgb[3]) {
    Vec3 c = texture_from_uv(*tex, u, v);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}

float orc_powf_det(float x, float y) { return orc_powf_spec(x, y); }
float orc_atanf_det(float x) { return orc_atanf_spec(x); }
float orc_powf_host(float x, float y) { return ::powf(x, y); }
void orc_powf_det_array(const float *x, const float *y, float *out, size_t n) { for (size_t i = 0; i < n; i++) out[i] = orc_powf_spec(x[i], y[i]); }
void orc_atanf_det_array(const float *x, float *out, size_t n) { for (size_t i = 0; i < n; i++) out[i] = orc_atanf_spec(x[i]); }
float orc_atanf_host(float x) { return ::atanf(x); }

uint32_t orc_rng_path_key(uint64_t seed, uint32_t pixel_index, uint32_t frame_index) { return rng_path_key(seed, pixel_index, frame_index); }
uint32_t orc_rng_child_key(uint32_t key, uint32_t branch) { return rng_child_key(key, branch); }
float orc_rng_uniform(uint32_t key, uint32_t draw, float low, float high) { return rng_uniform(key, draw, low, high); }

int32_t orc_uses_host_libm(void) {
#ifdef ORC_HOST_LIBM
    return 1;
#else
    return 0;
#endif
}

}  // extern "C"
