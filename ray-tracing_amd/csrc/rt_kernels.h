// rt_kernels.h — hand-written gfx950 kernels for the per-pixel hot path of mariusdanciu/ray-tracing.
//
//   rt_trace_kernel   Scene::pixel with ray_marching == false: camera ray generation
//                     (src/camera.rs:128-157), RayTracing::{albedo,color,color_diffuse,light,trace_ray}
//                     (src/ray_tracing/ray_tracing.rs:17-241), the six Intersection impls, Blinn-Phong,
//                     reflection / refraction rays, Fresnel, nearest-texel sampling, then
//                     render_chunk's accumulate / clamp / to_rgba (src/renderer.rs:59-75).
//   rt_march_kernel   Scene::pixel with ray_marching == true: RayMarching::{albedo,march_ray,sdfs,normal,
//                     occlusion,light} (src/ray_marching/ray_marching.rs:24-253) and the sdf() impls.
//
// Execution model: one lane per pixel, one wavefront per 8x8 pixel tile.  The closest-hit loop runs over
// the object array with a wave-uniform index, so object records arrive through scalar loads and the
// per-kind switch does not diverge.  The reference's recursion is an explicit state machine: a lane is
// either descending (tracing one ray and shading its hit) or unwinding pending combines.  Reflective
// frames (p_light, roughness) live in LDS, one column per lane; refractive frames (two children) live
// in a small per-lane array.  Intersection candidates only produce t; normal / UV / point are computed
// once for the closest one, with the same arithmetic the reference applies to every candidate.
#pragma once
#include "../../include/rt_mi355x.h"
#include "rt_dev_math.h"
#include "rt_dev_scene.h"

#pragma clang fp contract(off)

#define RT_EPSILON 0.0001f                        // src/ray.rs:9
#define RT_INV_PI (1.0f / 3.14159274101257324f)   // src/utils/sphere.rs:13
#define RT_F32_MAX 3.40282347e+38f

// ------------------------------------------------------------------------------------------------
// per-lane counters, summed per workgroup at kernel end
// ------------------------------------------------------------------------------------------------
template <bool EVENTS>
struct Cnt {
    uint32_t rays;
    uint32_t ev[EVENTS ? RT_CNT_COUNT : 1];
    RT_DEV void clear() {
        rays = 0;
#pragma unroll
        for (int i = 0; i < (EVENTS ? RT_CNT_COUNT : 1); i++) ev[i] = 0;
    }
    RT_DEV void add(int slot) { if (EVENTS) ev[slot] += 1u; }
    RT_DEV void ray(int slot) { rays += 1u; if (EVENTS) ev[slot] += 1u; }
};

// Diagnostic build (-DRT_DIAG): wave-cycle stamps around the sections of rt_trace_kernel.  Never enabled in the
// shipped library; the stamps serialise the instruction stream, so only the shares are meaningful.
enum { DG_TOTAL = 0, DG_RAYGEN, DG_TRACE, DG_FINALIZE, DG_LIGHTS, DG_SHADOW_TRACE, DG_POW, DG_BLINN, DG_REFLRAY, DG_UNWIND, DG_STORE, DG_TEX, DG_N };
#ifdef RT_DIAG
struct Diag { uint32_t acc[DG_N]; };
__device__ Diag *g_diag_unused;
#define DG_DECL Diag dg; for (int i_ = 0; i_ < DG_N; i_++) dg.acc[i_] = 0u;
#define DG_BEGIN(sec) const unsigned long long dgt_##sec = __builtin_amdgcn_s_memtime();
#define DG_END(sec) dg.acc[sec] += (uint32_t)(__builtin_amdgcn_s_memtime() - dgt_##sec);
#define DG_ARG , Diag &dg
#define DG_PASS , dg
#else
#define DG_DECL
#define DG_BEGIN(sec)
#define DG_END(sec)
#define DG_ARG
#define DG_PASS
#endif

RT_DEV f3 ld3(const float *p) { return mk3(p[0], p[1], p[2]); }
// (M * vec4(v, 0)).xyz and (M * vec4(v, 1)).xyz in glam's order: ((c0*x + c1*y) + c2*z) + c3*w
RT_DEV f3 xf_dir(const DevXform &m, f3 v) {
    return mk3(((m.c0[0] * v.x + m.c1[0] * v.y) + m.c2[0] * v.z) + m.z[0], ((m.c0[1] * v.x + m.c1[1] * v.y) + m.c2[1] * v.z) + m.z[1],
               ((m.c0[2] * v.x + m.c1[2] * v.y) + m.c2[2] * v.z) + m.z[2]);
}
RT_DEV f3 xf_pt(const DevXform &m, f3 v) {
    return mk3(((m.c0[0] * v.x + m.c1[0] * v.y) + m.c2[0] * v.z) + m.c3[0], ((m.c0[1] * v.x + m.c1[1] * v.y) + m.c2[1] * v.z) + m.c3[1],
               ((m.c0[2] * v.x + m.c1[2] * v.y) + m.c2[2] * v.z) + m.c3[2]);
}

struct Mtl {
    float ambience, diffuse, specular, shininess;
    f3 albedo;
    int texture, kind;
    float roughness, transparency, refraction_index, reflectivity, emission_power;
};
RT_DEV Mtl load_mtl(const DevMaterial &m) {
    Mtl r;
    r.ambience = m.ambience; r.diffuse = m.diffuse; r.specular = m.specular; r.shininess = m.shininess;
    r.albedo = ld3(m.albedo); r.texture = m.texture; r.kind = m.kind; r.roughness = m.roughness;
    r.transparency = m.transparency; r.refraction_index = m.refraction_index; r.reflectivity = m.reflectivity;
    r.emission_power = m.emission_power;
    return r;
}

struct HitRec { f3 point, normal; float u, v; };

// ------------------------------------------------------------------------------------------------
// Texture::from_uv (src/objects.rs:121-147)
// ------------------------------------------------------------------------------------------------
RT_DEV float rt_textel(float p) {
    if (p < 0.0f) return 1.0f - rt_abs(__builtin_ceilf(p) - p);
    else if (p > 1.0f) return p - __builtin_floorf(p);
    return p;
}
RT_DEV f3 tex_from_uv(const DevTexture &t, float u, float v) {
    float uu = rt_textel(u), vv = rt_textel(v);
    uint32_t x = rt_as_u32((float)(t.width - 1u) * uu);
    uint32_t y = rt_as_u32((float)(t.height - 1u) * vv);
    uint32_t px = t.texels[y * t.width + x];
    const float RGB_RATIO = 1.0f / 255.0f;
    return mk3((float)(px & 255u) * RGB_RATIO, (float)((px >> 8) & 255u) * RGB_RATIO, (float)((px >> 16) & 255u) * RGB_RATIO);
}

// ------------------------------------------------------------------------------------------------
// candidate phase of the Intersection impls: Some/None and `distance` only
// ------------------------------------------------------------------------------------------------
// The ray in the sphere's object space (src/utils/sphere.rs:103-105), with a = rd . rd and cc = ro . ro of the quadratic.
// diag (DevObject::diag, spheres without rotation): the 3x3 part of inv_transform is diagonal and every other entry of its
// first three rows is (+-)0, so a row of the reference's product, ((d * v.x + 0 * v.y) + 0 * v.z) + c, is d * v.x + c bit for
// bit -- unless v has a non-finite component (0 * inf = NaN reaches every row) or the row's result is a zero (the sign of a
// zero sum depends on the signs of the zero terms).  Either case shows in values computed anyway: a non-finite component of
// v makes its own row, and with it a or cc, non-finite.  Such lanes take the full product.  Returns whether the short form held.
RT_DEV bool sphere_local_ray(const DevObject &o, bool diag, f3 ro_w, f3 rd_w, f3 &ro, f3 &rd, float &a, float &cc) {
    if (diag) {
        rd = mk3(o.inv.c0[0] * rd_w.x, o.inv.c1[1] * rd_w.y, o.inv.c2[2] * rd_w.z);
        ro = mk3(o.inv.c0[0] * ro_w.x + o.inv.c3[0], o.inv.c1[1] * ro_w.y + o.inv.c3[1], o.inv.c2[2] * ro_w.z + o.inv.c3[2]);
        a = dot(rd, rd);
        cc = dot(ro, ro);
        const float smallest = rt_min(rt_min(rt_min(rt_abs(rd.x), rt_abs(rd.y)), rt_abs(rd.z)), rt_min(rt_min(rt_abs(ro.x), rt_abs(ro.y)), rt_abs(ro.z)));
        if (smallest > 0.0f && a + cc < __builtin_inff()) return true;   // a, cc >= 0, or NaN
    }
    rd = xf_dir(o.inv, rd_w); ro = xf_pt(o.inv, ro_w);
    a = dot(rd, rd);
    cc = dot(ro, ro);
    return false;
}
template <bool DIAG = false>
RT_DEV bool cand_sphere(const DevObject &o, f3 ro_w, f3 rd_w, float &t) {  // src/utils/sphere.rs:99-125
    f3 rd, ro;
    float a, cc;
    sphere_local_ray(o, DIAG, ro_w, rd_w, ro, rd, a, cc);
    float b = 2.0f * dot(ro, rd);
    float c = cc - 1.0f;
    float disc = b * b - 4.0f * a * c;
    if (disc < 0.0f) return false;
    t = (-b - rt_sqrt(disc)) / (2.0f * a);
    return true;
}
RT_DEV bool cand_plane(const DevObject &o, f3 ro, f3 rd, float &t) {  // src/utils/plane.rs:42-59
    f3 n = ld3(o.p), pt = ld3(o.p + 3);
    float denom = dot(rd, n);
    if (rt_abs(denom) < 1e-6f) return false;
    t = dot(pt - ro, n) / denom;
    if (t < 0.0f) return false;
    f3 hp = ro + rd * t;
    if (o.p[8] != 0.0f) {
        if (rt_abs(hp.z) > o.p[7] || rt_abs(hp.x) > o.p[6]) return false;
    }
    return true;
}
RT_DEV bool cand_cuboid(const DevObject &o, f3 ro_w, f3 rd_w, float &t) {  // src/utils/cuboid.rs:85-105
    f3 rd = xf_dir(o.inv, rd_w), ro = xf_pt(o.inv, ro_w);
    f3 inv = 1.0f / rd;
    f3 t_min = (ld3(o.p + 3) - ro) * inv;
    f3 t_max = (ld3(o.p + 6) - ro) * inv;
    f3 t_enter = vmin(t_min, t_max), t_exit = vmax(t_min, t_max);
    float t_near = rt_max(rt_max(t_enter.x, t_enter.y), t_enter.z);
    float t_far = rt_min(rt_min(t_exit.x, t_exit.y), t_exit.z);
    if (t_near > t_far || t_far < 0.0f) return false;
    t = t_near;
    return true;
}
// sub: 0 = lateral surface, 1 = cap
RT_DEV bool cand_cylinder(const DevObject &o, f3 ro_w, f3 rd_w, float &t, int &sub) {  // src/utils/cylinder.rs:91-158
    f3 rd3 = xf_dir(o.inv, rd_w), ro3 = xf_pt(o.inv, ro_w);
    float a = (rd3.x * rd3.x) + (rd3.y * rd3.y);
    float b = 2.0f * ((ro3.x * rd3.x) + (ro3.y * rd3.y));
    float c = ((ro3.x * ro3.x) + (ro3.y * ro3.y)) - 1.0f;
    float disc = b * b - 4.0f * a * c;
    if (disc > 0.0f) {
        float t1 = (-b - rt_sqrt(disc)) / (2.0f * a);
        f3 h = ro3 + rd3 * t1;
        if (rt_abs(h.z) < 0.5f) { t = t1; sub = 0; return true; }
    }
    float inv_rd3z = -1.0f / rd3.z;
    float t1 = (ro3.z - 0.5f) * inv_rd3z;
    float t2 = (ro3.z + 0.5f) * inv_rd3z;
    f3 h1 = ro3 + rd3 * t1, h2 = ro3 + rd3 * t2;
    bool v1 = ((h1.x * h1.x) + (h1.y * h1.y)) < 1.0f;
    bool v2 = ((h2.x * h2.x) + (h2.y * h2.y)) < 1.0f;
    sub = 1;
    if (v1 && v2) { t = (t1 < t2) ? t1 : t2; return true; }
    if (v1) { t = t1; return true; }
    if (v2) { t = t2; return true; }
    return false;
}
// sub: 0 = lateral surface, 1 = cap
RT_DEV bool cand_cone(const DevObject &o, f3 ro_w, f3 rd_w, float &t, int &sub) {  // src/utils/cone.rs:74-151
    f3 rd3 = xf_dir(o.inv, rd_w), ro3 = xf_pt(o.inv, ro_w);
    float a = ((rd3.x * rd3.x) + (rd3.y * rd3.y)) - rd3.z * rd3.z;
    float b = 2.0f * (((ro3.x * rd3.x) + (ro3.y * rd3.y)) - ro3.z * rd3.z);
    float c = ((ro3.x * ro3.x) + (ro3.y * ro3.y)) - ro3.z * ro3.z;
    float disc = b * b - 4.0f * a * c;
    bool have_cone = false;
    float t_cone = 0.0f;
    if (disc > 0.0f) {
        float t1 = (-b - rt_sqrt(disc)) / (2.0f * a);
        f3 h = ro3 + rd3 * t1;
        if (h.z > 0.0f && h.z < 1.0f) { have_cone = true; t_cone = t1; }
    }
    float t1 = (ro3.z - 1.0f) / -rd3.z;
    f3 h1 = ro3 + rd3 * t1;
    bool valid = ((h1.x * h1.x) + (h1.y * h1.y)) < 1.0f;
    if (valid) {
        if (have_cone && t_cone < t1) { t = t_cone; sub = 0; return true; }
        t = t1; sub = 1; return true;
    }
    if (have_cone) { t = t_cone; sub = 0; return true; }
    return false;
}
RT_DEV bool cand_triangle(const DevObject &o, f3 ro, f3 rd, float &t) {  // src/utils/triangle.rs:36-68
    f3 v1 = ld3(o.p), v2 = ld3(o.p + 3), v3 = ld3(o.p + 6);
    f3 edge_ab = v2 - v1, edge_ac = v3 - v1;
    f3 normal = cross(edge_ab, edge_ac);
    float denom = dot(rd, normal);
    if (rt_abs(denom) < 1e-6f) return false;
    t = dot(v1 - ro, normal) / denom;
    f3 hp = ro + rd * t;
    f3 edge_bc = v3 - v2, edge_ca = v1 - v3;
    f3 a_test = cross(edge_ab, hp - v1), b_test = cross(edge_bc, hp - v2), c_test = cross(edge_ca, hp - v3);
    bool a_n = dot(a_test, normal) > 0.0f, b_n = dot(b_test, normal) > 0.0f, c_n = dot(c_test, normal) > 0.0f;
    return a_n && b_n && c_n;
}

RT_DEV int test_slot(int kind) {
    switch (kind) {
        case RT_SPHERE: return RT_CNT_TEST_SPHERE; case RT_TRIANGLE: return RT_CNT_TEST_TRIANGLE;
        case RT_CUBOID: return RT_CNT_TEST_CUBOID; case RT_PLANE: return RT_CNT_TEST_PLANE;
        case RT_CYLINDER: return RT_CNT_TEST_CYLINDER; default: return RT_CNT_TEST_CONE;
    }
}

// RayTracing::trace_ray (src/ray_tracing/ray_tracing.rs:59-78): linear scan, accept 0 < t < closest (strict),
// so the lowest index wins ties.  Returns the object index or -1.
// SP: the host has checked that every object of the scene is a sphere or a plane; the other candidate tests are left out.
template <bool EVENTS, bool SP = false>
RT_DEV int trace_closest(const DevScene &S, f3 ro, f3 rd, float &t_out, int &sub_out, Cnt<EVENTS> &cnt, int ray_slot) {
    const uint32_t n = S.n_objects;
    if (n == 0u) return -1;
    cnt.ray(ray_slot);
    float closest = RT_F32_MAX;
    int best = -1, best_sub = 0;
    const float rd2 = dot(rd, rd);
    // incoherent bounce rays of the diffuse mode almost never reject wave-wide: there the test is pure overhead
    const bool cull = !EVENTS && (S.diffuse == 0u || __builtin_amdgcn_readfirstlane(ray_slot) == RT_CNT_RAYS_PRIMARY);
    for (uint32_t i = 0; i < n; i++) {
        const DevObject &o = S.objects[i];
        if (cull && o.bounded) {
            // Wave-level rejection (timed variant only; the counting variant runs every candidate test the reference runs).
            // The object lies inside the sphere (c, R): a ray whose line passes farther than R from c, or that starts
            // outside the sphere with c behind it, can only report "no hit" or a negative distance, which the acceptance
            // rule below discards anyway.  |oc x rd|^2 = |oc|^2 |rd|^2 - (oc . rd)^2; the 1e-5 |oc|^2 term covers the
            // rounding of that difference.  The candidate test is skipped only when every lane of the wavefront rejects.
            const f3 oc = ld3(o.bs) - ro;
            const float oc2 = dot(oc, oc), b = dot(oc, rd);
            const float r2 = o.bs[3] + 1e-5f * oc2;
            const bool miss = (oc2 * rd2 - b * b > r2 * rd2) || (b < 0.0f && oc2 > r2);
            if (__ballot(!miss) == 0ull) continue;
        }
        float t = 0.0f;
        int sub = 0;
        bool ok = false;
        switch (o.kind) {
            case RT_SPHERE: cnt.add(RT_CNT_TEST_SPHERE); ok = o.diag ? cand_sphere<true>(o, ro, rd, t) : cand_sphere<false>(o, ro, rd, t); break;
            case RT_TRIANGLE: if (!SP) { cnt.add(RT_CNT_TEST_TRIANGLE); ok = cand_triangle(o, ro, rd, t); } break;
            case RT_CUBOID: if (!SP) { cnt.add(RT_CNT_TEST_CUBOID); ok = cand_cuboid(o, ro, rd, t); } break;
            case RT_PLANE: cnt.add(RT_CNT_TEST_PLANE); ok = cand_plane(o, ro, rd, t); break;
            case RT_CYLINDER: if (!SP) { cnt.add(RT_CNT_TEST_CYLINDER); ok = cand_cylinder(o, ro, rd, t, sub); } break;
            case RT_CONE: if (!SP) { cnt.add(RT_CNT_TEST_CONE); ok = cand_cone(o, ro, rd, t, sub); } break;
            default: break;  // Union / Substraction: Ray::hit's `_ => None` (src/ray.rs:175)
        }
        if (ok && t > 0.0f && t < closest) { closest = t; best = (int)i; best_sub = sub; }
    }
    t_out = closest;
    sub_out = best_sub;
    return best;
}

// One candidate test with a per-lane object (the object index differs between lanes: vector loads, divergent switch).
// `kind` comes from a hierarchy leaf: the object's kind, or RT_LEAF_DIAG_SPHERE for a sphere with DevObject::diag set
// (composites never are leaves, so their codes are free there).
template <bool EVENTS>
RT_DEV bool cand_any(int kind, const DevObject &o, f3 ro, f3 rd, float &t, int &sub, Cnt<EVENTS> &cnt) {
    sub = 0;
    switch (kind) {
        case RT_SPHERE: cnt.add(RT_CNT_TEST_SPHERE); return cand_sphere<false>(o, ro, rd, t);
        case RT_LEAF_DIAG_SPHERE: cnt.add(RT_CNT_TEST_SPHERE); return cand_sphere<true>(o, ro, rd, t);
        case RT_TRIANGLE: cnt.add(RT_CNT_TEST_TRIANGLE); return cand_triangle(o, ro, rd, t);
        case RT_CUBOID: cnt.add(RT_CNT_TEST_CUBOID); return cand_cuboid(o, ro, rd, t);
        case RT_PLANE: cnt.add(RT_CNT_TEST_PLANE); return cand_plane(o, ro, rd, t);
        case RT_CYLINDER: cnt.add(RT_CNT_TEST_CYLINDER); return cand_cylinder(o, ro, rd, t, sub);
        case RT_CONE: cnt.add(RT_CNT_TEST_CONE); return cand_cone(o, ro, rd, t, sub);
        default: return false;
    }
}

// trace_ray through the bounding-volume hierarchy.  Same candidate tests and the same acceptance rule as the linear
// scan; because objects are visited in a different order, "first index wins ties" is applied explicitly.
template <bool EVENTS>
RT_DEV int trace_closest_bvh(const DevScene &S, f3 ro, f3 rd, float &t_out, int &sub_out, Cnt<EVENTS> &cnt, int ray_slot) {
    cnt.ray(ray_slot);
    float closest = RT_F32_MAX;
    int best = -1, best_sub = 0;
    for (uint32_t k = 0; k < S.n_unbounded; k++) {  // planes and cones (no usable bounds, see rt_object_bounds): wave-uniform loop
        const uint32_t idx = S.unbounded[k];
        const DevObject &o = S.objects[idx];
        float t = 0.0f;
        int sub = 0;
        bool ok;
        if (o.kind == RT_PLANE) { cnt.add(RT_CNT_TEST_PLANE); ok = cand_plane(o, ro, rd, t); }
        else { cnt.add(RT_CNT_TEST_CONE); ok = cand_cone(o, ro, rd, t, sub); }
        if (ok && t > 0.0f && (t < closest || (t == closest && (int)idx < best))) { closest = t; best = (int)idx; best_sub = sub; }
    }
    const f3 inv = 1.0f / rd;
    const int n_nodes = (int)S.n_bvh_nodes;
    // one depth-first layout per sign octant of the direction: children are met near-first along each split axis
    const uint32_t oct = (rd.x < 0.0f ? 1u : 0u) | (rd.y < 0.0f ? 2u : 0u) | (rd.z < 0.0f ? 4u : 0u);
    const float4 *nodes = reinterpret_cast<const float4 *>(S.bvh_nodes) + (size_t)oct * (size_t)n_nodes * 2u;
    int i = 0;
    while (i < n_nodes) {
        const float4 a = nodes[2 * i];
        const float4 b = nodes[2 * i + 1];
        cnt.add(RT_CNT_BVH_NODE);
        // slab test; fminf/fmaxf drop the NaNs of 0 * inf so such slabs do not cull
        const float tx0 = (a.x - ro.x) * inv.x, tx1 = (b.x - ro.x) * inv.x;
        const float ty0 = (a.y - ro.y) * inv.y, ty1 = (b.y - ro.y) * inv.y;
        const float tz0 = (a.z - ro.z) * inv.z, tz1 = (b.z - ro.z) * inv.z;
        const float tmin = rt_max(rt_max(rt_min(tx0, tx1), rt_min(ty0, ty1)), rt_max(rt_min(tz0, tz1), 0.0f));
        const float tmax = rt_min(rt_min(rt_max(tx0, tx1), rt_max(ty0, ty1)), rt_max(tz0, tz1));
        const bool hit = tmin <= tmax && tmin <= closest;
        const int leaf = __float_as_int(b.w);
        if (hit && leaf >= 0) {  // one object per leaf: kind in the top bits, so its record is the only dependent load
            const int idx = leaf & 0x0FFFFFFF;
            float t = 0.0f;
            int sub = 0;
            if (cand_any<EVENTS>(leaf >> 28, S.objects[idx], ro, rd, t, sub, cnt) && t > 0.0f && (t < closest || (t == closest && idx < best))) {
                closest = t; best = idx; best_sub = sub;
            }
        }
        i = hit ? i + 1 : __float_as_int(a.w);
    }
    t_out = closest;
    sub_out = best_sub;
    return best;
}

template <bool EVENTS, bool BVH, bool SP = false>
RT_DEV int trace_ray(const DevScene &S, f3 ro, f3 rd, float &t_out, int &sub_out, Cnt<EVENTS> &cnt, int ray_slot) {
    if (BVH) return trace_closest_bvh<EVENTS>(S, ro, rd, t_out, sub_out, cnt, ray_slot);
    return trace_closest<EVENTS, SP>(S, ro, rd, t_out, sub_out, cnt, ray_slot);
}

// ------------------------------------------------------------------------------------------------
// finalise phase: RayHit of the closest candidate, same arithmetic as the reference's intersect()
// ------------------------------------------------------------------------------------------------
RT_DEV void fin_sphere(const DevObject &o, f3 ro_w, f3 rd_w, float t1, bool need_uv, HitRec &h) {  // sphere.rs:127-146
    f3 rd, ro;
    float a_unused, cc_unused;
    const bool diag = o.diag != 0;
    const bool short_form = sphere_local_ray(o, diag, ro_w, rd_w, ro, rd, a_unused, cc_unused);
    f3 l = ro + rd * t1;
    // the normal's transform by the diagonal, under the same two conditions (ro, rd and t1 are finite here, so a component
    // of l is finite or an infinity, never a NaN)
    f3 n = splat3(0.0f);
    bool n_done = false;
    if (short_form) {
        n = mk3(o.fwd.c0[0] * l.x, o.fwd.c1[1] * l.y, o.fwd.c2[2] * l.z);
        const float n_min = rt_min(rt_min(rt_abs(n.x), rt_abs(n.y)), rt_abs(n.z)), n_max = rt_max(rt_max(rt_abs(n.x), rt_abs(n.y)), rt_abs(n.z));
        n_done = n_min > 0.0f && n_max < __builtin_inff();
    }
    if (!n_done) n = xf_dir(o.fwd, l);
    h.normal = normalize(n);
    h.point = ro_w + rd_w * t1;
    if (need_uv) {
        float u = rt_atanf((l.x * l.x + l.y * l.y) / l.z);
        float v = rt_atanf(l.y / l.x);
        h.u = v * RT_INV_PI;
        h.v = u * RT_INV_PI;
    }
}
RT_DEV void fin_plane(const DevObject &o, f3 ro, f3 rd, float t, HitRec &h) {  // plane.rs:53-73
    f3 n = ld3(o.p);
    f3 hp = ro + rd * t;
    float sign = -1.0f;
    if (dot(rd, n) < 0.0f) sign = 1.0f;
    h.point = hp;
    h.normal = sign * n;
    h.u = hp.x * 0.1f;
    h.v = hp.z * 0.1f;
}
RT_DEV void fin_cuboid(const DevObject &o, f3 ro_w, f3 rd_w, float t_near, HitRec &h) {  // cuboid.rs:92-126
    f3 rd = xf_dir(o.inv, rd_w), ro = xf_pt(o.inv, ro_w);
    f3 inv = 1.0f / rd;
    f3 t_min = (ld3(o.p + 3) - ro) * inv;
    f3 t_max = (ld3(o.p + 6) - ro) * inv;
    f3 t_enter = vmin(t_min, t_max);
    f3 sg = mk3(rt_signum(rd.x), rt_signum(rd.y), rt_signum(rd.z));
    f3 st = mk3(t_enter.x < t_near ? 0.0f : 1.0f, t_enter.y < t_near ? 0.0f : 1.0f, t_enter.z < t_near ? 0.0f : 1.0f);
    f3 a = (-sg) * st;
    h.normal = xf_dir(o.fwd, a);
    h.point = ro_w + rd_w * t_near;
    f3 opos = xf_pt(o.inv, h.point);
    float ax = rt_abs(a.x), ay = rt_abs(a.y), az = rt_abs(a.z);
    h.u = (ax * opos.y + ay * opos.z) + az * opos.x;
    h.v = (ax * opos.z + ay * opos.x) + az * opos.y;
}
RT_DEV void fin_cylinder(const DevObject &o, f3 ro_w, f3 rd_w, float t, int sub, bool need_uv, HitRec &h) {  // cylinder.rs:108-173
    f3 rd3 = xf_dir(o.inv, rd_w), ro3 = xf_pt(o.inv, ro_w);
    f3 ht = ro3 + rd3 * t;
    h.point = ro_w + rd_w * t;
    if (sub == 0) {
        h.normal = normalize(xf_dir(o.fwd, mk3(ht.x, ht.y, 0.0f)));
        if (need_uv) h.u = rt_atanf(ht.y / ht.x);
        h.v = ht.z * 2.0f;
    } else {
        h.normal = normalize(xf_dir(o.fwd, mk3(0.0f, 0.0f, ht.z)));
        h.u = ht.x;
        h.v = ht.y;
    }
}
RT_DEV void fin_cone(const DevObject &o, f3 ro_w, f3 rd_w, float t, int sub, bool need_uv, HitRec &h) {  // cone.rs:93-148
    f3 rd3 = xf_dir(o.inv, rd_w), ro3 = xf_pt(o.inv, ro_w);
    f3 ht = ro3 + rd3 * t;
    h.point = ro_w + rd_w * t;
    if (sub == 0) {
        f3 n = mk3(ht.x, ht.y, -rt_sqrt(ht.x * ht.x + ht.y * ht.y));
        h.normal = normalize(xf_dir(o.fwd, n));
        if (need_uv) h.u = rt_atanf(ht.y / ht.x);
        h.v = ht.z * 2.0f + 1.0f;
    } else {
        h.normal = xf_dir(o.fwd, mk3(0.0f, 0.0f, 1.0f));
        h.u = ht.x;
        h.v = ht.y;
    }
}
RT_DEV void fin_triangle(const DevObject &o, f3 ro, f3 rd, float t, HitRec &h) {  // triangle.rs:49-88
    f3 v1 = ld3(o.p), v2 = ld3(o.p + 3), v3 = ld3(o.p + 6);
    f3 edge_ab = v2 - v1, edge_ac = v3 - v1;
    f3 n = cross(edge_ab, edge_ac);
    f3 hp = ro + rd * t;
    f3 edge_ca = v1 - v3;
    float sign = -rt_signum(dot(rd, n));
    h.u = dot(cross(edge_ab, hp - v1), n);
    h.v = dot(cross(edge_ca, hp - v3), n);
    h.point = hp;
    h.normal = normalize(sign * n);
}

template <bool EVENTS, bool SP = false>
RT_DEV void finalize_one(const DevObject &o, const DevMaterial &dm, f3 ro, f3 rd, float t, int sub, HitRec &h, Mtl &m, Cnt<EVENTS> &cnt) {
    m = load_mtl(dm);
    const bool need_uv = dm.texture >= 0 && dm.kind == RT_REFLECTIVE;
    if (SP) {
        if (o.kind == RT_SPHERE) { cnt.add(RT_CNT_HIT_SPHERE); fin_sphere(o, ro, rd, t, need_uv, h); }
        else { cnt.add(RT_CNT_HIT_PLANE); fin_plane(o, ro, rd, t, h); }
        return;
    }
    switch (o.kind) {
        case RT_SPHERE: cnt.add(RT_CNT_HIT_SPHERE); fin_sphere(o, ro, rd, t, need_uv, h); break;
        case RT_TRIANGLE: cnt.add(RT_CNT_HIT_TRIANGLE); fin_triangle(o, ro, rd, t, h); break;
        case RT_CUBOID: cnt.add(RT_CNT_HIT_CUBOID); fin_cuboid(o, ro, rd, t, h); break;
        case RT_PLANE: cnt.add(RT_CNT_HIT_PLANE); fin_plane(o, ro, rd, t, h); break;
        case RT_CYLINDER: cnt.add(RT_CNT_HIT_CYLINDER); fin_cylinder(o, ro, rd, t, sub, need_uv, h); break;
        default: cnt.add(RT_CNT_HIT_CONE); fin_cone(o, ro, rd, t, sub, need_uv, h); break;
    }
}

// Completes the closest hit of every lane.  Small scenes: lanes of a wave usually share a handful of winners, so the
// loop visits each distinct winner once with a wave-uniform object (scalar loads, no divergent switch).  Large scenes
// (BVH): winners are mostly distinct, so every lane loads its own object.
template <bool EVENTS, bool BVH, bool SP = false>
RT_DEV void finalize_hit(const DevScene &S, f3 ro, f3 rd, float t, int obj, int sub, HitRec &h, Mtl &m, Cnt<EVENTS> &cnt) {
    h.u = 0.0f; h.v = 0.0f;
    if (BVH) {
        const DevObject &o = S.objects[obj];
        finalize_one<EVENTS>(o, S.materials[o.material], ro, rd, t, sub, h, m, cnt);
        return;
    }
    int pending = obj;
    while (true) {
        int cur = __builtin_amdgcn_readfirstlane(pending);
        if (pending == cur) {
            const DevObject &o = S.objects[cur];
            finalize_one<EVENTS, SP>(o, S.materials[o.material], ro, rd, t, sub, h, m, cnt);
            break;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// shading
// ------------------------------------------------------------------------------------------------
// Ray::blinn_phong (src/ray.rs:68-85) with light.direction(hit.point) evaluated once (it is pure).
RT_DEV f3 blinn_phong(f3 ray_dir, f3 normal, f3 ldir, f3 color, const Mtl &m, PowTab pow_tab DG_ARG) {
    float coeff = dot(normal, -ldir);
    f3 ambience = m.ambience * color;
    f3 diffuse = m.diffuse * rt_max(coeff, 0.0f) * color;
    f3 half_angle = normalize(-ray_dir - ldir);
    DG_BEGIN(DG_POW)
    float shininess = rt_powf(rt_max(dot(normal, half_angle), 0.0f), m.shininess, pow_tab);
    DG_END(DG_POW)
    f3 specular = m.specular * shininess * color;
    return ambience + diffuse + specular;
}

// LightSource::{direction, distance} (src/light.rs:42-92): one subtraction / length feeds both.
RT_DEV void light_dir_dist(const DevLight &l, f3 point, f3 &dir, float &dist) {
    if (l.kind == RT_DIRECTIONAL) { dir = ld3(l.vec); dist = 1.0f; }
    else { f3 v = point - ld3(l.vec); float len = length(v); dir = v * (1.0f / len); dist = len; }
}

// RayTracing::light (src/ray_tracing/ray_tracing.rs:28-57).  The reference first sums all lights and then, per light,
// halves the sum if the shadow ray towards it hits another object.  Every halving is the same exact operation (x 0.5),
// so only their number matters: one loop shades a light and traces its shadow ray (its direction is computed once),
// and the halvings are applied after the loop.
template <bool EVENTS, bool BVH, bool SP = false>
RT_DEV f3 shade_lights(const DevScene &S, f3 ray_dir, const HitRec &h, f3 albedo, const Mtl &m, int obj_index, PowTab pow_tab, Cnt<EVENTS> &cnt DG_ARG) {
    cnt.add(RT_CNT_LIGHT_CALL);
    f3 l_acc = splat3(0.0f);
    const uint32_t nl = S.n_lights;
    const bool shadows = S.shadow_casting != 0u;
    const f3 so = h.point + RT_EPSILON * h.normal;
    uint32_t occluded = 0u;
    for (uint32_t i = 0; i < nl; i++) {
        const DevLight &l = S.lights[i];
        cnt.add(RT_CNT_LIGHT_SHADE);
        f3 ldir; float ldist;
        DG_BEGIN(DG_BLINN)
        light_dir_dist(l, h.point, ldir, ldist);
        f3 k = blinn_phong(ray_dir, h.normal, ldir, albedo, m, pow_tab DG_PASS);
        l_acc = l_acc + (k / (ldist * ldist)) * ld3(l.albedo) * l.intensity;
        DG_END(DG_BLINN)
        if (shadows) {
            float st; int ssub;
            DG_BEGIN(DG_SHADOW_TRACE)
            int idx = trace_ray<EVENTS, BVH, SP>(S, so, -ldir, st, ssub, cnt, RT_CNT_RAYS_SHADOW);
            DG_END(DG_SHADOW_TRACE)
            if (idx >= 0 && idx != obj_index) occluded += 1u;
        }
    }
    for (uint32_t j = 0; j < occluded; j++) l_acc = l_acc * 0.5f;
    DG_BEGIN(DG_POW)
    f3 res = vpowf(l_acc, 0.4166f, pow_tab);
    DG_END(DG_POW)
    return res;
}

RT_DEV f3 rt_reflect(f3 vec, f3 normal) { return vec - (2.0f * dot(vec, normal)) * normal; }  // geometry.rs:7-9

// Material::fresnel (src/objects.rs:178-205); returns kr
RT_DEV float rt_fresnel(f3 incident, f3 normal, float refraction_index, float reflectivity) {
    float n2 = refraction_index, n1 = 1.0f;
    float r0 = (n1 - n2) / (n1 + n2);
    r0 = r0 * r0;
    float cos_x = dot(normal, -incident);
    if (n1 > n2) {
        float n = n1 / n2;
        float sin_t2 = n * n * (1.0f - cos_x * cos_x);
        if (sin_t2 > 1.0f) return 1.0f;
        cos_x = rt_sqrt(1.0f - sin_t2);
    }
    float x = 1.0f - cos_x;
    float ret = r0 + (1.0f - r0) * x * x * x * x * x;
    return reflectivity + (1.0f - reflectivity) * ret;
}

// Ray::refraction_ray (src/ray.rs:138-165)
RT_DEV bool rt_refraction_ray(f3 ray_dir, const HitRec &h, float refraction_index, f3 &o, f3 &d) {
    f3 normal = h.normal;
    float eta_t = refraction_index, eta_i = 1.0f;
    float c1 = dot(ray_dir, h.normal);
    if (c1 < 0.0f) { c1 = -c1; }
    else { normal = -normal; eta_i = eta_t; eta_t = 1.0f; }
    float eta = eta_i / eta_t;
    float k = 1.0f - eta * eta * (1.0f - c1 * c1);
    if (k < 0.0f) return false;
    float c2 = rt_sqrt(k);
    d = eta * ray_dir + normal * (eta * c1 - c2);
    o = h.point - RT_EPSILON * normal;
    return true;
}

// Ray::reflection_ray (src/ray.rs:101-136)
RT_DEV void rt_reflection_ray(f3 ray_dir, const HitRec &h, float roughness, uint32_t key, bool diffuse, bool jitter, f3 &o, f3 &d) {
    if (!diffuse) {
        f3 factor = splat3(0.0f);
        if (jitter) factor = roughness * mk3(rt_rng_uniform(key, 0u, -0.5f, 0.5f), rt_rng_uniform(key, 1u, -0.5f, 0.5f), rt_rng_uniform(key, 2u, -0.5f, 0.5f));
        d = normalize(rt_reflect(ray_dir, h.normal + factor));
    } else {
        f3 rnd = mk3(rt_rng_uniform(key, 0u, -1.0f, 1.0f), rt_rng_uniform(key, 1u, -1.0f, 1.0f), rt_rng_uniform(key, 2u, -1.0f, 1.0f));
        d = normalize(h.normal + rnd);
    }
    o = h.point + h.normal * RT_EPSILON;
}

// Camera::calculate_ray_directions for one pixel (src/camera.rs:137-148)
RT_DEV f3 camera_ray(const DevCamera &C, uint32_t x, uint32_t y) {
    float p_ndc_x = (float)x * C.inv_w, p_ndc_y = (float)y * C.inv_h;
    float sx = 2.0f * p_ndc_x - 1.0f, sy = 2.0f * p_ndc_y - 1.0f;
    // inverse_perspective * vec4(sx, sy, 1, 1); the products by 1.0 are exact
    float tx = ((C.ip_c0[0] * sx + C.ip_c1[0] * sy) + C.ip_c2[0]) + C.ip_c3[0];
    float ty = ((C.ip_c0[1] * sx + C.ip_c1[1] * sy) + C.ip_c2[1]) + C.ip_c3[1];
    float tz = ((C.ip_c0[2] * sx + C.ip_c1[2] * sy) + C.ip_c2[2]) + C.ip_c3[2];
    float tw = ((C.ip_c0[3] * sx + C.ip_c1[3] * sy) + C.ip_c2[3]) + C.ip_c3[3];
    f3 v = mk3(tx / tw, ty / tw, tz / tw);
    f3 world = xf_dir(C.iv, v);
    return normalize(world - ld3(C.position));
}

// render_chunk's tail (src/renderer.rs:59-75) + to_rgba (:29-36)
RT_DEV void store_pixel(const DevFrame &F, uint32_t pos, f3 p) {
    float4 *acc = reinterpret_cast<float4 *>(F.accum) + pos;
    float4 c;
    if (F.accumulate) {
        float4 a = *acc;
        a.x = a.x + p.x; a.y = a.y + p.y; a.z = a.z + p.z; a.w = a.w + 1.0f;
        *acc = a;
        float fi = (float)F.frame_index;
        c.x = rt_clamp01_sse(a.x / fi); c.y = rt_clamp01_sse(a.y / fi); c.z = rt_clamp01_sse(a.z / fi); c.w = rt_clamp01_sse(a.w / fi);
    } else {
        c.x = rt_clamp01_sse(p.x); c.y = rt_clamp01_sse(p.y); c.z = rt_clamp01_sse(p.z); c.w = rt_clamp01_sse(1.0f);
        // written once and not read again by this launch: streamed past the caches, which keeps textures, hierarchy nodes and
        // object records resident (the four stores merge into one 16-byte store)
        if (!F.skip_accum_store) {   // wave-uniform
            __builtin_nontemporal_store(c.x, &acc->x); __builtin_nontemporal_store(c.y, &acc->y);
            __builtin_nontemporal_store(c.z, &acc->z); __builtin_nontemporal_store(c.w, &acc->w);
        }
    }
    uint32_t r = rt_as_u8(c.x * 255.0f), g = rt_as_u8(c.y * 255.0f), b = rt_as_u8(c.z * 255.0f), a8 = rt_as_u8(c.w + 255.0f);
    __builtin_nontemporal_store(r | (g << 8) | (b << 16) | (a8 << 24), &F.rgba[pos]);
}
// The same in two steps for a lane with several pixels: load_accum for all of them first, so that every accumulator load
// is in flight before the first dependent store, then store_pixel with the loaded entry.
RT_DEV float4 load_accum(const DevFrame &F, uint32_t pos) {
    if (F.accumulate) return reinterpret_cast<const float4 *>(F.accum)[pos];
    return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}
RT_DEV void store_pixel(const DevFrame &F, uint32_t pos, f3 p, float4 a) {
    float4 *acc = reinterpret_cast<float4 *>(F.accum) + pos;
    float4 c;
    if (F.accumulate) {
        a.x = a.x + p.x; a.y = a.y + p.y; a.z = a.z + p.z; a.w = a.w + 1.0f;
        *acc = a;
        float fi = (float)F.frame_index;
        c.x = rt_clamp01_sse(a.x / fi); c.y = rt_clamp01_sse(a.y / fi); c.z = rt_clamp01_sse(a.z / fi); c.w = rt_clamp01_sse(a.w / fi);
    } else {
        c.x = rt_clamp01_sse(p.x); c.y = rt_clamp01_sse(p.y); c.z = rt_clamp01_sse(p.z); c.w = rt_clamp01_sse(1.0f);
        if (!F.skip_accum_store) {
            __builtin_nontemporal_store(c.x, &acc->x); __builtin_nontemporal_store(c.y, &acc->y);
            __builtin_nontemporal_store(c.z, &acc->z); __builtin_nontemporal_store(c.w, &acc->w);
        }
    }
    uint32_t r = rt_as_u8(c.x * 255.0f), g = rt_as_u8(c.y * 255.0f), b = rt_as_u8(c.z * 255.0f), a8 = rt_as_u8(c.w + 255.0f);
    __builtin_nontemporal_store(r | (g << 8) | (b << 16) | (a8 << 24), &F.rgba[pos]);
}

// The tail of the same for a pixel whose samples of several accumulating frames were added by the caller, in frame order and
// starting from the entry's previous value (`self.accumulated[pos] += p` once per frame, src/renderer.rs:60): writes the entry
// back and packs it as the LAST of those frames does (:62-66 with that frame's frame_index).
RT_DEV void store_summed(const DevFrame &F, uint32_t pos, float4 a, uint32_t last_frame_index) {
    reinterpret_cast<float4 *>(F.accum)[pos] = a;
    const float fi = (float)last_frame_index;
    const float cx = rt_clamp01_sse(a.x / fi), cy = rt_clamp01_sse(a.y / fi), cz = rt_clamp01_sse(a.z / fi), cw = rt_clamp01_sse(a.w / fi);
    const uint32_t r = rt_as_u8(cx * 255.0f), g = rt_as_u8(cy * 255.0f), b = rt_as_u8(cz * 255.0f), a8 = rt_as_u8(cw + 255.0f);
    __builtin_nontemporal_store(r | (g << 8) | (b << 16) | (a8 << 24), &F.rgba[pos]);
}

// The workgroup's row in the counter table (2-D grid, see lane_pixel).
RT_DEV uint32_t rt_block_index() { return blockIdx.y * gridDim.x + blockIdx.x; }

// Sums the per-lane counters of a workgroup and writes one row of F.counters.
template <bool EVENTS>
RT_DEV void flush_counters(const DevFrame &F, Cnt<EVENTS> &cnt, uint32_t *s_cnt) {
    constexpr int N = (EVENTS ? RT_CNT_COUNT : 0) + 1;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t vals[N];
    vals[0] = cnt.rays;
#pragma unroll
    for (int i = 1; i < N; i++) vals[i] = cnt.ev[i - 1];
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint32_t v = vals[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0u && v != 0u) atomicAdd(&s_cnt[i], v);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < (uint32_t)N; i += blockDim.x) F.counters[rt_block_index() * (uint32_t)N + i] = s_cnt[i];
}

// pixel of this lane: a wavefront owns one 8x8 tile; blockIdx.y counts the tile rows from the bottom of the band,
// blockIdx.x * waves + wave is the tile column (a 2-D grid: no integer division).
// Workgroups start in launch order and a launch ends when its last workgroup does.  Rows from the bottom up put the sky --
// the cheap rows of most frames -- last, so the launch does not end on a few expensive workgroups with the chip idle
// (measured against top-down on the 4K workloads: scene3 -6 %, scene1 -7 %, scene4 -18 %, tex4 and scene2 +1 %).
RT_DEV bool lane_pixel(const DevCamera &C, const DevFrame &F, uint32_t &x, uint32_t &y) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t tx = blockIdx.x * (blockDim.x >> 6) + wave, ty = gridDim.y - 1u - blockIdx.y;
    x = tx * 8u + (lane & 7u);
    y = F.row_begin + ty * F.tile_row_step + (lane >> 3);
    return x < C.width && y < F.row_end;
}
enum { FK_REFL = 0, FK_REFR1 = 1, FK_REFR2 = 2, FK_PASS = 3 };

// ------------------------------------------------------------------------------------------------
// ray tracing kernel
//   REFR   : the scene has refractive materials (two-child frames are compiled in)
//   RDEPTH : capacity of the per-lane refractive frame array (>= max_ray_bounces when REFR)
//   BVH    : closest hits go through the bounding-volume hierarchy (scenes with many objects)
//   SP     : every object is a sphere or a plane (the reference's default scene): the other four primitives' candidate
//            tests and hit completions are compiled out (scene3 4K: 0.101 -> 0.096 ms); never combined with BVH
//   dynamic LDS: max_ray_bounces * 4 * blockDim.x floats (reflective frames, one column per lane)
// ------------------------------------------------------------------------------------------------
// second launch bound = waves per SIMD.  Plain variant: 7 waves (72 VGPRs, 16 bytes of spills; measured against 6 waves:
// tex4 -2.6 %, scene2 -2.5 %, scene3 -0.3 %; 8 waves spill heavily, +50 %).  Variants with refractive frames: 6 waves
// (80 VGPRs; 5 or 4 waves measured slower).  Hierarchy variants: the 96 registers of 5 waves (5 % faster than 6).
// The spheres-and-planes variant at 8 waves (64 VGPRs, 16 bytes of spills) measured 5 % slower than at 7.
// The march kernel below fits 8 waves per SIMD (64 VGPRs).
#ifndef RT_REFR_WAVES
#define RT_REFR_WAVES 6
#endif
template <bool EVENTS, bool REFR, int RDEPTH, bool BVH, bool SP = false>
__global__ void __launch_bounds__(256, BVH ? 5 : (REFR ? RT_REFR_WAVES : 7)) rt_trace_kernel(const DevScene S, const DevCamera C, const DevFrame F) {
    extern __shared__ float s_frames[];
    __shared__ uint32_t s_cnt[(EVENTS ? RT_CNT_COUNT : 0) + 1];
    __shared__ __attribute__((aligned(16))) double s_pow[RT_POW_LDS_DOUBLES];
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    uint32_t px, py;
    const bool in_frame = lane_pixel(C, F, px, py);
    // Sky tiles (rt_tile_sky_kernel): a workgroup of nothing else writes the value of a miss (ray_tracing.rs:238-240) and ends
    // -- no directions, no tables, no barrier.  (A sky tile inside a mixed workgroup takes the ordinary path: skipping it there
    // measured slower, the wave-uniform flag lives through the whole kernel.)
    // (not in the variants with refractive frames: they sit at their register budget, and the extra prologue cost the scene
    // with glass 3 % while 3 % of its workgroups are sky)
    constexpr bool kSkyTiles = !EVENTS && !BVH && !REFR;
    uint32_t sky_word = 0u;
    if (kSkyTiles && F.tile_sky) sky_word = F.tile_sky[rt_block_index()];
    if (kSkyTiles && sky_word == (0x01010101u >> (8u * (4u - (nthr >> 6))))) {
        if (in_frame) store_pixel(F, py * C.width + px, splat3(0.0f) + ld3(S.ambient) * splat3(1.0f));
        if (tid == 0u) {   // one primary ray per pixel of the workgroup's tiles that lies inside the frame (see lane_pixel)
            const uint32_t x0 = blockIdx.x * (nthr >> 6) * 8u, y0 = F.row_begin + (gridDim.y - 1u - blockIdx.y) * F.tile_row_step;
            const uint32_t x1 = x0 + (nthr >> 6) * 8u < C.width ? x0 + (nthr >> 6) * 8u : C.width, y1 = y0 + 8u < F.row_end ? y0 + 8u : F.row_end;
            // bit 31 marks the row as "rays given the value of a miss by the classification" for rt_get_stats
            F.counters[rt_block_index()] = ((x1 > x0 && y1 > y0) ? (x1 - x0) * (y1 - y0) : 0u) | 0x80000000u;
        }
        return;
    }
    // Camera.ray_directions[pos] (src/renderer.rs:49), requested before the tables are staged: the two memory round trips
    // of a wavefront's start overlap (lanes outside the frame take pixel 0's direction, which nobody uses)
    // (C.ray_dirs null, wave-uniform: the first frame after a camera change computes its directions itself, by the table's formula)
    f3 rd_first;
    if (C.ray_dirs) rd_first = ld3(C.ray_dirs + 3u * (size_t)(in_frame ? py * C.width + px : 0u));
    else rd_first = camera_ray(C, in_frame ? px : 0u, in_frame ? py : 0u);
    for (uint32_t i = tid; i < (uint32_t)((EVENTS ? RT_CNT_COUNT : 0) + 1); i += nthr) s_cnt[i] = 0u;
    rt_pow_lds_init(s_pow);
    __syncthreads();

    Cnt<EVENTS> cnt;
    cnt.clear();
    DG_DECL
    DG_BEGIN(DG_TOTAL)
    if (in_frame) {
        const uint32_t pos = py * C.width + px;
        cnt.add(RT_CNT_RAYGEN);
        cnt.add(RT_CNT_PIXEL_PACK);
        f3 ro = ld3(C.position);
        f3 rd = rd_first;
        f3 L = splat3(0.0f);      // light_color carried down
        f3 Cc = splat3(1.0f);     // contribution carried down
        f3 V = splat3(0.0f);      // value returned by the subtree just finished
        const f3 ambient = ld3(S.ambient);
        const uint32_t maxd = S.max_ray_bounces;
        const bool scene_diffuse = S.diffuse != 0u, jitter = S.enable_accumulation != 0u;
        bool dmode = scene_diffuse;  // true: inside color_diffuse, false: inside color
        uint32_t depth = 0u;
        uint32_t key = rt_rng_path_key(F.seed, pos, F.frame_index);
        // two bits per frame, frame d at bits [2d, 2d+1]; without refractive materials every frame of a path has the same
        // kind (the mode never changes), so the variable is only kept when REFR
        using Kinds = typename std::conditional<(RDEPTH <= 16), uint32_t, uint64_t>::type;   // 32 bits hold 16 frames
        Kinds kinds = 0;
        int ray_slot = RT_CNT_RAYS_PRIMARY;
        float rfr[REFR ? RDEPTH : 1][16];
        bool done = false;
        if (!REFR && scene_diffuse) {
            // color_diffuse without refractive materials: every node of the recursion has one child and returns that child's
            // value unchanged (ray_tracing.rs:111-114), so a path is a forward loop without frames.
            for (;;) {
                if (depth >= maxd) { V = L; break; }
                float t; int sub;
                const int obj = trace_ray<EVENTS, BVH, SP>(S, ro, rd, t, sub, cnt, ray_slot);
                if (obj < 0) { V = L + ambient * Cc; break; }
                HitRec h; Mtl m;
                finalize_hit<EVENTS, BVH, SP>(S, ro, rd, t, obj, sub, h, m, cnt);
                if (m.kind != RT_REFLECTIVE) { V = splat3(0.0f); break; }   // unreachable: REFR is set whenever a refractive material exists
                f3 albedo = m.albedo;
                if (m.texture >= 0) { cnt.add(RT_CNT_TEX_FETCH); albedo = tex_from_uv(S.textures[m.texture], h.u, h.v); }
                const f3 p = L + m.emission_power * albedo;
                cnt.add(RT_CNT_DIFFUSE_RAY);
                f3 no, nd;
                rt_reflection_ray(rd, h, m.roughness, key, true, jitter, no, nd);
                cnt.add(RT_CNT_COMBINE_REFL);
                ro = no; rd = nd; L = p; Cc = Cc * albedo;
                key = rt_rng_child_key(key, 0u);
                depth += 1u;
                ray_slot = RT_CNT_RAYS_REFLECTION;
            }
            done = true;
        }
        while (!done) {
            bool have_value = false;
            // ---- descend: evaluate color / color_diffuse at (ro, rd, depth, L, Cc)
            if (depth >= maxd) {
                V = L;
                have_value = true;
            } else {
                float t; int sub;
                DG_BEGIN(DG_TRACE)
                int obj = trace_ray<EVENTS, BVH, SP>(S, ro, rd, t, sub, cnt, ray_slot);
                DG_END(DG_TRACE)
                if (obj < 0) {
                    V = L + ambient * Cc;
                    have_value = true;
                } else {
                    HitRec h; Mtl m;
                    DG_BEGIN(DG_FINALIZE)
                    finalize_hit<EVENTS, BVH, SP>(S, ro, rd, t, obj, sub, h, m, cnt);
                    DG_END(DG_FINALIZE)
                    f3 albedo = m.albedo;
                    if (m.kind == RT_REFLECTIVE) {
                        DG_BEGIN(DG_TEX)
                        if (m.texture >= 0) { cnt.add(RT_CNT_TEX_FETCH); albedo = tex_from_uv(S.textures[m.texture], h.u, h.v); }
                        DG_END(DG_TEX)
                        f3 p;
                        DG_BEGIN(DG_LIGHTS)
                        if (!dmode) p = shade_lights<EVENTS, BVH, SP>(S, rd, h, albedo, m, obj, s_pow, cnt DG_PASS);
                        else p = L + m.emission_power * albedo;
                        DG_END(DG_LIGHTS)
                        cnt.add(scene_diffuse ? RT_CNT_DIFFUSE_RAY : RT_CNT_REFLECTION_RAY);
                        f3 no, nd;
                        DG_BEGIN(DG_REFLRAY)
                        rt_reflection_ray(rd, h, m.roughness, key, scene_diffuse, jitter, no, nd);
                        DG_END(DG_REFLRAY)
                        if (!dmode) {
                            float *fr = s_frames + (size_t)(depth * 4u) * nthr + tid;
                            fr[0] = p.x; fr[nthr] = p.y; fr[2u * nthr] = p.z; fr[3u * nthr] = m.roughness;
                            if (REFR) kinds = (kinds & ~((Kinds)3 << (2u * depth))) | ((Kinds)FK_REFL << (2u * depth));
                        } else {
                            cnt.add(RT_CNT_COMBINE_REFL);
                            if (REFR) kinds = (kinds & ~((Kinds)3 << (2u * depth))) | ((Kinds)FK_PASS << (2u * depth));
                        }
                        ro = no; rd = nd; L = p; Cc = Cc * albedo;
                        key = rt_rng_child_key(key, 0u);
                        depth += 1u;
                        ray_slot = RT_CNT_RAYS_REFLECTION;
                    } else if (REFR) {
                        cnt.add(RT_CNT_REFRACTION);
                        float kr = rt_fresnel(rd, h.normal, m.refraction_index, m.reflectivity);
                        f3 fo, fd;
                        bool has_refr = rt_refraction_ray(rd, h, m.refraction_index, fo, fd);
                        cnt.add(RT_CNT_REFLECTION_RAY);
                        f3 lo = h.point + RT_EPSILON * h.normal;
                        f3 ldv = rt_reflect(rd, h.normal);
                        f3 p;
                        if (!dmode) p = shade_lights<EVENTS, BVH, SP>(S, rd, h, albedo, m, obj, s_pow, cnt DG_PASS);
                        else p = L + m.emission_power * albedo;
                        f3 cchild = Cc * albedo;
                        float *f = rfr[depth];
                        f[12] = kr;
                        f[13] = __int_as_float(S.objects[obj].material);
                        f[15] = dmode ? 1.0f : 0.0f;
                        if (has_refr) {
                            f[0] = lo.x; f[1] = lo.y; f[2] = lo.z; f[3] = ldv.x; f[4] = ldv.y; f[5] = ldv.z;
                            f[6] = p.x; f[7] = p.y; f[8] = p.z; f[9] = cchild.x; f[10] = cchild.y; f[11] = cchild.z;
                            f[14] = __uint_as_float(rt_rng_child_key(key, 0u));
                            kinds = (kinds & ~((Kinds)3 << (2u * depth))) | ((Kinds)FK_REFR1 << (2u * depth));
                            ro = fo; rd = fd; Cc = cchild;  // L unchanged: the refraction child gets the parent's light_color
                            key = rt_rng_child_key(key, 1u);
                            dmode = false;                 // the refraction child always runs color()
                            ray_slot = RT_CNT_RAYS_REFRACTION;
                        } else {
                            f[0] = 0.0f; f[1] = 0.0f; f[2] = 0.0f;
                            kinds = (kinds & ~((Kinds)3 << (2u * depth))) | ((Kinds)FK_REFR2 << (2u * depth));
                            ro = lo; rd = ldv; L = p; Cc = cchild;
                            key = rt_rng_child_key(key, 0u);
                            ray_slot = RT_CNT_RAYS_REFLECTION;
                        }
                        depth += 1u;
                    } else {
                        V = splat3(0.0f);  // unreachable: REFR is set whenever a refractive material exists
                        have_value = true;
                    }
                }
            }
            // ---- unwind: feed V to the pending frames until a second child must be traced or the root returns
            DG_BEGIN(DG_UNWIND)
            while (have_value) {
                if (depth == 0u) { done = true; break; }
                const uint32_t d = depth - 1u;
                const uint32_t kind = REFR ? ((uint32_t)(kinds >> (2u * d)) & 3u) : (scene_diffuse ? (uint32_t)FK_PASS : (uint32_t)FK_REFL);
                if (kind == FK_PASS) { depth = d; continue; }
                if (kind == FK_REFL) {
                    const float *fr = s_frames + (size_t)(d * 4u) * nthr + tid;
                    f3 p = mk3(fr[0], fr[nthr], fr[2u * nthr]);
                    float rough = fr[3u * nthr];
                    cnt.add(RT_CNT_COMBINE_REFL);
                    V = p * rough + p * V * (1.0f - rough);  // ray_tracing.rs:194
                    depth = d;
                    continue;
                }
                if (REFR) {
                    float *f = rfr[d];
                    if (kind == FK_REFR1) {
                        ro = mk3(f[0], f[1], f[2]); rd = mk3(f[3], f[4], f[5]);
                        L = mk3(f[6], f[7], f[8]); Cc = mk3(f[9], f[10], f[11]);
                        key = __float_as_uint(f[14]);
                        dmode = f[15] != 0.0f;
                        f[0] = V.x; f[1] = V.y; f[2] = V.z;
                        kinds = (kinds & ~((Kinds)3 << (2u * d))) | ((Kinds)FK_REFR2 << (2u * d));
                        ray_slot = RT_CNT_RAYS_REFLECTION;
                        have_value = false;  // descend into the reflection child at the same depth
                    } else {
                        f3 refr = mk3(f[0], f[1], f[2]);
                        float kr = f[12];
                        const DevMaterial &dm = S.materials[__float_as_int(f[13])];
                        cnt.add(RT_CNT_COMBINE_REFR);
                        f3 col = V * kr + refr * (1.0f - kr) * dm.transparency;  // ray_tracing.rs:232-233 / :150-151
                        if (f[15] == 0.0f) col = col * ld3(dm.albedo);             // only color() multiplies by albedo (:235)
                        V = col;
                        depth = d;
                    }
                } else {
                    depth = d;
                }
            }
            DG_END(DG_UNWIND)
        }
        DG_BEGIN(DG_STORE)
        store_pixel(F, pos, V);
        DG_END(DG_STORE)
    }
    DG_END(DG_TOTAL)
#ifdef RT_DIAG
    if ((threadIdx.x & 63u) == 0u && F.diag) {
        const uint32_t gw_ = rt_block_index() * (blockDim.x >> 6) + (threadIdx.x >> 6);
        for (int i_ = 0; i_ < DG_N; i_++) F.diag[(size_t)gw_ * DG_N + i_] = (unsigned long long)dg.acc[i_];
    }
#endif
    flush_counters<EVENTS>(F, cnt, s_cnt);
}

// ------------------------------------------------------------------------------------------------
// ray marching
// ------------------------------------------------------------------------------------------------
struct SdfOut { float dist; f3 albedo, tro, trd; };

RT_DEV float rt_mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }                       // geometry.rs:34-36
RT_DEV float rt_interpolation(float d1, float d2, float k) { return rt_clamp(0.5f + 0.5f * (d2 - d1) / k, 0.0f, 1.0f); }  // :42-44
RT_DEV float rt_smooth_union(float d1, float d2, float k) {                                             // :46-50
    float h = rt_clamp(0.5f + 0.5f * (d2 - d1) / k, 0.0f, 1.0f);
    return rt_mix(d2, d1, h) - k * h * (1.0f - h);
}
RT_DEV float rt_smooth_step(float e0, float e1, float x) {  // :29-32
    float t = rt_clamp((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}

// The object-space ray of a leaf, (inv * ro, inv * rd), does not depend on the march parameter t, but the reference
// recomputes it in every sdf() call (~40 of the ~50 operations of a sphere's sdf).  During the march loop of one
// pixel the world ray is fixed, so the kernel computes each leaf's object-space ray once (rm_fill_cache) and keeps it
// in LDS: the direction per lane (one column per lane), the origin once per workgroup (every primary ray starts at
// the camera position).  Leaves are numbered in evaluation order, which is the same in every sdfs() call.
// One wavefront per workgroup: march lengths differ widely between tiles, and single-wave workgroups release their slots
// one by one (measured 1.8 % faster than 256 threads on the 4K ray-marching scene).
constexpr uint32_t kMarchThreads = 64u;  // workgroup size of the march kernel (compile-time: LDS offsets become immediates)
struct RayCache {
    float *trd;        // this lane's column: component c of slot s at trd[(s * 3 + c) * kMarchThreads]
    float *tro;        // workgroup-wide: component c of slot s at tro[s * 3 + c]
    uint32_t slot;
};
template <bool CACHED>
RT_DEV void leaf_ray(const DevObject &o, f3 ro, f3 rd, RayCache &rc, f3 &tro, f3 &trd) {
    if (CACHED) {
        const float *d = rc.trd + rc.slot * (3u * kMarchThreads);
        const float *q = rc.tro + rc.slot * 3u;
        trd = mk3(d[0], d[kMarchThreads], d[2u * kMarchThreads]);
        tro = mk3(q[0], q[1], q[2]);
        rc.slot += 1u;
    } else {
        trd = xf_dir(o.inv, rd); tro = xf_pt(o.inv, ro);
    }
}

// Object3D::sdf (src/objects.rs:33-44).  FULL: also albedo and transformed_ray (needed at the final march
// step only); LEVEL: remaining Union/Substraction nesting the instantiation can descend; CACHED: object-space rays
// come from the RayCache instead of being recomputed.
template <bool EVENTS, bool FULL, int LEVEL, bool CACHED>
struct Sdf {
    static RT_DEV void eval(const DevScene &S, const DevObject &o, f3 ro, f3 rd, float t, SdfOut &out, Cnt<EVENTS> &cnt, RayCache &rc) {
        switch (o.kind) {
            case RT_SPHERE: {  // sphere.rs:72-82
                cnt.add(RT_CNT_SDF_SPHERE);
                f3 trd, tro; leaf_ray<CACHED>(o, ro, rd, rc, tro, trd);
                f3 p = tro + trd * t;
                out.dist = length(p) - o.p[0];
                if (FULL) { out.albedo = ld3(S.materials[o.material].albedo); out.tro = tro; out.trd = trd; }
                break;
            }
            case RT_CUBOID: {  // cuboid.rs:53-70
                cnt.add(RT_CNT_SDF_CUBOID);
                f3 trd, tro; leaf_ray<CACHED>(o, ro, rd, rc, tro, trd);
                f3 p = tro + trd * t;
                const float corner_radius = 0.1f;
                f3 q = vabs(p) - ld3(o.p) + corner_radius;
                out.dist = length(vmax(q, splat3(0.0f))) + rt_min(rt_max(q.x, rt_max(q.y, q.z)), 0.0f) - corner_radius;
                if (FULL) { out.albedo = ld3(S.materials[o.material].albedo); out.tro = tro; out.trd = trd; }
                break;
            }
            case RT_PLANE: {  // plane.rs:31-37
                cnt.add(RT_CNT_SDF_PLANE);
                f3 p = ro + rd * t;
                out.dist = dot(p - ld3(o.p + 3), ld3(o.p));
                if (FULL) { out.albedo = ld3(S.materials[o.material].albedo); out.tro = ro; out.trd = rd; }
                break;
            }
            case RT_CYLINDER: {  // cylinder.rs:56-74
                cnt.add(RT_CNT_SDF_CYLINDER);
                f3 trd, tro; leaf_ray<CACHED>(o, ro, rd, rc, tro, trd);
                f3 p = tro + trd * t;
                const float corner_radius = 0.1f;
                float dx = (rt_sqrt((p.x * p.x) + (p.z * p.z)) - o.p[0]) + corner_radius;
                float dy = (rt_abs(p.y) - o.p[1] * 0.5f) + corner_radius;
                float mx = rt_max(dx, 0.0f), my = rt_max(dy, 0.0f);
                out.dist = rt_sqrt((mx * mx) + (my * my)) + rt_min(rt_max(dx, dy), 0.0f) - corner_radius;
                if (FULL) { out.albedo = ld3(S.materials[o.material].albedo); out.tro = tro; out.trd = trd; }
                break;
            }
            case RT_UNION: {  // union.rs:26-42
                if constexpr (LEVEL > 0) {
                    cnt.add(RT_CNT_SDF_UNION);
                    SdfOut h1, h2;
                    Sdf<EVENTS, FULL, LEVEL - 1, CACHED>::eval(S, S.objects[o.first], ro, rd, t, h1, cnt, rc);
                    Sdf<EVENTS, FULL, LEVEL - 1, CACHED>::eval(S, S.objects[o.second], ro, rd, t, h2, cnt, rc);
                    out.dist = rt_smooth_union(h1.dist, h2.dist, 0.7f);
                    if (FULL) {
                        float i = rt_interpolation(h1.dist, h2.dist, 0.7f);
                        float a = 1.0f - i;
                        out.albedo = h1.albedo * (1.0f - a) + h2.albedo * a;  // mix_vec3, geometry.rs:38-40
                        if (h1.dist < h2.dist) { out.tro = h1.tro; out.trd = h1.trd; } else { out.tro = h2.tro; out.trd = h2.trd; }
                    }
                } else {
                    out.dist = RT_F32_MAX; if (FULL) { out.albedo = splat3(0.0f); out.tro = splat3(0.0f); out.trd = splat3(0.0f); }
                }
                break;
            }
            case RT_SUBSTRACTION: {  // substraction.rs:26-37
                if constexpr (LEVEL > 0) {
                    cnt.add(RT_CNT_SDF_SUBSTRACTION);
                    SdfOut h1, h2;
                    const DevObject &o1 = S.objects[o.first];
                    Sdf<EVENTS, FULL, LEVEL - 1, CACHED>::eval(S, o1, ro, rd, t, h1, cnt, rc);
                    Sdf<EVENTS, FULL, LEVEL - 1, CACHED>::eval(S, S.objects[o.second], ro, rd, t, h2, cnt, rc);
                    out.dist = rt_max(-h2.dist, h1.dist);
                    if (FULL) { out.albedo = ld3(S.materials[o1.material].albedo); out.tro = h2.tro; out.trd = h2.trd; }
                } else {
                    out.dist = RT_F32_MAX; if (FULL) { out.albedo = splat3(0.0f); out.tro = splat3(0.0f); out.trd = splat3(0.0f); }
                }
                break;
            }
            default:  // cone.rs:55-57, triangle.rs:29-31
                out.dist = RT_F32_MAX;
                if (FULL) { out.albedo = splat3(0.0f); out.tro = splat3(0.0f); out.trd = splat3(0.0f); }
                break;
        }
    }
};

// RayMarching::sdfs (src/ray_marching/ray_marching.rs:24-107); returns obj_idx
template <bool EVENTS, bool FULL, bool CACHED = false>
RT_DEV uint32_t rm_sdfs(const DevScene &S, f3 ro, f3 rd, float t, SdfOut &best, Cnt<EVENTS> &cnt, RayCache &rc) {
    cnt.add(RT_CNT_SDFS_CALL);
    rc.slot = 0u;
    best.dist = RT_F32_MAX;
    best.albedo = splat3(0.0f); best.tro = splat3(0.0f); best.trd = splat3(0.0f);
    uint32_t obj_idx = 0u;
    const uint32_t n = S.n_sdfs;
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t idx = S.sdfs[k];
        const DevObject &o = S.objects[idx];
        const int kind = o.kind;
        if (kind == RT_TRIANGLE || kind == RT_CONE) continue;  // the `_ => {}` arm (:95)
        SdfOut d;
        Sdf<EVENTS, FULL, 2, CACHED>::eval(S, o, ro, rd, t, d, cnt, rc);
        if (d.dist < best.dist) {
            best.dist = d.dist;
            if (FULL) { best.albedo = d.albedo; best.tro = d.tro; best.trd = d.trd; }
            obj_idx = (kind == RT_UNION || kind == RT_SUBSTRACTION) ? (uint32_t)o.second : idx;
        }
    }
    return obj_idx;
}

// RayMarching::sdfs inside march_ray's loop (src/ray_marching/ray_marching.rs:186-199), where only the smallest distance
// matters.  An entry of Scene.sdfs whose distance function cannot have fallen below the running minimum since it was last
// evaluated is left out: along a ray, a distance function falls by at most `lip` per unit of the march parameter (the leaf
// functions are exact or conservative distances in their object spaces, so `lip` is the largest stretch of the object's
// inv_transform times |direction|; a smooth union or a subtraction falls no faster than its faster child), hence
//     value(t) >= value(t_k) - lip * (t - t_k)      for t >= t_k, the parameter of the entry's last evaluation
// (t only grows while a ray marches: a step that does not end the loop is >= 0.001).  With a margin for the rounding of the
// evaluations (2e-3, a thousand times their error at these magnitudes; lip comes inflated by 1e-3 from the host), an entry
// with bound > minimum-so-far cannot change `d.dist < best.dist` (:47-52), so the minimum -- the only thing march_ray reads
// -- is the reference's, bit for bit.  The decision is taken per wavefront (an evaluation costs the same for one lane as
// for sixty-four), every lane that marches takes part in an evaluation and renews its bound.  lb: this lane's column,
// entry k at lb[k * kMarchThreads], holding value(t_k) + lip * t_k - margin (second children of pairs: at n_sdfs + k).
// UPDATE: a march step at parameter t of the camera ray (bounds are taken at `at` = t and renewed).  Otherwise a tap of
// RayMarching::normal or ::occlusion (:109-158) at a point within `away` of the camera ray's point at parameter `at` (the last
// march step's): the bounds are taken there, lowered by lip * away, and left as they are.
template <bool CACHED, bool UPDATE>
RT_DEV float rm_sdfs_march(const DevScene &S, const DevFrame &F, f3 ro, f3 rd, float t, RayCache &rc, float *lb, float at, float away) {
    Cnt<false> nocnt;
    float best = RT_F32_MAX;
    const uint32_t n = S.n_sdfs;
    for (uint32_t k = 0; k < n; k++) {
        const float lip = F.march_lip[k];
        const float bound = (lb[k * kMarchThreads] - lip * at) - lip * away;
        if (__ballot(!(bound > best)) == 0ull) continue;   // (a NaN bound never skips)
        const DevObject &o = S.objects[S.sdfs[k]];
        const int kind = o.kind;
        if (kind == RT_TRIANGLE || kind == RT_CONE) continue;  // the `_ => {}` arm (:95)
        rc.slot = F.march_slot[k];
        SdfOut d;
        if ((F.march_pair >> k) & 1u) {
            // A Union or Substraction of two leaves: the first child always; the second only if, by its own bound, it can
            // influence the pair.  Union (src/utils/union.rs:26-42, geometry.rs:46-50): with d2 - d1 >= k the interpolation
            // factor clamps to 1 and smooth_union returns (d2 * 0 + d1 * 1) - k * 1 * 0, which is 0 + d1 (d2 > 0 whenever d1
            // is a zero).  Substraction (src/utils/substraction.rs:34): max(-d2, d1) is d1 when d2 > -d1.
            SdfOut d1;
            Sdf<false, false, 0, CACHED>::eval(S, S.objects[o.first], ro, rd, t, d1, nocnt, rc);
            const float lip2 = F.march_lip2[k];
            const float bound2 = (lb[(n + k) * kMarchThreads] - lip2 * at) - lip2 * away;
            const float matters_below = kind == RT_UNION ? d1.dist + 0.7f : -d1.dist;
            if (__ballot(!(bound2 > matters_below)) == 0ull) {
                d.dist = kind == RT_UNION ? (0.0f + d1.dist) - 0.0f : d1.dist;
            } else {
                rc.slot = F.march_slot2[k];
                SdfOut d2;
                Sdf<false, false, 0, CACHED>::eval(S, S.objects[o.second], ro, rd, t, d2, nocnt, rc);
                if (UPDATE) lb[(n + k) * kMarchThreads] = (d2.dist + lip2 * t) - 2e-3f;
                d.dist = kind == RT_UNION ? rt_smooth_union(d1.dist, d2.dist, 0.7f) : rt_max(-d2.dist, d1.dist);
            }
        } else {
            Sdf<false, false, 2, CACHED>::eval(S, o, ro, rd, t, d, nocnt, rc);
        }
        if (UPDATE) lb[k * kMarchThreads] = (d.dist + lip * t) - 2e-3f;
        if (d.dist < best) best = d.dist;
    }
    return best;
}

// Fills the RayCache for the world ray (ro, rd): visits the leaves in the order Sdf::eval does and stores their
// object-space rays.  ro is the same for every lane of the workgroup; thread 0 stores the transformed origins.
template <int LEVEL>
RT_DEV void rm_fill_leaf(const DevScene &S, const DevObject &o, f3 ro, f3 rd, RayCache &rc) {
    switch (o.kind) {
        case RT_SPHERE: case RT_CUBOID: case RT_CYLINDER: {
            const f3 trd = xf_dir(o.inv, rd), tro = xf_pt(o.inv, ro);
            float *d = rc.trd + rc.slot * (3u * kMarchThreads);
            d[0] = trd.x; d[kMarchThreads] = trd.y; d[2u * kMarchThreads] = trd.z;
            if (threadIdx.x == 0u) { float *q = rc.tro + rc.slot * 3u; q[0] = tro.x; q[1] = tro.y; q[2] = tro.z; }
            rc.slot += 1u;
            break;
        }
        case RT_UNION: case RT_SUBSTRACTION:
            if constexpr (LEVEL > 0) {
                rm_fill_leaf<LEVEL - 1>(S, S.objects[o.first], ro, rd, rc);
                rm_fill_leaf<LEVEL - 1>(S, S.objects[o.second], ro, rd, rc);
            }
            break;
        default: break;  // plane: no transform; cone / triangle: no sdf
    }
}
RT_DEV void rm_fill_cache(const DevScene &S, f3 ro, f3 rd, RayCache &rc) {
    rc.slot = 0u;
    const uint32_t n = S.n_sdfs;
    for (uint32_t k = 0; k < n; k++) {
        const DevObject &o = S.objects[S.sdfs[k]];
        if (o.kind == RT_TRIANGLE || o.kind == RT_CONE) continue;
        rm_fill_leaf<2>(S, o, ro, rd, rc);
    }
}

// CACHE: the march loop reads the leaves' object-space rays from LDS (dynamic LDS: n_leaf_slots * 3 * (kMarchThreads + 1)
// floats); chosen by the host when the scene's leaf count fits.  Launched with kMarchThreads threads per workgroup.
// SKIP (timed variant only, at most 8 entries in Scene.sdfs): march steps evaluate only the entries that can hold the
// minimum (rm_sdfs_march; dynamic LDS: 2 * n_sdfs * kMarchThreads more floats behind the ray cache).
template <bool EVENTS, bool CACHE, bool SKIP = false>
__global__ void __launch_bounds__(256, 8) rt_march_kernel(const DevScene S, const DevCamera C, const DevFrame F) {
    extern __shared__ float s_ray[];
    __shared__ uint32_t s_cnt[(EVENTS ? RT_CNT_COUNT : 0) + 1];
    __shared__ __attribute__((aligned(16))) double s_pow[RT_POW_LDS_DOUBLES];
    for (uint32_t i = threadIdx.x; i < (uint32_t)((EVENTS ? RT_CNT_COUNT : 0) + 1); i += blockDim.x) s_cnt[i] = 0u;
    rt_pow_lds_init(s_pow);
    Cnt<EVENTS> cnt;
    cnt.clear();
    Cnt<false> nocnt;
    nocnt.clear();
    DG_DECL
    uint32_t px, py;
    const bool in_frame = lane_pixel(C, F, px, py);
    const f3 ro = ld3(C.position);
    // Camera.ray_directions[pos] (src/renderer.rs:49); lanes outside the frame take pixel 0's direction, which nobody uses
    // (C.ray_dirs null: the first frame after a camera change computes its directions itself, by the table's formula)
    f3 rd;
    if (C.ray_dirs) rd = ld3(C.ray_dirs + 3u * (size_t)(in_frame ? py * C.width + px : 0u));
    else rd = camera_ray(C, in_frame ? px : 0u, in_frame ? py : 0u);
    RayCache rc;
    rc.slot = 0u;
    rc.trd = s_ray + threadIdx.x;
    rc.tro = s_ray + F.leaf_slots * (3u * kMarchThreads);
    if (CACHE) rm_fill_cache(S, ro, rd, rc);
    __syncthreads();
    if (in_frame) {
        const uint32_t pos = py * C.width + px;
        cnt.add(RT_CNT_RAYGEN);
        cnt.add(RT_CNT_PIXEL_PACK);
        f3 color = ld3(S.ambient);  // miss: ambient_color, no fade (:252)
        // march_ray (:181-203)
        cnt.ray(RT_CNT_RAYS_MARCH);
        float t = 0.0f, t_eval = 0.0f;
        bool hit = false;
        float *lb = s_ray + F.leaf_slots * (3u * kMarchThreads + 3u) + threadIdx.x;   // behind the ray cache
        if (SKIP) for (uint32_t k = 0; k < 2u * S.n_sdfs; k++) lb[k * kMarchThreads] = -RT_F32_MAX;   // nothing evaluated yet (entries, then second children)
        for (int i = 0; i < 255; i++) {
            if (t > 40.0f) break;
            cnt.add(RT_CNT_MARCH_STEP);
            SdfOut d;
            if (SKIP) d.dist = rm_sdfs_march<CACHE, true>(S, F, ro, rd, t, rc, lb, t, 0.0f);
            else rm_sdfs<EVENTS, false, CACHE>(S, ro, rd, t, d, cnt, rc);
            t_eval = t;
            t = t + d.dist;
            if (d.dist < 0.001f) { hit = true; break; }
        }
        if (hit) {
            cnt.add(RT_CNT_MARCH_SHADE);
            // the final step again with albedo / transformed_ray / object index (same inputs, same values)
            SdfOut rh;
            const uint32_t obj_idx = rm_sdfs<false, true, CACHE>(S, ro, rd, t_eval, rh, nocnt, rc);
            const float distance = t;
            f3 hitp = ro + rd * distance;
            // distance of hitp from the camera ray's point at t_eval, where the bounds of rm_sdfs_march are valid (the last step
            // may have been negative; |rd| is 1 up to rounding)
            const float last_step = rt_abs(distance - t_eval) * 1.001f + 1e-6f;
            // normal (:109-139)
            const float k = 0.5773f * 0.0005f;
            const f3 xyy = mk3(1.0f, -1.0f, -1.0f), yyx = mk3(-1.0f, -1.0f, 1.0f), yxy = mk3(-1.0f, 1.0f, -1.0f), xxx = mk3(1.0f, 1.0f, 1.0f);
            // four taps in a loop (one copy of the sdfs code); the sum keeps the reference's order ((a + b) + c) + d
            f3 nsum = splat3(0.0f);
#pragma unroll 1
            for (int tap = 0; tap < 4; tap++) {
                const f3 e = tap == 0 ? xyy : (tap == 1 ? yyx : (tap == 2 ? yxy : xxx));
                SdfOut a;
                if (SKIP) a.dist = rm_sdfs_march<false, false>(S, F, hitp, e, k, rc, lb, t_eval, last_step + 0.000501f);   // |e| k = 0.0005
                else rm_sdfs<EVENTS, false>(S, hitp, e, k, a, cnt, rc);
                const f3 term = e * a.dist;
                nsum = tap == 0 ? term : nsum + term;
            }
            f3 n = normalize(nsum);
            // material of the winning object (:212-214); the index differs per lane: vector loads
            const DevObject *po = S.objects + obj_idx;
            const int okind = po->kind;
            const int mat_idx = (okind == RT_UNION || okind == RT_SUBSTRACTION) ? 0 : po->material;
            Mtl m = load_mtl(S.materials[mat_idx]);
            f3 albedo = rh.albedo;
            if (m.texture >= 0) {  // tri-planar (:218-232, geometry.rs:56-68)
                cnt.add(RT_CNT_TRIPLANAR);
                DevXform inv = po->inv;
                f3 n1 = normalize(xf_dir(inv, n));
                f3 hit1 = rh.tro + rh.trd * distance;
                const DevTexture tex = S.textures[m.texture];
                const float scale = 0.5f;
                f3 tx = tex_from_uv(tex, hit1.y * scale, hit1.z * scale);
                f3 ty = tex_from_uv(tex, hit1.x * scale, hit1.z * scale);
                f3 tz = tex_from_uv(tex, hit1.x * scale, hit1.y * scale);
                f3 bw = vpowf(vabs(n1), 0.8f, s_pow);
                bw = bw / (bw.x + bw.y + bw.z);
                albedo = tx * bw.x + ty * bw.y + tz * bw.z;
            }
            // light (:160-179)
            cnt.add(RT_CNT_LIGHT_CALL);
            f3 l_acc = splat3(0.0f);
            for (uint32_t i = 0; i < S.n_lights; i++) {
                const DevLight &l = S.lights[i];
                cnt.add(RT_CNT_LIGHT_SHADE);
                f3 ldir; float ldist;
                light_dir_dist(l, hitp, ldir, ldist);
                f3 phong = blinn_phong(rd, n, ldir, albedo, m, s_pow DG_PASS);
                l_acc = l_acc + (phong / (ldist * ldist)) * ld3(l.albedo) * l.intensity;
            }
            color = vpowf(l_acc, 0.4545f, s_pow);
            // occlusion (:141-158)
            float occ = 0.0f, sca = 1.0f;
#pragma unroll 1
            for (int i = 0; i < 3; i++) {
                cnt.add(RT_CNT_OCCLUSION_TAP);
                float hr = 0.02f + 0.025f * (float)(i * i);
                SdfOut dd;
                if (SKIP) dd.dist = rm_sdfs_march<false, false>(S, F, hitp, n, hr, rc, lb, t_eval, last_step + hr * 1.001f);   // |n| = 1
                else rm_sdfs<EVENTS, false>(S, hitp, n, hr, dd, cnt, rc);
                occ = occ + (-(dd.dist - hr)) * sca;
                sca = sca * 0.85f;
            }
            float o = 1.0f - rt_clamp(occ, 0.0f, 1.0f);
            color = color * o;
            color = color * (1.0f - rt_smooth_step(1.0f, 20.0f, distance));
        }
        store_pixel(F, pos, color);
    }
    flush_counters<EVENTS>(F, cnt, s_cnt);
}

// ------------------------------------------------------------------------------------------------
// Sky tiles.  Not part of the reference: screen-space culling for rt_trace_kernel's linear-scan variants.  For every
// workgroup of a render launch, one thread decides per wavefront tile (8x8 pixels) whether ANY camera ray of the tile can
// reach ANY object; the render kernel gives the tiles that cannot the value of a miss without reading their directions.
// The test is conservative: the tile's rays lie inside the pyramid through the pixel centres one pixel outside its corners
// (directions are a projective image of the pixel grid, so the pyramid of a rectangle contains the rays of its interior);
// an object is unreachable when its bounding sphere (DevObject::bs, the sphere of the scan's own rejection test, enlarged
// once more here) lies outside one of the pyramid's four side planes, a plane when the sign of its denominator at all four
// corners rules out a positive distance.  Objects without finite bounds count as reachable.  Which candidate tests run never
// changes a pixel (see trace_closest); this only moves the decision from every ray to the tile.
// ------------------------------------------------------------------------------------------------
RT_DEV f3 camera_ray_at(const DevCamera &C, float fx, float fy) {   // camera_ray for a real-valued pixel position
    float sx = 2.0f * (fx * C.inv_w) - 1.0f, sy = 2.0f * (fy * C.inv_h) - 1.0f;
    float tx = ((C.ip_c0[0] * sx + C.ip_c1[0] * sy) + C.ip_c2[0]) + C.ip_c3[0];
    float ty = ((C.ip_c0[1] * sx + C.ip_c1[1] * sy) + C.ip_c2[1]) + C.ip_c3[1];
    float tz = ((C.ip_c0[2] * sx + C.ip_c1[2] * sy) + C.ip_c2[2]) + C.ip_c3[2];
    float tw = ((C.ip_c0[3] * sx + C.ip_c1[3] * sy) + C.ip_c2[3]) + C.ip_c3[3];
    f3 v = mk3(tx / tw, ty / tw, tz / tw);
    f3 world = xf_dir(C.iv, v);
    return normalize(world - ld3(C.position));
}
RT_DEV bool tile_is_sky(const DevScene &S, const DevCamera &C, uint32_t x0, uint32_t y0) {
    const f3 ro = ld3(C.position);
    const float xa = (float)x0 - 1.0f, xb = (float)x0 + 8.0f, ya = (float)y0 - 1.0f, yb = (float)y0 + 8.0f;
    f3 d[4] = {camera_ray_at(C, xa, ya), camera_ray_at(C, xb, ya), camera_ray_at(C, xb, yb), camera_ray_at(C, xa, yb)};
    const f3 dc = camera_ray_at(C, (float)x0 + 3.5f, (float)y0 + 3.5f);
    f3 nrm[4];
    bool usable = true;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        f3 n = cross(d[k], d[(k + 1) & 3]);
        const float len = length(n);
        if (!(len > 1e-12f)) usable = false;
        n = n * (1.0f / len);
        if (dot(n, dc) < 0.0f) n = -n;   // inward
        nrm[k] = n;
        if (!(dot(n, dc) > 0.0f)) usable = false;   // NaN or a degenerate pyramid
    }
    if (!usable) return false;
    for (uint32_t i = 0; i < S.n_objects; i++) {
        const DevObject &o = S.objects[i];
        const int kind = o.kind;
        if (kind == RT_UNION || kind == RT_SUBSTRACTION) continue;   // never hit by rays (src/ray.rs:175)
        bool reach = true;
        if (o.bounded) {
            const f3 oc = ld3(o.bs) - ro;
            const float radius = rt_sqrt(o.bs[3]) * 1.01f + 1e-4f * length(oc) + 1e-4f;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (dot(nrm[k], oc) < -radius) reach = false;
        }
        if (kind == RT_PLANE) {
            const f3 pn = ld3(o.p);
            const float num = dot(ld3(o.p + 3) - ro, pn);
            const float eps = 1e-5f * length(pn);
            bool behind = num != 0.0f, ahead = num != 0.0f;
            f3 lo = splat3(RT_F32_MAX), hi = splat3(-RT_F32_MAX);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float denom = dot(d[k], pn);
                behind = behind && (num > 0.0f ? denom < -eps : denom > eps);   // every corner: distance negative
                ahead = ahead && (num > 0.0f ? denom > eps : denom < -eps);     // every corner: distance positive
                const f3 hp = ro + d[k] * (num / denom);
                lo = vmin(lo, hp); hi = vmax(hi, hp);
            }
            if (behind) reach = false;
            // a plane with max_dist (src/utils/plane.rs:55-59): the tile's rays meet it inside the quadrilateral of the four
            // corner hits; entirely beyond the limit in x or in z, every hit is rejected
            if (ahead && o.p[8] != 0.0f) {
                const float mx = o.p[6] * 1.001f + 1e-3f, mz = o.p[7] * 1.001f + 1e-3f;
                if (lo.x > mx || hi.x < -mx || lo.z > mz || hi.z < -mz) reach = false;
            }
        } else if (!o.bounded) {
            return false;
        }
        if (reach) return false;
    }
    return true;
}
// Four threads per workgroup of a render launch with grid (grid_x, grid_y) and `waves` wavefronts per workgroup: thread k
// writes byte k of the workgroup's word, wavefront k's tile (tiles outside the frame count as sky: they have no pixels;
// bytes beyond `waves` are 0).
__global__ void __launch_bounds__(256) rt_tile_sky_kernel(const DevScene S, const DevCamera C, const DevFrame F, uint32_t grid_x, uint32_t grid_y, uint32_t waves,
                                                           uint8_t *__restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t idx = t >> 2, k = t & 3u;
    if (idx >= grid_x * grid_y) return;
    const uint32_t by = idx / grid_x, bx = idx - by * grid_x;
    const uint32_t y0 = F.row_begin + (grid_y - 1u - by) * F.tile_row_step, x0 = (bx * waves + k) * 8u;
    uint8_t sky = 0u;
    if (k < waves) sky = (x0 >= C.width || y0 >= F.row_end || tile_is_sky(S, C, x0, y0)) ? 1u : 0u;
    out[t] = sky;
}

// ------------------------------------------------------------------------------------------------
// Camera::calculate_ray_directions (src/camera.rs:128-157): the table of primary directions, once per camera change
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) rt_raygen_kernel(const DevCamera C, float *__restrict__ out) {
    const uint32_t x = blockIdx.x * 64u + (threadIdx.x & 63u), y = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (x >= C.width || y >= C.height) return;
    const f3 d = camera_ray(C, x, y);
    float *o = out + 3u * ((size_t)y * C.width + x);
    o[0] = d.x; o[1] = d.y; o[2] = d.z;
}

// ------------------------------------------------------------------------------------------------
// rt_update_objects for a handful of objects: the new records travel as kernel arguments, so the update needs no staging
// buffer and is ordered on the stream like a launch (the reference's update functions rewrite one or two objects per frame,
// src/bin/scene1.rs:13-24, src/bin/scene4.rs:14-26).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kPatchObjects = 8u;
struct PatchArgs {
    uint32_t n;
    uint32_t index[kPatchObjects];
    DevObject object[kPatchObjects];
};
__global__ void __launch_bounds__(64) rt_patch_objects_kernel(DevObject *objects, const PatchArgs P) {
    constexpr uint32_t kWords = sizeof(DevObject) / 4u;
    for (uint32_t k = 0; k < P.n; k++) {
        uint32_t *dst = reinterpret_cast<uint32_t *>(objects + P.index[k]);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(&P.object[k]);
        for (uint32_t w = threadIdx.x; w < kWords; w += 64u) dst[w] = src[w];
    }
}

// ------------------------------------------------------------------------------------------------
// resolve: accumulated / sample_count -> clamp -> to_rgba (src/renderer.rs:62-66, :29-36); HBM-bound, 20 B per pixel
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) rt_resolve_kernel(const float4 *__restrict__ accum, uint32_t *__restrict__ rgba, uint32_t first, uint32_t count, float samples) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const float4 a = accum[first + i];
        const float x = rt_clamp01_sse(a.x / samples), y = rt_clamp01_sse(a.y / samples), z = rt_clamp01_sse(a.z / samples), w = rt_clamp01_sse(a.w / samples);
        rgba[first + i] = rt_as_u8(x * 255.0f) | (rt_as_u8(y * 255.0f) << 8) | (rt_as_u8(z * 255.0f) << 16) | (rt_as_u8(w + 255.0f) << 24);
    }
}
