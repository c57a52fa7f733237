// rt_bvh.h — host-side construction of the bounding-volume hierarchy used for scenes with many objects.
//
// The reference scans every object for every ray (RayTracing::trace_ray, src/ray_tracing/ray_tracing.rs:59-78);
// at 10 000 objects that is ~10^4 candidate tests per ray.  The hierarchy only decides WHICH objects get the
// reference's candidate test: the test itself, its `0 < t < closest` acceptance and the lowest-index tie rule are
// unchanged, so the closest hit (and therefore every pixel) is identical to the linear scan's.
//
// Layout: nodes in depth-first order with a skip link (the node to continue with when the ray misses the box), so
// the traversal needs no stack.  Boxes are inflated slightly: the candidate tests run in each object's own space,
// and a hit they report must never be culled by rounding in the world-space box test.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "rt_dev_scene.h"

constexpr uint32_t kBvhMinObjects = 32;  // below this the linear scan is faster
constexpr uint32_t kBvhLeafSize = 4;

struct RtAabb {
    float lo[3], hi[3];
    void reset() { for (int k = 0; k < 3; k++) { lo[k] = 3.0e38f; hi[k] = -3.0e38f; } }
    void grow(const float p[3]) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
    void grow(const RtAabb &b) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
};

// World-space box of an object-space box [lo, hi] under the object's transform (column-major 4x4).
inline RtAabb rt_transform_box(const float *m, const float lo[3], const float hi[3]) {
    RtAabb b; b.reset();
    for (int c = 0; c < 8; c++) {
        float p[3] = {(c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]};
        float w[3];
        for (int r = 0; r < 3; r++) w[r] = m[r] * p[0] + m[4 + r] * p[1] + m[8 + r] * p[2] + m[12 + r];
        b.grow(w);
    }
    return b;
}

// Bounds of everything an object's intersect() can report, or false for unbounded / never-hit kinds.
inline bool rt_object_bounds(const rt_object &o, RtAabb &b) {
    switch (o.kind) {
        case RT_SPHERE: {  // unit sphere in object space
            const float lo[3] = {-1, -1, -1}, hi[3] = {1, 1, 1};
            b = rt_transform_box(o.transform, lo, hi); break;
        }
        case RT_CUBOID: {
            const float lo[3] = {-std::fabs(o.dimension[0]), -std::fabs(o.dimension[1]), -std::fabs(o.dimension[2])};
            const float hi[3] = {std::fabs(o.dimension[0]), std::fabs(o.dimension[1]), std::fabs(o.dimension[2])};
            b = rt_transform_box(o.transform, lo, hi); break;
        }
        case RT_CYLINDER: {  // r = 1, |z| <= 0.5 whatever radius / height say (src/utils/cylinder.rs:45-54)
            const float lo[3] = {-1, -1, -0.5f}, hi[3] = {1, 1, 0.5f};
            b = rt_transform_box(o.transform, lo, hi); break;
        }
        case RT_CONE: {  // x^2 + y^2 = z^2, 0 < z <= 1
            const float lo[3] = {-1, -1, 0}, hi[3] = {1, 1, 1};
            b = rt_transform_box(o.transform, lo, hi); break;
        }
        case RT_TRIANGLE:
            b.reset(); b.grow(o.v1); b.grow(o.v2); b.grow(o.v3); break;
        default:
            return false;  // Plane: unbounded; Union / Substraction: never hit by rays
    }
    for (int k = 0; k < 3; k++) {
        if (!(std::isfinite(b.lo[k]) && std::isfinite(b.hi[k]))) return false;
        float pad = 1e-4f * (b.hi[k] - b.lo[k]) + 1e-4f * std::max(std::fabs(b.lo[k]), std::fabs(b.hi[k])) + 1e-5f;
        b.lo[k] -= pad; b.hi[k] += pad;
    }
    return true;
}

inline bool rt_bvh_wanted(const rt_scene_desc *d) {
    uint32_t bounded = 0;
    RtAabb b;
    for (uint32_t i = 0; i < d->n_objects; i++) bounded += rt_object_bounds(d->objects[i], b) ? 1u : 0u;
    return bounded >= kBvhMinObjects;
}

struct RtBvhBuilder {
    const rt_scene_desc *d;
    std::vector<RtAabb> boxes;
    std::vector<uint32_t> ids;  // bounded object indices, permuted during the build
    std::vector<DevBvhNode> *nodes;
    std::vector<uint32_t> *prims;

    uint32_t build(uint32_t first, uint32_t count) {
        RtAabb box; box.reset();
        RtAabb cbox; cbox.reset();
        for (uint32_t i = first; i < first + count; i++) {
            const RtAabb &b = boxes[ids[i]];
            box.grow(b);
            float c[3] = {0.5f * (b.lo[0] + b.hi[0]), 0.5f * (b.lo[1] + b.hi[1]), 0.5f * (b.lo[2] + b.hi[2])};
            cbox.grow(c);
        }
        const uint32_t me = (uint32_t)nodes->size();
        nodes->push_back(DevBvhNode{});
        DevBvhNode n;
        for (int k = 0; k < 3; k++) { n.lo[k] = box.lo[k]; n.hi[k] = box.hi[k]; }
        if (count <= kBvhLeafSize) {
            // ascending object index inside a leaf keeps the candidate order of the linear scan
            std::sort(ids.begin() + first, ids.begin() + first + count);
            n.leaf = (int32_t)(((uint32_t)prims->size() << 4) | count);
            for (uint32_t i = first; i < first + count; i++) prims->push_back(ids[i]);
        } else {
            int axis = 0;
            float ext = cbox.hi[0] - cbox.lo[0];
            for (int k = 1; k < 3; k++) if (cbox.hi[k] - cbox.lo[k] > ext) { ext = cbox.hi[k] - cbox.lo[k]; axis = k; }
            const uint32_t mid = first + count / 2;
            std::nth_element(ids.begin() + first, ids.begin() + mid, ids.begin() + first + count, [&](uint32_t a, uint32_t b) {
                float ca = boxes[a].lo[axis] + boxes[a].hi[axis], cb = boxes[b].lo[axis] + boxes[b].hi[axis];
                return ca < cb || (ca == cb && a < b);
            });
            n.leaf = -1;
            build(first, mid - first);
            build(mid, first + count - mid);
        }
        n.skip = (int32_t)nodes->size();  // first node after this subtree in depth-first order
        (*nodes)[me] = n;
        return me;
    }
};

// nodes: depth-first with skip links; prims: object indices in leaf order; unbounded: objects tested linearly.
inline void rt_bvh_build(const rt_scene_desc *d, std::vector<DevBvhNode> &nodes, std::vector<uint32_t> &prims, std::vector<uint32_t> &unbounded) {
    RtBvhBuilder B;
    B.d = d; B.nodes = &nodes; B.prims = &prims;
    B.boxes.resize(d->n_objects);
    nodes.clear(); prims.clear(); unbounded.clear();
    for (uint32_t i = 0; i < d->n_objects; i++) {
        if (rt_object_bounds(d->objects[i], B.boxes[i])) B.ids.push_back(i);
        else if (d->objects[i].kind == RT_PLANE) unbounded.push_back(i);
    }
    if (!B.ids.empty()) B.build(0, (uint32_t)B.ids.size());
}
