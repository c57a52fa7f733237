// rt_bvh.h — host-side construction of the bounding-volume hierarchy used for scenes with many objects.
//
// The reference scans every object for every ray (RayTracing::trace_ray, src/ray_tracing/ray_tracing.rs:59-78);
// at 10 000 objects that is ~10^4 candidate tests per ray.  The hierarchy only decides WHICH objects get the
// reference's candidate test: the test itself, its `0 < t < closest` acceptance and the lowest-index tie rule are
// unchanged, so the closest hit (and therefore every pixel) is identical to the linear scan's.
//
// Layout: nodes in depth-first order with a skip link (the node to continue with when the ray misses the box), so
// the traversal needs no stack; one such layout per sign octant of the ray direction (see RtBvhBuilder).  Boxes are inflated slightly: the candidate tests run in each object's own space,
// and a hit they report must never be culled by rounding in the world-space box test.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "rt_dev_scene.h"

constexpr uint32_t kBvhMinObjects = 32;  // below this the linear scan is faster
constexpr uint32_t kBvhLeafSize = 1;  // one object per leaf measured fastest (a candidate test costs about five node visits)

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

// Build: top-down, binned surface-area heuristic on the longest centroid axis (median split when the heuristic finds
// no useful plane).  The tree is then written out EIGHT times, once per sign octant of the ray direction: in layout
// `oct` every inner node is followed by the child that a ray of that octant enters first along the node's split
// axis, then by the other child.  Near-first order lets `tmin <= closest` cull most of the far subtrees while the
// traversal stays stackless (skip links).  All layouts share the leaf -> object index list.
constexpr int kBvhBins = 16;
constexpr uint32_t kBvhLayouts = 8;

struct RtBvhBuilder {
    struct TreeNode { RtAabb box; int32_t left, right, axis; uint32_t first, count; };  // left < 0: leaf over prims[first, first + count)
    std::vector<RtAabb> boxes;
    std::vector<uint32_t> ids;  // bounded object indices, permuted during the build
    std::vector<TreeNode> tree;
    std::vector<uint32_t> *prims;
    std::vector<int> kinds;     // object kind by object index

    static float half_area(const RtAabb &b) {
        const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
    float centroid2(uint32_t id, int axis) const { return boxes[id].lo[axis] + boxes[id].hi[axis]; }  // twice the centroid

    int32_t build(uint32_t first, uint32_t count) {
        RtAabb box; box.reset();
        float clo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, chi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        for (uint32_t i = first; i < first + count; i++) {
            box.grow(boxes[ids[i]]);
            for (int k = 0; k < 3; k++) { const float c = centroid2(ids[i], k); clo[k] = std::min(clo[k], c); chi[k] = std::max(chi[k], c); }
        }
        const int32_t me = (int32_t)tree.size();
        tree.push_back(TreeNode{});
        TreeNode n; n.box = box; n.left = n.right = -1; n.axis = 0; n.first = 0; n.count = 0;
        if (count <= kBvhLeafSize) {
            // ascending object index inside a leaf keeps the candidate order of the linear scan
            std::sort(ids.begin() + first, ids.begin() + first + count);
            n.first = (uint32_t)prims->size(); n.count = count;
            for (uint32_t i = first; i < first + count; i++) prims->push_back(ids[i]);
        } else {
            int axis = 0;
            for (int k = 1; k < 3; k++) if (chi[k] - clo[k] > chi[axis] - clo[axis]) axis = k;
            uint32_t mid = first + count / 2;
            bool split_done = false;
            const float ext = chi[axis] - clo[axis];
            if (ext > 0.0f && std::isfinite(ext)) {
                RtAabb bin_box[kBvhBins]; uint32_t bin_n[kBvhBins];
                for (int b = 0; b < kBvhBins; b++) { bin_box[b].reset(); bin_n[b] = 0; }
                const float scale = (float)kBvhBins / ext;
                auto bin_of = [&](uint32_t id) { int b = (int)((centroid2(id, axis) - clo[axis]) * scale); return std::min(std::max(b, 0), kBvhBins - 1); };
                for (uint32_t i = first; i < first + count; i++) { const int b = bin_of(ids[i]); bin_box[b].grow(boxes[ids[i]]); bin_n[b]++; }
                float right_area[kBvhBins]; RtAabb acc; acc.reset();
                for (int b = kBvhBins - 1; b > 0; b--) { acc.grow(bin_box[b]); right_area[b] = half_area(acc); }  // only read where bins b.. are non-empty
                acc.reset();
                float best_cost = 3.0e38f; int best_plane = -1; uint32_t nl = 0;
                for (int b = 0; b < kBvhBins - 1; b++) {  // plane after bin b
                    if (bin_n[b]) acc.grow(bin_box[b]);
                    nl += bin_n[b];
                    const uint32_t nr = count - nl;
                    if (nl == 0 || nr == 0) continue;
                    const float cost = half_area(acc) * (float)nl + right_area[b + 1] * (float)nr;
                    if (cost < best_cost) { best_cost = cost; best_plane = b; }
                }
                if (best_plane >= 0) {
                    auto it = std::stable_partition(ids.begin() + first, ids.begin() + first + count, [&](uint32_t id) { return bin_of(id) <= best_plane; });
                    mid = (uint32_t)(it - ids.begin());
                    split_done = mid > first && mid < first + count;
                }
            }
            if (!split_done) {
                mid = first + count / 2;
                std::nth_element(ids.begin() + first, ids.begin() + mid, ids.begin() + first + count, [&](uint32_t a, uint32_t b) {
                    const float ca = centroid2(a, axis), cb = centroid2(b, axis);
                    return ca < cb || (ca == cb && a < b);
                });
            }
            n.axis = axis;
            n.left = build(first, mid - first);           // lower centroids along `axis`
            n.right = build(mid, first + count - mid);
        }
        tree[(size_t)me] = n;
        return me;
    }

    // Depth-first layout for one sign octant (bit k of `oct` set: the ray direction's k-th component is negative).
    void emit(int32_t t, uint32_t oct, std::vector<DevBvhNode> &out) const {
        const TreeNode &n = tree[(size_t)t];
        const size_t me = out.size();
        out.push_back(DevBvhNode{});
        DevBvhNode d;
        for (int k = 0; k < 3; k++) { d.lo[k] = n.box.lo[k]; d.hi[k] = n.box.hi[k]; }
        if (n.left < 0) {
            d.leaf = (int32_t)(((uint32_t)kinds[(*prims)[n.first]] << 28) | (*prims)[n.first]);  // one object per leaf: kind and index
        } else {
            d.leaf = -1;
            const bool negative = (oct >> n.axis) & 1u;
            emit(negative ? n.right : n.left, oct, out);
            emit(negative ? n.left : n.right, oct, out);
        }
        d.skip = (int32_t)out.size();  // first node after this subtree (each layout is emitted into its own array)
        out[me] = d;
    }
};

// nodes: kBvhLayouts depth-first layouts of `nodes.size() / kBvhLayouts` nodes each, skip links relative to the layout's
// first node; unbounded: objects tested linearly.
inline void rt_bvh_build(const rt_scene_desc *d, std::vector<DevBvhNode> &nodes, std::vector<uint32_t> &unbounded) {
    std::vector<uint32_t> prims;  // object indices in leaf order
    RtBvhBuilder B;
    B.prims = &prims;
    B.boxes.resize(d->n_objects);
    B.kinds.resize(d->n_objects);
    for (uint32_t i = 0; i < d->n_objects; i++) B.kinds[i] = d->objects[i].kind;
    nodes.clear(); unbounded.clear();
    for (uint32_t i = 0; i < d->n_objects; i++) {
        if (rt_object_bounds(d->objects[i], B.boxes[i])) B.ids.push_back(i);
        else if (d->objects[i].kind == RT_PLANE) unbounded.push_back(i);
    }
    if (B.ids.empty()) return;
    B.build(0, (uint32_t)B.ids.size());
    for (uint32_t oct = 0; oct < kBvhLayouts; oct++) {
        std::vector<DevBvhNode> one;
        one.reserve(B.tree.size());
        B.emit(0, oct, one);
        nodes.insert(nodes.end(), one.begin(), one.end());
    }
}
