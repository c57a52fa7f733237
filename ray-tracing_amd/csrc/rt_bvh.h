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

// Bounds of everything an object's intersect() can report, or false for unbounded / never-hit kinds and for cones (below).
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
        case RT_CONE:
            // No bounds, although the surface x^2 + y^2 = z^2, 0 < z <= 1 has some.  Cone::intersect (src/utils/cone.rs:73-152)
            // solves a quadratic whose leading coefficient dx^2 + dy^2 - dz^2 is ~0 for every ray direction near the cone's own
            // slope, wherever the ray is; there (-b - sqrt(disc)) / 2a is rounding noise divided by almost nothing, and when
            // that lands between z = 0 and z = 1 the reference reports a hit -- on rays that pass nowhere near the cone (seen
            // once in 7 * 10^8 pixels of random scenes).  Only testing every ray reproduces those, so cones are tested like
            // planes: always.
            return false;
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

// DevObject::diag: a sphere whose transform and inv_transform both have, in their first three rows, non-zero finite
// diagonal entries, a finite translation and (+-)0 everywhere else (what Sphere::new builds: no rotation,
// src/utils/sphere.rs:26-37, :61-70).
inline bool rt_object_is_diag_sphere(const rt_object &o) {
    if (o.kind != RT_SPHERE) return false;
    const float *ms[2] = {o.transform, o.inv_transform};
    for (const float *m : ms)
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 3; r++) {
                const float v = m[4 * c + r];
                if (!std::isfinite(v)) return false;
                if (c == r && v == 0.0f) return false;
                if (c < 3 && c != r && v != 0.0f) return false;
            }
    return true;
}
// Kind code a hierarchy leaf carries for object `o` (see RT_LEAF_DIAG_SPHERE).
inline int rt_leaf_kind(const rt_object &o) { return rt_object_is_diag_sphere(o) ? RT_LEAF_DIAG_SPHERE : (int)o.kind; }

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
    std::vector<int> kinds;     // leaf kind code by object index (rt_leaf_kind)
    uint32_t max_depth = 0;     // 0: unlimited; else no leaf lies deeper than this many inner nodes below the root (see build)

    static float half_area(const RtAabb &b) {
        const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
    float centroid2(uint32_t id, int axis) const { return boxes[id].lo[axis] + boxes[id].hi[axis]; }  // twice the centroid

    static uint32_t ceil_log2(uint32_t n) { uint32_t l = 0; while ((1u << l) < n) l++; return l; }

    // `depth`: inner nodes above this one.  With max_depth set, a subtree that could no longer be finished within the limit
    // by halving is split at the median (a balanced tree over c objects is ceil(log2 c) levels deep), so traversal stacks of
    // max_depth entries never overflow.
    int32_t build(uint32_t first, uint32_t count, uint32_t depth = 0) {
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
            const bool balanced = max_depth != 0u && depth + 1u + ceil_log2(count) >= max_depth;
            if (!balanced && ext > 0.0f && std::isfinite(ext)) {
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
            n.left = build(first, mid - first, depth + 1u);           // lower centroids along `axis`
            n.right = build(mid, first + count - mid, depth + 1u);
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
    for (uint32_t i = 0; i < d->n_objects; i++) B.kinds[i] = rt_leaf_kind(d->objects[i]);
    nodes.clear(); unbounded.clear();
    for (uint32_t i = 0; i < d->n_objects; i++) {
        if (rt_object_bounds(d->objects[i], B.boxes[i])) B.ids.push_back(i);
        else if (d->objects[i].kind == RT_PLANE || d->objects[i].kind == RT_CONE) unbounded.push_back(i);
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

// ------------------------------------------------------------------------------------------------
// Second form of the hierarchy, for the kernel that re-groups rays (rt_wavefront.h): only inner nodes, each holding the
// boxes of BOTH children, so one visit decides about two subtrees and a ray needs half as many dependent fetches.  The
// boxes are quantised to 16 bits (binary16) per coordinate relative to the centre of the root box: a node is 32 bytes (two 16-byte loads)
// instead of 64, and the whole tree of 10 000 objects is 320 KB.  Traversal uses a small per-lane stack (near child first),
// so the tree is depth-limited (kBvh2MaxDepth) and needs no per-octant copies.
// Objects so large that they would stretch the grid (a ground sphere of radius 1000 under spheres of radius 0.5) are left
// out of this tree and tested linearly, like the planes.
// ------------------------------------------------------------------------------------------------
#ifndef RT_WF_STACK
#define RT_WF_STACK 20
#endif
constexpr uint32_t kBvh2MaxDepth = RT_WF_STACK;     // inner levels; the kernel's traversal stack has this many entries
constexpr uint32_t kBvh2MaxOutliers = 8;
constexpr uint32_t kBvh2MaxNodes = 1u << 26;   // inner references are byte offsets in 31 bits

struct RtBvh2 {
    std::vector<DevBvh2Node> nodes;   // inner nodes, depth-first, root first
    std::vector<uint32_t> linear;     // planes and outliers, ascending object index
    float origin[3] = {0, 0, 0}, scale[3] = {1, 1, 1};   // decoded coordinate = origin + q * scale
    bool usable = false;              // at least two objects in the tree
};

inline void rt_bvh2_build(const rt_scene_desc *d, RtBvh2 &out) {
    out = RtBvh2();
    std::vector<uint32_t> prims;
    RtBvhBuilder B;
    B.prims = &prims;
    B.max_depth = kBvh2MaxDepth;
    B.boxes.resize(d->n_objects);
    B.kinds.resize(d->n_objects);
    std::vector<uint32_t> ids;
    for (uint32_t i = 0; i < d->n_objects; i++) {
        B.kinds[i] = rt_leaf_kind(d->objects[i]);
        if (rt_object_bounds(d->objects[i], B.boxes[i])) ids.push_back(i);
        else if (d->objects[i].kind == RT_PLANE || d->objects[i].kind == RT_CONE) out.linear.push_back(i);
    }
    // outliers: while taking one object out shrinks the root box's surface at least 16-fold, take it out
    auto root_of = [&](const std::vector<uint32_t> &v, uint32_t skip) { RtAabb r; r.reset(); for (uint32_t id : v) if (id != skip) r.grow(B.boxes[id]); return r; };
    for (uint32_t round = 0; round < kBvh2MaxOutliers && ids.size() > 2; round++) {
        const RtAabb root = root_of(ids, 0xFFFFFFFFu);
        uint32_t cand[6]; int nc = 0;
        for (int k = 0; k < 3; k++) {   // the objects that define the root box's six faces
            uint32_t lo_id = ids[0], hi_id = ids[0];
            for (uint32_t id : ids) { if (B.boxes[id].lo[k] < B.boxes[lo_id].lo[k]) lo_id = id; if (B.boxes[id].hi[k] > B.boxes[hi_id].hi[k]) hi_id = id; }
            cand[nc++] = lo_id; cand[nc++] = hi_id;
        }
        float best_area = RtBvhBuilder::half_area(root); uint32_t best_id = 0xFFFFFFFFu;
        for (int c = 0; c < nc; c++) {
            const float a = RtBvhBuilder::half_area(root_of(ids, cand[c]));
            if (a < best_area) { best_area = a; best_id = cand[c]; }
        }
        if (best_id == 0xFFFFFFFFu || !(best_area * 16.0f <= RtBvhBuilder::half_area(root))) break;
        ids.erase(std::find(ids.begin(), ids.end(), best_id));
        out.linear.push_back(best_id);
    }
    std::sort(out.linear.begin(), out.linear.end());
    if (ids.size() < 2) return;
    // a balanced tree over this many objects must fit the traversal stack, or the depth limit could not be kept
    if (RtBvhBuilder::ceil_log2((uint32_t)ids.size()) + 1u > kBvh2MaxDepth) return;
    B.ids = ids;
    B.build(0, (uint32_t)B.ids.size());
    // Coordinates are stored as binary16 numbers relative to the centre of the root box, +-32000 units at its faces.  The
    // spacing of binary16 is 2^-11 of the distance from the centre (16 units at the faces, finer towards the centre); every
    // plane first moves outward by one unit (2^-16 of the extent) and then to the next representable value outward.
    const RtAabb root = B.tree[0].box;
    double org[3], scl[3];
    for (int k = 0; k < 3; k++) {
        double ext = (double)root.hi[k] - (double)root.lo[k];
        if (!(ext > 0.0)) ext = 1e-6;
        scl[k] = (double)(float)(ext / 64000.0);
        org[k] = (double)(float)(0.5 * ((double)root.lo[k] + (double)root.hi[k]));
        out.scale[k] = (float)scl[k]; out.origin[k] = (float)org[k];
    }
    // binary16 with directed rounding: the nearest representable value at or beyond x in the given direction
    auto half_dir = [](double x, bool up) -> uint16_t {
        const bool neg = x < 0.0;
        const double a = neg ? -x : x;
        const bool mag_up = (up != neg);
        if (a == 0.0) return 0;
        uint32_t bits;
        if (a < 6.103515625e-05) {                       // below 2^-14: subnormal spacing 2^-24
            const double n = a * 16777216.0;
            bits = (uint32_t)(mag_up ? std::ceil(n) : std::floor(n));   // 1024 rolls over into the first normal number
        } else {
            int e; const double m = std::frexp(a, &e);   // a = m * 2^e, m in [0.5, 1)
            const double f = (m * 2.0 - 1.0) * 1024.0;
            uint32_t mant = (uint32_t)(mag_up ? std::ceil(f) : std::floor(f));
            uint32_t ex = (uint32_t)(e - 1 + 15);
            if (mant == 1024u) { mant = 0u; ex += 1u; }
            if (ex >= 31u) { ex = 30u; mant = 1023u; }   // 65504: never reached by coordinates within +-32032
            bits = (ex << 10) | mant;
        }
        return (uint16_t)((neg ? 0x8000u : 0u) | bits);
    };
    auto quant = [&](const RtAabb &b, uint16_t q[6]) {
        for (int k = 0; k < 3; k++) {
            const double lo = ((double)b.lo[k] - org[k]) / scl[k] - 1.0, hi = ((double)b.hi[k] - org[k]) / scl[k] + 1.0;
            q[2 * k] = half_dir(lo, false); q[2 * k + 1] = half_dir(hi, true);   // one dword per axis: lower | upper << 16
        }
    };
    // inner nodes in depth-first order
    std::vector<int32_t> index_of(B.tree.size(), -1);
    std::vector<int32_t> order;
    {
        std::vector<int32_t> stack{0};
        while (!stack.empty()) {
            const int32_t t = stack.back(); stack.pop_back();
            if (B.tree[(size_t)t].left < 0) continue;
            index_of[(size_t)t] = (int32_t)order.size(); order.push_back(t);
            stack.push_back(B.tree[(size_t)t].right); stack.push_back(B.tree[(size_t)t].left);
        }
    }
    auto ref_of = [&](int32_t t) -> int32_t {
        const RtBvhBuilder::TreeNode &n = B.tree[(size_t)t];
        if (n.left >= 0) return index_of[(size_t)t] * (int32_t)sizeof(DevBvh2Node);   // inner node: byte offset of the node
        const uint32_t obj = prims[n.first];
        return (int32_t)~(((uint32_t)B.kinds[obj] << 28) | obj);   // leaf: one object, named in the reference itself
    };
    out.nodes.resize(order.size());
    for (size_t i = 0; i < order.size(); i++) {
        const RtBvhBuilder::TreeNode &n = B.tree[(size_t)order[i]];
        DevBvh2Node g;
        quant(B.tree[(size_t)n.left].box, g.q);
        quant(B.tree[(size_t)n.right].box, g.q + 6);
        g.ref[0] = ref_of(n.left); g.ref[1] = ref_of(n.right);
        out.nodes[i] = g;
    }
    out.usable = !out.nodes.empty() && out.nodes.size() < kBvh2MaxNodes;
}
