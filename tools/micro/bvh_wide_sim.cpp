// CPU count of what a ray of the 10 000-sphere frame costs in hierarchy visits, for the tree the library builds
// (RtBvhBuilder, two children per node) and for the same tree with every node's grandchildren pulled up (four children
// per node, the usual collapse): node visits, box tests, 16-byte node loads (a lane loads at most 16 bytes per
// instruction, and diverged lanes each touch their own cache line), leaves whose box is hit, stack depth.
//
//   g++ -O2 -std=c++17 -o /tmp/bvh_wide_sim tools/micro/bvh_wide_sim.cpp && /tmp/bvh_wide_sim
//
// Rays: camera rays of the bench pose on a coarse grid, and "bounce" rays that leave a random small sphere's upper surface
// in a direction normal + random unit vector (what Ray::reflection_ray does for roughness 1, src/ray.rs:123-130), which is
// what the multi-sample launch traces.  Traversal as in wf_trace_pool: near child first, children culled by the closest
// hit so far, leaves tested (here: exact sphere test) when their box is hit.
#include <cstdio>
#include <random>
#include "../../ray-tracing_amd/csrc/rt_bvh.h"

struct Sph { float c[3], r; };
struct Ray { float o[3], d[3]; };
struct Stats { double visits = 0, boxes = 0, loads = 0, leaves = 0, max_sp = 0, rays = 0, sorted_pushes = 0; };

static bool hit_box(const RtAabb &b, const Ray &r, const float inv[3], float closest, float &tmin) {
    float t0 = 0.0f, t1 = closest;
    for (int k = 0; k < 3; k++) {
        float a = (b.lo[k] - r.o[k]) * inv[k], c = (b.hi[k] - r.o[k]) * inv[k];
        if (a > c) std::swap(a, c);
        t0 = std::max(t0, a); t1 = std::min(t1, c);
    }
    tmin = t0;
    return t0 <= t1;
}
static bool hit_sphere(const Sph &s, const Ray &r, float &t) {
    const float oc[3] = {r.o[0] - s.c[0], r.o[1] - s.c[1], r.o[2] - s.c[2]};
    const float a = r.d[0] * r.d[0] + r.d[1] * r.d[1] + r.d[2] * r.d[2];
    const float b = 2.0f * (oc[0] * r.d[0] + oc[1] * r.d[1] + oc[2] * r.d[2]);
    const float c = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2] - s.r * s.r;
    const float disc = b * b - 4.0f * a * c;
    if (disc < 0.0f) return false;
    t = (-b - std::sqrt(disc)) / (2.0f * a);
    return t > 0.0f;
}

struct Wide { int32_t child[4]; int n; };   // child: tree index (inner or leaf)

int main() {
    // the scene of rt::scenes::spheres10k (sizes and placement; the ground sphere is outside the tree, as in rt_bvh2_build)
    std::mt19937 gen(12345);
    std::uniform_real_distribution<float> U(0.0f, 1.0f);
    std::vector<Sph> sph;
    for (int i = 1; i < 10000; i++) {
        const float r = 0.2f + 0.4f * U(gen), x = -60.0f + 120.0f * U(gen), z = -60.0f + 120.0f * U(gen);
        sph.push_back(Sph{{x, r - 0.5f, z}, r});
    }
    const Sph ground{{0.0f, -1000.5f, 0.0f}, 1000.0f};
    std::vector<uint32_t> prims;
    RtBvhBuilder B;
    B.prims = &prims; B.max_depth = kBvh2MaxDepth;
    B.boxes.resize(sph.size()); B.kinds.assign(sph.size(), 0);
    for (size_t i = 0; i < sph.size(); i++) {
        for (int k = 0; k < 3; k++) { B.boxes[i].lo[k] = sph[i].c[k] - sph[i].r; B.boxes[i].hi[k] = sph[i].c[k] + sph[i].r; }
        B.ids.push_back((uint32_t)i);
    }
    B.build(0, (uint32_t)B.ids.size());
    const auto &T = B.tree;
    auto is_leaf = [&](int32_t t) { return T[(size_t)t].left < 0; };
    // four-child nodes: a node's children, with every inner child replaced by its own two children
    std::vector<Wide> wide(T.size());
    size_t inner2 = 0, inner4 = 0;
    for (size_t t = 0; t < T.size(); t++) {
        Wide w{}; w.n = 0;
        if (!is_leaf((int32_t)t)) {
            inner2++;
            for (int32_t c : {T[t].left, T[t].right}) {
                if (is_leaf(c)) w.child[w.n++] = c;
                else { w.child[w.n++] = T[(size_t)c].left; w.child[w.n++] = T[(size_t)c].right; }
            }
        }
        wide[t] = w;
    }
    {   // four-child nodes reachable from the root
        std::vector<int32_t> st{0};
        while (!st.empty()) { const int32_t t = st.back(); st.pop_back(); inner4++; for (int k = 0; k < wide[(size_t)t].n; k++) if (!is_leaf(wide[(size_t)t].child[k])) st.push_back(wide[(size_t)t].child[k]); }
    }
    std::printf("objects %zu, two-child nodes %zu (32 B each), four-child nodes %zu (64 B each)\n", sph.size(), inner2, inner4);

    // rays
    std::vector<Ray> cam, bounce;
    {
        const float pos[3] = {0.0f, 6.0f, 40.0f}; float f[3] = {0.0f, -0.15f, -1.0f};
        const float fl = std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]); for (float &v : f) v /= fl;
        const float right[3] = {-f[2], 0.0f, f[0]};   // forward x up, up = +y, not normalised exactly: fine for a count
        const float up[3] = {right[1] * f[2] - right[2] * f[1], right[2] * f[0] - right[0] * f[2], right[0] * f[1] - right[1] * f[0]};
        const float th = std::tan(0.5f * 45.0f * 3.14159265f / 180.0f), aspect = 16.0f / 9.0f;
        for (int y = 0; y < 108; y++) for (int x = 0; x < 192; x++) {
            const float u = ((x + 0.5f) / 192.0f * 2.0f - 1.0f) * th * aspect, v = (1.0f - (y + 0.5f) / 108.0f * 2.0f) * th;
            Ray r; for (int k = 0; k < 3; k++) { r.o[k] = pos[k]; r.d[k] = f[k] + u * right[k] + v * up[k]; }
            cam.push_back(r);
        }
        std::normal_distribution<float> N(0.0f, 1.0f);
        auto unit = [&](float v[3]) { float l; do { for (int k = 0; k < 3; k++) v[k] = N(gen); l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); } while (l < 1e-6f); for (int k = 0; k < 3; k++) v[k] /= l; };
        while (bounce.size() < 40000) {
            const Sph &s = sph[(size_t)(U(gen) * (float)sph.size()) % sph.size()];
            float n[3], q[3]; unit(n); if (n[1] < 0.0f) n[1] = -n[1];
            unit(q);
            Ray r; float l = 0.0f;
            for (int k = 0; k < 3; k++) { r.o[k] = s.c[k] + n[k] * (s.r + 1e-4f); r.d[k] = n[k] + q[k]; l += r.d[k] * r.d[k]; }
            if (l < 1e-6f) continue;
            l = std::sqrt(l); for (float &v : r.d) v /= l;
            bounce.push_back(r);
        }
    }

    auto run = [&](const std::vector<Ray> &rays, bool four, bool sort_children, Stats &S) {
        for (const Ray &r : rays) {
            float inv[3]; for (int k = 0; k < 3; k++) inv[k] = 1.0f / (std::fabs(r.d[k]) < 1e-20f ? 1e-20f : r.d[k]);
            float closest = 3.0e38f, t;
            if (hit_sphere(ground, r, t)) closest = t;   // the linear list's result seeds `closest`
            std::vector<int32_t> st{0};
            size_t max_sp = 0;
            while (!st.empty()) {
                const int32_t node = st.back(); st.pop_back();
                S.visits += 1;
                int32_t ch[4]; int n = 0;
                if (four) { n = wide[(size_t)node].n; for (int k = 0; k < n; k++) ch[k] = wide[(size_t)node].child[k]; S.loads += 4; }
                else { ch[0] = T[(size_t)node].left; ch[1] = T[(size_t)node].right; n = 2; S.loads += 2; }
                S.boxes += n;
                std::pair<float, int32_t> in[4]; int ni = 0;
                for (int k = 0; k < n; k++) {
                    float tmin;
                    if (!hit_box(T[(size_t)ch[k]].box, r, inv, closest, tmin)) continue;
                    if (is_leaf(ch[k])) {
                        S.leaves += 1;
                        const Sph &s = sph[prims[T[(size_t)ch[k]].first]];
                        if (hit_sphere(s, r, t) && t < closest) closest = t;
                    } else in[ni++] = {tmin, ch[k]};
                }
                if (sort_children) std::sort(in, in + ni, [](auto &a, auto &b) { return a.first > b.first; });   // farthest pushed first
                else if (ni == 2 && !four && in[0].first < in[1].first) std::swap(in[0], in[1]);
                if (ni > 1) S.sorted_pushes += ni - 1;
                for (int k = 0; k < ni; k++) st.push_back(in[k].second);
                max_sp = std::max(max_sp, st.size());
            }
            S.max_sp = std::max(S.max_sp, (double)max_sp);
            S.rays += 1;
        }
    };
    struct { const char *name; const std::vector<Ray> *rays; } sets[2] = {{"camera rays", &cam}, {"bounce rays", &bounce}};
    for (auto &set : sets) {
        Stats a, b, c;
        run(*set.rays, false, true, a);
        run(*set.rays, true, true, b);
        run(*set.rays, true, false, c);
        auto pr = [&](const char *what, const Stats &s) {
            std::printf("  %-44s visits %6.2f  box tests %6.2f  16-byte loads %6.2f  leaf tests %5.2f  entries pushed %5.2f  deepest stack %2.0f\n", what,
                        s.visits / s.rays, s.boxes / s.rays, s.loads / s.rays, s.leaves / s.rays, s.sorted_pushes / s.rays, s.max_sp);
        };
        std::printf("%s (%.0f):\n", set.name, a.rays);
        pr("two children per node, near one first", a);
        pr("four children per node, sorted by entry", b);
        pr("four children per node, stored order", c);
    }
    return 0;
}
