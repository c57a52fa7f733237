// rt_wavefront.h — path tracing of large diffuse scenes with re-grouped rays.
//
// What it computes: Scene::pixel for scenes with `diffuse` set and no refractive material, i.e. RayTracing::color_diffuse's
// reflective arm (src/ray_tracing/ray_tracing.rs:80-114, :155-157) over RayTracing::trace_ray (:59-78), then render_chunk's
// accumulate / clamp / to_rgba (src/renderer.rs:59-75).  color_diffuse returns its child's value unchanged, so a path is a
// forward loop: trace, on a miss return light + ambient * contribution, on a hit add the emission, draw the next direction.
//
// Why a second formulation next to rt_trace_kernel: with one pixel per lane for a whole path, the lanes of a wavefront
// disagree on how many bounces their path has and on how many hierarchy nodes each ray visits; on the 10 000-sphere
// workload 30 % of the lanes did useful work.  Here a wavefront owns a pool of kWfPool paths (a 64x8 pixel tile) and
// works on it bounce by bounce:
//   trace   lanes pull rays from the pool as they finish the previous one (refilled in batches, when enough lanes are
//           idle), so a short traversal does not wait for the longest one of its wavefront.  A lane whose ray reaches a
//           leaf parks until enough lanes have a leaf to test, so the candidate test runs with many lanes instead of a few.
//   shade   64 hits at a time, all lanes in the same code; the paths that go on are written back to the pool compacted
//           (ballot + prefix count), so the next bounce starts with full wavefronts whatever the paths' lengths.
// The pool (ray, carried light, contribution, RNG key, pixel) lives in a per-wavefront slice of a device buffer that is
// rewritten every bounce and stays cache resident; the (distance, object) results of a trace pass live in LDS.
// Every path does the arithmetic it does in rt_trace_kernel, in the same order: frames are bit-identical.
#pragma once
#include "rt_kernels.h"

#pragma clang fp contract(off)

#ifndef RT_WF_POOL
#define RT_WF_POOL 512
#endif
#ifndef RT_WF_WALK_EIGHTHS
#define RT_WF_WALK_EIGHTHS 5   // the walk loop goes on while this many eighths of the lanes that hold a ray are walking
#endif
constexpr uint32_t kWfPool = RT_WF_POOL;              // paths per wavefront pool = pixels per tile
constexpr uint32_t kWfTileW = kWfPool / 8u, kWfTileH = 8u;   // tile = sub-tiles of 8x8 pixels side by side
// Traversal stack entries per lane (>= kBvh2MaxDepth of rt_bvh.h).  LDS is granted in blocks of 1280 bytes: 20 entries x 64
// lanes x 4 bytes are exactly four blocks, 32 wavefronts per CU; 21 or 22 entries take a fifth block and leave room for 25
// (measured: 22 entries 325.1 ms per 8K frame, 20 entries 320.0, and 317.7 compiled for 8 wavefronts per SIMD).
#ifndef RT_WF_STACK
#define RT_WF_STACK 20
#endif
constexpr uint32_t kWfStack = RT_WF_STACK;
constexpr uint32_t kWfHitMiss = 0xFFFFFFFFu, kWfHitInvalid = 0xFFFFFFFEu;

#ifndef RT_WF_REFILL
#define RT_WF_REFILL 16      // idle lanes that trigger a refill from the pool
#endif
#ifndef RT_WF_WAVES
#define RT_WF_WAVES 8        // wavefronts per SIMD the kernel is compiled for (64 registers, no spills)
#endif

// Diagnostic build (-DRT_WF_STATS): per phase, how often it ran and with how many lanes (never enabled in the shipped library).
enum { WS_REFILL = 0, WS_WALK, WS_LEAF, WS_FINISH, WS_SHADE, WS_PREPARE, WS_N };
#ifdef RT_WF_STATS
#define WS_ADD(ph, mask) do { const uint64_t m_ = (mask); if (threadIdx.x == 0u) { s_stat[2 * (ph)] += 1u; s_stat[2 * (ph) + 1] += (uint32_t)__popcll(m_); } } while (0)
#define WS_PARAM , uint32_t *s_stat
#define WS_PASS , s_stat
#else
#define WS_ADD(ph, mask) do { } while (0)
#define WS_PARAM
#define WS_PASS
#endif

// Layout of WfArgs::ctrl.  Work is handed out as items from one counter.  Along the launch order of the tile rows the items
// shrink: rows [0, R1) are dealt as whole tiles (512 paths), [R1, R2) as halves (256), [R2, R3) as quarters (128), the rest as
// eighths (64), so that the launch does not end on a few wavefronts that started a large, expensive tile late.
// ITEMS_c = first item index of size class c + 1 (ITEMS_3 = number of items), TILE_c = first whole tile of class c.
enum { kWfCtrlNext = 0, kWfCtrlRowOffset = 1, kWfCtrlItems0 = 2, kWfCtrlItems1 = 3, kWfCtrlItems2 = 4, kWfCtrlItems3 = 5,
       kWfCtrlTile2 = 6, kWfCtrlTile3 = 7, kWfCtrlRows = 16 };

constexpr float kWfSlack = 4.76837158e-7f;   // 2^-21: bound on the relative rounding error of a decoded plane distance

struct WfArgs {
    float4 *a;   // per pool entry: ray origin xyz, RNG key
    float4 *b;   //                 ray direction xyz, pixel index
    float4 *c;   //                 carried light xyz, contribution x
    float4 *d;   //                 contribution y, z; closest hit so far: distance, object | sub << 30 (kWfHitMiss: none)
    uint32_t *ctrl;   // kWfCtrl* header words, then [kWfCtrlRows + r] = rays traced in tile row r
    uint32_t tiles_x, tiles_y;
    unsigned long long *stats;   // diagnostic builds only
    // rt_wf_chain_kernel: per pixel slot, planes a (ray origin, RNG key), b (ray direction, contribution x), c (carried light,
    // contribution y) as above -- only written for bounce rays -- and
    float *d1;        // contribution z
    float2 *h;        // closest hit so far: distance, object word (see kChPrimaryBit)
    float4 *e;        // the pixel's accumulator entry so far (Renderer.accumulated[pos] plus the samples added)
    uint2 *g;         // pixel index, samples of this launch already added | bounce depth of the current path << 24
    float4 *k0, *k1, *k2;   // what every sample of the slot's pixel shares, from the pixel's camera ray (see the kernel): origin of the
                      // first bounce ray + light x, normal at the first hit + light y, contribution + light z
    uint32_t samples; // accumulating frames this launch renders: frame_index = F.frame_index ... F.frame_index + samples - 1
    // rt_wf_diffuse_kernel: closest hit of every pixel's camera ray (distance, object | sub << 30 or kWfHitMiss), kept from one
    // frame to the next of the same camera and scene: trace_ray(camera ray) does not depend on the frame (src/renderer.rs:49)
    float2 *first_hit;
    uint32_t first_hit_mode;   // kFirstHitOff / kFirstHitWrite (this launch traces the camera rays and fills the table) / kFirstHitRead
};
enum { kFirstHitOff = 0, kFirstHitWrite = 1, kFirstHitRead = 2 };

// pixel of pool slot j in stage 0: sub-tile j / 64, lane-in-tile j % 64 as in lane_pixel
RT_DEV bool wf_slot_pixel(const DevCamera &C, const DevFrame &F, uint32_t x0, uint32_t y0, uint32_t j, uint32_t &pos) {
    const uint32_t l = j & 63u;
    const uint32_t x = x0 + (j >> 6) * 8u + (l & 7u), y = y0 + (l >> 3);
    pos = y * C.width + x;
    return x < C.width && y < F.row_end;
}

RT_DEV float rt_max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
RT_DEV float rt_min3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }

RT_DEV uint32_t wf_pack_hit(int best, int best_sub) { return best < 0 ? kWfHitMiss : ((uint32_t)best | ((uint32_t)best_sub << 30)); }

// trace_ray's scan over the objects outside the hierarchy (planes, outliers): wave-uniform loop, scalar loads, the
// reference's acceptance rule.  Runs with all lanes busy (shade phase), not inside the traversal.
RT_DEV void wf_linear_objects(const DevScene &S, f3 ro, f3 rd, float &closest, int &best, int &best_sub) {
    Cnt<false> nocnt;
    for (uint32_t k = 0; k < S.n_bvh2_linear; k++) {
        const uint32_t idx = S.bvh2_linear[k];
        const DevObject &o = S.objects[idx];
        float t = 0.0f;
        int sub = 0;
        if (cand_any<false>(o.kind, o, ro, rd, t, sub, nocnt) && t > 0.0f && (t < closest || (t == closest && (int)idx < best))) { closest = t; best = (int)idx; best_sub = sub; }
    }
}

// trace_ray for the n rays of the pool (stage 0: the tile's camera rays).  In: pd[j].zw = closest hit among the linear
// objects; out: pd[j].zw = closest hit overall (t bits, object | sub << 30).
// CHAIN (rt_wf_chain_kernel): the pool is the wavefront's array of pixel slots, used in place.  ph[j] = (closest hit among the
// linear objects, object word) in, (closest hit overall, object word) out; an object word of kWfHitInvalid marks an empty slot,
// bit kChPrimaryBit a camera ray, whose origin and direction are not stored: pg[j].x is its pixel.
constexpr uint32_t kChPrimaryBit = 1u << 29, kChObjMask = 0x0FFFFFFFu;
RT_DEV uint32_t ch_pack_hit(int best, int best_sub) { return (best < 0 ? kChObjMask : (uint32_t)best) | ((uint32_t)best_sub << 30); }
template <bool CHAIN = false>
RT_DEV void wf_trace_pool(const DevScene &S, const DevCamera &C, const DevFrame &F, bool primary, uint32_t n, uint32_t x0, uint32_t y0,
                          const float4 *pa, const float4 *pb, float4 *pd, uint32_t *s_stack, uint32_t &rays WS_PARAM,
                          float2 *ph = nullptr, const uint2 *pg = nullptr) {
    const char *nodes = reinterpret_cast<const char *>(S.bvh2_nodes);   // inner references are byte offsets into this array
    const f3 gorg = ld3(S.bvh2_origin), gscl = ld3(S.bvh2_scale);
    uint32_t *stack = s_stack + threadIdx.x;   // entry e of this lane at stack[e * 64]
    uint32_t next = 0u;   // wave-uniform: first pool slot not handed out yet
    int ridx = -1;        // pool slot this lane is tracing, -1: idle
    int cur = -1;         // inner node to visit next, -1: none left
    int pend0 = 0, pend1 = 0;   // leaves waiting for their candidate test (negative references), 0: none
    uint32_t sp = 0u;
    f3 ro = splat3(0.0f), rd = splat3(0.0f), ka = splat3(0.0f), kn = splat3(0.0f), kf = splat3(0.0f);
    struct { uint32_t x, y, z; } rot = {0u, 0u, 0u};   // 16 where the direction component is negative
    int best = -1, best_sub = 0;
    float closest = RT_F32_MAX;
    Cnt<false> nocnt;
    for (;;) {
        const uint64_t busy = __ballot(ridx >= 0);
        if (next < n && (64 - __popcll(busy) >= RT_WF_REFILL || busy == 0ull)) {
            // ---- refill: idle lanes take the next pool slots in lane order
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(~busy >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)~busy, 0u));
            WS_ADD(WS_REFILL, ~busy);
            if (ridx < 0) {
                const uint32_t j = next + rank;
                if (j < n) {
                    bool valid = true;
                    float2 h0 = make_float2(0.0f, 0.0f);
                    if (CHAIN) {
                        h0 = ph[j];
                        const uint32_t hw = __float_as_uint(h0.y);
                        valid = hw != kWfHitInvalid;
                        if (valid && (hw & kChPrimaryBit)) {
                            ro = ld3(C.position); rd = ld3(C.ray_dirs + 3u * (size_t)pg[j].x);
                        } else if (valid) {
                            const float4 A = pa[j], B = pb[j];
                            ro = mk3(A.x, A.y, A.z); rd = mk3(B.x, B.y, B.z);
                        }
                    } else if (primary) {
                        uint32_t pos;
                        valid = wf_slot_pixel(C, F, x0, y0, j, pos);
                        if (valid) { ro = ld3(C.position); rd = ld3(C.ray_dirs + 3u * (size_t)pos); }
                    } else {
                        const float4 A = pa[j], B = pb[j];
                        ro = mk3(A.x, A.y, A.z); rd = mk3(B.x, B.y, B.z);
                    }
                    if (valid) {
                        if (!CHAIN) h0 = *reinterpret_cast<const float2 *>(&pd[j].z);
                        ridx = (int)j;
                        rays += 1u;
                        closest = h0.x;
                        const uint32_t hb = __float_as_uint(h0.y);
                        if (CHAIN) { best = (hb & kChObjMask) == kChObjMask ? -1 : (int)(hb & kChObjMask); best_sub = (int)((hb >> 30) & 1u); }
                        else { best = hb == kWfHitMiss ? -1 : (int)(hb & 0x0FFFFFFFu); best_sub = hb == kWfHitMiss ? 0 : (int)(hb >> 30); }
                        pend0 = 0; pend1 = 0; sp = 0u; cur = 0;
                        // box test on the quantised planes: t = (origin + q * scale - ro) / rd = q * ka + kb
                        // (the box test only steers the traversal: an approximate reciprocal is as good as the exact quotient).
                        // The rounding of ka, kb and the FMA is at most 2^-21 of |kb| + 32768 |ka| in t; entry planes are taken
                        // that much earlier (kn) and exit planes that much later (kf), so no box is missed however far the ray
                        // starts from the scene.
                        // A zero direction component would give 0 * inf = NaN planes, which never cull: such a ray (the camera rays
                        // of the frame's centre column, for one) would walk every node its other two slabs let through.  For the
                        // box test the component is at least 1e-20 in magnitude: the planes it meets lie at +-huge, finite t.
                        const f3 inv = mk3(__builtin_amdgcn_rcpf(__builtin_copysignf(rt_max(rt_abs(rd.x), 1e-20f), rd.x)),
                                           __builtin_amdgcn_rcpf(__builtin_copysignf(rt_max(rt_abs(rd.y), 1e-20f), rd.y)),
                                           __builtin_amdgcn_rcpf(__builtin_copysignf(rt_max(rt_abs(rd.z), 1e-20f), rd.z)));
                        ka = gscl * inv;
                        const f3 kb = (gorg - ro) * inv;
                        const f3 slack = mk3(kWfSlack * (rt_abs(kb.x) + 32768.0f * rt_abs(ka.x)), kWfSlack * (rt_abs(kb.y) + 32768.0f * rt_abs(ka.y)),
                                             kWfSlack * (rt_abs(kb.z) + 32768.0f * rt_abs(ka.z)));
                        kn = kb - slack;
                        kf = kb + slack;
                        rot.x = (__float_as_uint(rd.x) >> 31) << 4; rot.y = (__float_as_uint(rd.y) >> 31) << 4; rot.z = (__float_as_uint(rd.z) >> 31) << 4;
                    }
                }
            }
            const uint32_t take = 64u - (uint32_t)__popcll(busy);
            next = (n - next < take) ? n : next + take;
        }
        const uint64_t active = __ballot(ridx >= 0);
        if (active == 0ull) {
            if (next >= n) break;
            continue;   // only slots outside the frame were handed out: refill again
        }
        // ---- walk: lanes visit inner nodes until a child that is a leaf is hit (they park) or no node is left; the
        //      loop goes on while at least three quarters of the lanes that hold a ray are still walking
        const int walk_min = (__popcll(active) * RT_WF_WALK_EIGHTHS) >> 3;
        bool w = ridx >= 0 && cur >= 0 && pend0 == 0;
        const int walk_floor = walk_min > 1 ? walk_min : 1;
        const bool walked = __popcll(__ballot(w)) >= walk_floor;
        if (walked) do {
            WS_ADD(WS_WALK, __ballot(w));
            if (w) {
                const uint4 n0 = *reinterpret_cast<const uint4 *>(nodes + (uint32_t)cur);
                const uint4 n1 = *reinterpret_cast<const uint4 *>(nodes + (uint32_t)cur + 16u);
                // per axis one dword = lower | upper << 16; rotating it by 0 or 16 bits (the sign of the direction) puts the
                // plane the ray meets first into the low half, so no per-axis min / max is needed
                const uint32_t lx = __builtin_amdgcn_alignbit(n0.x, n0.x, rot.x), ly = __builtin_amdgcn_alignbit(n0.y, n0.y, rot.y), lz = __builtin_amdgcn_alignbit(n0.z, n0.z, rot.z);
                const uint32_t rx = __builtin_amdgcn_alignbit(n0.w, n0.w, rot.x), ry = __builtin_amdgcn_alignbit(n1.x, n1.x, rot.y), rz = __builtin_amdgcn_alignbit(n1.y, n1.y, rot.z);
                typedef _Float16 wf_h2 __attribute__((ext_vector_type(2)));
                const wf_h2 hlx = __builtin_bit_cast(wf_h2, lx), hly = __builtin_bit_cast(wf_h2, ly), hlz = __builtin_bit_cast(wf_h2, lz);
                const wf_h2 hrx = __builtin_bit_cast(wf_h2, rx), hry = __builtin_bit_cast(wf_h2, ry), hrz = __builtin_bit_cast(wf_h2, rz);
                // one v_fma_mix_f32 per plane (binary16 coordinate, binary32 ka / kb);
                // a NaN (0 * inf, inf - inf for rays parallel to an axis) is dropped by max3 / min3: such a slab does not cull
                const float ltmin = rt_max3(__builtin_fmaf((float)hlx.x, ka.x, kn.x), __builtin_fmaf((float)hly.x, ka.y, kn.y), rt_max(__builtin_fmaf((float)hlz.x, ka.z, kn.z), 0.0f));
                const float ltmax = rt_min3(__builtin_fmaf((float)hlx.y, ka.x, kf.x), __builtin_fmaf((float)hly.y, ka.y, kf.y), __builtin_fmaf((float)hlz.y, ka.z, kf.z));
                const float rtmin = rt_max3(__builtin_fmaf((float)hrx.x, ka.x, kn.x), __builtin_fmaf((float)hry.x, ka.y, kn.y), rt_max(__builtin_fmaf((float)hrz.x, ka.z, kn.z), 0.0f));
                const float rtmax = rt_min3(__builtin_fmaf((float)hrx.y, ka.x, kf.x), __builtin_fmaf((float)hry.y, ka.y, kf.y), __builtin_fmaf((float)hrz.y, ka.z, kf.z));
                const bool lhit = ltmin <= ltmax && ltmin <= closest, rhit = rtmin <= rtmax && rtmin <= closest;
                const int lref = (int)n1.z, rref = (int)n1.w;
                // leaves whose box is hit wait for their candidate test; inner children are visited near one first
                const bool lleaf = lhit && lref < 0, rleaf = rhit && rref < 0;
                pend0 = lleaf ? lref : (rleaf ? rref : 0);
                pend1 = (lleaf && rleaf) ? rref : 0;
                const bool lin = lhit && lref >= 0, rin = rhit && rref >= 0;
                const bool rfirst = rin && (!lin || rtmin < ltmin);   // the second child is the one to visit next
                cur = rfirst ? rref : (lin ? lref : -1);
                if (lin && rin) {            // both: the farther one waits on the stack (sp counts bytes of this lane's column)
                    *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(stack) + sp) = (uint32_t)(rfirst ? lref : rref);
                    sp += 256u;
                }
                if (!lin && !rin && sp > 0u) {   // neither: back to the last one that waits
                    sp -= 256u;
                    cur = (int)*reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(stack) + sp);
                }
                w = cur >= 0 && pend0 == 0;
            }
        } while (__popcll(__ballot(w)) >= walk_floor);
        // ---- candidate tests of the parked leaves' objects (the reference's intersect() up to `distance`)
        // (one test per lane and round: a lane with a second leaf stays parked and joins the next round, which then has
        // more lanes, instead of running a round of its own)
        if (__ballot(pend0 != 0) != 0ull) {
            WS_ADD(WS_LEAF, __ballot(pend0 != 0));
            if (pend0 != 0) {
                const uint32_t leaf = ~(uint32_t)pend0;
                const int idx = (int)(leaf & 0x0FFFFFFFu);
                float t = 0.0f;
                int sub = 0;
                if (cand_any<false>((int)(leaf >> 28), S.objects[idx], ro, rd, t, sub, nocnt) && t > 0.0f && (t < closest || (t == closest && idx < best))) {
                    closest = t; best = idx; best_sub = sub;
                }
                pend0 = pend1; pend1 = 0;
            }
        }
        // ---- rays with no node left are done
        WS_ADD(WS_FINISH, __ballot(ridx >= 0 && cur < 0 && pend0 == 0));
        if (ridx >= 0 && cur < 0 && pend0 == 0) {
            if (CHAIN) ph[ridx] = make_float2(closest, __uint_as_float(ch_pack_hit(best, best_sub)));
            else *reinterpret_cast<float2 *>(&pd[ridx].z) = make_float2(closest, __uint_as_float(wf_pack_hit(best, best_sub)));
            ridx = -1;
        }
    }
}

// One persistent wavefront per workgroup; tiles are handed out through an atomic counter, bottom rows first.
__global__ void __launch_bounds__(64, RT_WF_WAVES) rt_wf_diffuse_kernel(const DevScene S, const DevCamera C, const DevFrame F, const WfArgs W) {
    __shared__ uint32_t s_stack[kWfStack * 64u];
    const uint32_t lane = threadIdx.x;
    float4 *pa = W.a + (size_t)blockIdx.x * kWfPool, *pb = W.b + (size_t)blockIdx.x * kWfPool, *pc = W.c + (size_t)blockIdx.x * kWfPool;
    float4 *pd = W.d + (size_t)blockIdx.x * kWfPool;
    const uint32_t maxd = S.max_ray_bounces;
    const f3 ambient = ld3(S.ambient);
    const bool jitter = S.enable_accumulation != 0u;
    uint32_t rays = 0u;
    Cnt<false> nocnt;
    const uint32_t row_offset = W.ctrl[kWfCtrlRowOffset];
    const uint32_t items0 = W.ctrl[kWfCtrlItems0], items1 = W.ctrl[kWfCtrlItems1], items2 = W.ctrl[kWfCtrlItems2], items3 = W.ctrl[kWfCtrlItems3];
    const uint32_t tile2 = W.ctrl[kWfCtrlTile2], tile3 = W.ctrl[kWfCtrlTile3];
#ifdef RT_WF_STATS
    const unsigned long long ws_t0 = wall_clock64();
    __shared__ uint32_t s_stat[2 * WS_N];
    if (lane < 2u * WS_N) s_stat[lane] = 0u;
    __syncthreads();
#endif
    for (;;) {
        uint32_t item = 0u;
        if (lane == 0u) item = atomicAdd(&W.ctrl[kWfCtrlNext], 1u);
        item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item);
        if (item >= items3) break;
        // item -> (tile, part of the tile, size class): wave-uniform
        uint32_t cls, tile, part;
        if (item < items0) { cls = 0u; tile = item; part = 0u; }
        else if (item < items1) { const uint32_t q = item - items0; cls = 1u; tile = items0 + (q >> 1); part = q & 1u; }
        else if (item < items2) { const uint32_t q = item - items1; cls = 2u; tile = tile2 + (q >> 2); part = q & 3u; }
        else { const uint32_t q = item - items2; cls = 3u; tile = tile3 + (q >> 3); part = q & 7u; }
        const uint32_t n_slots = kWfPool >> cls;   // paths of this item: n_slots / 64 blocks of 8x8 pixels side by side
        const uint32_t trow = tile / W.tiles_x, tx = tile - trow * W.tiles_x;
                // tile rows top-down, starting at the row the order kernel chose from the previous launch's costs and wrapping around
        uint32_t rrow = trow + row_offset;
        if (rrow >= W.tiles_y) rrow -= W.tiles_y;
        const uint32_t x0 = tx * kWfTileW + part * (kWfTileW >> cls), y0 = F.row_begin + rrow * F.tile_row_step;
        const uint32_t rays_before = rays;
        // ---- camera rays against the linear objects, all lanes busy -- or their closest hits as an earlier frame found them
        const bool known = W.first_hit_mode == kFirstHitRead;
        for (uint32_t base = 0u; base < n_slots; base += 64u) {
            const uint32_t j = base + lane;
            uint32_t pos;
            float closest = RT_F32_MAX;
            uint32_t word = kWfHitInvalid;   // slot outside the frame: never traced, never shaded
            WS_ADD(WS_PREPARE, ~0ull);
            if (wf_slot_pixel(C, F, x0, y0, j, pos)) {
                if (known) {
                    const float2 h = W.first_hit[pos];
                    closest = h.x; word = __float_as_uint(h.y);
                } else {
                    int best = -1, best_sub = 0;
                    wf_linear_objects(S, ld3(C.position), ld3(C.ray_dirs + 3u * (size_t)pos), closest, best, best_sub);
                    word = wf_pack_hit(best, best_sub);
                }
            }
            pd[j] = make_float4(1.0f, 1.0f, closest, __uint_as_float(word));
        }
        __syncthreads();
        uint32_t n = n_slots;
        for (uint32_t depth = 0u; depth < maxd && n > 0u; depth++) {
            const bool primary = depth == 0u;
            if (!(primary && known)) wf_trace_pool(S, C, F, primary, n, x0, y0, pa, pb, pd, s_stack, rays WS_PASS);
            __syncthreads();
            // ---- shade: color_diffuse at every hit / miss of this bounce (ray_tracing.rs:87-114, :155-157)
            uint32_t out = 0u;
            for (uint32_t base = 0u; base < n; base += 64u) {
                const uint32_t j = base + lane;
                bool alive = j < n;
                float4 D = make_float4(1.0f, 1.0f, 0.0f, __uint_as_float(kWfHitInvalid));
                if (alive) D = pd[j];
                const uint32_t hobj = __float_as_uint(D.w);
                alive = alive && hobj != kWfHitInvalid;
                WS_ADD(WS_SHADE, __ballot(alive));
                f3 ro = splat3(0.0f), rd = splat3(0.0f), L = splat3(0.0f), Cc = splat3(1.0f);
                uint32_t key = 0u, pos = 0u;
                if (alive) {
                    if (primary) {
                        (void)wf_slot_pixel(C, F, x0, y0, j, pos);
                        ro = ld3(C.position); rd = ld3(C.ray_dirs + 3u * (size_t)pos);
                        key = rt_rng_path_key(F.seed, pos, F.frame_index);
                        if (W.first_hit_mode == kFirstHitWrite) W.first_hit[pos] = make_float2(D.z, D.w);
                    } else {
                        const float4 A = pa[j], B = pb[j], Cv = pc[j];
                        ro = mk3(A.x, A.y, A.z); key = __float_as_uint(A.w);
                        rd = mk3(B.x, B.y, B.z); pos = __float_as_uint(B.w);
                        L = mk3(Cv.x, Cv.y, Cv.z); Cc = mk3(Cv.w, D.x, D.y);
                    }
                }
                bool goes_on = false;
                f3 no = splat3(0.0f), nd = splat3(0.0f), p = splat3(0.0f), cc = splat3(0.0f);
                float nclosest = RT_F32_MAX;
                int nbest = -1, nbest_sub = 0;
                if (alive) {
                    if (hobj == kWfHitMiss) {
                        store_pixel(F, pos, L + ambient * Cc);
                    } else {
                        const int obj = (int)(hobj & 0x0FFFFFFFu), sub = (int)(hobj >> 30);
                        HitRec h; Mtl m;
                        h.u = 0.0f; h.v = 0.0f;
                        const DevObject &o = S.objects[obj];
                        finalize_one<false>(o, S.materials[o.material], ro, rd, D.z, sub, h, m, nocnt);
                        f3 albedo = m.albedo;
                        if (m.texture >= 0) albedo = tex_from_uv(S.textures[m.texture], h.u, h.v);
                        p = L + m.emission_power * albedo;
                        rt_reflection_ray(rd, h, m.roughness, key, true, jitter, no, nd);
                        cc = Cc * albedo;
                        if (depth + 1u >= maxd) store_pixel(F, pos, p);   // the child at the depth limit returns its light_color (:83-85)
                        else goes_on = true;
                    }
                }
                // the next ray against the linear objects (wave-uniform loop: kept outside the divergent code above)
                if (S.n_bvh2_linear != 0u && __ballot(goes_on) != 0ull) {
                    if (goes_on) wf_linear_objects(S, no, nd, nclosest, nbest, nbest_sub);
                }
                // every lane has read its slot: the compacted writes below never reach a slot that is still unread
                const uint64_t on = __ballot(goes_on);
                if (goes_on) {
                    const uint32_t k = out + __builtin_amdgcn_mbcnt_hi((uint32_t)(on >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)on, 0u));
                    pa[k] = make_float4(no.x, no.y, no.z, __uint_as_float(rt_rng_child_key(key, 0u)));
                    pb[k] = make_float4(nd.x, nd.y, nd.z, __uint_as_float(pos));
                    pc[k] = make_float4(p.x, p.y, p.z, cc.x);
                    pd[k] = make_float4(cc.y, cc.z, nclosest, __uint_as_float(wf_pack_hit(nbest, nbest_sub)));
                }
                out += (uint32_t)__popcll(on);
            }
            __syncthreads();   // pool writes of all lanes before the next bounce reads them
            n = out;
        }
        {   // cost of this tile row for the next launch's order
            uint32_t tr = rays - rays_before;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) tr += __shfl_xor(tr, off, 64);
            if (lane == 0u) atomicAdd(&W.ctrl[kWfCtrlRows + rrow], tr);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_xor(rays, off, 64);
    if (lane == 0u) F.counters[blockIdx.x] = rays;
#ifdef RT_WF_STATS
    __syncthreads();
    if (lane < 2u * WS_N && W.stats) atomicAdd(&W.stats[lane], (unsigned long long)s_stat[lane]);
    if (lane == 0u) {   // when this wavefront started and ended (100 MHz counter), left in its first pool slot for the host
        const unsigned long long t1 = wall_clock64();
        pa[0] = make_float4(__uint_as_float((uint32_t)ws_t0), __uint_as_float((uint32_t)(ws_t0 >> 32)), __uint_as_float((uint32_t)t1), __uint_as_float((uint32_t)(t1 >> 32)));
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// The same path tracer with per-pixel sample chains: ONE launch renders W.samples accumulating frames
// (frame_index = F.frame_index ... F.frame_index + samples - 1), the pixels of that many launches of the kernel above.
//
// A wavefront's pool entries no longer belong to one tile and one bounce.  Every entry is a PIXEL SLOT that always holds
// the current path of one pixel: when the path ends (miss, or depth limit) its value is added to the slot's copy of the
// pixel's accumulator entry and the slot continues with the SAME pixel's next sample; when the pixel has had all its
// samples, its accumulator entry and RGBA8 value are written -- once per launch instead of once per sample -- and the slot
// takes the next pixel of the launch (tiles of 64x8 pixels from one counter, pixel by pixel).  A pixel's samples are
// therefore traced one after the other and added in frame_index order, exactly the order of `self.accumulated[pos] += p`
// over consecutive frames (src/renderer.rs:59-66), and the random numbers are keyed by (seed, pixel, frame_index, bounce),
// so every path does the arithmetic it does in the kernels above: accumulator and frame are bit-identical to `samples`
// launches of them.
// What it buys: every slot holds a ray in every round, from the first round to the end of the launch, whatever the lengths
// of the paths (no drain per bounce and per tile, every trace pass runs over a full pool of rays at mixed depths, nothing is
// compacted or moved); no launch tail and no order kernel per sample; 36 bytes of frame traffic per pixel and LAUNCH instead
// of per sample; and a camera ray is not stored at all -- it is its pixel -- so only bounce rays pay the pool's bytes.
// One camera ray per PIXEL: Camera.ray_directions[pos] is the same in every frame (src/renderer.rs:49; nothing jitters it), so
// RayTracing::trace_ray of a pixel's camera ray returns the same hit for every sample.  It is traced for the pixel's first
// sample only; what the later samples need from it -- the hit point offset along the normal (origin of every first bounce ray,
// src/ray.rs:132), the normal, light_color and contribution after the hit (src/ray_tracing/ray_tracing.rs:101-109) -- is kept
// in the slot, and a later sample starts with its own random direction from there (src/ray.rs:123-130).  A pixel whose
// camera ray ends its path by itself (a miss, or a depth limit of 1) has the same value in every sample: the remaining
// samples are added on the spot.  The launch reports such camera rays apart from the ones it traced (F.counters, second half).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kChTileW = 64u, kChTilePixels = kChTileW * 8u;   // work item: 64x8 pixels = eight 8x8 blocks side by side
// Pixel slots per wavefront.  No slot is ever idle here, so the pool can be smaller than the other kernel's, and it has to
// be: every slot is rewritten and re-read each round, and the pools of all resident wavefronts together should stay inside
// the 256 MB last-level cache.
// Measured at 8K / 64 spp with 8192 resident wavefronts and 132 bytes per slot: 128 slots 203 ms, 192 slots 194 ms, 256 slots
// 222 ms, 384 slots 251 ms; 6144 wavefronts of 256 slots (the bytes of 8192 x 192) 212 ms.
#ifndef RT_CH_POOL
#define RT_CH_POOL 192
#endif
constexpr uint32_t kChPool = RT_CH_POOL;
static_assert(kChPool % 64u == 0u && kChPool >= 64u, "the shade pass walks the slots in batches of one per lane");
constexpr uint32_t kChMaxSamples = 1u << 24;   // samples per launch: the slot's sample counter shares a word with the depth

struct WfCursor { uint32_t x0, y0, i, n, exhausted; };   // wave-uniform: the item this wavefront is handing out, pixel by pixel

// Gives the next pixel of the launch to every lane that asks for one (`want`); false: the launch has no pixels left.
RT_DEV bool wf_claim_pixel(const DevCamera &C, const DevFrame &F, const WfArgs &W, WfCursor &cur, bool want, uint32_t &pos) {
    bool got = false;
    const uint32_t n_items = W.tiles_x * W.tiles_y;
    for (;;) {
        const uint64_t m = __ballot(want && !got);
        if (m == 0ull) break;
        if (cur.i >= cur.n) {
            if (cur.exhausted != 0u) break;
            uint32_t item = 0u;
            if (threadIdx.x == 0u) item = atomicAdd(&W.ctrl[kWfCtrlNext], 1u);
            item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item);
            if (item >= n_items) { cur.exhausted = 1u; break; }
            // tile rows from the bottom of the frame up, like the Whitted kernels: the sky -- pixels whose camera ray ends the
            // path, all samples settled at once -- comes last and fills the end of the launch (8K, 64 samples: 192.3 -> 189.9 ms;
            // 8 samples 30.0 -> 29.5 ms)
            const uint32_t trow_up = item / W.tiles_x, tx = item - trow_up * W.tiles_x;
            const uint32_t trow = W.tiles_y - 1u - trow_up;
            cur.x0 = tx * kChTileW; cur.y0 = F.row_begin + trow * F.tile_row_step; cur.i = 0u; cur.n = kChTilePixels;
        }
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        const uint32_t avail = cur.n - cur.i, asked = (uint32_t)__popcll(m);
        // a pixel of the item that lies outside the frame is nobody's: the lane asks again
        if (want && !got && rank < avail) got = wf_slot_pixel(C, F, cur.x0, cur.y0, cur.i + rank, pos);
        cur.i += asked < avail ? asked : avail;
    }
    return got;
}

__global__ void __launch_bounds__(64, RT_WF_WAVES) rt_wf_chain_kernel(const DevScene S, const DevCamera C, const DevFrame F, const WfArgs W) {
    __shared__ uint32_t s_stack[kWfStack * 64u];
    const uint32_t lane = threadIdx.x;
    const size_t slice = (size_t)blockIdx.x * kChPool;
    float4 *pa = W.a + slice, *pb = W.b + slice, *pc = W.c + slice, *pe = W.e + slice;
    float *pd1 = W.d1 + slice;
    float2 *ph = W.h + slice;
    uint2 *pg = W.g + slice;
    float4 *pk0 = W.k0 + slice, *pk1 = W.k1 + slice, *pk2 = W.k2 + slice;
    uint32_t reused = 0u;   // camera rays of later samples, settled from the pixel's first
    const uint32_t maxd = S.max_ray_bounces, n_samples = W.samples;
    const f3 ambient = ld3(S.ambient);
    const bool jitter = S.enable_accumulation != 0u;
    uint32_t rays = 0u;
    Cnt<false> nocnt;
#ifdef RT_WF_STATS
    __shared__ uint32_t s_stat[2 * WS_N];
    if (lane < 2u * WS_N) s_stat[lane] = 0u;
    __syncthreads();
#endif
    WfCursor cur = {0u, 0u, 0u, 0u, 0u};
    bool boot = true;   // first pass: every pixel slot is empty and takes its first pixel
    for (;;) {
        if (!boot) {
            wf_trace_pool<true>(S, C, F, false, kChPool, 0u, 0u, pa, pb, nullptr, s_stack, rays WS_PASS, ph, pg);
            __syncthreads();
        }
        // ---- shade: color_diffuse at every hit / miss of this round (ray_tracing.rs:87-114, :155-157), each path at its own depth
        uint32_t live = 0u;
        for (uint32_t base = 0u; base < kChPool; base += 64u) {
            const uint32_t j = base + lane;
            float2 H = make_float2(0.0f, __uint_as_float(kWfHitInvalid));
            uint2 G = make_uint2(0u, 0u);
            if (!boot) { H = ph[j]; G = pg[j]; }
            const uint32_t hw = __float_as_uint(H.y);
            const bool alive = hw != kWfHitInvalid;
            WS_ADD(WS_SHADE, __ballot(alive));
            uint32_t pos = G.x, sample = G.y & (kChMaxSamples - 1u), depth = G.y >> 24, key = 0u;
            bool goes_on = false, ended = false;
            f3 no = splat3(0.0f), nd = splat3(0.0f), p = splat3(0.0f), cc = splat3(1.0f), V = splat3(0.0f);
            if (alive) {
                f3 ro, rd, L = splat3(0.0f), Cc = splat3(1.0f);
                if (depth == 0u) {   // camera ray (src/renderer.rs:49-57; RayTracing::albedo seeds light 0, contribution 1)
                    ro = ld3(C.position); rd = ld3(C.ray_dirs + 3u * (size_t)pos);
                    key = rt_rng_path_key(F.seed, pos, F.frame_index + sample);
                } else {
                    const float4 A = pa[j], B = pb[j], Cv = pc[j];
                    ro = mk3(A.x, A.y, A.z); rd = mk3(B.x, B.y, B.z); L = mk3(Cv.x, Cv.y, Cv.z); Cc = mk3(B.w, Cv.w, pd1[j]);
                    key = __float_as_uint(A.w);
                }
                if ((hw & kChObjMask) == kChObjMask) {
                    V = L + ambient * Cc;
                    ended = true;
                } else {
                    const int obj = (int)(hw & kChObjMask), sub = (int)((hw >> 30) & 1u);
                    HitRec h; Mtl m;
                    h.u = 0.0f; h.v = 0.0f;
                    const DevObject &o = S.objects[obj];
                    finalize_one<false>(o, S.materials[o.material], ro, rd, H.x, sub, h, m, nocnt);
                    f3 albedo = m.albedo;
                    if (m.texture >= 0) albedo = tex_from_uv(S.textures[m.texture], h.u, h.v);
                    p = L + m.emission_power * albedo;
                    if (depth + 1u >= maxd) {   // the child at the depth limit returns its light_color (:83-85)
                        V = p;
                        ended = true;
                    } else {
                        rt_reflection_ray(rd, h, m.roughness, key, true, jitter, no, nd);
                        cc = Cc * albedo;
                        if (depth == 0u && n_samples > 1u) {   // the camera ray's hit: shared by every sample of this pixel
                            pk0[j] = make_float4(no.x, no.y, no.z, p.x);
                            pk1[j] = make_float4(h.normal.x, h.normal.y, h.normal.z, p.y);
                            pk2[j] = make_float4(cc.x, cc.y, cc.z, p.z);
                        }
                        key = rt_rng_child_key(key, 0u);
                        depth += 1u;
                        goes_on = true;
                    }
                }
            }
            // ---- a path that ended: its value joins the pixel's accumulator entry; the pixel goes on with its next sample or is written
            bool again = false, need_pixel = boot;
            if (ended) {
                sample += 1u;
                if (F.accumulate) {
                    float4 E = pe[j];
                    E.x = E.x + V.x; E.y = E.y + V.y; E.z = E.z + V.z; E.w = E.w + 1.0f;   // src/renderer.rs:60
                    if (depth == 0u) {   // the camera ray ended the path by itself: so it does in every other frame, with the same value
                        reused += n_samples - sample;
                        for (; sample < n_samples; sample++) { E.x = E.x + V.x; E.y = E.y + V.y; E.z = E.z + V.z; E.w = E.w + 1.0f; }
                    }
                    if (sample < n_samples) pe[j] = E;
                    else store_summed(F, pos, E, F.frame_index + n_samples - 1u);
                } else if (sample >= n_samples) {
                    store_pixel(F, pos, V);
                }
                if (sample < n_samples) again = true;
                else need_pixel = true;
            }
            if (again) {   // the pixel's next frame: its camera ray's hit is known, the path starts with its own first bounce ray
                const float4 K0 = pk0[j], K1 = pk1[j], K2 = pk2[j];
                HitRec h;
                h.normal = mk3(K1.x, K1.y, K1.z);
                key = rt_rng_path_key(F.seed, pos, F.frame_index + sample);
                const f3 rnd = mk3(rt_rng_uniform(key, 0u, -1.0f, 1.0f), rt_rng_uniform(key, 1u, -1.0f, 1.0f), rt_rng_uniform(key, 2u, -1.0f, 1.0f));
                nd = normalize(h.normal + rnd);   // Ray::reflection_ray, diffuse branch (src/ray.rs:123-130)
                no = mk3(K0.x, K0.y, K0.z);
                p = mk3(K0.w, K1.w, K2.w);
                cc = mk3(K2.x, K2.y, K2.z);
                key = rt_rng_child_key(key, 0u);
                depth = 1u;
                goes_on = true;
                reused += 1u;
            }
            bool respawn = false;
            // ---- empty pixel slots take the launch's next pixels
            if (__ballot(need_pixel) != 0ull) {
                uint32_t npos = 0u;
                const bool got = wf_claim_pixel(C, F, W, cur, need_pixel, npos);
                if (need_pixel && got) {
                    pos = npos; sample = 0u;
                    pe[j] = load_accum(F, pos);
                    respawn = true;
                }
            }
            if (respawn) {   // the camera ray of this pixel's next frame: only its pixel and its first-hit candidate are stored
                no = ld3(C.position); nd = ld3(C.ray_dirs + 3u * (size_t)pos);
                depth = 0u;
            }
            // the next ray against the linear objects (wave-uniform loop: kept outside the divergent code above)
            float nclosest = RT_F32_MAX;
            int nbest = -1, nbest_sub = 0;
            if (S.n_bvh2_linear != 0u && __ballot(goes_on || respawn) != 0ull) {
                if (goes_on || respawn) wf_linear_objects(S, no, nd, nclosest, nbest, nbest_sub);
            }
            if (goes_on) {
                pa[j] = make_float4(no.x, no.y, no.z, __uint_as_float(key));
                pb[j] = make_float4(nd.x, nd.y, nd.z, cc.x);
                pc[j] = make_float4(p.x, p.y, p.z, cc.y);
                pd1[j] = cc.z;
            }
            if (goes_on || respawn) {
                ph[j] = make_float2(nclosest, __uint_as_float(ch_pack_hit(nbest, nbest_sub) | (respawn ? kChPrimaryBit : 0u)));
                pg[j] = make_uint2(pos, sample | (depth << 24));
            } else if (alive || boot) {
                ph[j] = make_float2(0.0f, __uint_as_float(kWfHitInvalid));   // no pixel left for this slot
            }
            live += (uint32_t)__popcll(__ballot(goes_on || respawn));
        }
        __syncthreads();   // slot writes of all lanes before the next round reads them
        boot = false;
        if (live == 0u) break;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { rays += __shfl_xor(rays, off, 64); reused += __shfl_xor(reused, off, 64); }
    // first half of the table: trace_ray invocations of the reference (traced here or not); second half: the ones not traced
    if (lane == 0u) { F.counters[blockIdx.x] = rays + reused; F.counters[gridDim.x + blockIdx.x] = reused; }
#ifdef RT_WF_STATS
    __syncthreads();
    if (lane < 2u * WS_N && W.stats) atomicAdd(&W.stats[lane], (unsigned long long)s_stat[lane]);
#endif
}

// ------------------------------------------------------------------------------------------------
// Small diffuse scenes (linear scan): lanes that take the next pixel when their path ends.
//
// Same function as above -- color_diffuse's forward loop -- for scenes whose closest hits come from the wave-uniform scan
// over all objects.  There the scan itself keeps every lane busy whatever ray it holds, so paths need not be stored and
// re-grouped between bounces: a wavefront owns a strip of kRfTiles 8x8 tiles, every lane follows one path at a time in
// registers, and a lane whose path has ended (miss, or depth limit) writes its pixel and takes the strip's next pixel.
// One loop iteration is one trace_ray + shading step for all lanes, each at its own depth.  With one pixel per lane for the
// whole launch, a wavefront ran as long as its longest path (up to max_ray_bounces) while most paths end after one or two rays.
// Every path does the arithmetic it does in rt_trace_kernel, in the same order: frames are bit-identical.
// ------------------------------------------------------------------------------------------------
#ifndef RT_RF_TILES
#define RT_RF_TILES 4
#endif
#ifndef RT_RF_MIN_IDLE
#define RT_RF_MIN_IDLE 16    // idle lanes that trigger a refill
#endif
constexpr uint32_t kRfTiles = RT_RF_TILES;

template <bool SP>
__global__ void __launch_bounds__(64, 8) rt_refill_diffuse_kernel(const DevScene S, const DevCamera C, const DevFrame F) {
    // value of every finished path, by slot; the strip's pixels are written in one pass at the end, whole tiles at a time
    // (a store per finished path would put an accumulator read-modify-write round trip into every iteration of the loop)
    __shared__ float s_val[3u * kRfTiles * 64u];
    const uint32_t lane = threadIdx.x;
    const uint32_t ty = gridDim.y - 1u - blockIdx.y;   // tile rows from the bottom of the band up, as lane_pixel
    const uint32_t x0 = blockIdx.x * kRfTiles * 8u, y0 = F.row_begin + ty * F.tile_row_step;
    const uint32_t n = kRfTiles * 64u;
    const uint32_t maxd = S.max_ray_bounces;
    const f3 ambient = ld3(S.ambient);
    const bool jitter = S.enable_accumulation != 0u;
    Cnt<false> cnt;
    cnt.clear();
    uint32_t next = 0u;          // wave-uniform: first slot of the strip not handed out yet
    bool have = false;           // this lane holds a path
    bool first = true;           // wave-uniform: every ray of this iteration is a camera ray
    uint32_t slot = 0u, key = 0u, depth = 0u;
    f3 ro = splat3(0.0f), rd = splat3(0.0f), L = splat3(0.0f), Cc = splat3(1.0f);
    for (;;) {
        const uint64_t idle = __ballot(!have);
        if (next < n && ((uint32_t)__popcll(idle) >= (uint32_t)RT_RF_MIN_IDLE)) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            if (!have) {
                const uint32_t j = next + rank;
                uint32_t pos;
                if (j < n && wf_slot_pixel(C, F, x0, y0, j, pos)) {
                    have = true;
                    slot = j;
                    ro = ld3(C.position); rd = ld3(C.ray_dirs + 3u * (size_t)pos);
                    L = splat3(0.0f); Cc = splat3(1.0f);
                    key = rt_rng_path_key(F.seed, pos, F.frame_index);
                    depth = 0u;
                }
            }
            const uint32_t take = (uint32_t)__popcll(idle);
            next = (n - next < take) ? n : next + take;
        }
        if (__ballot(have) == 0ull) {
            if (next >= n) break;
            continue;   // only slots outside the frame were handed out
        }
        if (have) {
            float t; int sub;
            const int obj = trace_closest<false, SP>(S, ro, rd, t, sub, cnt, first ? RT_CNT_RAYS_PRIMARY : RT_CNT_RAYS_REFLECTION);
            f3 V = L + ambient * Cc;   // the value of a miss
            bool ended = true;
            if (obj >= 0) {
                HitRec h; Mtl m;
                finalize_hit<false, false, SP>(S, ro, rd, t, obj, sub, h, m, cnt);
                f3 albedo = m.albedo;
                if (m.texture >= 0) albedo = tex_from_uv(S.textures[m.texture], h.u, h.v);
                const f3 p = L + m.emission_power * albedo;
                V = p;                      // the child at the depth limit returns its light_color (ray_tracing.rs:83-85)
                if (depth + 1u < maxd) {
                    f3 no, nd;
                    rt_reflection_ray(rd, h, m.roughness, key, true, jitter, no, nd);
                    ro = no; rd = nd; L = p; Cc = Cc * albedo;
                    key = rt_rng_child_key(key, 0u);
                    depth += 1u;
                    ended = false;
                }
            }
            if (ended) {
                s_val[slot] = V.x; s_val[n + slot] = V.y; s_val[2u * n + slot] = V.z;
                have = false;
            }
        }
        first = false;
    }
    __syncthreads();
    float4 acc[kRfTiles];
#pragma unroll
    for (uint32_t k = 0u; k < kRfTiles; k++) {   // every accumulator load of the strip is in flight before the first store
        uint32_t pos;
        acc[k] = wf_slot_pixel(C, F, x0, y0, k * 64u + lane, pos) ? load_accum(F, pos) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
#pragma unroll
    for (uint32_t k = 0u; k < kRfTiles; k++) {
        const uint32_t j = k * 64u + lane;
        uint32_t pos;
        if (wf_slot_pixel(C, F, x0, y0, j, pos)) store_pixel(F, pos, mk3(s_val[j], s_val[n + j], s_val[2u * n + j]), acc[k]);
    }
    uint32_t rays = cnt.rays;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_xor(rays, off, 64);
    if (lane == 0u) F.counters[rt_block_index()] = rays;
}

// Chooses the tile-row order of the next rt_wf_diffuse_kernel launch and resets its work counter (one workgroup).
// Measured on the 10 000-sphere frame: which rows run first and last matters by 6 % -- starting every wavefront on the most
// expensive rows at once is the slowest order, ending on them leaves the chip half empty while the last tiles finish, and
// scattering rows loses the cache locality of neighbouring tiles.  So rows stay in top-down order, rotated so that the
// cheapest three eighths of the frame (by the rays the previous launch traced per row; a contiguous block, wrapping) come
// last.  Without history (first launch, or a new camera / row range) the order is plain top-down.  Pixels do not depend on it.
//
// The same pass sizes the items (see kWfCtrl*).  An item finishes before the launch does if its work is at most the work
// still to be handed out divided by the number of wavefronts; going down the launch order, the item size of a row is the
// largest one for which that holds (by the previous launch's rays per row; uniform rows without history), and it never grows
// again.  How small items may get depends on how many whole tiles there are per wavefront (see max_cls below).
// All of it runs in parallel over the rows (the kernel stands in front of every launch: done by one thread it took 54 us
// of a 1.4 ms sample at 4K): prefix sums by chunks and a scan of the chunk sums, the cheapest window by an atomic minimum,
// and -- because the size class along the launch order only ever grows -- the first row of every class as the minimum over
// the rows whose own comparison asks for at least that class.
constexpr uint32_t kWfMaxOrderRows = 4096u;   // taller frames run top-down with uniform rows
constexpr uint32_t kWfOrderThreads = 256u;
#ifndef RT_WF_GUIDE
#define RT_WF_GUIDE 1.0f     // safety factor on an item's work in that comparison
#endif
// exclusive prefix of v over the workgroup's threads; s_wave holds one word per wavefront
__device__ __forceinline__ uint32_t wf_order_scan(uint32_t v, uint32_t *s_wave) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t x = v;
#pragma unroll
    for (uint32_t off = 1u; off < 64u; off <<= 1) { const uint32_t y = __shfl_up(x, off, 64); if (lane >= off) x += y; }
    if (lane == 63u) s_wave[threadIdx.x >> 6] = x;
    __syncthreads();
    uint32_t base = 0u;
    for (uint32_t w = 0u; w < (threadIdx.x >> 6); w++) base += s_wave[w];
    __syncthreads();
    return base + x - v;
}
__global__ void __launch_bounds__(kWfOrderThreads) rt_wf_order_kernel(uint32_t *ctrl, uint32_t tiles_x, uint32_t tiles_y, uint32_t groups) {
    __shared__ uint32_t s_pre[kWfMaxOrderRows + 1u];    // s_pre[r]: rays of the rows before r
    __shared__ uint32_t s_work[kWfMaxOrderRows + 1u];   // the same over max(rays, 1): the work a row stands for
    __shared__ uint32_t s_wave[kWfOrderThreads / 64u];
    __shared__ unsigned long long s_best;
    __shared__ uint32_t s_offset, s_first[4];
    const uint32_t T = tiles_y, L = (T * 3u) / 8u;
    const bool history = T <= kWfMaxOrderRows;
    if (threadIdx.x == 0u) { s_offset = 0u; s_best = ~0ull; s_first[0] = 0u; s_first[1] = T; s_first[2] = T; s_first[3] = T; }
    uint32_t total = 0u;
    if (history) {
        for (uint32_t r = threadIdx.x; r < T; r += kWfOrderThreads) s_pre[r + 1u] = ctrl[kWfCtrlRows + r];
        __syncthreads();
        const uint32_t chunk = (T + kWfOrderThreads - 1u) / kWfOrderThreads;
        const uint32_t r0 = min(threadIdx.x * chunk, T), r1 = min(r0 + chunk, T);
        uint32_t a = 0u, b = 0u;
        for (uint32_t r = r0; r < r1; r++) { const uint32_t w = s_pre[r + 1u]; a += w; b += w > 0u ? w : 1u; }
        a = wf_order_scan(a, s_wave);
        b = wf_order_scan(b, s_wave);
        for (uint32_t r = r0; r < r1; r++) {
            const uint32_t w = s_pre[r + 1u];
            a += w; b += w > 0u ? w : 1u;
            s_pre[r + 1u] = a; s_work[r + 1u] = b;
        }
        if (threadIdx.x == 0u) { s_pre[0] = 0u; s_work[0] = 0u; }
        __syncthreads();
        total = s_pre[T];
        if (total != 0u && L > 0u) {
            for (uint32_t s0 = threadIdx.x; s0 < T; s0 += kWfOrderThreads) {   // window [s0, s0 + L), wrapping
                const uint32_t e = s0 + L;
                const uint32_t sum = e <= T ? s_pre[e] - s_pre[s0] : (total - s_pre[s0]) + s_pre[e - T];
                atomicMin(&s_best, ((unsigned long long)sum << 32) | s0);
            }
            __syncthreads();
            if (threadIdx.x == 0u) s_offset = ((uint32_t)(s_best & 0xFFFFFFFFull) + L) % T;
        }
    }
    __syncthreads();
    const uint32_t offset = s_offset;
    const bool costs = history && total != 0u;
    // smallest item size allowed: halves below four whole tiles per wavefront, quarters below one, eighths below a quarter
    // (measured on the 10 000-sphere frame: at 4K, 2.3 tiles per wavefront, halves gain 10 % and quarters 5 %; at 8K, 9 tiles
    // per wavefront, every smaller size loses 1-3 %: short pools drain at low occupancy)
    const uint32_t n_tiles = tiles_x * T;
    const uint32_t max_cls = n_tiles >= 4u * groups ? 0u : (n_tiles >= groups ? 1u : (4u * n_tiles >= groups ? 2u : 3u));
    for (uint32_t k = threadIdx.x; k < T && max_cls > 0u; k += kWfOrderThreads) {   // k: position in the launch order
        uint32_t w = 1u, remaining = T - k;   // work of row k and of the rows from k on
        if (costs) {
            const uint32_t r = k + offset < T ? k + offset : k + offset - T;
            w = s_work[r + 1u] - s_work[r];
            remaining = k + offset < T ? (s_work[T] - s_work[r]) + s_work[offset] : s_work[offset] - s_work[r];
        }
        const float item = RT_WF_GUIDE * (float)groups * (float)w / (float)tiles_x;   // groups x (work of one whole tile of this row)
        uint32_t cls = 0u;
        while (cls < max_cls && item > (float)remaining * (float)(1u << cls)) cls++;
        for (uint32_t c = 1u; c <= cls; c++) atomicMin(&s_first[c], k);
    }
    __syncthreads();
    if (threadIdx.x == 0u) {
        const uint32_t r1 = s_first[1], r2 = s_first[2], r3 = s_first[3];   // first launch-order row of each size class
        const uint32_t i0 = r1 * tiles_x, i1 = i0 + (r2 - r1) * tiles_x * 2u, i2 = i1 + (r3 - r2) * tiles_x * 4u, i3 = i2 + (T - r3) * tiles_x * 8u;
        ctrl[kWfCtrlNext] = 0u; ctrl[kWfCtrlRowOffset] = offset;
        ctrl[kWfCtrlItems0] = i0; ctrl[kWfCtrlItems1] = i1; ctrl[kWfCtrlItems2] = i2; ctrl[kWfCtrlItems3] = i3;
        ctrl[kWfCtrlTile2] = r2 * tiles_x; ctrl[kWfCtrlTile3] = r3 * tiles_x;
    }
    for (uint32_t r = threadIdx.x; r < T; r += blockDim.x) ctrl[kWfCtrlRows + r] = 0u;
}
