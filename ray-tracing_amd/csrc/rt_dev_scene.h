// rt_dev_scene.h — flat device-side records of Scene / Camera (reference: src/scene.rs:13-25,
// src/objects.rs:21-30,151-160, src/light.rs:29-33, src/camera.rs:6-20), shared by the host upload code
// and the kernels.
//
// Layout rule: everything a whole wavefront reads at the same index (objects in the closest-hit loop,
// lights, the sdf list, camera matrices) is an array of small records read through scalar loads into
// SGPRs; only per-lane data (texels, BVH nodes, frame buffers) goes through vector memory.
#pragma once
#include <stdint.h>

// (M * vec4(v, w)).xyz needs the x,y,z rows only: four columns of three floats, plus the column-3-times-zero
// term glam still adds for directions (w = 0), pre-multiplied on the host so the kernel adds it verbatim.
struct DevXform {
    float c0[3], c1[3], c2[3], c3[3];
    float z[3];  // c3 * 0.0f
};

struct DevObject {
    int32_t kind;      // rt_object_kind
    int32_t material;  // material index (0 for Union / Substraction)
    int32_t first, second;
    // Bounding sphere flag of the linear scan's wave-level rejection test (bs below); 0 for planes and composites: never rejected.
    int32_t bounded;
    // Sphere whose transform and inv_transform have a diagonal 3x3 part with (+-)0 everywhere else (no rotation:
    // Sphere::create, src/utils/sphere.rs:26-49): cand_sphere / fin_sphere then multiply by the diagonal only.
    int32_t diag;
    int32_t pad_[2];
    DevXform inv;  // inv_transform
    DevXform fwd;  // transform
    // kind-specific parameters:
    //   Sphere   p[0] radius
    //   Cuboid   p[0..2] dimension
    //   Plane    p[0..2] normal, p[3..5] point, p[6..7] max_dist, p[8] has_max_dist (0/1)
    //   Cylinder p[0] radius, p[1] height
    //   Triangle p[0..2] v1, p[3..5] v2, p[6..8] v3
    float p[12];
    // Bounding sphere of everything the object's intersect() can report (centre, squared radius with a safety margin), for
    // the linear scan's wave-level rejection test (used when `bounded`).
    float bs[4];
};

// Kind code of a hierarchy leaf that holds a sphere with DevObject::diag set.  Leaves name their object's kind next to its
// index; composites are never hit by rays (Ray::hit's `_ => None`, src/ray.rs:175) and never are leaves, so RT_UNION's code is
// free there.
#define RT_LEAF_DIAG_SPHERE 6

struct DevMaterial {
    float ambience, diffuse, specular, shininess;
    float albedo[3];
    int32_t texture;  // -1: none
    int32_t kind;     // rt_material_kind
    float roughness, transparency, refraction_index, reflectivity, emission_power;
    float pad[2];
};

struct DevLight {
    int32_t kind;  // rt_light_kind
    float albedo[3];
    float vec[3];
    float intensity;
};

// One texel per dword (R | G<<8 | B<<16), repacked from the caller's RGB8 once at upload.
struct DevTexture {
    const uint32_t *texels;
    uint32_t width, height;
};

// Bounding-volume hierarchy for scenes with many objects (not part of the reference, which scans all objects
// linearly; it only selects which objects get the reference's candidate test).  Nodes are stored in depth-first
// order; `skip` is the node to continue with when the ray misses the box, so traversal needs no stack.
struct DevBvhNode {
    float lo[3];
    int32_t skip;  // index of the first node after this subtree
    float hi[3];
    int32_t leaf;  // -1: inner node (first child = this + 1); else (object kind << 28) | object index
};

// Two-child node of the quantised hierarchy (rt_bvh.h, rt_bvh2_build): q[0..5] = the first child's box as (lower, upper)
// pairs for x, y, z (one dword per axis), q[6..11] the second child's.  Each q is a binary16 bit pattern: the coordinate
// is DevScene.bvh2_origin + q * bvh2_scale (origin = centre of the root box, +-32000 units at its faces), rounded outward,
// so that one mixed-precision FMA per plane decodes and intersects it.  ref >= 0: byte offset of an inner node in the node
// array, ref < 0: ~((object kind << 28) | object index).
struct DevBvh2Node {
    uint16_t q[12];
    int32_t ref[2];
};

struct DevScene {
    const DevObject *objects;
    const DevMaterial *materials;
    const DevLight *lights;
    const uint32_t *sdfs;
    const DevTexture *textures;
    const DevBvhNode *bvh_nodes;   // null: linear scan
    const uint32_t *unbounded;     // objects outside the hierarchy (planes), tested linearly
    uint32_t n_bvh_nodes, n_unbounded;
    const DevBvh2Node *bvh2_nodes; // null: the re-grouping kernel is not used for this scene
    const uint32_t *bvh2_linear;   // objects outside that tree (planes, outliers), tested linearly
    uint32_t n_bvh2_linear;
    float bvh2_origin[3], bvh2_scale[3];
    uint32_t n_objects, n_materials, n_lights, n_sdfs;
    float ambient[3];
    uint32_t max_ray_bounces;
    uint32_t shadow_casting, ray_marching, diffuse, enable_accumulation;
};

struct DevCamera {
    float ip_c0[4], ip_c1[4], ip_c2[4], ip_c3[4];  // inverse_perspective columns
    DevXform iv;                                   // inverse_view
    float position[3];
    uint32_t width, height;
    float inv_w, inv_h;  // 1 / width as f32, 1 / height as f32 (src/camera.rs:132-133)
    const float *ray_dirs;  // Camera.ray_directions (src/camera.rs:16): width*height*3 floats, filled by rt_raygen_kernel at rt_set_camera
};

struct DevFrame {
    uint32_t frame_index;
    uint32_t row_begin, row_end;
    uint32_t tile_row_step;  // rows between the starts of consecutive tile rows of this launch: 8, or 8 N for an interleaved share
    uint32_t leaf_slots;  // march kernel: object-space rays cached in LDS (0: none)
    // march kernel, entries of Scene.sdfs that a march step may leave out (see rm_sdfs_march); march_skip = 0: none are.
    // march_lip[k]: an upper bound of how fast entry k's distance function can fall along a ray, per unit of the march
    // parameter; march_slot[k]: index of the entry's first cached object-space ray.
    uint32_t march_skip;
    float march_lip[8];
    uint32_t march_slot[8];
    // entries that are a Union or a Substraction of two leaves (bit k of march_pair): the second child has a bound of its own
    // (march_lip2, cached ray march_slot2) and is left out while it cannot influence the pair's value
    uint32_t march_pair;
    float march_lip2[8];
    uint32_t march_slot2[8];
    int32_t accumulate;
    int32_t skip_accum_store;  // accumulate == 0 only: do not write the frame's linear value to the accumulator
    uint64_t seed;
    uint32_t *rgba;      // width*height packed R | G<<8 | B<<16 | A<<24
    float *accum;        // width*height float4
    uint32_t *counters;  // [gridDim.x][n_counters]
    unsigned long long *diag;  // diagnostic builds only (RT_DIAG): per-section wave-cycle sums
    // rt_trace_kernel, or null: one word per workgroup of this launch, byte k = 1 when no camera ray of wavefront k's tile can
    // reach any object (rt_tile_sky_kernel); such tiles get the value of a miss without reading their directions
    const uint32_t *tile_sky;
};
