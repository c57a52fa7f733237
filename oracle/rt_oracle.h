/*
 * rt_oracle.h — C API of the CPU parity oracle (TEST INFRASTRUCTURE).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The shipped path (librt_mi355x.so and the host mirror) never links, loads or calls it.
 *
 * PARITY UNPINNED: the reference (Rust, no toolchain here, no tests or golden vectors of its own)
 * cannot be run, so this restatement is not checked against reference outputs; see orc_math.h.
 *
 * The oracle consumes the same plain-data scene/camera description as the backend
 * (include/rt_mi355x.h) so that both sides of a parity test see identical inputs.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include "../include/rt_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_texture {
    uint32_t width;
    uint32_t height;
    const uint8_t *rgb8; /* Texture.bytes, src/objects.rs:105-111 */
} orc_texture;

/* One frame of Renderer::render (src/renderer.rs:81-151) over rows [row_begin,row_end) (0,0 = all):
 *   accum   in/out, width*height float4 (Renderer.accumulated)
 *   rgba    out, width*height*4 bytes
 *   linear  out or NULL, width*height float4: the raw Scene::pixel value of this frame
 *   counters out or NULL, RT_CNT_COUNT slots, added to
 *   n_threads worker threads over the reference's 200-chunk decomposition (src/app.rs:80)
 * Returns 0, or -1 if the description is invalid (what the reference would panic on). */
int32_t orc_render(const rt_scene_desc *scene, const orc_texture *textures, const rt_camera_desc *camera,
                   uint32_t frame_index, int32_t accumulate, uint64_t seed, uint32_t row_begin, uint32_t row_end,
                   float *accum, uint8_t *rgba, float *linear, uint64_t *counters, int32_t n_threads);

/* The same frame with Camera.ray_directions supplied (the reference computes that table once per camera change,
 * src/camera.rs:128-157, and render_chunk only reads it, src/renderer.rs:49); ray_dirs = width*height*3 floats. */
int32_t orc_render_with_directions(const rt_scene_desc *scene, const orc_texture *textures, const rt_camera_desc *camera,
                                   const float *ray_dirs, uint32_t frame_index, int32_t accumulate, uint64_t seed,
                                   uint32_t row_begin, uint32_t row_end, float *accum, uint8_t *rgba, float *linear,
                                   uint64_t *counters, int32_t n_threads);

/* Renderer::render's chunk loop exactly as written (src/renderer.rs:104-143), whole frames only: with width*height not a
 * multiple of num_chunks the short last chunk takes the directions and accumulator entries of the pixels at
 * `chunk index * its own length` (src/renderer.rs:115-117) and writes them to the end of the frame.  orc_render gives every
 * pixel its own direction instead, as the GPU backend does.  Returns -2 where the reference panics (fewer than num_chunks
 * pixels: chunks_mut(0)), -1 for an invalid description. */
int32_t orc_render_reference_chunks(const rt_scene_desc *scene, const orc_texture *textures, const rt_camera_desc *camera,
                                    uint32_t frame_index, int32_t accumulate, uint64_t seed, uint32_t num_chunks, float *accum,
                                    uint8_t *rgba);

/* Camera::calculate_ray_directions (src/camera.rs:128-157): out = width*height*3 floats, row-major. */
void orc_ray_directions(const rt_camera_desc *camera, float *out);

/* Camera::calculate_ray_directions for one pixel (src/camera.rs:137-148). */
void orc_camera_ray(const rt_camera_desc *camera, uint32_t x, uint32_t y, float dir[3]);

/* Camera::update's matrices (src/camera.rs:88-93,117-123) from Camera's public fields. */
void orc_camera_matrices(const float position[3], const float forward[3], const float up[3], float fov, float near_,
                         float far_, uint32_t width, uint32_t height, float inverse_perspective[16],
                         float inverse_view[16]);

/* transform / inv_transform as the object's update() builds them
 * (sphere.rs:61-70, cuboid.rs:41-51, cylinder.rs:45-54, cone.rs:44-53); identity for the other kinds. */
void orc_object_transform(const rt_object *obj, float transform[16], float inv_transform[16]);

/* Intersection::intersect of one object (src/ray.rs:167-177).  Returns 1 for Some(hit) and fills
 * out[9] = distance, point.xyz, normal.xyz, u, v; returns 0 for None. */
int32_t orc_intersect(const rt_object *obj, const float origin[3], const float dir[3], float out[9]);

/* RayMarching::sdfs (src/ray_marching/ray_marching.rs:24-107): out[11] = distance, albedo.xyz,
 * transformed_ray.origin.xyz, transformed_ray.direction.xyz, obj_idx (as float). */
int32_t orc_sdfs(const rt_scene_desc *scene, const float origin[3], const float dir[3], float t, float out[11]);

/* Texture::from_uv (src/objects.rs:130-137). */
void orc_texture_from_uv(const orc_texture *tex, float u, float v, float rgb[3]);

/* The deterministic libm pair and the host libm, for the agreement tests. */
float orc_powf_det(float x, float y);
float orc_atanf_det(float x);
float orc_powf_host(float x, float y);
float orc_atanf_host(float x);
/* The deterministic pair over arrays: out[i] = powf(x[i], y[i]) / atanf(x[i]). */
void orc_powf_det_array(const float *x, const float *y, float *out, size_t n);
void orc_atanf_det_array(const float *x, float *out, size_t n);

/* The counter-based generator standing in for ThreadRng: uniform f32 in [low, high) drawn as
 * rand 0.8.5 maps a u32 (Cargo.lock:671-683): (bits >> 9) * 2^-23 * (high - low) + low. */
uint32_t orc_rng_path_key(uint64_t seed, uint32_t pixel_index, uint32_t frame_index);
uint32_t orc_rng_child_key(uint32_t key, uint32_t branch);
float orc_rng_uniform(uint32_t key, uint32_t draw, float low, float high);

/* 1 if built with -DORC_HOST_LIBM. */
int32_t orc_uses_host_libm(void);

#ifdef __cplusplus
}
#endif
#endif
