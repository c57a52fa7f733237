// rt_host.h — host-side mirror of the reference's public API for the rendering hot path, written in
// C++ because the reference is compiled code and no Rust toolchain exists in this build.
//
// Names, argument order, defaults and error behaviour follow the Rust types so that code (and tests)
// written against the reference read the same here:
//   Scene / Material / Texture / Object3D constructors   src/scene.rs, src/objects.rs, src/utils/*.rs
//   Light                                               src/light.rs
//   Camera                                              src/camera.rs
//   Renderer::render                                    src/renderer.rs:81-151
// The per-pixel work itself is not here: Renderer::render hands it to librt_mi355x.so through the
// C ABI of include/rt_mi355x.h.  "new" is a C++ keyword, so the Rust `X::new(..)` is `X::create(..)`.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rt_mi355x.h"

namespace rt {

struct Vec2 { float x = 0, y = 0; };
struct Vec3 { float x = 0, y = 0, z = 0; };
struct Vec4 { float x = 0, y = 0, z = 0, w = 0; };
inline Vec2 vec2(float x, float y) { return Vec2{x, y}; }
inline Vec3 vec3(float x, float y, float z) { return Vec3{x, y, z}; }

// glam::Mat4 layout: m[4*col + row]
struct Mat4 {
    float m[16];
    static Mat4 identity();
    static Mat4 from_translation(Vec3 t);
    static Mat4 from_scale(Vec3 s);
    static Mat4 from_rotation_x(float a);
    static Mat4 from_rotation_y(float a);
    static Mat4 from_rotation_z(float a);
    static Mat4 perspective_rh(float fov_y, float aspect, float z_near, float z_far);
    static Mat4 look_at_lh(Vec3 eye, Vec3 center, Vec3 up);
    Mat4 inverse() const;
    Mat4 operator*(const Mat4 &r) const;
    Vec4 operator*(Vec4 v) const;
};

extern const float DEGREES;  // src/utils/geometry.rs:5

// src/objects.rs:86-95
struct MaterialType {
    rt_material_kind kind = RT_REFLECTIVE;
    float roughness = 1.0f;
    float transparency = 0.0f, refraction_index = 0.0f, reflectivity = 0.0f;
    static MaterialType Reflective(float roughness);
    static MaterialType Refractive(float transparency, float refraction_index, float reflectivity);
};

// src/objects.rs:151-175 (same defaults as `impl Default for Material`)
struct Material {
    float ambience = 0.2f, diffuse = 0.7f, specular = 0.5f, shininess = 5.0f;
    Vec3 albedo{};
    int texture = -1;  // Option<usize>::None
    MaterialType kind = MaterialType::Reflective(1.0f);
    float emission_power = 0.0f;
};

// src/objects.rs:104-111; bytes are tightly packed RGB8 as ImageUtils::load_image leaves them
// (src/utils/image.rs:13-29).
struct Texture {
    std::string path;
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> bytes;
};

// ImageUtils::load_image (src/utils/image.rs:13-29) for the formats whose decoding is exact: PNG (8-bit grey, RGB,
// palette, with or without alpha; 1/2/4-bit grey and palette; not interlaced) and binary PPM.  Alpha is dropped and grey
// replicated, as `into_rgb8()` does.  JPEG files are refused (their pixels depend on the decoder).  Returns "" or a message.
std::string load_image(const std::string &path, Texture &out);

// src/light.rs:3-33
struct Light {
    rt_light_kind kind = RT_POSITIONAL;
    Vec3 albedo{}, vec{};
    float radius = 0.0f, intensity = 0.0f;
    static Light Directional(Vec3 albedo, Vec3 direction, float intensity);
    static Light Positional(Vec3 albedo, Vec3 position, float intensity);
    static Light SphericalPositional(Vec3 albedo, Vec3 position, float radius, float intensity);
};

// src/objects.rs:21-30: one value type for the eight variants; `o` is the flat record the C ABI takes.
struct Object3D {
    rt_object o{};
    // recomputes transform / inv_transform from position, rotation_axis, radius, height — the
    // variant's update() (sphere.rs:61, cuboid.rs:41, cylinder.rs:45, cone.rs:44); no-op otherwise
    Object3D &update();
};
struct Sphere {    // src/utils/sphere.rs:26-59
    static Object3D create(Vec3 origin, float radius, size_t material_index);
    static Object3D new_sphere_with_rotation(Vec3 origin, Vec3 rotation_axis, float radius, size_t material_index);
};
struct Plane {     // src/utils/plane.rs:17-29; max_dist == nullptr is Option::None
    static Object3D create(Vec3 normal, Vec3 point, const Vec2 *max_dist, size_t material_index);
};
struct Cuboid {    // src/utils/cuboid.rs:23-39
    static Object3D create(Vec3 position, Vec3 rotation_axis, Vec3 dimension, size_t material_index);
};
struct Cylinder {  // src/utils/cylinder.rs:25-43 (height before rotation before radius)
    static Object3D create(Vec3 position, float height, Vec3 rotation_axis, float radius, size_t material_index);
};
struct Cone {      // src/utils/cone.rs:25-43 (radius, height, rotation)
    static Object3D create(Vec3 position, float radius, float height, Vec3 rotation_axis, size_t material_index);
};
struct Triangle {  // src/utils/triangle.rs:20-27
    static Object3D create(Vec3 v1, Vec3 v2, Vec3 v3, size_t material_index);
};
struct Union {     // src/utils/union.rs:18-20
    static Object3D create(size_t first, size_t second);
};
struct Substraction {  // src/utils/substraction.rs:18-20
    static Object3D create(size_t first, size_t second);
};

// src/scene.rs:13-78
struct Scene {
    std::vector<Light> lights;
    Vec3 ambient_color{};
    std::vector<Object3D> objects;
    std::vector<size_t> sdfs;
    std::vector<Material> materials;
    std::vector<Texture> textures;
    uint8_t max_ray_bounces = 0;
    bool shadow_casting = false, ray_marching = false, diffuse = false, enable_accumulation = false;
    bool (*update_func)(Scene &, float) = nullptr;

    static Scene create(std::vector<Object3D> objects, std::vector<Material> materials);  // max_ray_bounces = 4
    Scene with_light(const Light &l) const;
    Scene with_texture(const Texture &t) const;
    Scene with_textures(const std::vector<Texture> &t) const;
};

// Plain-data view of a Scene for rt_set_scene; owns the arrays the desc points into.
struct FlatScene {
    std::vector<rt_object> objects;
    std::vector<rt_material> materials;
    std::vector<rt_light> lights;
    std::vector<uint32_t> sdfs;
    rt_scene_desc desc{};
};
void flatten(const Scene &scene, FlatScene &out);

// src/camera.rs:54-61
struct CameraEvent {
    enum Kind { Resize, RotateXY, Up, Down, Left, Right } kind;
    size_t w = 0, h = 0;
    Vec2 delta{};
};

// src/camera.rs:6-126.  ray_directions (the per-pixel table) does not exist here: rt_set_camera builds it on the
// device from inverse_perspective / inverse_view / position.
struct Camera {
    size_t width = 800, height = 600;
    float fov = 45.0f, near_ = -1.0f, far_ = -100.0f;
    Vec3 position{0.0f, 0.0f, 3.0f}, forward_direction{0.0f, 0.0f, -1.0f}, up{0.0f, 1.0f, 0.0f};
    Mat4 view = Mat4::identity(), inverse_view = Mat4::identity();
    Mat4 perspective = Mat4::identity(), inverse_perspective = Mat4::identity();

    static Camera create();
    static Camera new_with_pos(Vec3 position, Vec3 look_at);
    void update(const std::vector<CameraEvent> &events, float ts);
    rt_camera_desc desc() const;
};

// src/renderer.rs:13-151.  `accumulated` lives on the GPU; read_accumulated() syncs it on demand.
struct Renderer {
    bool enable_accumulation = false;
    uint32_t max_frames_rendering = 1000;
    uint32_t frame_index = 1;
    uint64_t seed = 0x5EED;  // keys the reproducible generator that stands in for ThreadRng
    int device_id = 0;

    Renderer() = default;
    ~Renderer();
    Renderer(const Renderer &) = delete;
    Renderer &operator=(const Renderer &) = delete;

    // Same contract as the reference: fills img (w*h*4, RGBA) or leaves it untouched on the
    // early-out (src/renderer.rs:98-102); returns "" for Ok(()) and the message for Err(String).
    // The SDL texture argument is dropped: uploading img to the display stays in the caller.
    // num_chunks is accepted and ignored (the GPU dispatch replaces the rayon chunking).
    std::string render(Scene &scene, std::vector<uint8_t> &img, const Camera &camera, bool updated, size_t num_chunks);
    std::string read_accumulated(std::vector<float> &out_vec4);
    rt_ctx *context() { return ctx_; }

  private:
    rt_ctx *ctx_ = nullptr;
    size_t uploaded_textures_ = 0;
    bool scene_uploaded_ = false;
    FlatScene last_;           // what the device holds: an `updated` frame re-uploads only the objects that moved
    uint64_t full_uploads_ = 0, object_updates_ = 0;

  public:
    uint64_t full_uploads() const { return full_uploads_; }      // rt_set_scene calls so far
    uint64_t object_updates() const { return object_updates_; }  // rt_update_objects calls so far
};

// The reference's demo scenes as data (src/bin/scene1.rs .. scene4.rs, construction-time state) and the
// two synthetic benchmark scenes.  Texture slots are declared (n_textures) but carry no bytes: the caller
// attaches them with Scene::with_texture / rt_set_texture (load_image above, or its own decoder for JPEG).
namespace scenes {
struct Built { Scene scene; Camera camera; bool renderer_accumulation = false; };
Built scene1();
Built scene2();
Built scene3();
Built scene4();
Built tex4();
Built spheres10k(uint32_t n_spheres = 10000, uint64_t seed = 0x5EED);
bool by_name(const std::string &name, Built &out);
}  // namespace scenes

}  // namespace rt

// C view of the same mirror for non-C++ drivers (ctypes).
extern "C" {
typedef struct rth_scene rth_scene;
rth_scene *rth_scene_build(const char *name);
void rth_scene_free(rth_scene *s);
const rt_scene_desc *rth_scene_desc(rth_scene *s);
// camera of the scene's binary, resized to (width, height) through Camera::update
void rth_scene_camera(rth_scene *s, uint32_t width, uint32_t height, rt_camera_desc *out);
int32_t rth_scene_renderer_accumulation(rth_scene *s);
// Camera::new_with_pos + Resize
void rth_camera(const float position[3], const float look_at[3], uint32_t width, uint32_t height, rt_camera_desc *out);
// rebuilds one object's matrices the way its update() does (for cross-checks against the oracle)
void rth_object_update(rt_object *o);
// drives Renderer::render for n_frames frames on `device_id`; textures are attached first.
// Returns 0 or -1 (message in err, up to err_len bytes).
int32_t rth_render_frames(rth_scene *s, uint32_t width, uint32_t height, uint32_t n_frames, int32_t accumulate,
                          uint32_t n_textures, const uint32_t *tex_w, const uint32_t *tex_h, const uint8_t *const *tex_rgb8,
                          int32_t device_id, uint8_t *img_out, float *accum_out, char *err, size_t err_len);
// App3D::run's frame loop without the window (src/app.rs:219-248): per frame the scene's update function (ts = frame * dt),
// then Renderer::render.  img_out receives n_frames frames back to back, objects_out (may be NULL) the flattened objects
// after each update (n_frames * n_objects); the two counters report how the scene reached the device.
int32_t rth_render_animation(rth_scene *s, uint32_t width, uint32_t height, uint32_t n_frames, float dt, uint32_t n_textures,
                             const uint32_t *tex_w, const uint32_t *tex_h, const uint8_t *const *tex_rgb8, int32_t device_id,
                             uint8_t *img_out, rt_object *objects_out, uint64_t *full_uploads, uint64_t *object_updates,
                             char *err, size_t err_len);
// load_image: *rgb8 receives width*height*3 bytes allocated by the library (release with rth_free_image)
int32_t rth_load_image(const char *path, uint32_t *width, uint32_t *height, uint8_t **rgb8, char *err, size_t err_len);
void rth_free_image(uint8_t *rgb8);
}
