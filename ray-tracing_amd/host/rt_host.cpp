// rt_host.cpp — implementation of the host-side mirror declared in rt_host.h.
#include "rt_host.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace rt {

const float DEGREES = 3.14159274101257324f / 180.0f;

// ------------------------------------------------------------------------------------------------
// Mat4: glam 0.28 formulas, column-major storage
// ------------------------------------------------------------------------------------------------
static inline Vec3 sub(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline Vec3 add(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline Vec3 neg(Vec3 a) { return {-a.x, -a.y, -a.z}; }
static inline Vec3 scale(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline float dot3(Vec3 a, Vec3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
static inline Vec3 cross3(Vec3 a, Vec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
static inline Vec3 normalize3(Vec3 a) { return scale(a, 1.0f / std::sqrt(dot3(a, a))); }

Mat4 Mat4::identity() {
    Mat4 r; std::memset(r.m, 0, sizeof(r.m)); r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r;
}
Mat4 Mat4::from_translation(Vec3 t) { Mat4 r = identity(); r.m[12] = t.x; r.m[13] = t.y; r.m[14] = t.z; return r; }
Mat4 Mat4::from_scale(Vec3 s) { Mat4 r = identity(); r.m[0] = s.x; r.m[5] = s.y; r.m[10] = s.z; return r; }
Mat4 Mat4::from_rotation_x(float a) {
    float s = std::sin(a), c = std::cos(a);
    Mat4 r = identity(); r.m[5] = c; r.m[6] = s; r.m[9] = -s; r.m[10] = c; return r;
}
Mat4 Mat4::from_rotation_y(float a) {
    float s = std::sin(a), c = std::cos(a);
    Mat4 r = identity(); r.m[0] = c; r.m[2] = -s; r.m[8] = s; r.m[10] = c; return r;
}
Mat4 Mat4::from_rotation_z(float a) {
    float s = std::sin(a), c = std::cos(a);
    Mat4 r = identity(); r.m[0] = c; r.m[1] = s; r.m[4] = -s; r.m[5] = c; return r;
}
Mat4 Mat4::perspective_rh(float fov_y, float aspect, float z_near, float z_far) {
    float sf = std::sin(0.5f * fov_y), cf = std::cos(0.5f * fov_y);
    float h = cf / sf, w = h / aspect, r = z_far / (z_near - z_far);
    Mat4 m; std::memset(m.m, 0, sizeof(m.m));
    m.m[0] = w; m.m[5] = h; m.m[10] = r; m.m[11] = -1.0f; m.m[14] = r * z_near;
    return m;
}
Mat4 Mat4::look_at_lh(Vec3 eye, Vec3 center, Vec3 up) {
    // look_at_lh(eye, center, up) == look_to_rh(eye, -(center - eye), up)
    Vec3 f = normalize3(neg(sub(center, eye)));
    Vec3 s = normalize3(cross3(f, up));
    Vec3 u = cross3(s, f);
    Mat4 m;
    m.m[0] = s.x; m.m[1] = u.x; m.m[2] = -f.x; m.m[3] = 0.0f;
    m.m[4] = s.y; m.m[5] = u.y; m.m[6] = -f.y; m.m[7] = 0.0f;
    m.m[8] = s.z; m.m[9] = u.z; m.m[10] = -f.z; m.m[11] = 0.0f;
    m.m[12] = -dot3(eye, s); m.m[13] = -dot3(eye, u); m.m[14] = dot3(eye, f); m.m[15] = 1.0f;
    return m;
}
Vec4 Mat4::operator*(Vec4 v) const {
    float o[4];
    for (int r = 0; r < 4; r++) o[r] = ((m[r] * v.x + m[4 + r] * v.y) + m[8 + r] * v.z) + m[12 + r] * v.w;
    return Vec4{o[0], o[1], o[2], o[3]};
}
Mat4 Mat4::operator*(const Mat4 &b) const {
    Mat4 out;
    for (int j = 0; j < 4; j++) {
        Vec4 c = (*this) * Vec4{b.m[4 * j], b.m[4 * j + 1], b.m[4 * j + 2], b.m[4 * j + 3]};
        out.m[4 * j] = c.x; out.m[4 * j + 1] = c.y; out.m[4 * j + 2] = c.z; out.m[4 * j + 3] = c.w;
    }
    return out;
}
// cofactor inverse in glam's term order: 18 2x2 sub-determinants, 4 cofactor columns, 1/det
Mat4 Mat4::inverse() const {
    const float *a = m;
    auto det2 = [&](int c0, int r0, int c1, int r1, int c2, int r2, int c3, int r3) {
        return a[4 * c0 + r0] * a[4 * c1 + r1] - a[4 * c2 + r2] * a[4 * c3 + r3];
    };
    float coef00 = det2(2, 2, 3, 3, 3, 2, 2, 3), coef02 = det2(1, 2, 3, 3, 3, 2, 1, 3), coef03 = det2(1, 2, 2, 3, 2, 2, 1, 3);
    float coef04 = det2(2, 1, 3, 3, 3, 1, 2, 3), coef06 = det2(1, 1, 3, 3, 3, 1, 1, 3), coef07 = det2(1, 1, 2, 3, 2, 1, 1, 3);
    float coef08 = det2(2, 1, 3, 2, 3, 1, 2, 2), coef10 = det2(1, 1, 3, 2, 3, 1, 1, 2), coef11 = det2(1, 1, 2, 2, 2, 1, 1, 2);
    float coef12 = det2(2, 0, 3, 3, 3, 0, 2, 3), coef14 = det2(1, 0, 3, 3, 3, 0, 1, 3), coef15 = det2(1, 0, 2, 3, 2, 0, 1, 3);
    float coef16 = det2(2, 0, 3, 2, 3, 0, 2, 2), coef18 = det2(1, 0, 3, 2, 3, 0, 1, 2), coef19 = det2(1, 0, 2, 2, 2, 0, 1, 2);
    float coef20 = det2(2, 0, 3, 1, 3, 0, 2, 1), coef22 = det2(1, 0, 3, 1, 3, 0, 1, 1), coef23 = det2(1, 0, 2, 1, 2, 0, 1, 1);
    float fac0[4] = {coef00, coef00, coef02, coef03}, fac1[4] = {coef04, coef04, coef06, coef07};
    float fac2[4] = {coef08, coef08, coef10, coef11}, fac3[4] = {coef12, coef12, coef14, coef15};
    float fac4[4] = {coef16, coef16, coef18, coef19}, fac5[4] = {coef20, coef20, coef22, coef23};
    float vec0[4] = {a[4], a[0], a[0], a[0]}, vec1[4] = {a[5], a[1], a[1], a[1]};
    float vec2_[4] = {a[6], a[2], a[2], a[2]}, vec3_[4] = {a[7], a[3], a[3], a[3]};
    const float sign_a[4] = {1.0f, -1.0f, 1.0f, -1.0f}, sign_b[4] = {-1.0f, 1.0f, -1.0f, 1.0f};
    Mat4 inv;
    for (int i = 0; i < 4; i++) {
        inv.m[i] = ((vec1[i] * fac0[i] - vec2_[i] * fac1[i]) + vec3_[i] * fac2[i]) * sign_a[i];
        inv.m[4 + i] = ((vec0[i] * fac0[i] - vec2_[i] * fac3[i]) + vec3_[i] * fac4[i]) * sign_b[i];
        inv.m[8 + i] = ((vec0[i] * fac1[i] - vec1[i] * fac3[i]) + vec3_[i] * fac5[i]) * sign_a[i];
        inv.m[12 + i] = ((vec0[i] * fac2[i] - vec1[i] * fac4[i]) + vec2_[i] * fac5[i]) * sign_b[i];
    }
    float d0 = a[0] * inv.m[0], d1 = a[1] * inv.m[4], d2 = a[2] * inv.m[8], d3 = a[3] * inv.m[12];
    float det = d0 + d1 + d2 + d3;
    float rcp = 1.0f / det;
    for (int i = 0; i < 16; i++) inv.m[i] = inv.m[i] * rcp;
    return inv;
}

// ------------------------------------------------------------------------------------------------
// materials, lights, objects
// ------------------------------------------------------------------------------------------------
MaterialType MaterialType::Reflective(float roughness) { MaterialType m; m.kind = RT_REFLECTIVE; m.roughness = roughness; return m; }
MaterialType MaterialType::Refractive(float transparency, float refraction_index, float reflectivity) {
    MaterialType m; m.kind = RT_REFRACTIVE; m.roughness = 0.0f;
    m.transparency = transparency; m.refraction_index = refraction_index; m.reflectivity = reflectivity; return m;
}
Light Light::Directional(Vec3 albedo, Vec3 direction, float intensity) { Light l; l.kind = RT_DIRECTIONAL; l.albedo = albedo; l.vec = direction; l.intensity = intensity; return l; }
Light Light::Positional(Vec3 albedo, Vec3 position, float intensity) { Light l; l.kind = RT_POSITIONAL; l.albedo = albedo; l.vec = position; l.intensity = intensity; return l; }
Light Light::SphericalPositional(Vec3 albedo, Vec3 position, float radius, float intensity) {
    Light l; l.kind = RT_SPHERICAL_POSITIONAL; l.albedo = albedo; l.vec = position; l.radius = radius; l.intensity = intensity; return l;
}

static void st3(float *d, Vec3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
static Vec3 ld3(const float *d) { return {d[0], d[1], d[2]}; }
static rt_object blank(rt_object_kind kind, size_t material_index) {
    rt_object o; std::memset(&o, 0, sizeof(o));
    o.kind = kind; o.material_index = (int32_t)material_index;
    Mat4 id = Mat4::identity();
    std::memcpy(o.transform, id.m, sizeof(id.m)); std::memcpy(o.inv_transform, id.m, sizeof(id.m));
    return o;
}

Object3D &Object3D::update() {
    Vec3 pos = ld3(o.position), rot = ld3(o.rotation_axis);
    Mat4 t;
    switch (o.kind) {
        case RT_SPHERE:
            t = Mat4::from_translation(pos) * Mat4::from_rotation_x(rot.x * DEGREES) * Mat4::from_rotation_y(rot.y * DEGREES) *
                Mat4::from_rotation_z(rot.z * DEGREES) * Mat4::from_scale(vec3(o.radius, o.radius, o.radius));
            break;
        case RT_CUBOID:
        case RT_CYLINDER:  // the reference leaves the cylinder's scale out of its transform
            t = Mat4::from_translation(pos) * Mat4::from_rotation_x(rot.x * DEGREES) * Mat4::from_rotation_y(rot.y * DEGREES) *
                Mat4::from_rotation_z(rot.z * DEGREES);
            break;
        case RT_CONE:
            t = Mat4::from_translation(pos) * Mat4::from_rotation_x(rot.x * DEGREES) * Mat4::from_rotation_y(rot.y * DEGREES) *
                Mat4::from_rotation_z(rot.z * DEGREES) * Mat4::from_scale(vec3(o.radius, o.radius, o.height));
            break;
        default:
            return *this;
    }
    Mat4 inv = t.inverse();
    std::memcpy(o.transform, t.m, sizeof(t.m)); std::memcpy(o.inv_transform, inv.m, sizeof(inv.m));
    return *this;
}

Object3D Sphere::create(Vec3 origin, float radius, size_t material_index) {
    Object3D r; r.o = blank(RT_SPHERE, material_index); st3(r.o.position, origin); r.o.radius = radius; return r.update();
}
Object3D Sphere::new_sphere_with_rotation(Vec3 origin, Vec3 rotation_axis, float radius, size_t material_index) {
    Object3D r; r.o = blank(RT_SPHERE, material_index); st3(r.o.position, origin); st3(r.o.rotation_axis, rotation_axis);
    r.o.radius = radius; return r.update();
}
Object3D Plane::create(Vec3 normal, Vec3 point, const Vec2 *max_dist, size_t material_index) {
    Object3D r; r.o = blank(RT_PLANE, material_index); st3(r.o.normal, normal); st3(r.o.point, point);
    if (max_dist) { r.o.has_max_dist = 1; r.o.max_dist[0] = max_dist->x; r.o.max_dist[1] = max_dist->y; }
    return r;
}
Object3D Cuboid::create(Vec3 position, Vec3 rotation_axis, Vec3 dimension, size_t material_index) {
    Object3D r; r.o = blank(RT_CUBOID, material_index); st3(r.o.position, position); st3(r.o.rotation_axis, rotation_axis);
    st3(r.o.dimension, dimension); return r.update();
}
Object3D Cylinder::create(Vec3 position, float height, Vec3 rotation_axis, float radius, size_t material_index) {
    Object3D r; r.o = blank(RT_CYLINDER, material_index); st3(r.o.position, position); st3(r.o.rotation_axis, rotation_axis);
    r.o.height = height; r.o.radius = radius; return r.update();
}
Object3D Cone::create(Vec3 position, float radius, float height, Vec3 rotation_axis, size_t material_index) {
    Object3D r; r.o = blank(RT_CONE, material_index); st3(r.o.position, position); st3(r.o.rotation_axis, rotation_axis);
    r.o.height = height; r.o.radius = radius; return r.update();
}
Object3D Triangle::create(Vec3 v1, Vec3 v2, Vec3 v3, size_t material_index) {
    Object3D r; r.o = blank(RT_TRIANGLE, material_index); st3(r.o.v1, v1); st3(r.o.v2, v2); st3(r.o.v3, v3); return r;
}
Object3D Union::create(size_t first, size_t second) {
    Object3D r; r.o = blank(RT_UNION, 0); r.o.first = (int32_t)first; r.o.second = (int32_t)second; return r;
}
Object3D Substraction::create(size_t first, size_t second) {
    Object3D r; r.o = blank(RT_SUBSTRACTION, 0); r.o.first = (int32_t)first; r.o.second = (int32_t)second; return r;
}

// ------------------------------------------------------------------------------------------------
// Scene
// ------------------------------------------------------------------------------------------------
Scene Scene::create(std::vector<Object3D> objects, std::vector<Material> materials) {
    Scene s; s.objects = std::move(objects); s.materials = std::move(materials); s.max_ray_bounces = 4; return s;
}
Scene Scene::with_light(const Light &l) const { Scene s = *this; s.lights.push_back(l); return s; }
Scene Scene::with_texture(const Texture &t) const { Scene s = *this; s.textures.push_back(t); return s; }
Scene Scene::with_textures(const std::vector<Texture> &t) const { Scene s = *this; s.textures.insert(s.textures.end(), t.begin(), t.end()); return s; }

void flatten(const Scene &scene, FlatScene &out) {
    out.objects.clear(); out.materials.clear(); out.lights.clear(); out.sdfs.clear();
    for (const Object3D &o : scene.objects) out.objects.push_back(o.o);
    for (const Material &m : scene.materials) {
        rt_material f; std::memset(&f, 0, sizeof(f));
        f.ambience = m.ambience; f.diffuse = m.diffuse; f.specular = m.specular; f.shininess = m.shininess;
        st3(f.albedo, m.albedo); f.texture = m.texture; f.kind = m.kind.kind; f.roughness = m.kind.roughness;
        f.transparency = m.kind.transparency; f.refraction_index = m.kind.refraction_index; f.reflectivity = m.kind.reflectivity;
        f.emission_power = m.emission_power;
        out.materials.push_back(f);
    }
    for (const Light &l : scene.lights) {
        rt_light f; std::memset(&f, 0, sizeof(f));
        f.kind = l.kind; st3(f.albedo, l.albedo); st3(f.vec, l.vec); f.radius = l.radius; f.intensity = l.intensity;
        out.lights.push_back(f);
    }
    for (size_t i : scene.sdfs) out.sdfs.push_back((uint32_t)i);
    rt_scene_desc &d = out.desc; std::memset(&d, 0, sizeof(d));
    d.objects = out.objects.data(); d.materials = out.materials.data(); d.lights = out.lights.data(); d.sdfs = out.sdfs.data();
    d.n_objects = (uint32_t)out.objects.size(); d.n_materials = (uint32_t)out.materials.size();
    d.n_lights = (uint32_t)out.lights.size(); d.n_sdfs = (uint32_t)out.sdfs.size();
    d.n_textures = (uint32_t)scene.textures.size();
    st3(d.ambient_color, scene.ambient_color);
    d.max_ray_bounces = scene.max_ray_bounces; d.shadow_casting = scene.shadow_casting; d.ray_marching = scene.ray_marching;
    d.diffuse = scene.diffuse; d.enable_accumulation = scene.enable_accumulation;
}

// ------------------------------------------------------------------------------------------------
// Camera (src/camera.rs:63-126)
// ------------------------------------------------------------------------------------------------
Camera Camera::create() { return Camera{}; }
Camera Camera::new_with_pos(Vec3 position, Vec3 look_at) {
    Camera c; c.position = position; c.forward_direction = normalize3(look_at); return c;
}
void Camera::update(const std::vector<CameraEvent> &events, float ts) {
    Vec3 right_direction = cross3(forward_direction, up);
    const float speed = 7.0f, rotation_speed = 7.0f;
    for (const CameraEvent &e : events) {
        switch (e.kind) {
            case CameraEvent::Up: position = add(position, scale(scale(forward_direction, speed), ts)); break;
            case CameraEvent::Down: position = sub(position, scale(scale(forward_direction, speed), ts)); break;
            case CameraEvent::Left: position = sub(position, scale(scale(right_direction, speed), ts)); break;
            case CameraEvent::Right: position = add(position, scale(scale(right_direction, speed), ts)); break;
            case CameraEvent::Resize:
                width = e.w; height = e.h;
                perspective = Mat4::perspective_rh(fov, (float)e.w / (float)e.h, near_, far_);
                inverse_perspective = perspective.inverse();
                break;
            case CameraEvent::RotateXY: {
                float pitch_delta = e.delta.y * rotation_speed, yaw_delta = e.delta.x * rotation_speed;
                Mat4 rotation = Mat4::from_rotation_x(-pitch_delta * DEGREES) * Mat4::from_rotation_y(-yaw_delta * DEGREES);
                Vec4 fd = rotation * Vec4{forward_direction.x, forward_direction.y, forward_direction.z, 1.0f};
                forward_direction = vec3(fd.x, fd.y, fd.z);
                break;
            }
        }
    }
    view = Mat4::look_at_lh(position, add(position, forward_direction), up);
    inverse_view = view.inverse();
}
rt_camera_desc Camera::desc() const {
    rt_camera_desc d; std::memset(&d, 0, sizeof(d));
    d.width = (uint32_t)width; d.height = (uint32_t)height; st3(d.position, position);
    std::memcpy(d.inverse_perspective, inverse_perspective.m, sizeof(d.inverse_perspective));
    std::memcpy(d.inverse_view, inverse_view.m, sizeof(d.inverse_view));
    return d;
}

// ------------------------------------------------------------------------------------------------
// Renderer (src/renderer.rs:81-151)
// ------------------------------------------------------------------------------------------------
Renderer::~Renderer() { if (ctx_) rt_destroy(ctx_); }

// True when two flattened scenes differ at most in the contents of their object records.
static bool same_except_object_records(const FlatScene &a, const FlatScene &b) {
    auto bytes_equal = [](const auto &x, const auto &y) {
        return x.size() == y.size() && (x.empty() || std::memcmp(x.data(), y.data(), x.size() * sizeof(x[0])) == 0);
    };
    if (a.objects.size() != b.objects.size() || !bytes_equal(a.materials, b.materials) || !bytes_equal(a.lights, b.lights) ||
        !bytes_equal(a.sdfs, b.sdfs))
        return false;
    const rt_scene_desc &p = a.desc, &q = b.desc;
    return p.n_textures == q.n_textures && std::memcmp(p.ambient_color, q.ambient_color, sizeof(p.ambient_color)) == 0 &&
           p.max_ray_bounces == q.max_ray_bounces && p.shadow_casting == q.shadow_casting && p.ray_marching == q.ray_marching &&
           p.diffuse == q.diffuse && p.enable_accumulation == q.enable_accumulation;
}

std::string Renderer::render(Scene &scene, std::vector<uint8_t> &img, const Camera &camera, bool updated, size_t /*num_chunks*/) {
    auto fail = [&](const char *what) { return std::string(what) + ": " + (ctx_ ? rt_last_error(ctx_) : "no context"); };
    if (!ctx_) {
        if (rt_create(device_id, &ctx_) != RT_OK) return "rt_create failed (no usable MI355X device?)";
        updated = true;
    }
    bool reset = false;
    if (updated) {  // :93-96 — and the scene/camera may have changed since the previous frame
        reset = true;
        frame_index = 1;
        for (size_t i = uploaded_textures_; i < scene.textures.size(); i++) {
            const Texture &t = scene.textures[i];
            if (t.bytes.size() != (size_t)t.width * t.height * 3) return "texture " + std::to_string(i) + " has no RGB8 bytes";
            if (rt_set_texture(ctx_, (uint32_t)i, t.width, t.height, t.bytes.data()) != RT_OK) return fail("rt_set_texture");
            uploaded_textures_ = i + 1;
        }
        // The binaries' update functions (src/bin/scene1.rs:13-24, scene4.rs:14-26) move a few objects per frame:
        // when only object records changed (same kinds, same composite children), send just that range.
        FlatScene flat; flatten(scene, flat);
        size_t lo = flat.objects.size(), hi = 0;
        bool same_shape = scene_uploaded_ && same_except_object_records(last_, flat);
        if (same_shape) {
            for (size_t i = 0; i < flat.objects.size(); i++) {
                if (std::memcmp(&flat.objects[i], &last_.objects[i], sizeof(rt_object)) == 0) continue;
                const rt_object &a = flat.objects[i], &b = last_.objects[i];
                if (a.kind != b.kind || a.first != b.first || a.second != b.second) { same_shape = false; break; }
                lo = std::min(lo, i); hi = std::max(hi, i + 1);
            }
        }
        if (!same_shape) {
            if (rt_set_scene(ctx_, &flat.desc) != RT_OK) return fail("rt_set_scene");
            full_uploads_ += 1;
        } else if (lo < hi) {
            if (rt_update_objects(ctx_, (uint32_t)lo, (uint32_t)(hi - lo), flat.objects.data() + lo) != RT_OK) return fail("rt_update_objects");
            object_updates_ += 1;
        }
        last_ = std::move(flat);
        last_.desc.objects = last_.objects.data(); last_.desc.materials = last_.materials.data();
        last_.desc.lights = last_.lights.data(); last_.desc.sdfs = last_.sdfs.data();
        rt_camera_desc cd = camera.desc();
        if (rt_set_camera(ctx_, &cd) != RT_OK) return fail("rt_set_camera");
        scene_uploaded_ = true;
    }
    if (frame_index > max_frames_rendering || (frame_index > 1 && !enable_accumulation)) return "";  // :98-102
    if (!scene_uploaded_) return "render called before the first updated frame";
    if (img.size() != camera.width * camera.height * 4) return "img has the wrong length";
    if (rt_render(ctx_, frame_index, enable_accumulation ? 1 : 0, reset ? 1 : 0, seed, img.data(), img.size()) != RT_OK)
        return fail("rt_render");
    frame_index += 1;  // :149
    return "";
}

std::string Renderer::read_accumulated(std::vector<float> &out_vec4) {
    if (!ctx_) return "no context";
    if (rt_read_accum(ctx_, out_vec4.data(), out_vec4.size() / 4) != RT_OK) return rt_last_error(ctx_);
    return "";
}

}  // namespace rt

// ------------------------------------------------------------------------------------------------
// C view
// ------------------------------------------------------------------------------------------------
struct rth_scene {
    rt::scenes::Built built;
    rt::FlatScene flat;
};

extern "C" {

rth_scene *rth_scene_build(const char *name) {
    rth_scene *s = new rth_scene();
    if (!name || !rt::scenes::by_name(name, s->built)) { delete s; return nullptr; }
    rt::flatten(s->built.scene, s->flat);
    return s;
}
void rth_scene_free(rth_scene *s) { delete s; }
const rt_scene_desc *rth_scene_desc(rth_scene *s) { return s ? &s->flat.desc : nullptr; }
void rth_scene_camera(rth_scene *s, uint32_t width, uint32_t height, rt_camera_desc *out) {
    rt::Camera c = s->built.camera;
    rt::CameraEvent e; e.kind = rt::CameraEvent::Resize; e.w = width; e.h = height;
    c.update({e}, 0.0f);
    *out = c.desc();
}
int32_t rth_scene_renderer_accumulation(rth_scene *s) { return s->built.renderer_accumulation ? 1 : 0; }
void rth_camera(const float position[3], const float look_at[3], uint32_t width, uint32_t height, rt_camera_desc *out) {
    rt::Camera c = rt::Camera::new_with_pos(rt::vec3(position[0], position[1], position[2]), rt::vec3(look_at[0], look_at[1], look_at[2]));
    rt::CameraEvent e; e.kind = rt::CameraEvent::Resize; e.w = width; e.h = height;
    c.update({e}, 0.0f);
    *out = c.desc();
}
void rth_object_update(rt_object *o) { rt::Object3D x; x.o = *o; x.update(); *o = x.o; }

// App3D::run's frame loop without the window (src/app.rs:219-248): every frame calls the scene's update function with
// ts = frame * dt, then Renderer::render with its return value as `updated`.  Writes the n_frames RGBA8 frames back to
// back into img_out and, per frame, the flattened objects after the update into objects_out (n_frames * n_objects).
int32_t rth_render_animation(rth_scene *s, uint32_t width, uint32_t height, uint32_t n_frames, float dt, uint32_t n_textures,
                             const uint32_t *tex_w, const uint32_t *tex_h, const uint8_t *const *tex_rgb8, int32_t device_id,
                             uint8_t *img_out, rt_object *objects_out, uint64_t *full_uploads, uint64_t *object_updates,
                             char *err, size_t err_len) {
    auto fail = [&](const std::string &m) { if (err && err_len) { std::strncpy(err, m.c_str(), err_len - 1); err[err_len - 1] = 0; } return -1; };
    if (!s || !img_out) return fail("null argument");
    rt::Scene scene = s->built.scene;
    scene.textures.clear();
    for (uint32_t i = 0; i < n_textures; i++) {
        rt::Texture t; t.width = tex_w[i]; t.height = tex_h[i];
        t.bytes.assign(tex_rgb8[i], tex_rgb8[i] + (size_t)tex_w[i] * tex_h[i] * 3);
        scene = scene.with_texture(t);
    }
    rt::Camera camera = s->built.camera;
    rt::CameraEvent e; e.kind = rt::CameraEvent::Resize; e.w = width; e.h = height;
    camera.update({e}, 0.0f);
    rt::Renderer renderer; renderer.device_id = device_id;
    const size_t frame_bytes = (size_t)width * height * 4;
    std::vector<uint8_t> img(frame_bytes, 0);
    for (uint32_t f = 0; f < n_frames; f++) {
        bool updated = f == 0;
        if (scene.update_func && scene.update_func(scene, (float)f * dt)) updated = true;
        std::string r = renderer.render(scene, img, camera, updated, 200);
        if (!r.empty()) return fail(r);
        std::memcpy(img_out + (size_t)f * frame_bytes, img.data(), frame_bytes);
        if (objects_out) {
            rt::FlatScene flat; rt::flatten(scene, flat);
            std::memcpy(objects_out + (size_t)f * flat.objects.size(), flat.objects.data(), flat.objects.size() * sizeof(rt_object));
        }
    }
    if (full_uploads) *full_uploads = renderer.full_uploads();
    if (object_updates) *object_updates = renderer.object_updates();
    return 0;
}

int32_t rth_render_frames(rth_scene *s, uint32_t width, uint32_t height, uint32_t n_frames, int32_t accumulate, uint32_t n_textures,
                          const uint32_t *tex_w, const uint32_t *tex_h, const uint8_t *const *tex_rgb8, int32_t device_id,
                          uint8_t *img_out, float *accum_out, char *err, size_t err_len) {
    auto fail = [&](const std::string &m) { if (err && err_len) { std::strncpy(err, m.c_str(), err_len - 1); err[err_len - 1] = 0; } return -1; };
    if (!s || !img_out) return fail("null argument");
    rt::Scene scene = s->built.scene;
    scene.textures.clear();
    for (uint32_t i = 0; i < n_textures; i++) {
        rt::Texture t; t.width = tex_w[i]; t.height = tex_h[i];
        t.bytes.assign(tex_rgb8[i], tex_rgb8[i] + (size_t)tex_w[i] * tex_h[i] * 3);
        scene = scene.with_texture(t);
    }
    rt::Camera camera = s->built.camera;
    rt::CameraEvent e; e.kind = rt::CameraEvent::Resize; e.w = width; e.h = height;
    camera.update({e}, 0.0f);
    rt::Renderer renderer; renderer.device_id = device_id; renderer.enable_accumulation = accumulate != 0;
    std::vector<uint8_t> img((size_t)width * height * 4, 0);
    for (uint32_t f = 0; f < n_frames; f++) {
        std::string r = renderer.render(scene, img, camera, f == 0, 200);
        if (!r.empty()) return fail(r);
    }
    std::memcpy(img_out, img.data(), img.size());
    if (accum_out) {
        std::vector<float> acc((size_t)width * height * 4);
        std::string r = renderer.read_accumulated(acc);
        if (!r.empty()) return fail(r);
        std::memcpy(accum_out, acc.data(), acc.size() * sizeof(float));
    }
    return 0;
}

}  // extern "C"
