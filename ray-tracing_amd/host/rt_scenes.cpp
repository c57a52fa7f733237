// rt_scenes.cpp — the reference's four demo scenes and two synthetic benchmark scenes, as data.
//
// Each builder restates the literals of one reference binary at construction time, and sets
// Scene::update_func to the binary's per-frame update (applied only by callers that animate):  scene1 src/bin/scene1.rs:27-175, scene2 src/bin/scene2.rs:24-62,
// scene3 src/bin/scene3.rs:20-73, scene4 src/bin/scene4.rs:29-118.
// tex4 and spheres10k have no counterpart in the reference; they exercise texture sampling on all four
// shipped images and a 10 000-object scene respectively.
#include <cmath>

#include "rt_host.h"

namespace rt {
namespace scenes {

static Material mat(float ambience, float diffuse, float shininess, float specular, Vec3 albedo, MaterialType kind, int texture = -1,
                    float emission = 0.0f) {
    Material m; m.ambience = ambience; m.diffuse = diffuse; m.shininess = shininess; m.specular = specular;
    m.albedo = albedo; m.kind = kind; m.texture = texture; m.emission_power = emission; return m;
}

// The bins load chess.png, wood.png, stone3.jpg, earth_clouds.jpg into slots 0..3.
static void declare_texture_slots(Scene &s, size_t n) {
    static const char *names[4] = {"chess.png", "wood.png", "stone3.jpg", "earth_clouds.jpg"};
    for (size_t i = 0; i < n; i++) { Texture t; t.path = names[i % 4]; s.textures.push_back(t); }
}

// src/bin/scene1.rs:13-24: the cuboid (object 3) tumbles
static bool update_scene1(Scene &s, float /*ts*/) {
    const float speed = 0.2f;
    if (s.objects.size() > 3 && s.objects[3].o.kind == RT_CUBOID) {
        rt_object &c = s.objects[3].o;
        c.rotation_axis[0] += 2.0f * speed;
        c.rotation_axis[2] += 4.0f * speed;
        c.rotation_axis[1] += 2.0f * speed;
        s.objects[3].update();
    }
    return true;
}
// src/bin/scene2.rs:9-21: the first cuboid tumbles
static bool update_scene2(Scene &s, float /*ts*/) {
    const float speed = 0.2f;
    for (Object3D &obj : s.objects) {
        if (obj.o.kind != RT_CUBOID) continue;
        obj.o.rotation_axis[0] += 2.0f * speed;
        obj.o.rotation_axis[2] += 4.0f * speed;
        obj.o.rotation_axis[1] += 2.0f * speed;
        obj.update();
        break;
    }
    return true;
}
// src/bin/scene3.rs:10-12
static bool update_scene3(Scene &, float) { return false; }
// src/bin/scene4.rs:14-26: the sphere (object 2) bobs, the cylinder (object 3) spins
static bool update_scene4(Scene &s, float ts) {
    const float speed = 0.5f;
    if (s.objects.size() > 2 && s.objects[2].o.kind == RT_SPHERE) {
        s.objects[2].o.position[1] = std::sin(ts) * 2.0f * speed + 0.8f;
        s.objects[2].update();
    }
    if (s.objects.size() > 3 && s.objects[3].o.kind == RT_CYLINDER) {
        s.objects[3].o.rotation_axis[1] += 2.0f * speed;
        s.objects[3].update();
    }
    return true;
}

Built scene1() {
    Vec2 bounds = vec2(5.0f, 5.0f);
    std::vector<Object3D> objs = {
        Sphere::create(vec3(1.2f, 0.0f, 2.5f), 0.5f, 0),
        Plane::create(vec3(0.0f, 1.0f, 0.0f), vec3(0.0f, -0.5f, 0.0f), &bounds, 1),
        Sphere::new_sphere_with_rotation(vec3(3.0f, 0.5f, 0.8f), vec3(-90.0f, 0.0f, 0.0f), 0.7f, 2),
        Cuboid::create(vec3(-1.0f, 1.3f, 2.0f), vec3(0.0f, 0.0f, 0.0f), vec3(0.6f, 1.0f, 0.2f), 3),
        Sphere::create(vec3(1.5f, 0.0f, 0.0f), 0.5f, 4),
        Cone::create(vec3(2.3f, 0.7f, 2.0f), 0.5f, 1.0f, vec3(120.0f, 0.0f, 0.0f), 5),
        Cylinder::create(vec3(2.3f, 0.0f, 3.0f), 1.0f, vec3(90.0f, 0.0f, 0.0f), 0.4f, 6),
        Triangle::create(vec3(1.5f, 1.0f, 0.0f), vec3(1.5f, 0.0f, 0.0f), vec3(2.5f, 1.0f, 0.0f), 1),
    };
    std::vector<Material> mats = {
        mat(0.4f, 0.3f, 12.0f, 3.0f, vec3(1.0f, 1.0f, 1.0f), MaterialType::Refractive(1.0f, 1.08f, 0.1f)),
        mat(0.4f, 0.1f, 15.0f, 0.8f, vec3(0.4f, 0.4f, 0.4f), MaterialType::Reflective(0.8f), 0),
        mat(0.2f, 0.8f, 200.0f, 1.2f, vec3(0.0f, 0.2f, 0.9f), MaterialType::Reflective(0.6f), 3),
        mat(0.4f, 0.8f, 70.0f, 1.1f, vec3(0.5f, 0.5f, 0.5f), MaterialType::Reflective(0.8f), 1),
        mat(0.4f, 0.8f, 80.0f, 0.4f, vec3(0.8f, 0.6f, 0.1f), MaterialType::Reflective(0.4f)),
        mat(0.5f, 0.1f, 80.0f, 0.1f, vec3(0.3f, 0.7f, 0.5f), MaterialType::Reflective(0.4f), 0),
        mat(0.6f, 0.3f, 40.0f, 0.8f, vec3(0.1f, 0.5f, 0.9f), MaterialType::Reflective(0.4f)),
    };
    Built b;
    b.scene = Scene::create(objs, mats);
    declare_texture_slots(b.scene, 4);
    b.scene = b.scene.with_light(Light::Positional(vec3(1.0f, 1.0f, 1.0f), vec3(2.0f, 2.0f, 2.0f), 6.0f))
                  .with_light(Light::Positional(vec3(1.0f, 1.0f, 1.0f), vec3(3.0f, 2.0f, -2.0f), 6.0f));
    b.scene.diffuse = false;
    b.scene.shadow_casting = false;
    b.camera = Camera::new_with_pos(vec3(3.8536084f, 0.75215954f, 4.388293f), vec3(-0.76750606f, -0.05052291f, -0.6390541f));
    return b;
}

Built scene2() {
    Built b;
    b.scene.max_ray_bounces = 5;
    b.scene.ambient_color = vec3(0.0f, 0.0f, 0.0f);
    b.scene.diffuse = true;
    b.scene.objects = {
        Sphere::create(vec3(0.0f, -100.5f, 0.0f), 100.0f, 0),
        Sphere::create(vec3(10.0f, 15.0f, -40.0f), 20.0f, 1),
        Sphere::create(vec3(0.0f, 0.5f, -0.5f), 1.0f, 2),
    };
    Material m0; m0.albedo = vec3(0.9f, 0.9f, 0.2f); m0.kind = MaterialType::Reflective(1.0f); m0.emission_power = 0.0f;
    Material m1; m1.albedo = vec3(0.9f, 0.5f, 0.2f); m1.kind = MaterialType::Reflective(1.0f); m1.emission_power = 14.0f;
    Material m2; m2.albedo = vec3(0.0f, 0.3f, 0.7f); m2.kind = MaterialType::Reflective(1.0f); m2.emission_power = 0.8f;
    b.scene.materials = {m0, m1, m2};
    b.renderer_accumulation = true;
    b.camera = Camera::new_with_pos(vec3(3.8536084f, 0.75215954f, 4.388293f), vec3(-0.76750606f, -0.05052291f, -0.6390541f));
    return b;
}

Built scene3() {
    Vec2 bounds = vec2(5.0f, 5.0f);
    std::vector<Object3D> objs = {
        Plane::create(vec3(0.0f, 1.0f, 0.0f), vec3(0.0f, 0.0f, 0.0f), &bounds, 0),
        Sphere::create(vec3(0.0f, 0.5f, 0.0f), 0.5f, 1),
    };
    std::vector<Material> mats = {
        mat(1.6f, 0.2f, 5.0f, 0.8f, vec3(0.4f, 0.4f, 0.4f), MaterialType::Reflective(1.0f)),
        mat(0.4f, 0.7f, 60.0f, 1.5f, vec3(0.1f, 0.5f, 0.9f), MaterialType::Reflective(0.4f)),
    };
    Built b;
    b.scene = Scene::create(objs, mats);
    declare_texture_slots(b.scene, 4);
    b.scene = b.scene.with_light(Light::SphericalPositional(vec3(1.0f, 0.5f, 1.0f), vec3(1.0f, 3.0f, 2.0f), 1.0f, 8.0f))
                  .with_light(Light::SphericalPositional(vec3(0.0f, 0.7f, 1.0f), vec3(-2.0f, 3.0f, -2.0f), 1.0f, 8.0f));
    b.scene.diffuse = false;
    b.scene.shadow_casting = true;
    b.camera = Camera::new_with_pos(vec3(0.0f, 2.0f, 5.0f), vec3(0.0f, 0.0f, -5.0f));
    return b;
}

Built scene4() {
    Vec2 bounds = vec2(5.0f, 5.0f);
    std::vector<Object3D> objs = {
        Union::create(1, 2),
        Plane::create(vec3(0.0f, 1.0f, 0.0f), vec3(0.0f, 0.0f, 0.0f), &bounds, 0),
        Sphere::create(vec3(0.0f, -1.0f, -2.0f), 1.0f, 1),
        Cylinder::create(vec3(-1.0f, 0.8f, 0.2f), 0.2f, vec3(0.0f, 0.0f, 45.0f), 1.0f, 2),
        Cuboid::create(vec3(-1.0f, 1.5f, 0.2f), vec3(0.0f, 20.0f, 0.0f), vec3(0.5f, 1.0f, 0.5f), 1),
        Sphere::new_sphere_with_rotation(vec3(2.0f, 1.0f, -2.0f), vec3(-90.0f, 0.0f, 0.0f), 1.0f, 3),
        Substraction::create(3, 4),
    };
    std::vector<Material> mats = {
        mat(0.5f, 0.2f, 5.0f, 0.8f, vec3(1.0f, 0.8f, 0.6f), MaterialType::Reflective(1.0f)),
        mat(0.3f, 0.3f, 60.0f, 0.2f, vec3(1.0f, 0.8f, 0.6f), MaterialType::Reflective(1.0f)),
        mat(0.4f, 0.4f, 50.0f, 2.5f, vec3(0.0f, 0.4f, 1.0f), MaterialType::Reflective(1.0f), 2),
        mat(0.3f, 0.4f, 84.0f, 0.8f, vec3(0.0f, 0.4f, 1.0f), MaterialType::Reflective(1.0f), 3),
    };
    Built b;
    b.scene = Scene::create(objs, mats);
    b.scene.sdfs = {0, 3, 5};
    declare_texture_slots(b.scene, 4);
    Vec3 d = vec3(-1.0f, -1.0f, -2.0f);
    float inv_len = 1.0f / std::sqrt((d.x * d.x) + (d.y * d.y) + (d.z * d.z));  // Vec3::normalize
    b.scene = b.scene.with_light(Light::Directional(vec3(1.0f, 1.0f, 1.0f), vec3(d.x * inv_len, d.y * inv_len, d.z * inv_len), 2.0f));
    b.scene.diffuse = false;
    b.scene.ray_marching = true;
    b.scene.shadow_casting = true;
    b.camera = Camera::new_with_pos(vec3(0.0f, 2.0f, 4.0f), vec3(0.0f, 0.0f, -1.0f));
    return b;
}

// Four unit spheres textured with the four shipped images over a chess plane.
Built tex4() {
    Vec2 bounds = vec2(8.0f, 8.0f);
    std::vector<Object3D> objs = {
        Plane::create(vec3(0.0f, 1.0f, 0.0f), vec3(0.0f, -0.5f, 0.0f), &bounds, 4),
        Sphere::new_sphere_with_rotation(vec3(-3.0f, 0.5f, 0.0f), vec3(-90.0f, 0.0f, 0.0f), 1.0f, 0),
        Sphere::new_sphere_with_rotation(vec3(-1.0f, 0.5f, 0.0f), vec3(-90.0f, 0.0f, 0.0f), 1.0f, 1),
        Sphere::new_sphere_with_rotation(vec3(1.0f, 0.5f, 0.0f), vec3(-90.0f, 0.0f, 0.0f), 1.0f, 2),
        Sphere::new_sphere_with_rotation(vec3(3.0f, 0.5f, 0.0f), vec3(-90.0f, 0.0f, 0.0f), 1.0f, 3),
    };
    std::vector<Material> mats = {
        mat(0.3f, 0.8f, 60.0f, 1.0f, vec3(0.5f, 0.5f, 0.5f), MaterialType::Reflective(0.7f), 0),
        mat(0.3f, 0.8f, 60.0f, 1.0f, vec3(0.5f, 0.5f, 0.5f), MaterialType::Reflective(0.7f), 1),
        mat(0.3f, 0.8f, 60.0f, 1.0f, vec3(0.5f, 0.5f, 0.5f), MaterialType::Reflective(0.7f), 2),
        mat(0.3f, 0.8f, 60.0f, 1.0f, vec3(0.5f, 0.5f, 0.5f), MaterialType::Reflective(0.7f), 3),
        mat(0.4f, 0.1f, 15.0f, 0.8f, vec3(0.4f, 0.4f, 0.4f), MaterialType::Reflective(0.8f), 0),
    };
    Built b;
    b.scene = Scene::create(objs, mats);
    b.scene.max_ray_bounces = 5;
    declare_texture_slots(b.scene, 4);
    b.scene = b.scene.with_light(Light::Positional(vec3(1.0f, 1.0f, 1.0f), vec3(2.0f, 2.0f, 2.0f), 6.0f))
                  .with_light(Light::Positional(vec3(1.0f, 1.0f, 1.0f), vec3(3.0f, 2.0f, -2.0f), 6.0f));
    b.camera = Camera::new_with_pos(vec3(0.0f, 2.0f, 7.0f), vec3(0.0f, -0.2f, -1.0f));
    return b;
}

// PCG32 (XSH-RR 64/32), the generator that places the synthetic spheres.
struct Pcg32 {
    uint64_t state = 0, inc = 0;
    explicit Pcg32(uint64_t seed, uint64_t seq = 54u) {
        inc = (seq << 1u) | 1u; next(); state += seed; next();
    }
    uint32_t next() {
        uint64_t old = state;
        state = old * 6364136223846793005ULL + inc;
        uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
    }
    float uniform(float lo, float hi) { return lo + (hi - lo) * ((float)(next() >> 8) * (1.0f / 16777216.0f)); }
};

// Ground sphere + (n_spheres - 1) small spheres on it; 16 diffuse materials, the last one emissive so that
// every 16th sphere is a light; sky-blue ambient so that escaping paths contribute.
Built spheres10k(uint32_t n_spheres, uint64_t seed) {
    Pcg32 rng(seed);
    Built b;
    b.scene.max_ray_bounces = 5;
    b.scene.diffuse = true;
    b.scene.ambient_color = vec3(0.4f * 0.3f, 0.7f * 0.3f, 1.0f * 0.3f);
    for (int i = 0; i < 16; i++) {
        Material m;
        m.albedo = vec3(rng.uniform(0.1f, 0.9f), rng.uniform(0.1f, 0.9f), rng.uniform(0.1f, 0.9f));
        m.kind = MaterialType::Reflective(1.0f);
        m.emission_power = (i == 15) ? 6.0f : 0.0f;
        b.scene.materials.push_back(m);
    }
    b.scene.objects.push_back(Sphere::create(vec3(0.0f, -1000.5f, 0.0f), 1000.0f, 0));
    for (uint32_t i = 1; i < n_spheres; i++) {
        float r = rng.uniform(0.2f, 0.6f);
        float x = rng.uniform(-60.0f, 60.0f), z = rng.uniform(-60.0f, 60.0f);
        b.scene.objects.push_back(Sphere::create(vec3(x, r - 0.5f, z), r, i % 16));
    }
    b.renderer_accumulation = true;
    b.camera = Camera::new_with_pos(vec3(0.0f, 6.0f, 40.0f), vec3(0.0f, -0.15f, -1.0f));
    return b;
}

bool by_name(const std::string &name, Built &out) {
    if (name == "scene1") { out = scene1(); out.scene.update_func = update_scene1; return true; }
    if (name == "scene2") { out = scene2(); out.scene.update_func = update_scene2; return true; }
    if (name == "scene3") { out = scene3(); out.scene.update_func = update_scene3; return true; }
    if (name == "scene4") { out = scene4(); out.scene.update_func = update_scene4; return true; }
    if (name == "tex4") { out = tex4(); return true; }
    if (name == "spheres10k") { out = spheres10k(); return true; }
    if (name == "spheres1k") { out = spheres10k(1000); return true; }
    if (name == "spheres200") { out = spheres10k(200); return true; }
    return false;
}

}  // namespace scenes
}  // namespace rt
