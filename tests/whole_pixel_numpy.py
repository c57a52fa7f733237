"""Whole pixels of the reference renderer re-derived in numpy float32, straight from the Rust sources.

This is a second, independent restatement of the per-pixel function for the two scene families the reference ships with
a deterministic path -- Whitted ray tracing of spheres and planes with shadows and reflection (src/bin/scene3.rs), and
ray marching of the SDF list of src/bin/scene4.rs with its normal, occlusion and tri-planar texturing.  It shares no code
with oracle/ (C++) or the HIP kernels: scalar numpy expressions in the order the Rust source writes them.  Inputs are the
same plain-data scene and camera records the C ABI takes (matrices as glam stores them, column-major).

Transcendentals (powf, atan) come from numpy here, from a specified algorithm in the oracle: values agree to about one
unit in the last place, so comparisons use a small tolerance instead of bit equality.
"""
import numpy as np

F = np.float32
EPSILON = F(0.0001)            # src/ray.rs:9
F32_MAX = np.finfo(np.float32).max


def v3(x, y, z):
    return np.array([x, y, z], dtype=np.float32)


def dot(a, b):                 # glam Vec3::dot: x*x' + y*y' + z*z', left to right
    return F(F(a[0] * b[0]) + F(a[1] * b[1])) + F(a[2] * b[2])


def length(a):
    return np.sqrt(dot(a, a), dtype=np.float32)


def normalize(a):              # glam: self * length().recip()
    return a * (F(1.0) / length(a))


def mat_mul_vec4(m, x, y, z, w):
    """glam Mat4 * Vec4 for a column-major 16-float matrix: ((c0*x + c1*y) + c2*z) + c3*w, xyz rows."""
    c = np.asarray(m, dtype=np.float32).reshape(4, 4)   # c[j] = column j
    return ((c[0] * F(x) + c[1] * F(y)) + c[2] * F(z)) + c[3] * F(w)


def reflect(vec, normal):      # src/utils/geometry.rs:7-9
    return vec - (F(2.0) * dot(vec, normal)) * normal


def powf3(v, e):
    return np.power(v, F(e), dtype=np.float32)


# ---------------------------------------------------------------------------------------------------------------------
# camera (src/camera.rs:128-157)
# ---------------------------------------------------------------------------------------------------------------------
def camera_ray(cam, x, y):
    inv_w, inv_h = F(1.0) / F(cam.width), F(1.0) / F(cam.height)
    sx = F(2.0) * (F(x) * inv_w) - F(1.0)
    sy = F(2.0) * (F(y) * inv_h) - F(1.0)
    target = mat_mul_vec4(cam.inverse_perspective[:], sx, sy, 1.0, 1.0)
    v = target[:3] / target[3]
    world = mat_mul_vec4(cam.inverse_view[:], v[0], v[1], v[2], 0.0)[:3]
    pos = v3(*cam.position[:])
    return pos, normalize(world - pos)


# ---------------------------------------------------------------------------------------------------------------------
# intersect() of the two kinds the Whitted test scene uses
# ---------------------------------------------------------------------------------------------------------------------
INV_PI = F(1.0) / F(np.pi)     # src/utils/sphere.rs:13


def atan(x):
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.arctan(F(x), dtype=np.float32)


def fdiv(a, b):                # f32 division with IEEE results for a zero divisor
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.divide(F(a), F(b), dtype=np.float32)


def cross(a, b):               # glam Vec3::cross
    return v3(a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1])


def signum(x):                 # f32::signum: +-1 by the sign bit, NaN stays NaN
    if np.isnan(x):
        return F(x)
    return F(-1.0) if np.signbit(x) else F(1.0)


def fmax(a, b):                # f32::max
    return F(np.fmax(F(a), F(b)))


def fmin(a, b):
    return F(np.fmin(F(a), F(b)))


def sphere_intersect(o, ro, rd):   # src/utils/sphere.rs:98-147 -> (t, point, normal, u, v) or None
    d = mat_mul_vec4(o.inv_transform[:], rd[0], rd[1], rd[2], 0.0)[:3]
    p = mat_mul_vec4(o.inv_transform[:], ro[0], ro[1], ro[2], 1.0)[:3]
    a = dot(d, d)
    b = F(2.0) * dot(p, d)
    c = dot(p, p) - F(1.0)
    disc = b * b - F(4.0) * a * c
    if disc < 0:
        return None
    t1 = (-b - np.sqrt(disc, dtype=np.float32)) / (F(2.0) * a)
    l = p + d * t1
    normal = normalize(mat_mul_vec4(o.transform[:], l[0], l[1], l[2], 0.0)[:3])
    u = atan(fdiv(F(l[0] * l[0]) + F(l[1] * l[1]), l[2]))
    v = atan(fdiv(l[1], l[0]))
    return t1, ro + rd * t1, normal, v * INV_PI, u * INV_PI


def plane_intersect(o, ro, rd):    # src/utils/plane.rs:41-74
    n, pt = v3(*o.normal[:]), v3(*o.point[:])
    denom = dot(rd, n)
    if abs(denom) < 1e-6:
        return None
    t = dot(pt - ro, n) / denom
    if t < 0:
        return None
    hp = ro + rd * t
    if o.has_max_dist and (abs(hp[2]) > F(o.max_dist[1]) or abs(hp[0]) > F(o.max_dist[0])):
        return None
    sign = F(1.0) if dot(rd, n) < 0 else F(-1.0)
    return t, hp, sign * n, hp[0] * F(0.1), hp[2] * F(0.1)


def local_ray(o, ro, rd):
    return (mat_mul_vec4(o.inv_transform[:], ro[0], ro[1], ro[2], 1.0)[:3], mat_mul_vec4(o.inv_transform[:], rd[0], rd[1], rd[2], 0.0)[:3])


def cuboid_intersect(o, ro, rd):   # src/utils/cuboid.rs:84-127
    lo, ld = local_ray(o, ro, rd)
    dim = v3(*o.dimension[:])
    b_min = ((v3(dim[0], 0, 0) * F(-1.0) + v3(0, dim[1], 0) * F(-1.0)) + v3(0, 0, dim[2]) * F(-1.0)) + v3(0, 0, 0) * F(0.0)   # :47-48
    b_max = ((v3(dim[0], 0, 0) * F(1.0) + v3(0, dim[1], 0) * F(1.0)) + v3(0, 0, dim[2]) * F(1.0)) + v3(0, 0, 0) * F(0.0)
    inv = np.array([fdiv(1.0, ld[i]) for i in range(3)], dtype=np.float32)
    with np.errstate(invalid="ignore"):
        t_min = (b_min - lo) * inv
        t_max = (b_max - lo) * inv
    t_enter = np.array([fmin(t_min[i], t_max[i]) for i in range(3)], dtype=np.float32)
    t_exit = np.array([fmax(t_min[i], t_max[i]) for i in range(3)], dtype=np.float32)
    t_near = fmax(fmax(t_enter[0], t_enter[1]), t_enter[2])
    t_far = fmin(fmin(t_exit[0], t_exit[1]), t_exit[2])
    if t_near > t_far or t_far < 0:
        return None
    step = np.array([F(0.0) if t_enter[i] < t_near else F(1.0) for i in range(3)], dtype=np.float32)   # geometry.rs:21-27
    a = -np.array([signum(ld[i]) for i in range(3)], dtype=np.float32) * step
    normal = mat_mul_vec4(o.transform[:], a[0], a[1], a[2], 0.0)[:3]
    hp = ro + rd * t_near
    op = mat_mul_vec4(o.inv_transform[:], hp[0], hp[1], hp[2], 1.0)[:3]
    ax, ay, az = abs(a[0]), abs(a[1]), abs(a[2])
    u = (ax * op[1] + ay * op[2]) + az * op[0]
    v = (ax * op[2] + ay * op[0]) + az * op[1]
    return t_near, hp, normal, u, v


def cylinder_intersect(o, ro, rd):   # src/utils/cylinder.rs:89-174
    ro3, rd3 = local_ray(o, ro, rd)
    a = F(rd3[0] * rd3[0]) + F(rd3[1] * rd3[1])
    b = F(2.0) * (F(ro3[0] * rd3[0]) + F(ro3[1] * rd3[1]))
    c = (F(ro3[0] * ro3[0]) + F(ro3[1] * ro3[1])) - F(1.0)
    disc = b * b - F(4.0) * a * c
    if disc > 0:
        t1 = fdiv(-b - np.sqrt(disc, dtype=np.float32), F(2.0) * a)
        h = ro3 + rd3 * t1
        if abs(h[2]) < 0.5:
            normal = normalize(mat_mul_vec4(o.transform[:], h[0], h[1], 0.0, 0.0)[:3])
            return t1, ro + rd * t1, normal, atan(fdiv(h[1], h[0])), h[2] * F(2.0)
    inv_z = fdiv(-1.0, rd3[2])
    t1 = (ro3[2] - F(0.5)) * inv_z
    t2 = (ro3[2] + F(0.5)) * inv_z
    with np.errstate(invalid="ignore"):
        h1, h2 = ro3 + rd3 * t1, ro3 + rd3 * t2
    ok1 = (F(h1[0] * h1[0]) + F(h1[1] * h1[1])) < 1
    ok2 = (F(h2[0] * h2[0]) + F(h2[1] * h2[1])) < 1
    if ok1 and ok2:
        t = t1 if t1 < t2 else t2
    elif ok1:
        t = t1
    elif ok2:
        t = t2
    else:
        return None
    h = ro3 + rd3 * t
    normal = normalize(mat_mul_vec4(o.transform[:], 0.0, 0.0, h[2], 0.0)[:3])
    return t, ro + rd * t, normal, h[0], h[1]


def cone_intersect(o, ro, rd):       # src/utils/cone.rs:72-152
    ro3, rd3 = local_ray(o, ro, rd)
    a = (F(rd3[0] * rd3[0]) + F(rd3[1] * rd3[1])) - rd3[2] * rd3[2]
    b = F(2.0) * ((F(ro3[0] * rd3[0]) + F(ro3[1] * rd3[1])) - ro3[2] * rd3[2])
    c = (F(ro3[0] * ro3[0]) + F(ro3[1] * ro3[1])) - ro3[2] * ro3[2]
    disc = b * b - F(4.0) * a * c
    cone = None
    if disc > 0:
        t1 = fdiv(-b - np.sqrt(disc, dtype=np.float32), F(2.0) * a)
        h = ro3 + rd3 * t1
        if h[2] > 0 and h[2] < 1:
            n = v3(h[0], h[1], -np.sqrt(F(h[0] * h[0]) + F(h[1] * h[1]), dtype=np.float32))
            normal = normalize(mat_mul_vec4(o.transform[:], n[0], n[1], n[2], 0.0)[:3])
            cone = (t1, ro + rd * t1, normal, atan(fdiv(h[1], h[0])), h[2] * F(2.0) + F(1.0))
    t1 = fdiv(ro3[2] - F(1.0), -rd3[2])
    with np.errstate(invalid="ignore"):
        h1 = ro3 + rd3 * t1
    if (F(h1[0] * h1[0]) + F(h1[1] * h1[1])) < 1:
        if cone is not None and cone[0] < t1:
            return cone
        normal = mat_mul_vec4(o.transform[:], 0.0, 0.0, 1.0, 0.0)[:3]
        return t1, ro + rd * t1, normal, h1[0], h1[1]
    return cone


def triangle_intersect(o, ro, rd):   # src/utils/triangle.rs:35-88
    p1, p2, p3 = v3(*o.v1[:]), v3(*o.v2[:]), v3(*o.v3[:])
    edge_ab, edge_ac = p2 - p1, p3 - p1
    normal = cross(edge_ab, edge_ac)
    denom = dot(rd, normal)
    if abs(denom) < 1e-6:
        return None
    t = dot(p1 - ro, normal) / denom
    hp = ro + rd * t
    edge_bc, edge_ca = p3 - p2, p1 - p3
    a_test, b_test, c_test = cross(edge_ab, hp - p1), cross(edge_bc, hp - p2), cross(edge_ca, hp - p3)
    if not (dot(a_test, normal) > 0 and dot(b_test, normal) > 0 and dot(c_test, normal) > 0):
        return None
    sign = -signum(dot(rd, normal))
    return t, hp, normalize(sign * normal), dot(cross(edge_ab, hp - p1), normal), dot(cross(edge_ca, hp - p3), normal)


class Whitted:
    """RayTracing::{albedo, color, light, trace_ray} (src/ray_tracing/ray_tracing.rs:17-78,160-241) for scenes of spheres and
    planes with untextured Reflective materials (what src/bin/scene3.rs builds) and Refractive ones (Fresnel weight,
    refraction and reflection children, total internal reflection)."""

    def __init__(self, pkg, desc, textures=()):
        self.pkg, self.d, self.tex = pkg, desc, textures
        self.objects = [desc.objects[i] for i in range(desc.n_objects)]
        self.materials = [desc.materials[i] for i in range(desc.n_materials)]
        self.lights = [desc.lights[i] for i in range(desc.n_lights)]
        self.rays = 0
        self.textured_hits = 0
        self.kinds_hit = set()

    def trace_ray(self, ro, rd):
        self.rays += 1
        closest, best = F32_MAX, None
        for idx, o in enumerate(self.objects):
            P = self.pkg                # Ray::hit, src/ray.rs:167-177
            fn = {P.RT_SPHERE: sphere_intersect, P.RT_PLANE: plane_intersect, P.RT_CUBOID: cuboid_intersect, P.RT_CYLINDER: cylinder_intersect,
                  P.RT_CONE: cone_intersect, P.RT_TRIANGLE: triangle_intersect}.get(o.kind)
            h = fn(o, ro, rd) if fn else None
            if h is not None and h[0] > 0 and h[0] < closest:
                closest, best = h[0], (h, idx)
        return best

    def light_dir_dist(self, l, point):   # src/light.rs:42-92
        if l.kind == self.pkg.RT_DIRECTIONAL:
            return v3(*l.vec[:]), F(1.0)
        v = point - v3(*l.vec[:])
        return normalize(v), length(v)

    def blinn_phong(self, rd, point, normal, l, color, m):   # src/ray.rs:68-85
        ldir, _ = self.light_dir_dist(l, point)
        coeff = dot(normal, -ldir)
        ambience = F(m.ambience) * color
        diffuse = F(m.diffuse) * max(coeff, F(0.0)) * color
        half_angle = normalize(-rd - ldir)
        shininess = np.power(max(dot(normal, half_angle), F(0.0)), F(m.shininess), dtype=np.float32)
        specular = F(m.specular) * shininess * color
        return ambience + diffuse + specular

    def light(self, rd, hit, albedo, m, obj_index):   # ray_tracing.rs:28-57
        point, normal = hit[1], hit[2]
        l_acc = v3(0, 0, 0)
        for l in self.lights:
            k = self.blinn_phong(rd, point, normal, l, albedo, m)
            _, dist = self.light_dir_dist(l, point)
            l_acc = l_acc + (k / (dist * dist)) * v3(*l.albedo[:]) * F(l.intensity)
        if self.d.shadow_casting:
            for l in self.lights:
                ldir, _ = self.light_dir_dist(l, point)
                s = self.trace_ray(point + EPSILON * normal, -ldir)
                if s is not None and s[1] != obj_index:
                    l_acc = l_acc * F(0.5)
        return powf3(l_acc, 0.4166)

    def fresnel(self, incident, normal, refraction_index, reflectivity):   # Material::fresnel, src/objects.rs:178-205
        n2, n1 = F(refraction_index), F(1.0)
        r0 = (n1 - n2) / (n1 + n2)
        r0 = r0 * r0
        cos_x = dot(normal, -incident)
        if n1 > n2:
            n = n1 / n2
            sin_t2 = n * n * (F(1.0) - cos_x * cos_x)
            if sin_t2 > 1.0:
                self.total_internal += 1
                return F(1.0)
            cos_x = np.sqrt(F(1.0) - sin_t2, dtype=np.float32)
        x = F(1.0) - cos_x
        ret = r0 + (F(1.0) - r0) * x * x * x * x * x
        return F(reflectivity) + (F(1.0) - F(reflectivity)) * ret

    def refraction_ray(self, rd, point, hit_normal, refraction_index):   # Ray::refraction_ray, src/ray.rs:138-165
        normal = hit_normal
        eta_t, eta_i = F(refraction_index), F(1.0)
        c1 = dot(rd, hit_normal)
        if c1 < 0:
            c1 = -c1
        else:
            normal = -normal
            eta_i, eta_t = eta_t, F(1.0)
        eta = eta_i / eta_t
        k = F(1.0) - eta * eta * (F(1.0) - c1 * c1)
        if k < 0:
            return None
        c2 = np.sqrt(k, dtype=np.float32)
        direction = eta * rd + normal * (eta * c1 - c2)
        return point - EPSILON * normal, direction

    def color_refractive(self, ro, rd, depth, light_color, contribution, hit, obj_index, m):   # ray_tracing.rs:197-236
        point, normal = hit[1], hit[2]
        albedo = v3(*m.albedo[:])
        self.refractive_nodes += 1
        refraction_color = v3(0, 0, 0)
        kr = self.fresnel(rd, normal, m.refraction_index, m.reflectivity)
        rr = self.refraction_ray(rd, point, normal, m.refraction_index)
        if rr is not None:
            refraction_color = self.color(rr[0], rr[1], depth + 1, light_color, contribution * albedo)
        p_light = self.light(rd, hit, albedo, m, obj_index)
        reflection_color = self.color(point + EPSILON * normal, reflect(rd, normal), depth + 1, p_light, contribution * albedo)
        color = reflection_color * kr + refraction_color * (F(1.0) - kr) * F(m.transparency)
        return color * albedo

    total_internal = 0
    refractive_nodes = 0

    def color(self, ro, rd, depth, light_color, contribution):   # ray_tracing.rs:160-241
        if depth >= self.d.max_ray_bounces:
            return light_color
        h = self.trace_ray(ro, rd)
        if h is None:
            return light_color + v3(*self.d.ambient_color[:]) * contribution
        hit, obj_index = h
        self.kinds_hit.add(int(self.objects[obj_index].kind))
        m = self.materials[self.objects[obj_index].material_index]
        if m.kind == self.pkg.RT_REFRACTIVE:
            return self.color_refractive(ro, rd, depth, light_color, contribution, hit, obj_index, m)
        assert m.kind == self.pkg.RT_REFLECTIVE
        albedo = v3(*m.albedo[:])
        if m.texture >= 0:             # ray_tracing.rs:177-179
            self.textured_hits += 1
            albedo = from_uv(self.tex[m.texture], hit[3], hit[4])
        p_light = self.light(rd, hit, albedo, m, obj_index)
        point, normal = hit[1], hit[2]
        rdir = normalize(reflect(rd, normal + v3(0, 0, 0)))        # src/ray.rs:101-122, no jitter
        reflected = self.color(point + normal * EPSILON, rdir, depth + 1, p_light, contribution * albedo)
        rough = F(m.roughness)
        return p_light * rough + p_light * reflected * (F(1.0) - rough)

    def pixel(self, cam, x, y):
        ro, rd = camera_ray(cam, x, y)
        return self.color(ro, rd, 0, v3(0, 0, 0), v3(1, 1, 1))


# ---------------------------------------------------------------------------------------------------------------------
# ray marching (src/ray_marching/ray_marching.rs:24-253 and the sdf() implementations)
# ---------------------------------------------------------------------------------------------------------------------
def clamp01(x):
    return min(max(x, F(0.0)), F(1.0))


def textel(p):                 # src/objects.rs:121-128
    if p < 0:
        return F(1.0) - abs(np.ceil(p) - p)
    if p > 1:
        return p - np.floor(p)
    return p


def from_uv(tex, u, v):        # src/objects.rs:130-147; tex: (h, w, 3) uint8
    hgt, wid = tex.shape[:2]
    x = int(F(wid - 1) * textel(F(u)))
    y = int(F(hgt - 1) * textel(F(v)))
    return tex[y, x].astype(np.float32) * F(1.0 / 255.0)


class Marcher:
    def __init__(self, pkg, desc, textures):
        self.pkg, self.d, self.tex = pkg, desc, textures
        self.objects = [desc.objects[i] for i in range(desc.n_objects)]
        self.materials = [desc.materials[i] for i in range(desc.n_materials)]
        self.lights = [desc.lights[i] for i in range(desc.n_lights)]
        self.sdf_list = [desc.sdfs[i] for i in range(desc.n_sdfs)]
        self.textured_hits = 0

    def transform_ray(self, o, ro, rd):
        return (mat_mul_vec4(o.inv_transform[:], ro[0], ro[1], ro[2], 1.0)[:3],
                mat_mul_vec4(o.inv_transform[:], rd[0], rd[1], rd[2], 0.0)[:3])

    def sdf(self, idx, ro, rd, t):
        """Object3D::sdf (src/objects.rs:33-44) -> (distance, albedo, transformed ray origin, direction)."""
        o = self.objects[idx]
        P = self.pkg
        if o.kind == P.RT_SPHERE:          # sphere.rs:72-82
            tro, trd = self.transform_ray(o, ro, rd)
            p = tro + trd * t
            return length(p) - F(o.radius), v3(*self.materials[o.material_index].albedo[:]), tro, trd
        if o.kind == P.RT_PLANE:           # plane.rs:31-37
            p = ro + rd * t
            return dot(p - v3(*o.point[:]), v3(*o.normal[:])), v3(*self.materials[o.material_index].albedo[:]), ro, rd
        if o.kind == P.RT_CYLINDER:        # cylinder.rs:56-74
            tro, trd = self.transform_ray(o, ro, rd)
            p = tro + trd * t
            cr = F(0.1)
            dx = (np.sqrt(F(p[0] * p[0]) + F(p[2] * p[2]), dtype=np.float32) - F(o.radius)) + cr
            dy = (abs(p[1]) - F(o.height) * F(0.5)) + cr
            mx, my = max(dx, F(0.0)), max(dy, F(0.0))
            dist = np.sqrt(F(mx * mx) + F(my * my), dtype=np.float32) + min(max(dx, dy), F(0.0)) - cr
            return dist, v3(*self.materials[o.material_index].albedo[:]), tro, trd
        if o.kind == P.RT_UNION:           # union.rs:26-42, geometry.rs:34-50
            d1, a1, o1, r1 = self.sdf(o.first, ro, rd, t)
            d2, a2, o2, r2 = self.sdf(o.second, ro, rd, t)
            k = F(0.7)
            i = clamp01(F(0.5) + F(0.5) * (d2 - d1) / k)
            a = F(1.0) - i
            col = a1 * (F(1.0) - a) + a2 * a
            h = clamp01(F(0.5) + F(0.5) * (d2 - d1) / k)
            d = (d2 * (F(1.0) - h) + d1 * h) - k * h * (F(1.0) - h)
            return (d, col, o1, r1) if d1 < d2 else (d, col, o2, r2)
        raise NotImplementedError(o.kind)

    def sdfs(self, ro, rd, t):             # ray_marching.rs:24-107
        best, obj_idx = (F32_MAX, v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0)), 0
        for idx in self.sdf_list:
            o = self.objects[idx]
            if o.kind in (self.pkg.RT_TRIANGLE, self.pkg.RT_CONE):
                continue
            d = self.sdf(idx, ro, rd, F(t))
            if d[0] < best[0]:
                best = d
                obj_idx = o.second if o.kind in (self.pkg.RT_UNION, self.pkg.RT_SUBSTRACTION) else idx
        return obj_idx, best

    def pixel(self, cam, x, y):            # RayMarching::albedo, ray_marching.rs:205-253
        ro, rd = camera_ray(cam, x, y)
        t, hit = F(0.0), None
        for _ in range(255):               # march_ray, :181-203
            if t > 40.0:
                break
            obj_idx, rh = self.sdfs(ro, rd, t)
            t = t + rh[0]
            if rh[0] < 0.001:
                hit = (obj_idx, rh)
                break
        if hit is None:
            return v3(*self.d.ambient_color[:])
        obj_idx, rh = hit
        p = ro + rd * t
        k = F(0.5773) * F(0.0005)          # normal, :109-139
        es = [v3(1, -1, -1), v3(-1, -1, 1), v3(-1, 1, -1), v3(1, 1, 1)]
        terms = [e * self.sdfs(p, e, k)[1][0] for e in es]
        n = normalize(((terms[0] + terms[1]) + terms[2]) + terms[3])
        o = self.objects[obj_idx]
        mat_idx = 0 if o.kind in (self.pkg.RT_UNION, self.pkg.RT_SUBSTRACTION) else o.material_index
        m = self.materials[mat_idx]
        albedo = rh[1]
        if m.texture >= 0:                 # tri-planar, :218-232 and geometry.rs:56-68
            self.textured_hits += 1
            n1 = normalize(mat_mul_vec4(o.inv_transform[:], n[0], n[1], n[2], 0.0)[:3])
            hit1 = rh[2] + rh[3] * t
            tex = self.tex[m.texture]
            s = F(0.5)
            tx = from_uv(tex, hit1[1] * s, hit1[2] * s)
            ty = from_uv(tex, hit1[0] * s, hit1[2] * s)
            tz = from_uv(tex, hit1[0] * s, hit1[1] * s)
            bw = powf3(np.abs(n1), 0.8)
            bw = bw / (F(bw[0] + bw[1]) + bw[2])
            albedo = (tx * bw[0] + ty * bw[1]) + tz * bw[2]
        l_acc = v3(0, 0, 0)                # light, :160-179
        w = Whitted(self.pkg, self.d)
        for l in self.lights:
            phong = w.blinn_phong(rd, p, n, l, albedo, m)
            _, dist = w.light_dir_dist(l, p)
            l_acc = l_acc + (phong / (dist * dist)) * v3(*l.albedo[:]) * F(l.intensity)
        color = powf3(l_acc, 0.4545)
        occ, sca = F(0.0), F(1.0)          # occlusion, :141-158
        for i in range(3):
            hr = F(0.02) + F(0.025) * F(i * i)
            dd = self.sdfs(p, n, hr)[1][0]
            occ = occ + (-(dd - hr)) * sca
            sca = sca * F(0.85)
        color = color * (F(1.0) - clamp01(occ))
        tt = clamp01((t - F(1.0)) / (F(20.0) - F(1.0)))    # smooth_step(1, 20, distance), geometry.rs:29-32
        return color * (F(1.0) - tt * tt * (F(3.0) - F(2.0) * tt))


def to_rgba(p):                            # src/renderer.rs:29-36, :67-68
    c = np.clip(np.append(p, F(1.0)), F(0.0), F(1.0)).astype(np.float32)
    return np.array([int(c[0] * F(255.0)), int(c[1] * F(255.0)), int(c[2] * F(255.0)), min(int(c[3] + F(255.0)), 255)], dtype=np.uint8)
