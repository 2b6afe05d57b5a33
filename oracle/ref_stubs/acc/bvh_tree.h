// Stand-in for rayint's acc/bvh_tree.h (an un-vendored download).  The reference uses the tree purely as a boolean
// any-hit query (calculate_data_costs.cpp:200-209); the intersection arithmetic is DEFINED by the oracle, so this class
// forwards every query -- with the origin, direction, tmin and tmax exactly as the reference's code computed them -- to
// a hook the test installs (the oracle's orc_ray_hit).  What oracle/_ref pins is the reference's ray SET-UP and the
// order / early exit of the three vertex rays, not an intersection routine.  Test infrastructure only.
#ifndef MVS_REF_STUB_ACC_BVH_TREE_H
#define MVS_REF_STUB_ACC_BVH_TREE_H
#include <cstdint>
#include <stdexcept>
#include <vector>
namespace acc {
struct RayHook {
    int (*fn)(const void* bvh, const void* mesh, const float* origin, const float* dir, float tmin, float tmax, int brute);
    const void* bvh;
    const void* mesh;
    int brute;
    std::uint64_t calls;
};
inline RayHook& ray_hook() { static RayHook h = {nullptr, nullptr, nullptr, 0, 0}; return h; }
template <typename IdxType, typename Vec3fType>
class BVHTree {
public:
    struct Ray { Vec3fType origin; Vec3fType dir; float tmin; float tmax; };
    struct Hit { float t; IdxType idx; Vec3fType bcoords; };
    BVHTree(std::vector<IdxType> const&, std::vector<Vec3fType> const&) {}
    bool intersect(Ray const& ray, Hit*) const {
        RayHook& h = ray_hook();
        if (!h.fn) throw std::runtime_error("oracle/_ref: no ray hook installed");
        __atomic_fetch_add(&h.calls, (std::uint64_t)1, __ATOMIC_RELAXED);   // (the OpenMP build of oracle/_ref calls this from many threads)
        const float o[3] = {ray.origin[0], ray.origin[1], ray.origin[2]}, d[3] = {ray.dir[0], ray.dir[1], ray.dir[2]};
        return h.fn(h.bvh, h.mesh, o, d, ray.tmin, ray.tmax, h.brute) != 0;
    }
};
}  // namespace acc
#endif
