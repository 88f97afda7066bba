/* flatten.h -- host-side result of repacking a scene for the device (see device_layout.h). */
#ifndef RT_FLATTEN_H
#define RT_FLATTEN_H

#include <string>
#include <vector>

#include "hip_trace.h"
#include "device_layout.h"

struct FlatScene {
    std::vector<WideNode> nodes;     /* float boxes: intermediate */
    std::vector<QNode> qnodes;       /* what is uploaded */
    std::vector<uint32_t> leafIndex;
    std::vector<Float4> trianglePlane, triangleCorner, sphere, plane, material, light;   /* triangleCorner: 3 per
                                       triangle, staged on the device only while leaf records are expanded */
    std::vector<uint32_t> sphereMaterial, planeMaterial, triangleMaterial;
    uint32_t planeCount = 0, sphereCount = 0, triangleCount = 0, lightCount = 0, materialCount = 0;
    uint32_t treeLevels = 0;   /* inner levels on the longest root-to-leaf path (>= 1) */
    float extent = 32.0f;      /* largest |coordinate| of a sphere, triangle or plane offset, at least 32 */
    uint32_t leafCount = 0;    /* non-empty leaves */
    float sphereGrow = 0.0f;   /* added to every sphere's squared radius for its bounds (the private hierarchy) */
};

/* Sphere boxes are built to cover the reference arithmetic for ray origins within this many scene sizes (FlatScene::extent)
 * of the origin of coordinates. */
#define RT_SPHERE_REACH 4.0f
/* The largest distance D between a ray origin within RT_SPHERE_REACH scene sizes of the origin of coordinates (a test on
 * the largest coordinate, so a cube: its corner lies sqrt(3) * RT_SPHERE_REACH sizes out) and the centre of a sphere of
 * the scene (up to sqrt(3) sizes out the other way), in scene sizes. */
#define RT_SPHERE_DISTANCE (1.7320508f * (RT_SPHERE_REACH + 1.0f))

/* accel: a HipTraceAccel value; `tree` is read only for HIP_TRACE_ACCEL_CALLER_OCTREE.
 * Returns 0, or a negative HipTraceStatus with `error` set. */
int rt_flatten(const Scene* scene, const Octree* tree, uint32_t accel, FlatScene& out, std::string& error);


/* Walks `flat` from the root and fills the counters of `report` (see hip_trace.h). */
void rt_check_hierarchy(const Scene* scene, const FlatScene& flat, uint32_t accel, HipTraceHierarchyReport& report);

#endif
