/*
 * rt_types.h -- plain-old-data scene model consumed by the hip_trace C ABI.
 *
 * These records are byte-for-byte the records the reference renderer keeps on
 * the host and uploads verbatim to its device (reference: src/utils/vec3.h:10-17,
 * src/material.h:6-14, src/plane.h:8-12, src/sphere.h:8-12, src/triangle.h:8-11,
 * src/pointlight.h:6-10, src/ray.h:7-10, src/camera.h:7-15, src/scene.h:14-36,
 * src/octree.h:10-34).  The type and field names are the
 * reference's so that host code written against its headers compiles unchanged.
 *
 * Each block is skipped when the reference header that defines the same type
 * has already been included (its include guard is tested, never defined here),
 * so a translation unit of the reference renderer includes its own headers
 * first and this file afterwards; stand-alone code includes only this file.
 * The layout asserted at the bottom holds either way.
 */
#ifndef RT_TYPES_H
#define RT_TYPES_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
#define RT_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define RT_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

#ifndef RAYTRACER_VEC3_H
typedef union {
    struct { float x, y, z; };
    struct { float r, g, b; };
} Vec3;
#endif

#ifndef RAYTRACER_RAY_H
typedef struct {
    Vec3 origin;
    Vec3 direction;
} Ray;
#endif

#ifndef RAYTRACER_MATERIAL_H
typedef struct {
    float reflectionIndex;   /* 0 = matte, 1 = perfect mirror                   */
    float refractionIndex;   /* >0 = dielectric                                 */
    float ambientWeight;     /* the three weights and the exponent are only     */
    float diffuseWeight;     /* read by the device shading model of the         */
    float specularWeight;    /* reference, not by its CPU model                 */
    float specularExponent;
    Vec3 color;
} Material;
#endif

#ifndef RAYTRACER_PLANE_H
typedef struct {
    uint32_t materialIndex;
    Vec3 normal;
    float distanceFromOrigin;
} Plane;
#endif

#ifndef RAYTRACER_SPHERE_H
typedef struct {
    uint32_t materialIndex;
    Vec3 position;
    float radius;
} Sphere;
#endif

#ifndef RAYTRACER_TRIANGLE_H
typedef struct {
    uint32_t materialIndex;
    Vec3 v0, v1, v2;
} Triangle;
#endif

#ifndef RAYTRACER_POINTLIGHT_H
typedef struct {
    Vec3 position;
    Vec3 emissionColor;
    float strength;
} PointLight;
#endif

#ifndef RAYTRACER_CAMERA_H
typedef struct {
    Vec3 position;
    Vec3 x, y, z;                 /* orthonormal basis, z points away from lookAt */
    Vec3 lookAt;
    uint32_t width, height;       /* pixels */
    float FOV, aspectRatio, focalLength, apertureSize;
    Vec3 renderTargetCenter;
    float renderTargetWidth, renderTargetHeight, renderTargetDistance;
} Camera;
#endif

#ifndef RAYTRACER_OBJECT_H
typedef struct {
    uint32_t capacity;
    uint32_t triangleCount;
    Triangle* triangles;
} Object;
#endif

#if !defined(RAYTRACER_IMAGE_H) && !defined(RT_IMAGE_DECLARED)
#define RT_IMAGE_DECLARED
/* reference: src/utils/image.h:8-12 -- pixels 0xAABBGGRR, top row first */
typedef struct {
    uint32_t width, height;
    uint32_t* buffer;
    uint32_t bufferSize;
} Image;
#endif

#ifndef RAYTRACER_SCENE_H
typedef struct {
    Camera* camera;

    uint32_t materialCapacity;
    uint32_t materialCount;
    Material* materials;          /* materials[0] is the background: "no hit"    */

    uint32_t planeCapacity;
    uint32_t planeCount;
    Plane* planes;

    uint32_t sphereCapacity;
    uint32_t sphereCount;
    Sphere* spheres;

    uint32_t triangleCapacity;
    uint32_t triangleCount;
    Triangle* triangles;

    uint32_t pointLightCapacity;
    uint32_t pointLightCount;
    PointLight* pointLights;
} Scene;
#endif

#ifndef RAYTRACER_OCTREE_H
#define MIN_ELEMENTS_PER_NODE 8
#define NODE_INDEX_UNDEF -1
typedef struct {
    Vec3 bottomLeftFrontCorner;   /* minimum corner */
    Vec3 topRightBackCorner;      /* maximum corner */
} BoundingBox;

typedef struct {
    BoundingBox boundingBox;
    uint32_t sphereIndexOffset;   /* leaf only: run of sphere ids in indexes[]   */
    uint32_t sphereIndexCount;
    uint32_t triangleIndexOffset; /* leaf only: run of triangle ids in indexes[] */
    uint32_t triangleIndexCount;
    int32_t childNodeIndexes[8];  /* leaf iff [0] == NODE_INDEX_UNDEF; otherwise
                                     eight consecutive node ids                  */
} OctreeNode;

typedef struct {
    OctreeNode* nodes;
    uint32_t nodeCount;
    uint32_t nodeCapacity;
    uint32_t* indexes;
    uint32_t indexCount;
    uint32_t indexCapacity;
} Octree;
#endif

RT_STATIC_ASSERT(sizeof(Vec3) == 12, "Vec3 layout");
RT_STATIC_ASSERT(sizeof(Ray) == 24, "Ray layout");
RT_STATIC_ASSERT(sizeof(Material) == 36, "Material layout");
RT_STATIC_ASSERT(sizeof(Plane) == 20, "Plane layout");
RT_STATIC_ASSERT(sizeof(Sphere) == 20, "Sphere layout");
RT_STATIC_ASSERT(sizeof(Triangle) == 40, "Triangle layout");
RT_STATIC_ASSERT(sizeof(PointLight) == 28, "PointLight layout");
RT_STATIC_ASSERT(sizeof(Camera) == 108, "Camera layout");
RT_STATIC_ASSERT(offsetof(Camera, width) == 60, "Camera.width offset");
RT_STATIC_ASSERT(offsetof(Camera, renderTargetCenter) == 84, "Camera.renderTargetCenter offset");
RT_STATIC_ASSERT(sizeof(BoundingBox) == 24, "BoundingBox layout");
RT_STATIC_ASSERT(sizeof(OctreeNode) == 72, "OctreeNode layout");
RT_STATIC_ASSERT(offsetof(OctreeNode, childNodeIndexes) == 40, "OctreeNode.child offset");
RT_STATIC_ASSERT(sizeof(Scene) == 88, "Scene layout (LP64)");
RT_STATIC_ASSERT(sizeof(Octree) == 32, "Octree layout (LP64)");

#endif /* RT_TYPES_H */
