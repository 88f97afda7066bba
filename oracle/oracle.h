/*
 * oracle.h -- interface of the CPU parity oracle (test infrastructure, not product).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stddef.h>

#include "rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_NO_PRIMITIVE 0xFFFFFFFFu

enum {
    /* Visit every node whose box the ray meets, children pushed 0..7, no distance pruning
     * (reference: src/kernel.cl:552-601).  Used for work counts. */
    ORACLE_WALK_REFERENCE = 0,
    /* Same walk, but skip a node whose box is entered beyond the current best distance.
     * Same result, fewer visits.  Without an octree (brute force): the triangle loop runs block-wise
     * over a struct-of-arrays copy of the scene, every triangle's normal computed once per frame
     * instead of once per test -- the same fp32 operations, same bits, an order of magnitude less time;
     * ORACLE_WALK_REFERENCE keeps the loops exactly as src/raytracer.c:195-209 has them. */
    ORACLE_WALK_PRUNED = 1
};

enum {
    ORACLE_JITTER_ZERO = 0,    /* random_bilateral() == 0: hard shadows */
    /* random_bilateral() drawn from a counter-based hash keyed by (pixel, position of the shading
     * point in the ray tree, light, component): the reference's unit-sphere light jitter
     * (src/raytracer.c:278-279) made deterministic and order-independent.  See oracle_jitter_hash. */
    ORACLE_JITTER_HASHED = 1
};

enum {
    ORACLE_SHADING_CPU = 0,     /* src/raytracer.c:274-310: one shadow ray per light, exponent 64 */
    /* src/kernel.cl:671-714: four jittered shadow rays per light, ambient/diffuse/specular weights
     * and the material's exponent, attenuation 1 / (1 + 4 pi d^2), and that kernel's way of
     * averaging the four samples.  The reference can only run this on an OpenCL device, so this
     * model is checked against nothing but itself: PARITY UNPINNED.  Its random draws are hashed
     * (pixel, sub-pixel sample, tree position, light, shadow sample, component) instead of the
     * kernel's per-pixel xorshift stream, and mapped to [0,1) in 2^17 steps as src/kernel.cl:85-87. */
    ORACLE_SHADING_KERNEL = 1
};

typedef struct {
    uint32_t maxDepth;   /* maxRecursionDepth of raytracer_raycast */
    uint32_t walk;       /* ORACLE_WALK_*, octree mode only */
    float boxPad;        /* widening of every octree box, 0 = the reference's slab test */
    int32_t threads;     /* <= 0: all OpenMP threads */
    uint32_t jitter;     /* ORACLE_JITTER_* */
    uint32_t shadingModel; /* ORACLE_SHADING_* */
    uint32_t raysPerPixel; /* 0 or 1: one primary ray; n > 1: the sub-pixel grid of src/gpu.c:446-465 */
} OracleOptions;

/* Pixels (x0 + i*stepX, y0 + j*stepY), i < countX, j < countY; outputs are countX*countY
 * arrays in that order.  The whole frame is {0, 0, 1, 1, width, height}. */
typedef struct {
    uint32_t x0, y0, stepX, stepY, countX, countY;
} OracleWindow;

typedef struct {
    uint64_t primary, shadow, reflection, refraction;   /* rays traced */
    uint64_t boxTests, nodesVisited;                     /* octree mode */
    uint64_t planeTests, sphereTests, triangleTests;     /* primitive tests started */
    uint64_t triangleDistance;                           /* ... that got past the parallel test */
    uint64_t triangleInside;                             /* ... that ran the edge tests */
    uint64_t lightEvaluations;                           /* unoccluded light samples shaded */
} OracleCounters;

typedef struct {
    uint32_t* rgba8;        /* each nullable */
    float*    rgb;          /* 3 floats per pixel, before clamping */
    uint8_t*  hitMask;
    uint32_t* primitiveId;
    float*    hitDistance;
    OracleCounters counters; /* out */
} OracleFrame;

/* octree == NULL: brute force over all primitives, the reference CPU algorithm. */
int oracle_render(const Scene* scene, const Octree* octree, const Camera* camera,
                  const OracleOptions* options, const OracleWindow* window, OracleFrame* out);

/* Position of a shading point in a pixel's ray tree: the primary hit is 1, the hit of the ray
 * reflected at node p is 2p, of the ray refracted at p is 2p + 1.
 * The 32 bits behind one random_bilateral() draw; the draw is -1 + 2 * ((h >> 8) / 2^24), the
 * expression of src/utils/random.c:5-11 with (h >> 8) / 2^24 in place of rand() / RAND_MAX. */
uint32_t oracle_jitter_hash(uint32_t pixel, uint32_t path, uint32_t light, uint32_t component);

/* For every pixel of the window, the tree positions of its shading points (hits with a
 * material) in the order the reference's recursion shades them: refraction subtree, reflection
 * subtree, then the node itself (src/raytracer.c:235-310).  offsets has countX*countY + 1
 * entries; paths receives up to `capacity` entries.  Returns the total number of shading points,
 * or -1 on bad arguments (nothing beyond `capacity` is written). */
long oracle_shading_order(const Scene* scene, const Octree* octree, const Camera* camera,
                          const OracleOptions* options, const OracleWindow* window,
                          uint32_t* offsets, uint32_t* paths, size_t capacity);

uint32_t oracle_pack_rgba8(float r, float g, float b);

int oracle_intersect_plane(const Plane* plane, const Ray* ray, float* t);
int oracle_intersect_sphere(const Sphere* sphere, const Ray* ray, float* t);
int oracle_intersect_triangle(const Triangle* triangle, const Ray* ray, float* t);
int oracle_intersect_box(const BoundingBox* box, const Ray* ray);
float oracle_fresnel(Vec3 dir, Vec3 normal, float refractionIndex);
Vec3 oracle_refract(Vec3 dir, Vec3 normal, float refractionIndex);
Vec3 oracle_reflect(Vec3 dir, Vec3 normal);
Ray oracle_primary_ray(const Camera* camera, uint32_t x, uint32_t y);

#ifdef __cplusplus
}
#endif
#endif
