/* light_grid.h -- per-light direction grids for occlusion queries (built at upload, see light_grid.hip). */
#ifndef RT_LIGHT_GRID_H
#define RT_LIGHT_GRID_H

#include <cstddef>
#include <cstdint>
#include <string>

#include "device_layout.h"

#define RT_GRID_SLOP 1e-5              /* scene sizes: how far beside a primitive a ray's exact line may pass and still count */
#define RT_GRID_MIN_DILATION 2e-6      /* least widening of a projection, in the [-1, 1] coordinates of a cube face */
#define RT_GRID_REACH 4.0f             /* scene sizes: rays that start farther from the origin do not use the grids */
#define RT_GRID_NEAR 0.01f             /* primitives within this many scene sizes of a light are tested for every ray */
#define RT_GRID_MAX_BUCKETS 4096u      /* a primitive seen in more buckets than this and than 1/RT_GRID_MAX_SHARE of */
#define RT_GRID_MAX_SHARE 16u          /* the light's buckets is tested for every ray instead of being listed       */
#define RT_GRID_MAX_PER_PRIMITIVE (1u << 18)   /* ... and so is one seen in more buckets than this at any resolution: the work
                                                  of entering one primitive into one grid stays bounded                     */
#define RT_GRID_ALWAYS 32u             /* capacity of a light's "tested for every ray" list; beyond it the light has no grid */
#define RT_GRID_MIN_RESOLUTION 16u     /* buckets per cube-face edge, at least ...                                  */
#define RT_GRID_MAX_RESOLUTION 4096u   /* ... and at most                                                           */
#define RT_GRID_MEMORY_SHARE 0.25      /* of the device memory free at upload: what the grids may take, and no more */
#define RT_GRID_SCENE_MULTIPLE 16.0    /* than this many times the scene's own device copy plus 1 GiB               */
#define RT_GRID_INFO_WORDS 8u          /* per light: has a grid, entries of its always list, first reference (lo, hi), distance levels
                                          per unit length (float bits); three spare words */
#define RT_GRID_DENSITY 25.0           /* buckets per primitive aimed at, over the whole cube: on the 1M-triangle frame with 8 lights
                                          12 / 16 / 20 / 25 / 32 give 2.28 / 2.19 / 2.14 / 2.08 / 2.08 ms for 0.46 / 0.57 / 0.67 /
                                          0.80 / 0.97 GB */
#define RT_GRID_MAX_EXPECTED 30.0      /* a light whose rays would meet longer lists than this keeps walking the hierarchy */
#define RT_GRID_MAX_REFERENCES 0xF0000000ull   /* one light's references are indexed with 32 bits */

struct LightGridInput {        /* device pointers of the uploaded scene */
    const QNode* nodes;         /* a hierarchy that stores every primitive once: triangles are referred to by their position
                                   in its leaf arrays.  NULL: by their own number */
    uint32_t nodeCount;
    const uint32_t* leafIndex;
    uint32_t entryCount;        /* spheres + triangles */
    uint32_t sphereCount, triangleCount;
    const Float4* triangleCorner;
    const Float4* sphere;
    const Float4* light;
    uint32_t lightCount;
    float extent;               /* largest |coordinate| of the scene, at least 32 */
    float sphereGrow;           /* added to a sphere's squared radius for its bounds */
    size_t sceneBytes;          /* size of the scene's device copy without the grids */
};

struct LightGridBuild {
    LightGrid view{};           /* what the kernels read; all NULL = no grid */
    size_t bytes = 0;
    size_t peakBytes = 0;       /* what the build held at its largest (the grids plus a 4-byte cursor per bucket) */
    uint64_t referenceCount = 0;
    uint32_t enabledLights = 0;
};

/* Builds the grids on the current device (synchronous).  Returns 0 -- with an empty `out` when a grid is not worth
 * having, would not fit its share of the memory or an allocation failed -- or -1 with `error` set when another HIP
 * call failed. */
int rt_light_grid_build(const LightGridInput& in, LightGridBuild& out, std::string& error);
void rt_light_grid_release(LightGridBuild& grid);

#endif
