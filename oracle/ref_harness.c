/*
 * ref_harness.c -- drives the reference renderer's own CPU code, compiled unmodified
 * from /root/reference/src, so that the restatement in cpu_trace.c can be checked
 * against it and golden frames can be generated from it.
 *
 * TEST INFRASTRUCTURE ONLY; built into oracle/_ref/libref.so by oracle/Makefile when
 * the reference tree is present.  Nothing from the reference is copied: its
 * raytracer.c is pulled in with #include so that its file-static intersectors are
 * reachable, and its other modules are compiled from where they lie.
 *
 * What this file adds, because the reference has no such code on the CPU side:
 *   - random_bilateral()/random_unilateral() returning 0 instead of rand()
 *     (replaces src/utils/random.c at link time: zero shadow jitter, thread-safe);
 *   - a per-pixel loop with the primary-ray construction of the reference's device
 *     kernel (src/kernel.cl:804-843), written with the reference's vec3 functions;
 *   - a recording closest-hit query that calls the reference's intersectors in its
 *     loop order (src/raytracer.c:163-209) and reports which primitive won.
 */
#include <float.h>
#include <string.h>

#include "raytracer.c"   /* reference: src/raytracer.c, via -I/root/reference/src */
#include "octree.c"      /* reference: src/octree.c, for its file-static octree_buildNode */
#include "camera.h"

/* The link seam for src/utils/random.c.  Zero mode: every draw is 0.  Hashed mode: the n-th
 * draw made while one pixel is shaded belongs to shading event n / (3 * lights), light
 * (n / 3) % lights, component n % 3 (src/raytracer.c:274-279 draws three values per light, lights
 * in order); the caller supplies the tree position of each event in shading order, and the draw
 * is the hash of (pixel, position, light, component) mapped as src/utils/random.c:5-11 maps
 * rand(). */
static _Thread_local struct {
    int hashed;
    uint32_t pixel, lights, events, draws;
    const uint32_t* positions;
    int overrun;
} seam;

static uint32_t seam_hash(uint32_t pixel, uint32_t path, uint32_t light, uint32_t component) {
    uint32_t v = (pixel * 0x9E3779B1u) ^ (path * 0x85EBCA77u) ^ (light * 0xC2B2AE3Du) ^ (component * 0x27D4EB2Fu);
    v ^= v >> 16; v *= 0x85EBCA6Bu;
    v ^= v >> 13; v *= 0xC2B2AE35u;
    v ^= v >> 16;
    return v;
}

float random_unilateral(void) {
    if (!seam.hashed) return 0.5f;
    const uint32_t n = seam.draws++;
    const uint32_t event = n / (3u * seam.lights);
    if (event >= seam.events) { seam.overrun = 1; return 0.5f; }
    const uint32_t h = seam_hash(seam.pixel, seam.positions[event], (n / 3u) % seam.lights, n % 3u);
    return (float)(h >> 8) / 16777216.0f;
}
float random_bilateral(void) {
    if (!seam.hashed) return 0.0f;
    return -1.0f + 2.0f * random_unilateral();
}

typedef struct { uint32_t x0, y0, stepX, stepY, countX, countY; } RefWindow;

static Ray ref_primary_ray(const Camera* camera, uint32_t x, uint32_t y) {
    /* one sample per pixel: i = j = 0, deltaX = pixelWidth, deltaY = pixelHeight */
    uint32_t i = 0, j = 0;
    float pixelWidth = 1.0f / (float)(camera->width);
    float pixelHeight = 1.0f / (float)(camera->height);
    float deltaX = pixelWidth, deltaY = pixelHeight;
    float PosX = -1.0f + 2.0f * ((float)x / (camera->width));
    float PosY = -1.0f + 2.0f * ((float)y / (camera->height));
    Vec3 OffsetY = vec3_mul(camera->y, (PosY - pixelHeight + j * deltaY) * camera->renderTargetHeight / 2.0f);
    Vec3 OffsetX = vec3_mul(camera->x, (PosX - pixelWidth + i * deltaX) * camera->renderTargetWidth / 2.0f);
    Vec3 renderTargetPos = vec3_sub(vec3_add(camera->renderTargetCenter, OffsetX), OffsetY);
    Ray ray = { camera->position, vec3_norm(vec3_sub(renderTargetPos, camera->position)) };
    Vec3 focalPoint = vec3_add(ray.origin, vec3_mul(ray.direction, camera->focalLength));
    Vec3 randomOffset;
    randomOffset.x = random_bilateral() / 2.0f;
    randomOffset.y = random_bilateral() / 2.0f;
    randomOffset.z = random_bilateral() / 2.0f;
    ray.origin = vec3_add(ray.origin, vec3_mul(randomOffset, camera->apertureSize));
    ray.direction = vec3_norm(vec3_sub(focalPoint, ray.origin));
    return ray;
}

static void ref_closest(Scene* scene, Ray* ray, float* tOut, uint32_t* idOut, uint32_t* materialOut) {
    float best = FLT_MAX;
    uint32_t id = 0xFFFFFFFFu, material = 0;
    Vec3 normal;
    for (uint32_t k = 0; k < scene->planeCount; k++) {
        float t = FLT_MAX;
        if (raytracer_intersectPlane(&scene->planes[k], ray, &t, &normal) && t < best) {
            best = t; id = k; material = scene->planes[k].materialIndex;
        }
    }
    for (uint32_t k = 0; k < scene->sphereCount; k++) {
        float t = FLT_MAX;
        if (raytracer_intersectSphere(&scene->spheres[k], ray, &t, &normal) && t < best) {
            best = t; id = scene->planeCount + k; material = scene->spheres[k].materialIndex;
        }
    }
    for (uint32_t k = 0; k < scene->triangleCount; k++) {
        float t = FLT_MAX;
        if (raytracer_intersectTriangle(&scene->triangles[k], ray, &t, &normal) && t < best) {
            best = t; id = scene->planeCount + scene->sphereCount + k; material = scene->triangles[k].materialIndex;
        }
    }
    *tOut = best; *idOut = id; *materialOut = material;
}

/* Renders the window with raytracer_raycast.  `scene->camera` is pointed at `camera` for the
 * duration of the call because the reference shader reads the eye position from it. */
int ref_render(Scene* scene, Camera* camera, uint32_t maxDepth, const RefWindow* window,
               float* rgb, uint8_t* hitMask, uint32_t* primitiveId, float* hitDistance) {
    Camera* saved = scene->camera;
    scene->camera = camera;
    const uint32_t nx = window->countX, ny = window->countY;
#pragma omp parallel for schedule(dynamic, 1)
    for (uint32_t j = 0; j < ny; j++) {
        for (uint32_t i = 0; i < nx; i++) {
            uint32_t x = window->x0 + i * window->stepX, y = window->y0 + j * window->stepY;
            size_t at = (size_t)j * nx + i;
            Ray ray = ref_primary_ray(camera, x, y);
            if (rgb) {
                Vec3 color = raytracer_raycast(scene, &ray, maxDepth);
                Vec3 sum = (Vec3){0};
                sum = vec3_add(sum, vec3_mul(color, 1.0f));
                rgb[3 * at] = sum.r; rgb[3 * at + 1] = sum.g; rgb[3 * at + 2] = sum.b;
            }
            if (hitMask || primitiveId || hitDistance) {
                float t; uint32_t id, material;
                ref_closest(scene, &ray, &t, &id, &material);
                if (hitMask) hitMask[at] = material != 0;
                if (primitiveId) primitiveId[at] = id;
                if (hitDistance) hitDistance[at] = t;
            }
        }
    }
    scene->camera = saved;
    return 0;
}

/* raytracer_raycast with hashed light jitter.  offsets/positions: per pixel of the window, the tree
 * positions of its shading events in the reference's shading order (from the restatement, which
 * must then agree with the reference on the SHAPE of every ray tree too: a pixel whose draw count
 * differs from 3 * lights * events is counted in *disagreements). */
int ref_render_jittered(Scene* scene, Camera* camera, uint32_t maxDepth, const RefWindow* window,
                        const uint32_t* offsets, const uint32_t* positions, float* rgb, uint32_t* disagreements) {
    Camera* saved = scene->camera;
    scene->camera = camera;
    const uint32_t nx = window->countX, ny = window->countY;
    uint32_t bad = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : bad)
    for (uint32_t j = 0; j < ny; j++) {
        for (uint32_t i = 0; i < nx; i++) {
            uint32_t x = window->x0 + i * window->stepX, y = window->y0 + j * window->stepY;
            size_t at = (size_t)j * nx + i;
            seam.hashed = 0;                       /* the aperture draws stay 0 */
            Ray ray = ref_primary_ray(camera, x, y);
            seam.hashed = 1;
            seam.pixel = y * camera->width + x;
            seam.lights = scene->pointLightCount;
            seam.positions = positions + offsets[at];
            seam.events = offsets[at + 1] - offsets[at];
            seam.draws = 0;
            seam.overrun = 0;
            Vec3 color = raytracer_raycast(scene, &ray, maxDepth);
            if (seam.overrun || seam.draws != 3u * seam.lights * seam.events) bad++;
            seam.hashed = 0;
            Vec3 sum = (Vec3){0};
            sum = vec3_add(sum, vec3_mul(color, 1.0f));
            rgb[3 * at] = sum.r; rgb[3 * at + 1] = sum.g; rgb[3 * at + 2] = sum.b;
        }
    }
    scene->camera = saved;
    if (disagreements) *disagreements = bad;
    return 0;
}

/* Thin exports of the reference's own functions. */
Scene* ref_scene_init(uint32_t width, uint32_t height) { return scene_init(width, height); }
void ref_scene_destroy(Scene* scene) { scene_destroy(scene); }
Octree* ref_octree_build(Scene* scene) { return octree_buildFromScene(scene); }

/* octree_buildFromScene as the reference wrote it (src/octree.c:384-419) starts with room for
 * one node and, per octree_buildNode call, doubles the node array once if nodeCount + 1 does
 * not fit (src/octree.c:207-210) -- but a split then takes eight ids at once
 * (src/octree.c:270-277), so while the tree is still small a child can be written past the
 * end of the array (root -> capacity 2, first child -> 4, its first child has id 9 -> capacity
 * 8 < 10).  That is a heap overflow whose outcome depends on the allocator.  This entry point
 * runs the same steps around the reference's unmodified octree_buildNode with the node array
 * sized up front, which is the result the reference intends. */
Octree* ref_octree_build_safe(Scene* scene, uint32_t nodeCapacity) {
    Octree* octree = malloc(sizeof(Octree));
    if (!octree) return NULL;
    octree->nodeCapacity = nodeCapacity < 16 ? 16 : nodeCapacity;
    octree->nodeCount = 0;
    octree->nodes = malloc(sizeof(OctreeNode) * octree->nodeCapacity);
    octree->indexCapacity = 2000;
    octree->indexCount = 0;
    octree->indexes = malloc(sizeof(uint32_t) * octree->indexCapacity);
    BoundingBox rootBoundingBox = octree_calculateRootBoundingBox(scene);
    uint32_t* sphereIndexes = malloc(sizeof(uint32_t) * (scene->sphereCount + 1));
    for (uint32_t i = 0; i < scene->sphereCount; i++) sphereIndexes[i] = i;
    uint32_t* triangleIndexes = malloc(sizeof(uint32_t) * (scene->triangleCount + 1));
    for (uint32_t i = 0; i < scene->triangleCount; i++) triangleIndexes[i] = i;
    int32_t rootId = octree->nodeCount++;
    octree_buildNode(octree, scene, rootId, sphereIndexes, scene->sphereCount, triangleIndexes,
                     scene->triangleCount, rootBoundingBox);
    free(sphereIndexes);
    free(triangleIndexes);
    octree_shrinkToFit(octree);
    return octree;
}
void ref_octree_destroy(Octree* octree) { octree_destroy(octree); }
void ref_camera_setup(Camera* camera) { camera_setup(camera); }
Camera* ref_camera_create(Vec3 position, Vec3 lookAt, uint32_t width, uint32_t height, float FOV, float aperture) {
    return camera_create(position, lookAt, width, height, FOV, aperture);
}
void ref_move_camera(Camera* camera, int upDown, int side, int frontal) { move_camera(camera, upDown, side, frontal); }
Vec3 ref_raycast(Scene* scene, Ray* ray, uint32_t maxDepth) { return raytracer_raycast(scene, ray, maxDepth); }
Ray ref_primary(const Camera* camera, uint32_t x, uint32_t y) { return ref_primary_ray(camera, x, y); }

int ref_intersect_plane(Plane* plane, Ray* ray, float* t) { Vec3 n; return raytracer_intersectPlane(plane, ray, t, &n); }
int ref_intersect_sphere(Sphere* sphere, Ray* ray, float* t) { Vec3 n; return raytracer_intersectSphere(sphere, ray, t, &n); }
int ref_intersect_triangle(Triangle* triangle, Ray* ray, float* t) { Vec3 n; return raytracer_intersectTriangle(triangle, ray, t, &n); }
float ref_fresnel(Vec3 dir, Vec3 normal, float index) { return raytracer_fresnel(dir, normal, index); }
Vec3 ref_refract(Vec3 dir, Vec3 normal, float index) { return raytracer_refract(dir, normal, index); }
Vec3 ref_reflect(Vec3 dir, Vec3 normal) { return vec3_reflect(dir, normal); }
float ref_clamp(float in, float lo, float hi) { return math_clamp(in, lo, hi); }
