/*
 * rt_scene.c -- scene container, camera and the default scene.
 *
 * Stand-alone equivalents of the reference's scene and camera modules
 * (reference: src/scene.c:7-290, src/camera.c:9-58): same records, same growth
 * behaviour visible to callers (counts, element order), same camera arithmetic.
 */
#include <stdlib.h>
#include <string.h>

#include "rt_host_internal.h"
#include "rt_vecmath.h"

enum { RT_INITIAL_CAPACITY = 200 };

/* The adders return nothing, as the reference's do (reference: src/scene.h:38-44), so an element that could not be
 * stored for want of memory is counted here instead of vanishing without a trace. */
static uint32_t droppedElements;
uint32_t rt_scene_droppedElements(void) { return __atomic_load_n(&droppedElements, __ATOMIC_RELAXED); }
static void rt_count_dropped(void) { __atomic_add_fetch(&droppedElements, 1u, __ATOMIC_RELAXED); }

/* Grows *items (element size `size`) so that one more element fits. */
/* Returns NULL (array untouched, *capacity unchanged) when memory runs out or the count would overflow. */
static void* rt_reserve_one(void* items, uint32_t* capacity, uint32_t count, size_t size) {
    if (items && count + 1 <= *capacity && count != UINT32_MAX) return items;
    if (count == UINT32_MAX) return NULL;
    uint32_t grown = *capacity ? (*capacity > UINT32_MAX / 2u ? UINT32_MAX : *capacity * 2u) : 1u;
    if (grown <= count) grown = count + 1;
    void* moved = realloc(items, size * (size_t)grown);
    if (!moved) return NULL;
    *capacity = grown;
    return moved;
}

static void* rt_trim(void* items, uint32_t* capacity, uint32_t count, size_t size) {
    if (*capacity <= count) return items;
    *capacity = count;
    if (count == 0) { free(items); return NULL; }
    void* moved = realloc(items, size * count);
    return moved ? moved : items;
}

Scene* rt_scene_create(void) {
    Scene* scene = (Scene*)calloc(1, sizeof(Scene));
    if (!scene) return NULL;
    scene->materialCapacity = scene->planeCapacity = scene->sphereCapacity =
        scene->triangleCapacity = scene->pointLightCapacity = RT_INITIAL_CAPACITY;
    scene->materials = (Material*)malloc(sizeof(Material) * RT_INITIAL_CAPACITY);
    scene->planes = (Plane*)malloc(sizeof(Plane) * RT_INITIAL_CAPACITY);
    scene->spheres = (Sphere*)malloc(sizeof(Sphere) * RT_INITIAL_CAPACITY);
    scene->triangles = (Triangle*)malloc(sizeof(Triangle) * RT_INITIAL_CAPACITY);
    scene->pointLights = (PointLight*)malloc(sizeof(PointLight) * RT_INITIAL_CAPACITY);
    if (!scene->materials || !scene->planes || !scene->spheres || !scene->triangles || !scene->pointLights) {
        rt_scene_destroy(scene);
        return NULL;
    }
    return scene;
}

uint32_t rt_scene_addMaterial(Scene* scene, Material material) {
    Material* grown = (Material*)rt_reserve_one(scene->materials, &scene->materialCapacity,
                                                scene->materialCount, sizeof(Material));
    if (!grown) { rt_count_dropped(); return UINT32_MAX; }        /* out of memory: nothing added */
    scene->materials = grown;
    scene->materials[scene->materialCount] = material;
    return scene->materialCount++;
}

void rt_scene_addPlane(Scene* scene, Plane plane) {
    Plane* grown = (Plane*)rt_reserve_one(scene->planes, &scene->planeCapacity, scene->planeCount, sizeof(Plane));
    if (!grown) { rt_count_dropped(); return; }   /* out of memory: the scene keeps what it has */
    scene->planes = grown;
    scene->planes[scene->planeCount++] = plane;
}

void rt_scene_addSphere(Scene* scene, Sphere sphere) {
    Sphere* grown = (Sphere*)rt_reserve_one(scene->spheres, &scene->sphereCapacity, scene->sphereCount, sizeof(Sphere));
    if (!grown) { rt_count_dropped(); return; }   /* out of memory: the scene keeps what it has */
    scene->spheres = grown;
    scene->spheres[scene->sphereCount++] = sphere;
}

void rt_scene_addTriangle(Scene* scene, Triangle triangle) {
    Triangle* grown = (Triangle*)rt_reserve_one(scene->triangles, &scene->triangleCapacity, scene->triangleCount, sizeof(Triangle));
    if (!grown) { rt_count_dropped(); return; }   /* out of memory: the scene keeps what it has */
    scene->triangles = grown;
    scene->triangles[scene->triangleCount++] = triangle;
}

void rt_scene_addObject(Scene* scene, Object object) {
    for (uint32_t i = 0; i < object.triangleCount; i++) rt_scene_addTriangle(scene, object.triangles[i]);
}

void rt_scene_addPointLight(Scene* scene, PointLight pointLight) {
    PointLight* grown = (PointLight*)rt_reserve_one(scene->pointLights, &scene->pointLightCapacity,
                                                    scene->pointLightCount, sizeof(PointLight));
    if (!grown) { rt_count_dropped(); return; }   /* out of memory: the scene keeps what it has */
    scene->pointLights = grown;
    scene->pointLights[scene->pointLightCount++] = pointLight;
}

void rt_scene_shrinkToFit(Scene* scene) {
    scene->materials = (Material*)rt_trim(scene->materials, &scene->materialCapacity, scene->materialCount, sizeof(Material));
    scene->planes = (Plane*)rt_trim(scene->planes, &scene->planeCapacity, scene->planeCount, sizeof(Plane));
    scene->spheres = (Sphere*)rt_trim(scene->spheres, &scene->sphereCapacity, scene->sphereCount, sizeof(Sphere));
    scene->triangles = (Triangle*)rt_trim(scene->triangles, &scene->triangleCapacity, scene->triangleCount, sizeof(Triangle));
    scene->pointLights = (PointLight*)rt_trim(scene->pointLights, &scene->pointLightCapacity, scene->pointLightCount, sizeof(PointLight));
}

void rt_scene_destroy(Scene* scene) {
    if (!scene) return;
    rt_camera_destroy(scene->camera);
    free(scene->materials);
    free(scene->planes);
    free(scene->spheres);
    free(scene->triangles);
    free(scene->pointLights);
    free(scene);
}

/* ------------------------------------------------------------------ camera */

/* reference: src/camera.c:9-32 */
void rt_camera_setup(Camera* camera) {
    camera->z = v3_unit(v3_sub(camera->position, camera->lookAt));
    camera->x = v3_unit(v3_cross(v3(0.0f, 1.0f, 0.0f), camera->z));
    camera->y = v3_unit(v3_cross(camera->z, camera->x));

    float targetW = 1.0f, targetH = 1.0f;
    if (camera->width > camera->height) {
        targetH = targetW * 1.0f / camera->aspectRatio;
    } else if (camera->height > camera->width) {
        targetW = targetH * camera->aspectRatio;
    }
    camera->renderTargetWidth = targetW;
    camera->renderTargetHeight = targetH;

    float diagonal = sqrtf((targetW * targetW) + (targetH * targetH));
    camera->renderTargetDistance = diagonal * (2.0f * tanf(rt_deg2rad(camera->FOV) / 2.0f));
    camera->renderTargetCenter =
        v3_sub(camera->position, v3_scale(camera->z, camera->renderTargetDistance));
}

/* reference: src/camera.c:34-46 */
Camera* rt_camera_create(Vec3 position, Vec3 lookAt, uint32_t width, uint32_t height,
                         float FOV, float apertureSize) {
    Camera* camera = (Camera*)calloc(1, sizeof(Camera));
    if (!camera) return NULL;
    camera->position = position;
    camera->lookAt = lookAt;
    camera->width = width;
    camera->height = height;
    camera->aspectRatio = (float)width / (float)height;
    camera->FOV = FOV;
    camera->focalLength = v3_length(v3_sub(position, lookAt));
    camera->apertureSize = apertureSize;
    rt_camera_setup(camera);
    return camera;
}

/* reference: src/camera.c:48-52 (half a unit per step along x, then y, then z) */
void rt_move_camera(Camera* camera, int upDown, int side, int frontal) {
    camera->position = v3_add(camera->position, v3_scale(camera->x, side * 0.5f));
    camera->position = v3_add(camera->position, v3_scale(camera->y, upDown * 0.5f));
    camera->position = v3_add(camera->position, v3_scale(camera->z, frontal * 0.5f));
}

void rt_camera_destroy(Camera* camera) { free(camera); }

/* ------------------------------------------------------------ default scene */

typedef struct { float refl, refr, amb, diff, spec, expo, r, g, b; } MaterialRow;

/* reference: src/scene.c:43-105 -- index 0 is the background ("no hit"). */
static const MaterialRow kDefaultMaterials[] = {
    /* background */ {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f},
    /* grey       */ {0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 1.0f, 0.4f, 0.4f, 0.4f},
    /* redMirror  */ {1.0f, 0.0f, 0.2f, 1.0f, 1.0f, 64.0f, 1.0f, 0.0f, 0.0f},
    /* mirror     */ {1.0f, 0.0f, 0.2f, 1.0f, 1.0f, 64.0f, 1.0f, 1.0f, 1.0f},
    /* glass      */ {1.0f, 1.4f, 0.0f, 0.0f, 0.0f, 1.0f, 1.0f, 1.0f, 1.0f},
    /* yellow     */ {0.0f, 0.0f, 0.2f, 1.0f, 1.0f, 64.0f, 1.0f, 0.6549f, 0.1019f},
    /* metal      */ {0.3f, 0.0f, 0.2f, 1.0f, 1.0f, 64.0f, 0.81f, 0.83f, 0.84f},
};

typedef struct { uint32_t material; float nx, ny, nz, dist; } PlaneRow;

/* reference: src/scene.c:107-135 -- floor, front, back, left, right (all grey) */
static const PlaneRow kDefaultPlanes[] = {
    {RT_MAT_GREY, 0.0f, 1.0f, 0.0f, 0.0f},
    {RT_MAT_GREY, 0.0f, 0.0f, 1.0f, 50.0f},
    {RT_MAT_GREY, 0.0f, 0.0f, 1.0f, -50.0f},
    {RT_MAT_GREY, 1.0f, 0.0f, 0.0f, -50.0f},
    {RT_MAT_GREY, 1.0f, 0.0f, 0.0f, 50.0f},
};

void rt_scene_add_default_materials(Scene* scene) {
    for (size_t i = 0; i < sizeof kDefaultMaterials / sizeof kDefaultMaterials[0]; i++) {
        const MaterialRow* row = &kDefaultMaterials[i];
        Material m;
        memset(&m, 0, sizeof m);
        m.reflectionIndex = row->refl;
        m.refractionIndex = row->refr;
        m.ambientWeight = row->amb;
        m.diffuseWeight = row->diff;
        m.specularWeight = row->spec;
        m.specularExponent = row->expo;
        m.color = v3(row->r, row->g, row->b);
        rt_scene_addMaterial(scene, m);
    }
}

void rt_scene_add_default_planes(Scene* scene) {
    for (size_t i = 0; i < sizeof kDefaultPlanes / sizeof kDefaultPlanes[0]; i++) {
        const PlaneRow* row = &kDefaultPlanes[i];
        Plane p;
        memset(&p, 0, sizeof p);
        p.materialIndex = row->material;
        p.normal = v3(row->nx, row->ny, row->nz);
        p.distanceFromOrigin = row->dist;
        rt_scene_addPlane(scene, p);
    }
}

Camera* rt_default_camera(uint32_t width, uint32_t height) {
    /* reference: src/scene.c:35-39 */
    return rt_camera_create(v3(40.0f, 2.0f, 0.0f), v3(0.0f, 0.0f, 0.0f), width, height, 110.0f, 0.0f);
}

/* reference: src/scene.c:32-206 */
Scene* rt_scene_init(uint32_t width, uint32_t height) {
    Scene* scene = rt_scene_create();
    if (!scene) return NULL;
    scene->camera = rt_default_camera(width, height);
    rt_scene_add_default_materials(scene);
    rt_scene_add_default_planes(scene);

    static const struct { uint32_t material; float x, y, z, radius; } spheres[] = {
        {RT_MAT_RED_MIRROR, -3.0f, 1.0f, 0.0f, 1.0f},
        {RT_MAT_MIRROR, 0.0f, 1.5f, 0.0f, 1.0f},
        {RT_MAT_GLASS, 3.0f, 1.0f, 3.0f, 1.0f},
    };
    for (size_t i = 0; i < sizeof spheres / sizeof spheres[0]; i++) {
        Sphere s;
        memset(&s, 0, sizeof s);
        s.materialIndex = spheres[i].material;
        s.position = v3(spheres[i].x, spheres[i].y, spheres[i].z);
        s.radius = spheres[i].radius;
        rt_scene_addSphere(scene, s);
    }

    Triangle t;
    memset(&t, 0, sizeof t);
    t.materialIndex = RT_MAT_RED_MIRROR;
    t.v0 = v3(2.0f, 0.0f, 0.0f);
    t.v1 = v3(4.0f, 0.0f, 0.0f);
    t.v2 = v3(3.0f, 1.0f, 0.0f);
    rt_scene_addTriangle(scene, t);

    PointLight light;
    memset(&light, 0, sizeof light);
    light.position = v3(0.0f, 20.0f, 10.0f);
    light.emissionColor = v3(1.0f, 1.0f, 1.0f);
    light.strength = 10000.0f;
    rt_scene_addPointLight(scene, light);

    rt_scene_shrinkToFit(scene);
    return scene;
}
