/*
 * rt_synth.c -- deterministic synthetic scenes for benchmarks and parity tests.
 *
 * No counterpart in the reference, which ships one hard-coded scene
 * (reference: src/scene.c:32-206) and no mesh assets.  The generators reuse
 * that scene's materials, room planes and camera and add a procedural
 * height-field mesh, a ring of point lights and optionally a field of spheres.
 * Everything is a pure function of the arguments: the same call yields the same
 * bytes on every machine that shares this libm.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "rt_host_internal.h"
#include "rt_vecmath.h"

/* 32-bit integer hash of a grid vertex (murmur3 finaliser over a seeded mix). */
static uint32_t rt_vertex_hash(uint32_t ix, uint32_t iz) {
    uint32_t h = 0x9E3779B9u ^ (ix * 0x85EBCA6Bu) ^ (iz * 0xC2B2AE35u + 0x27D4EB2Fu);
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

/* PCG32 (XSH-RR), the minimal reference variant. */
typedef struct { uint64_t state, inc; } Pcg32;

static uint32_t pcg32_next(Pcg32* rng) {
    uint64_t old = rng->state;
    rng->state = old * 6364136223846793005ULL + rng->inc;
    uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
}

static void pcg32_seed(Pcg32* rng, uint64_t seed, uint64_t seq) {
    rng->state = 0u;
    rng->inc = (seq << 1u) | 1u;
    pcg32_next(rng);
    rng->state += seed;
    pcg32_next(rng);
}

static float pcg32_unit(Pcg32* rng) { return (float)(pcg32_next(rng) >> 8) * (1.0f / 16777216.0f); }

static Vec3 rt_grid_vertex(uint32_t ix, uint32_t iz, uint32_t quadsX, uint32_t quadsZ) {
    float x = -25.0f + 50.0f * (float)ix / (float)quadsX;
    float z = -25.0f + 50.0f * (float)iz / (float)quadsZ;
    double bump = (double)(rt_vertex_hash(ix, iz) >> 8) * (1.0 / 16777216.0);
    float y = (float)(1.5 + 0.75 * sin(0.37 * (double)x) * cos(0.29 * (double)z) + 0.25 * bump);
    return v3(x, y, z);
}

static void rt_add_light_ring(Scene* scene, uint32_t count, float radius, float height, float phase) {
    for (uint32_t i = 0; i < count; i++) {
        double angle = 2.0 * 3.14159265358979323846 * ((double)i + (double)phase) / (double)count;
        PointLight light;
        memset(&light, 0, sizeof light);
        light.position = v3((float)(radius * cos(angle)), height, (float)(radius * sin(angle)));
        light.emissionColor = v3(1.0f, 1.0f, 1.0f);
        light.strength = 10000.0f;
        rt_scene_addPointLight(scene, light);
    }
}

Scene* rt_scene_synth_mesh(uint32_t width, uint32_t height, uint32_t quadsX, uint32_t quadsZ,
                           uint32_t maxTriangles, uint32_t lightCount, uint32_t sphereCount) {
    if (quadsX == 0 || quadsZ == 0) return NULL;
    /* triangle numbers are 32-bit throughout (reference: src/scene.h:26-28) */
    if (2ull * quadsX * quadsZ > UINT32_MAX && (maxTriangles == 0 || maxTriangles > UINT32_MAX)) return NULL;
    Scene* scene = rt_scene_create();
    if (!scene) return NULL;
    scene->camera = rt_default_camera(width, height);
    rt_scene_add_default_materials(scene);
    rt_scene_add_default_planes(scene);

    uint64_t total = 2ull * quadsX * quadsZ;
    if (maxTriangles && total > maxTriangles) total = maxTriangles;
    free(scene->triangles);
    scene->triangles = (Triangle*)malloc(sizeof(Triangle) * (size_t)total);
    scene->triangleCapacity = (uint32_t)total;
    scene->triangleCount = 0;
    if (!scene->triangles) { rt_scene_destroy(scene); return NULL; }

    static const uint32_t cycle[3] = {RT_MAT_METAL, RT_MAT_YELLOW, RT_MAT_RED_MIRROR};
    for (uint32_t iz = 0; iz < quadsZ && scene->triangleCount < total; iz++) {
        for (uint32_t ix = 0; ix < quadsX && scene->triangleCount < total; ix++) {
            Vec3 p00 = rt_grid_vertex(ix, iz, quadsX, quadsZ);
            Vec3 p10 = rt_grid_vertex(ix + 1, iz, quadsX, quadsZ);
            Vec3 p01 = rt_grid_vertex(ix, iz + 1, quadsX, quadsZ);
            Vec3 p11 = rt_grid_vertex(ix + 1, iz + 1, quadsX, quadsZ);
            uint32_t material = cycle[(ix + iz) % 3u];
            Triangle lower = {material, p00, p10, p01};
            scene->triangles[scene->triangleCount++] = lower;
            if (scene->triangleCount < total) {
                Triangle upper = {material, p10, p11, p01};
                scene->triangles[scene->triangleCount++] = upper;
            }
        }
    }

    if (lightCount <= 32) {
        rt_add_light_ring(scene, lightCount, 30.0f, 20.0f, 0.0f);
    } else {
        uint32_t outer = lightCount / 2;
        rt_add_light_ring(scene, outer, 30.0f, 20.0f, 0.0f);
        rt_add_light_ring(scene, lightCount - outer, 15.0f, 30.0f, 0.5f);
    }

    if (sphereCount) {
        Pcg32 rng;
        pcg32_seed(&rng, 42u, 54u);
        uint32_t cols = (uint32_t)ceil(sqrt((double)sphereCount * 1.6));
        uint32_t rows = (sphereCount + cols - 1) / cols;
        static const uint32_t sphereCycle[3] = {RT_MAT_MIRROR, RT_MAT_GLASS, RT_MAT_METAL};
        for (uint32_t i = 0; i < sphereCount; i++) {
            uint32_t cx = i % cols, cz = i / cols;
            float pitchX = 48.0f / (float)cols, pitchZ = 48.0f / (float)rows;
            Sphere s;
            memset(&s, 0, sizeof s);
            s.materialIndex = sphereCycle[i % 3u];
            s.radius = 0.2f + 0.4f * pcg32_unit(&rng);
            s.position.x = -24.0f + pitchX * ((float)cx + 0.25f + 0.5f * pcg32_unit(&rng));
            s.position.z = -24.0f + pitchZ * ((float)cz + 0.25f + 0.5f * pcg32_unit(&rng));
            s.position.y = 3.0f + 5.0f * pcg32_unit(&rng);
            rt_scene_addSphere(scene, s);
        }
    }
    rt_scene_shrinkToFit(scene);
    return scene;
}

Scene* rt_scene_synth_s1(uint32_t width, uint32_t height) {
    return rt_scene_synth_mesh(width, height, 1000, 500, 0, 8, 0);
}

Scene* rt_scene_synth_s2(uint32_t width, uint32_t height) {
    return rt_scene_synth_mesh(width, height, 3163, 1581, 10000000u, 64, 1000);
}

void rt_camera_orbit(Camera* camera, uint32_t frame, uint32_t frameCount) {
    double f = frameCount > 1 ? (double)frame / (double)(frameCount - 1) : 0.0;
    double angle = 2.0 * 3.14159265358979323846 * (double)frame / (double)(frameCount ? frameCount : 1);
    camera->position = v3((float)(40.0 * cos(angle)), (float)(2.0 + 10.0 * f), (float)(40.0 * sin(angle)));
    camera->lookAt = v3(0.0f, 0.0f, 0.0f);
    camera->focalLength = v3_length(v3_sub(camera->position, camera->lookAt));
    rt_camera_setup(camera);
}
