/*
 * cpu_trace.c -- CPU restatement of the reference renderer's per-pixel trace path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle and the CPU
 * baseline; the product (libhiptrace) never links, loads or calls it.  It is
 * written from the reference's algorithm, operation by operation, in fp32 with
 * no fused multiply-add (compile with -ffp-contract=off), so that on identical
 * inputs it yields the same bits as the reference's C code compiled by the same
 * compiler.  tests/test_oracle_golden.py checks exactly that against the
 * reference's own sources compiled unmodified (oracle/_ref), and
 * tests/golden/ holds frames generated from that build.
 *
 * What follows what:
 *   vector helpers            reference: src/utils/vec3.c:4-106, src/utils/math.c:3-5
 *   plane / sphere / triangle reference: src/raytracer.c:70-161
 *   closest hit, brute force  reference: src/raytracer.c:163-209
 *   fresnel / refract         reference: src/raytracer.c:9-56 (argument-order quirk kept)
 *   shading + recursion       reference: src/raytracer.c:211-318
 *   primary rays, pixel write reference: src/kernel.cl:804-857 (the CPU side has no pixel loop)
 *   box test, octree walks    reference: src/kernel.cl:428-472, 504-601
 *
 * Shadow-ray jitter: the reference adds a random unit vector to the light
 * direction (src/raytracer.c:278-279, random_bilateral from src/utils/random.c).
 * With ORACLE_JITTER_ZERO random_bilateral() is the constant 0, which makes the offset
 * the zero vector (vec3_norm leaves it untouched).  With ORACLE_JITTER_HASHED each draw
 * comes from a hash of (pixel, tree position, light, component) instead of rand(): the
 * same soft shadows, but independent of the order in which rays are traced.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "oracle.h"

#define ORACLE_EPSILON 0.00001f
#define ORACLE_PI ((float)3.14159265358979323846)

/* ------------------------------------------------------------ vector helpers */

static inline Vec3 vec(float x, float y, float z) { Vec3 v; v.x = x; v.y = y; v.z = z; return v; }
static inline Vec3 add(Vec3 a, Vec3 b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
static inline Vec3 sub(Vec3 a, Vec3 b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; return a; }
static inline Vec3 mul(Vec3 a, float s) { a.x *= s; a.y *= s; a.z *= s; return a; }
static inline float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline Vec3 cross(Vec3 a, Vec3 b) {
    return vec(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline Vec3 hadamard(Vec3 a, Vec3 b) { return vec(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline float length(Vec3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
static inline Vec3 norm(Vec3 a) {
    float len = length(a);
    if (len != 0) { a.x /= len; a.y /= len; a.z /= len; }
    return a;
}
/* reference: src/utils/vec3.c:100-106 */
static inline Vec3 reflect(Vec3 incoming, Vec3 normal) {
    Vec3 reversed = mul(incoming, -1);
    return norm(sub(mul(normal, 2.0f * dot(normal, reversed)), reversed));
}
/* reference: src/utils/math.c:3-5 -- (in, min, max) */
static inline float clampf(float in, float lo, float hi) { return in < lo ? lo : (in > hi ? hi : in); }

/* random_bilateral() of the zero-jitter mode. */
static inline float jitter_bilateral(void) { return 0.0f; }   /* the aperture draws: always 0 */

uint32_t oracle_jitter_hash(uint32_t pixel, uint32_t path, uint32_t light, uint32_t component) {
    uint32_t v = (pixel * 0x9E3779B1u) ^ (path * 0x85EBCA77u) ^ (light * 0xC2B2AE3Du) ^ (component * 0x27D4EB2Fu);
    v ^= v >> 16; v *= 0x85EBCA6Bu;
    v ^= v >> 13; v *= 0xC2B2AE35u;
    v ^= v >> 16;
    return v;
}

/* reference: src/utils/random.c:5-11 -- unilateral = (float)rand() / (float)RAND_MAX,
 * bilateral = -1 + 2 * unilateral; here rand()/RAND_MAX is (h >> 8) / 2^24. */
static inline float hashed_bilateral(uint32_t pixel, uint32_t path, uint32_t light, uint32_t component) {
    float unilateral = (float)(oracle_jitter_hash(pixel, path, light, component) >> 8) / 16777216.0f;
    return -1.0f + 2.0f * unilateral;
}

/* ------------------------------------------------------------- per-thread state */

struct HitRecord;
typedef struct {
    const Scene* scene;
    const Octree* octree;      /* NULL = brute force over all primitives */
    Vec3 cameraPosition;
    uint32_t maxDepth;
    uint32_t walk;
    float boxPad;
    uint32_t* stack;           /* node stack, (levels * 8 + 8) entries */
    OracleCounters n;
    uint32_t jitter, pixel;         /* ORACLE_JITTER_*, y * width + x of the pixel being traced */
    uint32_t model, sample;         /* ORACLE_SHADING_*, index of the sub-pixel sample being traced */
    uint32_t* order;                /* when not NULL: receives the tree positions shaded, in order */
    size_t orderCount, orderCapacity;
    const struct TriangleTable* triangleTable;   /* when not NULL: struct-of-arrays copy of the triangles for block-wise brute force */
    Ray* batchRays;                 /* with a triangle table: room for one shadow ray per light ... */
    struct HitRecord* batchHits;    /* ... and its closest hit */
} Tracer;

/* ------------------------------------------------------------------ intersectors */

/* reference: src/raytracer.c:70-86 */
static int hit_plane(const Plane* plane, const Ray* ray, float* t_out) {
    float denominator = dot(plane->normal, ray->direction);
    if ((denominator < -ORACLE_EPSILON) || (denominator > ORACLE_EPSILON)) {
        float cosAngle = dot(plane->normal, ray->origin);
        float t = (-plane->distanceFromOrigin - cosAngle) / denominator;
        if (t > 0) { *t_out = t; return 1; }
    }
    return 0;
}

/* reference: src/raytracer.c:88-116.  The square root of a negative discriminant is NaN,
 * which fails the > comparison, so no explicit sign test is needed. */
static int hit_sphere(const Sphere* sphere, const Ray* ray, float* t_out) {
    Vec3 rel = sub(ray->origin, sphere->position);
    float a = dot(ray->direction, ray->direction);
    float b = 2.0f * dot(ray->direction, rel);
    float c = dot(rel, rel) - sphere->radius * sphere->radius;
    float denominator = 2.0f * a;
    float root = sqrtf(b * b - 4.0f * a * c);
    if (root > ORACLE_EPSILON) {
        float tpos = (-b + root) / denominator;
        float tneg = (-b - root) / denominator;
        float t = tpos;
        if ((tneg > 0) && (tneg < tpos)) t = tneg;
        if (t > 0) { *t_out = t; return 1; }
    }
    return 0;
}

static Vec3 sphere_normal(const Sphere* sphere, const Ray* ray, float t) {
    Vec3 hitPoint = add(ray->origin, mul(ray->direction, t));
    return norm(sub(hitPoint, sphere->position));
}

static Vec3 triangle_normal(const Triangle* tri) {
    return norm(cross(sub(tri->v1, tri->v0), sub(tri->v2, tri->v0)));
}

/* reference: src/raytracer.c:118-161 -- plane hit, then three inside-outside edge tests. */
static int hit_triangle(Tracer* tr, const Triangle* tri, const Ray* ray, float* t_out) {
    Vec3 normal = triangle_normal(tri);
    float normalDotDir = dot(normal, ray->direction);
    if (fabs(normalDotDir) < ORACLE_EPSILON) return 0;
    tr->n.triangleDistance++;
    float d = dot(normal, tri->v0);
    float t = -(dot(normal, ray->origin) - d) / normalDotDir;
    if (t > 0) {
        tr->n.triangleInside++;
        Vec3 hitPoint = add(ray->origin, mul(ray->direction, t));
        Vec3 c;
        c = cross(sub(tri->v1, tri->v0), sub(hitPoint, tri->v0));
        if (dot(normal, c) < 0) return 0;
        c = cross(sub(tri->v2, tri->v1), sub(hitPoint, tri->v1));
        if (dot(normal, c) < 0) return 0;
        c = cross(sub(tri->v0, tri->v2), sub(hitPoint, tri->v2));
        if (dot(normal, c) < 0) return 0;
        *t_out = t;
        return 1;
    }
    return 0;
}

/* Brute force over millions of triangles, for pixel samples of the large scenes.  The test above,
 * evaluated for a block of triangles at a time over struct-of-arrays copies of the scene: the two
 * quantities that depend on the triangle alone -- its normal and dot(normal, v0), reference:
 * src/raytracer.c:119-121,127 -- come from a table filled once per frame by triangle_normal()
 * itself, and the block loop has no early exits, so the compiler can evaluate eight triangles per
 * instruction.  Every value is produced by the same fp32 operations in the same order as in
 * hit_triangle (no fused multiply-add: the file is compiled with -ffp-contract=off and the vector
 * clone is built for AVX2 without FMA), and IEEE arithmetic gives each lane the bits the scalar
 * code gets; an early `return 0` there becomes a cleared flag here.  tests/test_oracle_golden.py
 * holds both forms against the reference's frames. */
struct TriangleTable {
    float* block;      /* one allocation: 13 rows of `count` floats */
    size_t count;
    const float *nx, *ny, *nz, *d, *ax, *ay, *az, *bx, *by, *bz, *cx, *cy, *cz;
};
#define BRUTE_BLOCK 1024u

#if defined(__x86_64__) && defined(__GNUC__)
__attribute__((target_clones("avx2", "default")))
#endif
void oracle_triangle_block(const struct TriangleTable* T, size_t first, uint32_t count, Vec3 o, Vec3 dir,
                           float* restrict tOut, uint32_t* restrict hitOut, uint32_t* restrict examinedOut, uint32_t* restrict frontOut) {
    const float* restrict nx = T->nx + first; const float* restrict ny = T->ny + first; const float* restrict nz = T->nz + first;
    const float* restrict dd = T->d + first;
    const float* restrict ax = T->ax + first; const float* restrict ay = T->ay + first; const float* restrict az = T->az + first;
    const float* restrict bx = T->bx + first; const float* restrict by = T->by + first; const float* restrict bz = T->bz + first;
    const float* restrict cx = T->cx + first; const float* restrict cy = T->cy + first; const float* restrict cz = T->cz + first;
    for (uint32_t i = 0; i < count; i++) {
        const float normalDotDir = nx[i] * dir.x + ny[i] * dir.y + nz[i] * dir.z;
        const uint32_t examined = !(fabsf(normalDotDir) < ORACLE_EPSILON);
        const float t = -((nx[i] * o.x + ny[i] * o.y + nz[i] * o.z) - dd[i]) / normalDotDir;
        const uint32_t front = examined & (uint32_t)(t > 0);
        const float px = o.x + dir.x * t, py = o.y + dir.y * t, pz = o.z + dir.z * t;
        /* edge v0->v1 against P - v0, v1->v2 against P - v1, v2->v0 against P - v2 */
        float ex = bx[i] - ax[i], ey = by[i] - ay[i], ez = bz[i] - az[i];
        float qx = px - ax[i], qy = py - ay[i], qz = pz - az[i];
        const float s0 = nx[i] * (ey * qz - ez * qy) + ny[i] * (ez * qx - ex * qz) + nz[i] * (ex * qy - ey * qx);
        ex = cx[i] - bx[i]; ey = cy[i] - by[i]; ez = cz[i] - bz[i];
        qx = px - bx[i]; qy = py - by[i]; qz = pz - bz[i];
        const float s1 = nx[i] * (ey * qz - ez * qy) + ny[i] * (ez * qx - ex * qz) + nz[i] * (ex * qy - ey * qx);
        ex = ax[i] - cx[i]; ey = ay[i] - cy[i]; ez = az[i] - cz[i];
        qx = px - cx[i]; qy = py - cy[i]; qz = pz - cz[i];
        const float s2 = nx[i] * (ey * qz - ez * qy) + ny[i] * (ez * qx - ex * qz) + nz[i] * (ex * qy - ey * qx);
        tOut[i] = t;
        hitOut[i] = front & (uint32_t)!(s0 < 0) & (uint32_t)!(s1 < 0) & (uint32_t)!(s2 < 0);
        examinedOut[i] = examined;
        frontOut[i] = front;
    }
}

/* reference: src/kernel.cl:428-472.  A 0/0 slab bound is NaN and falls through every
 * comparison, i.e. accepts.  `pad` widens the box (0 = the reference's test). */
static int ray_meets_box(const Ray* ray, const BoundingBox* box, float pad, float* entry) {
    float txmin = ((box->bottomLeftFrontCorner.x - pad) - ray->origin.x) / ray->direction.x;
    float txmax = ((box->topRightBackCorner.x + pad) - ray->origin.x) / ray->direction.x;
    if (txmin > txmax) { float tmp = txmin; txmin = txmax; txmax = tmp; }
    float tymin = ((box->bottomLeftFrontCorner.y - pad) - ray->origin.y) / ray->direction.y;
    float tymax = ((box->topRightBackCorner.y + pad) - ray->origin.y) / ray->direction.y;
    if (tymin > tymax) { float tmp = tymin; tymin = tymax; tymax = tmp; }
    if ((txmin > tymax) || (tymin > txmax)) return 0;
    if (tymin > txmin) txmin = tymin;
    if (tymax < txmax) txmax = tymax;
    float tzmin = ((box->bottomLeftFrontCorner.z - pad) - ray->origin.z) / ray->direction.z;
    float tzmax = ((box->topRightBackCorner.z + pad) - ray->origin.z) / ray->direction.z;
    if (tzmin > tzmax) { float tmp = tzmin; tzmin = tzmax; tzmax = tmp; }
    if ((txmin > tzmax) || (tzmin > txmax)) return 0;
    if (tzmin > txmin) txmin = tzmin;
    *entry = txmin;
    return 1;
}

/* ------------------------------------------------------------------ closest hit */

typedef struct HitRecord {
    float t;              /* FLT_MAX when nothing was met */
    uint32_t id;          /* ORACLE_NO_PRIMITIVE when nothing was met */
    uint32_t material;
} Hit;

/* The reference scans planes, spheres, triangles in index order and replaces the
 * running minimum on strict < (src/raytracer.c:170,186,202), so among equal distances the
 * smallest id wins.  An octree walk meets primitives in another order (and some of them
 * more than once), hence the explicit (t, id) comparison. */
static inline void consider(Hit* best, float t, uint32_t id, uint32_t material) {
    if (t < best->t || (t == best->t && id < best->id)) {
        best->t = t; best->id = id; best->material = material;
    }
}

static void closest_in_octree(Tracer* tr, const Ray* ray, Hit* best) {
    const Scene* scene = tr->scene;
    const Octree* tree = tr->octree;
    const uint32_t sphereBase = scene->planeCount;
    const uint32_t triangleBase = scene->planeCount + scene->sphereCount;
    uint32_t top = 0;
    tr->stack[top++] = 0;
    while (top > 0) {
        const OctreeNode* node = &tree->nodes[tr->stack[--top]];
        float entry;
        tr->n.boxTests++;
        if (!ray_meets_box(ray, &node->boundingBox, tr->boxPad, &entry)) continue;
        if (tr->walk == ORACLE_WALK_PRUNED && entry > best->t) continue;
        tr->n.nodesVisited++;
        if (node->childNodeIndexes[0] != NODE_INDEX_UNDEF) {
            for (int c = 0; c < 8; c++) tr->stack[top++] = (uint32_t)node->childNodeIndexes[c];
            continue;
        }
        for (uint32_t i = 0; i < node->sphereIndexCount; i++) {
            uint32_t s = tree->indexes[node->sphereIndexOffset + i];
            float t;
            tr->n.sphereTests++;
            if (hit_sphere(&scene->spheres[s], ray, &t)) consider(best, t, sphereBase + s, scene->spheres[s].materialIndex);
        }
        for (uint32_t i = 0; i < node->triangleIndexCount; i++) {
            uint32_t k = tree->indexes[node->triangleIndexOffset + i];
            float t;
            tr->n.triangleTests++;
            if (hit_triangle(tr, &scene->triangles[k], ray, &t)) consider(best, t, triangleBase + k, scene->triangles[k].materialIndex);
        }
    }
}

/* The triangle loop of the brute force for `n` rays at once: each block of the table is tested for
 * every ray while it is in cache.  best[r] holds the running minimum of ray r (planes and spheres
 * already considered); per ray the triangles are met in ascending id order, as in the reference's
 * loop, and replaced on strict < (src/raytracer.c:202). */
static void closest_triangles_blockwise(Tracer* tr, const Ray* rays, uint32_t n, Hit* best) {
    const Scene* scene = tr->scene;
    float t[BRUTE_BLOCK];
    uint32_t hit[BRUTE_BLOCK], examined[BRUTE_BLOCK], front[BRUTE_BLOCK];
    const uint32_t triangleBase = scene->planeCount + scene->sphereCount;
    for (uint32_t first = 0; first < scene->triangleCount; first += BRUTE_BLOCK) {
        const uint32_t count = scene->triangleCount - first < BRUTE_BLOCK ? scene->triangleCount - first : BRUTE_BLOCK;
        for (uint32_t r = 0; r < n; r++) {
            oracle_triangle_block(tr->triangleTable, first, count, rays[r].origin, rays[r].direction, t, hit, examined, front);
            uint32_t nExamined = 0, nFront = 0;
            for (uint32_t i = 0; i < count; i++) { nExamined += examined[i]; nFront += front[i]; }
            tr->n.triangleTests += count;
            tr->n.triangleDistance += nExamined;
            tr->n.triangleInside += nFront;
            Hit* b = &best[r];
            for (uint32_t i = 0; i < count; i++)
                if (hit[i] && t[i] < b->t) { b->t = t[i]; b->id = triangleBase + first + i; b->material = scene->triangles[first + i].materialIndex; }
        }
    }
}

static Hit closest_hit(Tracer* tr, const Ray* ray) {
    const Scene* scene = tr->scene;
    Hit best = {FLT_MAX, ORACLE_NO_PRIMITIVE, 0};
    for (uint32_t i = 0; i < scene->planeCount; i++) {
        float t;
        tr->n.planeTests++;
        if (hit_plane(&scene->planes[i], ray, &t)) consider(&best, t, i, scene->planes[i].materialIndex);
    }
    if (tr->octree) {
        closest_in_octree(tr, ray, &best);
        return best;
    }
    for (uint32_t i = 0; i < scene->sphereCount; i++) {
        float t;
        tr->n.sphereTests++;
        if (hit_sphere(&scene->spheres[i], ray, &t)) consider(&best, t, scene->planeCount + i, scene->spheres[i].materialIndex);
    }
    if (tr->triangleTable) {
        closest_triangles_blockwise(tr, ray, 1, &best);
        return best;
    }
    for (uint32_t i = 0; i < scene->triangleCount; i++) {
        float t;
        tr->n.triangleTests++;
        if (hit_triangle(tr, &scene->triangles[i], ray, &t))
            consider(&best, t, scene->planeCount + scene->sphereCount + i, scene->triangles[i].materialIndex);
    }
    return best;
}

/* closest_hit for several rays of a brute-force frame with a triangle table (the shadow rays of one
 * shading point): the same queries, one pass over the triangles instead of one per ray. */
static void closest_hit_batch(Tracer* tr, const Ray* rays, uint32_t n, Hit* best) {
    const Scene* scene = tr->scene;
    for (uint32_t r = 0; r < n; r++) {
        Hit b = {FLT_MAX, ORACLE_NO_PRIMITIVE, 0};
        for (uint32_t i = 0; i < scene->planeCount; i++) {
            float t;
            tr->n.planeTests++;
            if (hit_plane(&scene->planes[i], &rays[r], &t)) consider(&b, t, i, scene->planes[i].materialIndex);
        }
        for (uint32_t i = 0; i < scene->sphereCount; i++) {
            float t;
            tr->n.sphereTests++;
            if (hit_sphere(&scene->spheres[i], &rays[r], &t)) consider(&b, t, scene->planeCount + i, scene->spheres[i].materialIndex);
        }
        best[r] = b;
    }
    closest_triangles_blockwise(tr, rays, n, best);
}

/* The reference lights a point iff distanceToLight < (closest hit distance of the shadow ray)
 * (src/raytracer.c:286-292), i.e. the point is dark iff some primitive has t <= distanceToLight.
 * In octree mode the walk stops at the first such primitive (as src/kernel.cl:504-550 does
 * with <), which gives the same boolean. */
#define HIDES(t, strict, limit) ((strict) ? (t) < (limit) : (t) <= (limit))
static int occluded_by(Tracer* tr, const Ray* ray, float distanceToLight, int strict) {
    const Scene* scene = tr->scene;
    if (!tr->octree) {
        Hit h = closest_hit(tr, ray);
        return HIDES(h.t, strict, distanceToLight);
    }
    for (uint32_t i = 0; i < scene->planeCount; i++) {
        float t;
        tr->n.planeTests++;
        if (hit_plane(&scene->planes[i], ray, &t) && HIDES(t, strict, distanceToLight)) return 1;
    }
    const Octree* tree = tr->octree;
    uint32_t top = 0;
    tr->stack[top++] = 0;
    while (top > 0) {
        const OctreeNode* node = &tree->nodes[tr->stack[--top]];
        float entry;
        tr->n.boxTests++;
        if (!ray_meets_box(ray, &node->boundingBox, tr->boxPad, &entry)) continue;
        if (tr->walk == ORACLE_WALK_PRUNED && entry > distanceToLight) continue;
        tr->n.nodesVisited++;
        if (node->childNodeIndexes[0] != NODE_INDEX_UNDEF) {
            for (int c = 0; c < 8; c++) tr->stack[top++] = (uint32_t)node->childNodeIndexes[c];
            continue;
        }
        for (uint32_t i = 0; i < node->sphereIndexCount; i++) {
            float t;
            tr->n.sphereTests++;
            if (hit_sphere(&scene->spheres[tree->indexes[node->sphereIndexOffset + i]], ray, &t) && HIDES(t, strict, distanceToLight)) return 1;
        }
        for (uint32_t i = 0; i < node->triangleIndexCount; i++) {
            float t;
            tr->n.triangleTests++;
            if (hit_triangle(tr, &scene->triangles[tree->indexes[node->triangleIndexOffset + i]], ray, &t) && HIDES(t, strict, distanceToLight)) return 1;
        }
    }
    return 0;
}

static int occluded(Tracer* tr, const Ray* ray, float distanceToLight) { return occluded_by(tr, ray, distanceToLight, 0); }
/* the device model's any-hit queries use t < distanceToLight (src/kernel.cl:474-486, 504-550) */
static int occluded_strict(Tracer* tr, const Ray* ray, float distanceToLight) { return occluded_by(tr, ray, distanceToLight, 1); }

static Vec3 hit_normal(const Scene* scene, const Ray* ray, const Hit* hit) {
    if (hit->id < scene->planeCount) return scene->planes[hit->id].normal;
    uint32_t k = hit->id - scene->planeCount;
    if (k < scene->sphereCount) return sphere_normal(&scene->spheres[k], ray, hit->t);
    return triangle_normal(&scene->triangles[k - scene->sphereCount]);
}

/* --------------------------------------------------------- fresnel and refraction */

/* reference: src/raytracer.c:34-56.  The reference calls math_clamp(-1, 1, cos) although the
 * signature is (in, min, max); the result is the constant 1 for every direction.  Kept. */
static float fresnel(Vec3 dir, Vec3 normal, float refractionIndex) {
    float kr;
    float cosi = clampf(-1, 1, dot(dir, normal));
    float etai = 1, etat = refractionIndex;
    if (cosi > 0) { float tmp = etat; etat = etai; etai = tmp; }
    float a = 1.0f - cosi * cosi;
    float sint = etai / etat * sqrtf(0.f > a ? 0.f : a);
    if (sint >= 1) {
        kr = 1;
    } else {
        float b = 1.0f - sint * sint;
        float cost = sqrtf(0.f > b ? 0.f : b);
        cosi = fabsf(cosi);
        float Rs = ((etat * cosi) - (etai * cost)) / ((etat * cosi) + (etai * cost));
        float Rp = ((etai * cosi) - (etat * cost)) / ((etai * cosi) + (etat * cost));
        kr = (Rs * Rs + Rp * Rp) / 2;
    }
    return kr;
}

/* reference: src/raytracer.c:9-32, same argument-order quirk. */
static Vec3 refract(Vec3 direction, Vec3 normal, float refractionIndex) {
    float cosi = clampf(-1, 1, dot(direction, normal));
    float etai = 1, etat = refractionIndex;
    Vec3 n = normal;
    if (cosi < 0) {
        cosi = -cosi;
    } else {
        float tmp = etai; etai = etat; etat = tmp;
        n = mul(n, -1);
    }
    float eta = etai / etat;
    float k = 1 - eta * eta * (1 - cosi * cosi);
    if (k < 0) return vec(0, 0, 0);
    return norm(add(mul(direction, eta), mul(n, eta * cosi - sqrtf(k))));
}

/* reference: src/raytracer.c:66-68 */
static inline void nudge_forward(Ray* ray) {
    ray->origin = add(ray->origin, mul(ray->direction, 1.0f / 1000.0f));
}

/* Draws of the device shading model: the hash keyed additionally by sub-pixel sample and shadow
 * sample, mapped as src/kernel.cl:85-91 maps its xorshift output (17 bits). */
static inline float kernel_bilateral(const Tracer* tr, uint32_t path, uint32_t light, uint32_t shadowSample, uint32_t component) {
    if (tr->jitter != ORACLE_JITTER_HASHED) return 0.0f;
    uint32_t h = oracle_jitter_hash(tr->pixel, path | (tr->sample << 16), light | (shadowSample << 16), component);
    float unilateral = (float)(h % (1u << 17)) / (float)(1 << 17);
    return -1.0f + 2.0f * unilateral;
}

/* ------------------------------------------------------------------ shading */

/* reference: src/raytracer.c:277-284 -- the shadow ray from a shading point towards light i. */
static Ray cpu_model_shadow_ray(const Tracer* tr, const PointLight* light, Vec3 hitPoint, uint32_t path, uint32_t i, float* distanceToLight) {
    Vec3 hitToLight = sub(light->position, hitPoint);
    Vec3 draws = vec(0, 0, 0);
    if (tr->jitter == ORACLE_JITTER_HASHED)
        draws = vec(hashed_bilateral(tr->pixel, path, i, 0), hashed_bilateral(tr->pixel, path, i, 1),
                    hashed_bilateral(tr->pixel, path, i, 2));
    Vec3 offset = norm(draws);
    hitToLight = add(hitToLight, offset);
    *distanceToLight = length(hitToLight);

    Ray shadowRay;
    shadowRay.origin = hitPoint;
    shadowRay.direction = norm(hitToLight);
    nudge_forward(&shadowRay);
    return shadowRay;
}

/* reference: src/raytracer.c:211-314 */
static Vec3 shade_ray(Tracer* tr, const Ray* ray, uint32_t depth, uint32_t path, Hit* primaryOut) {
    const Scene* scene = tr->scene;
    Vec3 outColor = vec(0, 0, 0);
    if (depth >= tr->maxDepth) return outColor;

    Hit hit = closest_hit(tr, ray);
    if (primaryOut) *primaryOut = hit;
    if (!hit.material) return outColor;

    const Material* material = &scene->materials[hit.material];
    Vec3 normal = hit_normal(scene, ray, &hit);
    Vec3 hitPoint = add(ray->origin, mul(ray->direction, hit.t));

    if (material->refractionIndex > 0) {
        float kr = fresnel(ray->direction, normal, material->refractionIndex);
        Vec3 refractionColor = vec(0, 0, 0);
        if (kr < 1) {
            Ray refracted;
            refracted.origin = hitPoint;
            refracted.direction = refract(ray->direction, normal, material->refractionIndex);
            nudge_forward(&refracted);
            if (depth + 1 < tr->maxDepth) tr->n.refraction++;
            refractionColor = shade_ray(tr, &refracted, depth + 1, 2 * path + 1, NULL);
        }
        Ray reflected;
        reflected.origin = hitPoint;
        reflected.direction = reflect(ray->direction, normal);
        nudge_forward(&reflected);
        if (depth + 1 < tr->maxDepth) tr->n.reflection++;
        Vec3 reflectionColor = shade_ray(tr, &reflected, depth + 1, 2 * path, NULL);
        outColor = add(outColor, add(mul(reflectionColor, kr), mul(refractionColor, (1 - kr))));
    } else if (material->reflectionIndex > 0) {
        Ray reflected;
        reflected.origin = hitPoint;
        reflected.direction = reflect(ray->direction, normal);
        nudge_forward(&reflected);
        if (depth + 1 < tr->maxDepth) tr->n.reflection++;
        Vec3 reflectionColor = shade_ray(tr, &reflected, depth + 1, 2 * path, NULL);
        outColor = add(outColor, mul(reflectionColor, material->reflectionIndex));
    }

    if (tr->order) {
        if (tr->orderCount < tr->orderCapacity) tr->order[tr->orderCount] = path;
        tr->orderCount++;
    }
    if (tr->model == ORACLE_SHADING_KERNEL) {
        /* reference: src/kernel.cl:671-714 */
        for (uint32_t i = 0; i < scene->pointLightCount; i++) {
            const PointLight* light = &scene->pointLights[i];
            const uint32_t shadowRays = 4;
            Vec3 directLighting = vec(0, 0, 0);
            for (uint32_t x = 0; x < shadowRays; x++) {
                Vec3 hitToLight = sub(light->position, hitPoint);
                Vec3 randomOffset;
                randomOffset.x = kernel_bilateral(tr, path, i, x, 0);
                randomOffset.y = kernel_bilateral(tr, path, i, x, 1);
                randomOffset.z = kernel_bilateral(tr, path, i, x, 2);
                randomOffset = norm(randomOffset);
                hitToLight = add(hitToLight, randomOffset);
                float distanceToLight = length(hitToLight);
                float distanceToLightSquared = hitToLight.x * hitToLight.x + hitToLight.y * hitToLight.y + hitToLight.z * hitToLight.z;
                Ray shadowRay;
                shadowRay.origin = hitPoint;
                shadowRay.direction = norm(hitToLight);
                nudge_forward(&shadowRay);
                tr->n.shadow++;
                /* the kernel's any-hit walks report a hit for t < distanceToLight (src/kernel.cl:474-486,
                 * 504-550); the CPU model's closest-hit comparison hides the light for t <= distance */
                if (!occluded_strict(tr, &shadowRay, distanceToLight)) {
                    tr->n.lightEvaluations++;
                    float cosAngle = dot(shadowRay.direction, normal);
                    cosAngle = clampf(cosAngle, 0.0f, 1.0f);
                    float lightAttenuation = 1.0f / (1.0f + 4 * ORACLE_PI * distanceToLightSquared);
                    float lightStrength = light->strength * lightAttenuation;
                    Vec3 ambient = mul(light->emissionColor, material->ambientWeight * lightStrength);
                    Vec3 diffuse = mul(light->emissionColor, material->diffuseWeight * cosAngle * lightStrength);
                    Vec3 toView = norm(sub(tr->cameraPosition, hitPoint));
                    Vec3 toLight = mul(shadowRay.direction, -1);
                    Vec3 mirrored = reflect(toLight, normal);
                    cosAngle = dot(toView, mirrored);
                    cosAngle = powf(cosAngle, material->specularExponent);
                    Vec3 specular = mul(light->emissionColor, material->specularWeight * cosAngle * lightStrength);
                    directLighting = add(directLighting, mul(add(ambient, add(diffuse, specular)), (1 - material->reflectionIndex)));
                }
                /* src/kernel.cl:711-712: divided and added inside the sample loop, so the running sum
                 * is divided again by every later sample */
                directLighting.x /= (float)shadowRays; directLighting.y /= (float)shadowRays; directLighting.z /= (float)shadowRays;
                outColor = add(outColor, directLighting);
            }
        }
        return hadamard(outColor, material->color);
    }
    /* brute force over a triangle table: the shadow rays of this point are traced together, so that
     * the triangles are streamed once per point instead of once per light (same rays, same queries) */
    const int batched = tr->triangleTable != NULL && tr->batchRays != NULL && scene->pointLightCount > 1;
    if (batched) {
        for (uint32_t i = 0; i < scene->pointLightCount; i++) {
            float unused;
            tr->batchRays[i] = cpu_model_shadow_ray(tr, &scene->pointLights[i], hitPoint, path, i, &unused);
        }
        closest_hit_batch(tr, tr->batchRays, scene->pointLightCount, tr->batchHits);
    }
    for (uint32_t i = 0; i < scene->pointLightCount; i++) {
        const PointLight* light = &scene->pointLights[i];
        float distanceToLight;
        Ray shadowRay = cpu_model_shadow_ray(tr, light, hitPoint, path, i, &distanceToLight);

        tr->n.shadow++;
        if (batched ? !HIDES(tr->batchHits[i].t, 0, distanceToLight) : !occluded(tr, &shadowRay, distanceToLight)) {
            tr->n.lightEvaluations++;
            float cosAngle = dot(shadowRay.direction, normal);
            cosAngle = clampf(cosAngle, 0.0f, 1.0f);
            float lightStrength = (light->strength / (4 * ORACLE_PI * distanceToLight * distanceToLight));
            Vec3 diffuse = mul(light->emissionColor, cosAngle * lightStrength);

            Vec3 toView = norm(sub(tr->cameraPosition, hitPoint));
            Vec3 toLight = mul(shadowRay.direction, -1);
            Vec3 mirrored = reflect(toLight, normal);
            cosAngle = dot(toView, mirrored);
            cosAngle = powf(cosAngle, 64);
            Vec3 specular = mul(light->emissionColor, cosAngle * lightStrength);

            outColor = add(outColor, mul(add(diffuse, specular), (1 - material->reflectionIndex)));
        }
    }
    return hadamard(outColor, material->color);
}

/* ---------------------------------------------------------- primary rays and pixels */

/* reference: src/kernel.cl:804-838 with one sample per pixel (i = j = 0, deltaX = pixelWidth,
 * deltaY = pixelHeight, src/gpu.c:450-457) and the three aperture draws equal to 0. */
/* The sub-pixel grid: reference: src/gpu.c:446-465. */
typedef struct { uint32_t raysX, raysY; float deltaX, deltaY, contribution; } SampleGrid;
static SampleGrid sample_grid(const Camera* camera, uint32_t raysPerPixel) {
    SampleGrid g;
    if (raysPerPixel == 0) raysPerPixel = 1;
    g.contribution = 1.0f / (float)raysPerPixel;
    float pixelWidth = 1.0f / (float)(camera->width);
    float pixelHeight = 1.0f / (float)(camera->height);
    float rootTerm = sqrtf(pixelWidth / pixelHeight * raysPerPixel + powf(pixelWidth - pixelHeight, 2) / 4 * powf(pixelHeight, 2));
    g.raysX = 1; g.raysY = 1; g.deltaX = pixelWidth; g.deltaY = pixelHeight;
    if (raysPerPixel > 1) {
        g.raysX = (uint32_t)(rootTerm - (pixelWidth - pixelHeight / 2 * pixelHeight));
        if (g.raysX == 0) g.raysX = 1;               /* the reference would divide by zero here */
        g.raysY = (uint32_t)(raysPerPixel / g.raysX);
        g.deltaX = pixelWidth / g.raysX;
        g.deltaY = pixelHeight / g.raysY;
    }
    return g;
}

static Ray primary_ray_sample(const Tracer* tr, const Camera* camera, uint32_t x, uint32_t y, uint32_t i, uint32_t j,
                              const SampleGrid* g) {
    float pixelWidth = 1.0f / (float)(camera->width);
    float pixelHeight = 1.0f / (float)(camera->height);
    float posX = -1.0f + 2.0f * ((float)x / (camera->width));
    float posY = -1.0f + 2.0f * ((float)y / (camera->height));
    Vec3 offsetY = mul(camera->y, (posY - pixelHeight + j * g->deltaY) * camera->renderTargetHeight / 2.0f);
    Vec3 offsetX = mul(camera->x, (posX - pixelWidth + i * g->deltaX) * camera->renderTargetWidth / 2.0f);
    Vec3 target = sub(add(camera->renderTargetCenter, offsetX), offsetY);
    Ray ray;
    ray.origin = camera->position;
    ray.direction = norm(sub(target, camera->position));
    Vec3 focalPoint = add(ray.origin, mul(ray.direction, camera->focalLength));
    /* depth of field (src/kernel.cl:832-838): drawn only by the device model; tree position 0 */
    Vec3 lens = vec(0, 0, 0);
    if (tr && tr->model == ORACLE_SHADING_KERNEL)
        lens = vec(kernel_bilateral(tr, 0, 0, 0, 0) / 2.0f, kernel_bilateral(tr, 0, 0, 0, 1) / 2.0f,
                   kernel_bilateral(tr, 0, 0, 0, 2) / 2.0f);
    ray.origin = add(ray.origin, mul(lens, camera->apertureSize));
    ray.direction = norm(sub(focalPoint, ray.origin));
    return ray;
}

static Ray primary_ray(const Camera* camera, uint32_t x, uint32_t y) {
    float pixelWidth = 1.0f / (float)(camera->width);
    float pixelHeight = 1.0f / (float)(camera->height);
    float posX = -1.0f + 2.0f * ((float)x / (camera->width));
    float posY = -1.0f + 2.0f * ((float)y / (camera->height));
    Vec3 offsetY = mul(camera->y, (posY - pixelHeight + 0 * pixelHeight) * camera->renderTargetHeight / 2.0f);
    Vec3 offsetX = mul(camera->x, (posX - pixelWidth + 0 * pixelWidth) * camera->renderTargetWidth / 2.0f);
    Vec3 target = sub(add(camera->renderTargetCenter, offsetX), offsetY);
    Ray ray;
    ray.origin = camera->position;
    ray.direction = norm(sub(target, camera->position));
    Vec3 focalPoint = add(ray.origin, mul(ray.direction, camera->focalLength));
    Vec3 lens = vec(jitter_bilateral() / 2.0f, jitter_bilateral() / 2.0f, jitter_bilateral() / 2.0f);
    ray.origin = add(ray.origin, mul(lens, camera->apertureSize));
    ray.direction = norm(sub(focalPoint, ray.origin));
    return ray;
}

/* reference: src/kernel.cl:850-856 clamps to [0,1] and stores alpha 1 into an 8-bit
 * normalised RGBA texture (src/gpu.c:586); read back as 0xAABBGGRR (src/utils/image.c:55-59).
 * Conversion is round-to-nearest-even of c*255; NaN stores 0. */
static inline uint32_t to_unorm8(float c) {
    c = clampf(c, 0.0f, 1.0f);
    if (c != c) return 0;
    return (uint32_t)rintf(c * 255.0f);
}

uint32_t oracle_pack_rgba8(float r, float g, float b) {
    return to_unorm8(r) | (to_unorm8(g) << 8) | (to_unorm8(b) << 16) | 0xFF000000u;
}

static uint32_t tree_levels(const Octree* tree) {
    uint32_t* level = (uint32_t*)calloc(tree->nodeCount ? tree->nodeCount : 1, sizeof(uint32_t));
    uint32_t deepest = 0;
    if (!level) return 64;
    for (uint32_t n = 0; n < tree->nodeCount; n++) {
        const OctreeNode* node = &tree->nodes[n];
        if (level[n] > deepest) deepest = level[n];
        if (node->childNodeIndexes[0] == NODE_INDEX_UNDEF) continue;
        for (int c = 0; c < 8; c++) {
            uint32_t child = (uint32_t)node->childNodeIndexes[c];
            if (child < tree->nodeCount) level[child] = level[n] + 1;
        }
    }
    free(level);
    return deepest + 1;
}

static void counters_add(OracleCounters* into, const OracleCounters* from) {
    uint64_t* a = (uint64_t*)into;
    const uint64_t* b = (const uint64_t*)from;
    for (size_t i = 0; i < sizeof(OracleCounters) / sizeof(uint64_t); i++) a[i] += b[i];
}

int oracle_render(const Scene* scene, const Octree* octree, const Camera* camera,
                  const OracleOptions* options, const OracleWindow* window, OracleFrame* out) {
    if (!scene || !camera || !options || !window || !out) return -1;
    if (options->maxDepth == 0 || window->stepX == 0 || window->stepY == 0) return -1;
    const uint32_t nx = window->countX, ny = window->countY;
    const uint32_t stackEntries = octree ? tree_levels(octree) * 8u + 8u : 1u;
    int threads = options->threads;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
    OracleCounters total;
    memset(&total, 0, sizeof total);
    int failed = 0;
    size_t runLength = ((size_t)nx * ny) / ((size_t)threads * 16u);
    if (runLength < 1) runLength = 1;
    if (runLength > nx) runLength = nx;
    const long long runCount = (long long)(((size_t)nx * ny + runLength - 1) / runLength);
    const SampleGrid grid = sample_grid(camera, options->raysPerPixel);
    /* brute force with ORACLE_WALK_PRUNED: block-wise over a struct-of-arrays copy of the triangles */
    struct TriangleTable table;
    memset(&table, 0, sizeof table);
    if (!octree && options->walk == ORACLE_WALK_PRUNED && scene->triangleCount > 0) {
        const size_t n = scene->triangleCount;
        table.block = (float*)malloc(sizeof(float) * 13u * n);
        if (!table.block) return -2;
        table.count = n;
        float* row[13];
        for (int r = 0; r < 13; r++) row[r] = table.block + (size_t)r * n;
        table.nx = row[0]; table.ny = row[1]; table.nz = row[2]; table.d = row[3];
        table.ax = row[4]; table.ay = row[5]; table.az = row[6];
        table.bx = row[7]; table.by = row[8]; table.bz = row[9];
        table.cx = row[10]; table.cy = row[11]; table.cz = row[12];
#pragma omp parallel for num_threads(threads) schedule(static)
        for (size_t i = 0; i < n; i++) {
            const Triangle* tri = &scene->triangles[i];
            const Vec3 normal = triangle_normal(tri);
            row[0][i] = normal.x; row[1][i] = normal.y; row[2][i] = normal.z;
            row[3][i] = dot(normal, tri->v0);
            row[4][i] = tri->v0.x; row[5][i] = tri->v0.y; row[6][i] = tri->v0.z;
            row[7][i] = tri->v1.x; row[8][i] = tri->v1.y; row[9][i] = tri->v1.z;
            row[10][i] = tri->v2.x; row[11][i] = tri->v2.y; row[12][i] = tri->v2.z;
        }
    }

#pragma omp parallel num_threads(threads)
    {
        Tracer tr;
        memset(&tr, 0, sizeof tr);
        tr.scene = scene;
        tr.octree = octree;
        tr.cameraPosition = camera->position;
        tr.maxDepth = options->maxDepth;
        tr.walk = options->walk;
        tr.boxPad = options->boxPad;
        tr.jitter = options->jitter;
        tr.model = options->shadingModel;
        tr.triangleTable = table.block ? &table : NULL;
        if (tr.triangleTable && scene->pointLightCount > 0) {
            tr.batchRays = (Ray*)malloc(sizeof(Ray) * scene->pointLightCount);
            tr.batchHits = (Hit*)malloc(sizeof(Hit) * scene->pointLightCount);
        }
        tr.stack = (uint32_t*)malloc(sizeof(uint32_t) * stackEntries);
        if (!tr.stack) {
#pragma omp atomic write
            failed = 1;
        } else {
            /* pixels are handed out in runs of a row at most, and singly when the window is small (a
             * sample of a few dozen pixels of a ten-million-triangle scene takes seconds per pixel) */
#pragma omp for schedule(dynamic, 1)
            for (long long run = 0; run < runCount; run++) {
                for (size_t at = (size_t)run * runLength; at < (size_t)(run + 1) * runLength && at < (size_t)nx * ny; at++) {
                    uint32_t j = (uint32_t)(at / nx), i = (uint32_t)(at % nx);
                    uint32_t y = window->y0 + j * window->stepY;
                    uint32_t x = window->x0 + i * window->stepX;
                    Hit primary = {FLT_MAX, ORACLE_NO_PRIMITIVE, 0};
                    tr.pixel = y * camera->width + x;
                    Vec3 color = vec(0, 0, 0);
                    /* reference: src/kernel.cl:817-843 -- the supersampling loops */
                    for (uint32_t sj = 0; sj < grid.raysY; sj++)
                        for (uint32_t si = 0; si < grid.raysX; si++) {
                            tr.sample = sj * grid.raysX + si;
                            Ray ray = primary_ray_sample(&tr, camera, x, y, si, sj, &grid);
                            tr.n.primary++;
                            Vec3 sampleColor = shade_ray(&tr, &ray, 0, 1, tr.sample == 0 ? &primary : NULL);
                            color = add(color, mul(sampleColor, grid.contribution));
                        }
                    if (out->rgba8) out->rgba8[at] = oracle_pack_rgba8(color.r, color.g, color.b);
                    if (out->rgb) { out->rgb[3 * at] = color.r; out->rgb[3 * at + 1] = color.g; out->rgb[3 * at + 2] = color.b; }
                    if (out->hitMask) out->hitMask[at] = primary.material != 0;
                    if (out->primitiveId) out->primitiveId[at] = primary.id;
                    if (out->hitDistance) out->hitDistance[at] = primary.t;
                }
            }
        }
#pragma omp critical
        counters_add(&total, &tr.n);
        free(tr.stack);
        free(tr.batchRays);
        free(tr.batchHits);
    }
    free(table.block);
    out->counters = total;
    return failed ? -2 : 0;
}

long oracle_shading_order(const Scene* scene, const Octree* octree, const Camera* camera,
                          const OracleOptions* options, const OracleWindow* window,
                          uint32_t* offsets, uint32_t* paths, size_t capacity) {
    if (!scene || !camera || !options || !window || !offsets || (!paths && capacity)) return -1;
    if (options->maxDepth == 0 || window->stepX == 0 || window->stepY == 0) return -1;
    Tracer tr;
    memset(&tr, 0, sizeof tr);
    tr.scene = scene;
    tr.octree = octree;
    tr.cameraPosition = camera->position;
    tr.maxDepth = options->maxDepth;
    tr.walk = options->walk;
    tr.boxPad = options->boxPad;
    tr.jitter = options->jitter;     /* the order does not depend on it: jitter moves shadow rays only */
    tr.stack = (uint32_t*)malloc(sizeof(uint32_t) * (octree ? tree_levels(octree) * 8u + 8u : 1u));
    if (!tr.stack) return -1;
    uint32_t dummy = 0;
    tr.order = paths ? paths : &dummy;
    tr.orderCapacity = capacity;
    size_t at = 0;
    for (uint32_t j = 0; j < window->countY; j++)
        for (uint32_t i = 0; i < window->countX; i++) {
            uint32_t x = window->x0 + i * window->stepX, y = window->y0 + j * window->stepY;
            offsets[at++] = (uint32_t)tr.orderCount;
            Ray ray = primary_ray(camera, x, y);
            tr.pixel = y * camera->width + x;
            shade_ray(&tr, &ray, 0, 1, NULL);
        }
    offsets[at] = (uint32_t)tr.orderCount;
    free(tr.stack);
    return (long)tr.orderCount;
}

/* Single-ray entry points for known-answer tests of each intersector. */
int oracle_intersect_plane(const Plane* plane, const Ray* ray, float* t) { return hit_plane(plane, ray, t); }
int oracle_intersect_sphere(const Sphere* sphere, const Ray* ray, float* t) { return hit_sphere(sphere, ray, t); }
int oracle_intersect_triangle(const Triangle* triangle, const Ray* ray, float* t) {
    Tracer tr;
    memset(&tr, 0, sizeof tr);
    return hit_triangle(&tr, triangle, ray, t);
}
int oracle_intersect_box(const BoundingBox* box, const Ray* ray) {
    float entry;
    return ray_meets_box(ray, box, 0.0f, &entry);
}
float oracle_fresnel(Vec3 dir, Vec3 normal, float refractionIndex) { return fresnel(dir, normal, refractionIndex); }
Vec3 oracle_refract(Vec3 dir, Vec3 normal, float refractionIndex) { return refract(dir, normal, refractionIndex); }
Vec3 oracle_reflect(Vec3 dir, Vec3 normal) { return reflect(dir, normal); }
Ray oracle_primary_ray(const Camera* camera, uint32_t x, uint32_t y) { return primary_ray(camera, x, y); }
