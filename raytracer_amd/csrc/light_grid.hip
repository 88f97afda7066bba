/*
 * light_grid.hip -- per-light direction grids for occlusion queries towards point lights.
 *
 * A shadow ray (reference: src/raytracer.c:277-292) runs from a shading point P to a light L, and every point of it
 * is seen from L in the one direction (P - L).  So for each light the scene's spheres and triangles are binned, once
 * per upload, by the directions under which L sees them: a cube around the light, each face cut into R x R buckets,
 * every bucket listing the primitives whose projection from L reaches it.  An occlusion query then reads the list of
 * the bucket that (P - L) falls in and tests those primitives only -- the same conservative filter and the same exact
 * tests as a hierarchy walk applies, so the answer is the same; the walk of six or seven levels of boxes per ray is
 * replaced by one bucket lookup.  No counterpart in the reference, which walks its octree for shadow rays
 * (src/kernel.cl:504-550).
 *
 * What makes the lists complete.  The exact tests decide with fp32 arithmetic on a ray that is itself rounded: its
 * direction (a relative 1e-7, so up to 1e-6 scene sizes sideways over the length of a ray), its intersection points and
 * the edge tests on them (a few units in the last place of the scene's largest coordinate).  Together: a primitive can
 * answer "hit" for a ray whose exact line from the light misses it by up to some 3e-6 scene sizes.  The grid allows
 * RT_GRID_SLOP = 1e-5 scene sizes: every primitive is entered into all buckets that its projection, widened by that
 * distance as seen from the light (slop / distance to the primitive, times three for the stretch of an angle on a
 * cube face), touches.  This holds for rays that start within RT_GRID_REACH scene sizes of the origin; the kernel
 * sends any other ray through the hierarchy.  Near the light a distance in space is a large angle: primitives that come
 * within RT_GRID_NEAR scene sizes of the light, and those that would fill more than a 1/RT_GRID_MAX_SHARE of its
 * buckets, go on the light's short "always" list and are tested for every ray towards it; a light whose always list
 * would exceed RT_GRID_ALWAYS entries gets no grid and its rays walk the hierarchy as before.
 */
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "light_grid.h"

#define RT_GRID_BLOCK 256

namespace {

struct Bounds2 { int u0, u1, w0, w1; };   /* bucket ranges, inclusive; u0 > u1 = nothing */

/* A convex outline on a cube face, corners in order; n = 0: none (every bucket of the range counts). */
struct Outline {
    double u[12], w[12];
    double turn;          /* +1: counter-clockwise, -1: clockwise */
    int n;
};

/* Can the outline reach the bucket (iu, iw), widened by `dilation` on every side?  False only when one of the
 * outline's edges has the whole widened bucket on its outer side. */
__device__ bool outline_reaches(const Outline& o, int iu, int iw, int resolution, double dilation) {
    if (o.n == 0) return true;
    const double cell = 2.0 / resolution;
    const double x0 = -1.0 + iu * cell - dilation, x1 = -1.0 + (iu + 1) * cell + dilation;
    const double y0 = -1.0 + iw * cell - dilation, y1 = -1.0 + (iw + 1) * cell + dilation;
    for (int i = 0; i < o.n; i++) {
        const int j = i + 1 == o.n ? 0 : i + 1;
        /* inside the edge: turn * cross(b - a, p - a) >= 0, a linear function a * x + b * y + c of p */
        const double a = -o.turn * (o.w[j] - o.w[i]), b = o.turn * (o.u[j] - o.u[i]);
        const double c = -(a * o.u[i] + b * o.w[i]);
        const double best = a * (a >= 0.0 ? x1 : x0) + b * (b >= 0.0 ? y1 : y0) + c;
        /* the margin stands for the rounding of these few operations on values of size <= 2 */
        if (best < -1e-12) return false;
    }
    return true;
}

/* A convex polygon in the coordinates of one cube face: x towards the face, (u, w) across it. */
struct Polygon {
    double x[12], u[12], w[12];
    int n;
};

/* Keeps the part of `p` where a*x + b*u + c*w >= 0 (a plane through the light). */
__device__ void clip(Polygon& p, double a, double b, double c) {
    Polygon q;
    q.n = 0;
    for (int i = 0; i < p.n; i++) {
        const int j = i + 1 == p.n ? 0 : i + 1;
        const double di = a * p.x[i] + b * p.u[i] + c * p.w[i], dj = a * p.x[j] + b * p.u[j] + c * p.w[j];
        if (di >= 0.0) { q.x[q.n] = p.x[i]; q.u[q.n] = p.u[i]; q.w[q.n] = p.w[i]; q.n++; }
        if ((di >= 0.0) != (dj >= 0.0)) {
            const double s = di / (di - dj);
            q.x[q.n] = p.x[i] + s * (p.x[j] - p.x[i]); q.u[q.n] = p.u[i] + s * (p.u[j] - p.u[i]); q.w[q.n] = p.w[i] + s * (p.w[j] - p.w[i]);
            q.n++;
        }
    }
    p = q;
}

/* Buckets of face `face` (0..5: +x -x +y -y +z -z) that the polygon with corners v[0..n) (relative to the light) can be
 * seen in.  Returns false when it comes so close to the light that no bucket range can be trusted (`tooNear`). */
__device__ bool face_buckets(const double (*v)[3], int n, int face, int resolution, double nearLimit, double dilation, Bounds2& out, Outline* outline) {
    const int axis = face >> 1;
    const double sign = (face & 1) ? -1.0 : 1.0;
    const int ua = (axis + 1) % 3, wa = (axis + 2) % 3;
    Polygon p;
    p.n = n;
    for (int i = 0; i < n; i++) { p.x[i] = sign * v[i][axis]; p.u[i] = v[i][ua]; p.w[i] = v[i][wa]; }
    /* the face's pyramid, a little wider than the face so that directions on its rim are listed on both sides */
    const double rim = 1.0 + 4.0 * dilation;
    clip(p, rim, -1.0, 0.0);
    if (p.n) clip(p, rim, 1.0, 0.0);
    if (p.n) clip(p, rim, 0.0, -1.0);
    if (p.n) clip(p, rim, 0.0, 1.0);
    out.u0 = 1; out.u1 = 0; out.w0 = 1; out.w1 = 0;
    if (outline) outline->n = 0;
    if (p.n == 0) return true;
    double ulo = 2.0, uhi = -2.0, wlo = 2.0, whi = -2.0;
    for (int i = 0; i < p.n; i++) {
        if (!(p.x[i] > nearLimit)) return false;          /* at the light itself: every direction */
        const double u = p.u[i] / p.x[i], w = p.w[i] / p.x[i];
        ulo = fmin(ulo, u); uhi = fmax(uhi, u); wlo = fmin(wlo, w); whi = fmax(whi, w);
        if (outline) { outline->u[i] = u; outline->w[i] = w; }
    }
    if (outline) {
        /* a perspective view of a convex polygon in front of the eye is a convex polygon with the same corners in the
         * same order; which way round it runs is the sign of its area */
        double area = 0.0;
        for (int i = 0; i < p.n; i++) {
            const int j = i + 1 == p.n ? 0 : i + 1;
            area += outline->u[i] * outline->w[j] - outline->u[j] * outline->w[i];
        }
        /* seen (nearly) edge-on the outline says nothing reliable: the bucket range alone decides then */
        outline->n = fabs(area) > 1e-9 ? p.n : 0;
        outline->turn = area > 0.0 ? 1.0 : -1.0;
    }
    const double half = 0.5 * resolution;
    const int last = resolution - 1;
    out.u0 = max(0, min(last, (int)floor((ulo - dilation + 1.0) * half)));
    out.u1 = max(0, min(last, (int)floor((uhi + dilation + 1.0) * half)));
    out.w0 = max(0, min(last, (int)floor((wlo - dilation + 1.0) * half)));
    out.w1 = max(0, min(last, (int)floor((whi + dilation + 1.0) * half)));
    return true;
}

struct BuildParams {
    const QNode* nodes;
    uint32_t nodeCount;
    uint32_t sphereCount, triangleCount;   /* read when nodes == nullptr */
    const uint32_t* leafIndex;
    const Float4* triangleCorner;
    const Float4* sphere;
    const Float4* light;          /* 2 per light, (position, strength) first */
    uint32_t lightCount;
    uint32_t resolution;
    float sphereGrow;             /* added to r^2: see flatten.cpp, bounds_of(Sphere) */
    float nearLimit;              /* primitives closer to a light than this go on its always list */
    float extent;                 /* largest |coordinate| of the scene */
    uint32_t* counts;             /* [lightCount * (6 R^2 + 1)]: pass 0 counts, pass 1 uses it as write cursors */
    uint32_t* refs;               /* pass 1 */
    const uint32_t* info;         /* pass 1: where each light's stretch of refs begins */
    uint32_t* always;             /* [lightCount * RT_GRID_ALWAYS] */
    uint32_t* alwaysCount;        /* [lightCount], pass 0 only */
    uint32_t firstLight;          /* the light of grid row 0 (a look at one light alone) */
    uint32_t* farthest;           /* [lightCount], pass 0: the largest distance of a listed primitive, as the bits of a float */
    float levelSlop;              /* pass 1: taken off a primitive's distance before it is rounded down to a level */
};

/* Calls emit(bucket) for every bucket of light `l` that primitive `ref` belongs to, or returns false if it has to go on
 * the always list.  corners: the primitive's convex outline(s) relative to the light -- one triangle, or the six faces
 * of a sphere's bounding cube. */
template <typename Emit>
__device__ bool for_each_bucket(const double (*polygons)[4][3], const int* sizes, int polygonCount, double nearest,
                                const BuildParams& b, Emit emit) {
    if (nearest < (double)b.nearLimit) return false;
    /* the widening: an uncertainty in space seen from the primitive's distance, as an angle, on a cube face (where an
     * angle is stretched up to three times) */
    const double dilation = fmax(RT_GRID_MIN_DILATION, 3.0 * RT_GRID_SLOP * (double)b.extent / nearest);
    const uint32_t perFace = b.resolution * b.resolution;
    /* first pass over the faces: ranges and the total; a primitive that is everywhere is not worth listing */
    Bounds2 range[6];
    Outline outline[6];
    uint64_t total = 0;
    for (int face = 0; face < 6; face++) {
        Bounds2 all; all.u0 = 1; all.u1 = 0; all.w0 = 1; all.w1 = 0;
        outline[face].n = 0;
        for (int k = 0; k < polygonCount; k++) {
            Bounds2 one;
            if (!face_buckets(polygons[k], sizes[k], face, (int)b.resolution, 1e-3 * (double)b.nearLimit, dilation, one, polygonCount == 1 ? &outline[face] : nullptr)) return false;
            if (one.u0 > one.u1) continue;
            if (all.u0 > all.u1) all = one;
            else { all.u0 = min(all.u0, one.u0); all.u1 = max(all.u1, one.u1); all.w0 = min(all.w0, one.w0); all.w1 = max(all.w1, one.w1); }
        }
        range[face] = all;
        if (all.u0 <= all.u1) total += (uint64_t)(all.u1 - all.u0 + 1) * (uint64_t)(all.w1 - all.w0 + 1);
    }
    if (total > min((uint64_t)RT_GRID_MAX_PER_PRIMITIVE, max((uint64_t)RT_GRID_MAX_BUCKETS, (uint64_t)perFace * 6u / RT_GRID_MAX_SHARE))) return false;
    for (int face = 0; face < 6; face++) {
        const Bounds2& r = range[face];
        if (r.u0 > r.u1) continue;
        for (int w = r.w0; w <= r.w1; w++)
            for (int u = r.u0; u <= r.u1; u++)
                if (outline_reaches(outline[face], u, w, (int)b.resolution, dilation)) emit((uint32_t)face * perFace + (uint32_t)w * b.resolution + (uint32_t)u);
    }
    return true;
}

/* Enters one primitive into light l's grid.  PASS 0 counts per bucket and collects the always lists, PASS 1 writes the
 * references.  ref: what the kernel finds the primitive by (see LightGrid). */
template <int PASS>
__device__ void enter(const BuildParams& b, uint32_t l, bool isSphere, uint32_t k, uint32_t ref) {
    const Float4 light = b.light[2 * l];
    const double L[3] = {light.x, light.y, light.z};
    const uint32_t buckets = 6u * b.resolution * b.resolution;
    uint32_t* const cursor = b.counts + (size_t)l * (buckets + 1u);
    uint32_t* const list = PASS == 0 ? nullptr : b.refs + ((size_t)b.info[RT_GRID_INFO_WORDS * l + 2] | ((size_t)b.info[RT_GRID_INFO_WORDS * l + 3] << 32));
    double polygons[6][4][3];
    int sizes[6];
    int polygonCount;
    double nearest;
    if (isSphere) {
        const Float4 sp = b.sphere[k];
        const double r = sqrt((double)sp.w * (double)sp.w + (double)b.sphereGrow) * 1.0001;
        const double c[3] = {sp.x - L[0], sp.y - L[1], sp.z - L[2]};
        nearest = sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) - 1.7321 * r;     /* to the bounding cube */
        /* the six faces of the bounding cube; the sphere is seen inside their union */
        polygonCount = 6;
        for (int f = 0; f < 6; f++) {
            const int axis = f >> 1, ua = (axis + 1) % 3, wa = (axis + 2) % 3;
            const double side = (f & 1) ? -r : r;
            const double du[4] = {-r, r, r, -r}, dw[4] = {-r, -r, r, r};
            sizes[f] = 4;
            for (int q = 0; q < 4; q++) { polygons[f][q][axis] = c[axis] + side; polygons[f][q][ua] = c[ua] + du[q]; polygons[f][q][wa] = c[wa] + dw[q]; }
        }
    } else {
        polygonCount = 1;
        sizes[0] = 3;
        double centre[3] = {0, 0, 0};
        for (int q = 0; q < 3; q++) {
            const Float4 p = b.triangleCorner[3 * (size_t)k + q];
            polygons[0][q][0] = p.x - L[0]; polygons[0][q][1] = p.y - L[1]; polygons[0][q][2] = p.z - L[2];
            for (int a = 0; a < 3; a++) centre[a] += polygons[0][q][a] / 3.0;
        }
        double radius = 0.0;
        for (int q = 0; q < 3; q++) {
            double d2 = 0.0;
            for (int a = 0; a < 3; a++) d2 += (polygons[0][q][a] - centre[a]) * (polygons[0][q][a] - centre[a]);
            radius = fmax(radius, sqrt(d2));
        }
        nearest = sqrt(centre[0] * centre[0] + centre[1] * centre[1] + centre[2] * centre[2]) - radius;   /* to the bounding sphere */
    }
    bool listed;
    if (PASS == 0) {
        listed = for_each_bucket(polygons, sizes, polygonCount, nearest, b, [&](uint32_t bucket) { atomicAdd(&cursor[bucket], 1u); });
        /* positive floats compare like their bits */
        if (listed) atomicMax(&b.farthest[l], __float_as_uint((float)nearest));
    } else {
        /* the distance level: `nearest` is a lower bound of the distance of every point of the primitive from the light */
        const double perUnit = (double)__uint_as_float(b.info[RT_GRID_INFO_WORDS * l + 4]);
        const double levelAt = floor(fmax(0.0, nearest * (1.0 - 1e-6) - (double)b.levelSlop) * perUnit);
        const uint32_t level = (uint32_t)fmin((double)(GRID_REF_LEVELS - 1u), levelAt);
        const uint32_t word = ref | (level << GRID_REF_LEVEL_SHIFT);
        listed = for_each_bucket(polygons, sizes, polygonCount, nearest, b, [&](uint32_t bucket) { list[atomicAdd(&cursor[bucket], 1u)] = word; });
    }
    if (!listed && PASS == 0) {
        const uint32_t slot = atomicAdd(&b.alwaysCount[l], 1u);
        if (slot < RT_GRID_ALWAYS) b.always[l * RT_GRID_ALWAYS + slot] = ref;
    }
}

/* grid.y = light.  With a hierarchy that stores every primitive once (b.nodes set): one thread per (wide node, child),
 * a leaf child enters each of its primitives, a triangle under its position in the leaf arrays.  Otherwise one thread
 * per primitive, spheres first, a triangle under its own number. */
template <int PASS>
__global__ __launch_bounds__(RT_GRID_BLOCK) void grid_kernel(const BuildParams b) {
    const uint64_t item = (uint64_t)blockIdx.x * RT_GRID_BLOCK + threadIdx.x;
    const uint32_t l = b.firstLight + blockIdx.y;
    if (b.nodes == nullptr) {
        if (item < b.sphereCount) enter<PASS>(b, l, true, (uint32_t)item, GRID_REF_SPHERE | (uint32_t)item);
        else if (item < (uint64_t)b.sphereCount + b.triangleCount) enter<PASS>(b, l, false, (uint32_t)(item - b.sphereCount), (uint32_t)(item - b.sphereCount));
        return;
    }
    if (item >= (uint64_t)b.nodeCount * 8u) return;
    const uint32_t kind = b.nodes[item >> 3].kind[item & 7u], link = b.nodes[item >> 3].link[item & 7u];
    if (kind & QNODE_INNER) return;
    const uint32_t sphereCount = kind >> 11, triangleCount = kind & 0x7FFu;
    for (uint32_t i = 0; i < sphereCount + triangleCount; i++) {
        const uint32_t k = b.leafIndex[link + i];
        if (i < sphereCount) enter<PASS>(b, l, true, k, GRID_REF_SPHERE | k);
        else enter<PASS>(b, l, false, k, link + i);
    }
}

/* totals[2 l] += the counts of light l's buckets, totals[2 l + 1] += their squares (grid.y = light). */
__global__ void sum_kernel(const uint32_t* counts, size_t perLight, unsigned long long* totals) {
    const uint32_t l = blockIdx.y;
    unsigned long long mine = 0, squares = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < perLight; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long n = counts[(size_t)l * perLight + i];
        mine += n;
        squares += n * n;
    }
    if (mine) { atomicAdd(&totals[2 * l], mine); atomicAdd(&totals[2 * l + 1], squares); }
}

/* Diagnosis (RT_LIGHT_GRID_VERBOSE): longest[0] = the longest list, longest[1 + k] = lists longer than 16 << k. */
__global__ void longest_kernel(const uint32_t* counts, size_t count, uint32_t* longest) {
    uint32_t mine = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t n = counts[i];
        mine = max(mine, n);
        for (uint32_t k = 0; k < 6u; k++) if (n > (16u << k)) atomicAdd(&longest[1 + k], 1u);
    }
    atomicMax(&longest[0], mine);
}

/* longest[l] = the longest list of light l (grid.y = light). */
__global__ void longest_per_light_kernel(const uint32_t* counts, size_t perLight, uint32_t* longest) {
    const uint32_t l = blockIdx.y;
    uint32_t mine = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < perLight; i += (size_t)gridDim.x * blockDim.x)
        mine = max(mine, counts[(size_t)l * perLight + i]);
    if (mine) atomicMax(&longest[l], mine);
}

/* Sorts every list of light l (grid.y) by its entries' words, nearest distance level first.  `ends`: the light's run of
 * write cursors after the filling pass, each at the end of its bucket's list, which is where the next bucket's begins. */
__global__ void sort_lists_kernel(const uint32_t* ends, size_t buckets, const uint32_t* info, uint32_t* refs) {
    const uint32_t l = blockIdx.y;
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= buckets) return;
    /* a light without a grid is never read, and its lists may be of any length; a light with one has none longer than 255 */
    if (info[RT_GRID_INFO_WORDS * l] == 0u) return;
    const uint32_t* const run = ends + (size_t)l * (buckets + 1);
    const uint32_t begin = g ? run[g - 1] : 0u, end = run[g];
    uint32_t* const list = refs + ((size_t)info[RT_GRID_INFO_WORDS * l + 2] | ((size_t)info[RT_GRID_INFO_WORDS * l + 3] << 32)) + begin;
    const uint32_t n = end - begin;
    for (uint32_t i = 1; i < n; i++) {       /* lists are short: by insertion */
        const uint32_t word = list[i];
        uint32_t at = i;
        while (at > 0 && list[at - 1] > word) { list[at] = list[at - 1]; at--; }
        list[at] = word;
    }
}

/* The index the kernels read (device_layout.h: LightGrid): one 12-byte block per eight buckets, from a light's run of
 * list starts (buckets + 1 of them, the last being the light's total).  grid.y = light. */
__global__ void pack_kernel(const uint32_t* start, size_t buckets, uint32_t blocksPerLight, uint32_t* blocks) {
    const uint32_t l = blockIdx.y;
    const size_t block = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (block >= blocksPerLight) return;
    const uint32_t* const run = start + (size_t)l * (buckets + 1);
    const size_t g0 = block * 8;
    uint32_t length[8];
    for (int k = 0; k < 8; k++) {
        const size_t g = g0 + k;
        length[k] = g < buckets ? min(255u, run[g + 1] - run[g]) : 0u;
    }
    uint32_t* const out = blocks + ((size_t)l * blocksPerLight + block) * 3;
    out[0] = run[g0 < buckets ? g0 : buckets];
    out[1] = length[0] | (length[1] << 8) | (length[2] << 16) | (length[3] << 24);
    out[2] = length[4] | (length[5] << 8) | (length[6] << 16) | (length[7] << 24);
}

}  // namespace

#define GRID_TRY(call)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) { failure = e_; error = std::string(#call) + ": " + hipGetErrorString(e_); goto failed; } \
    } while (0)

int rt_light_grid_build(const LightGridInput& in, LightGridBuild& out, std::string& error) {
    out = LightGridBuild{};
    if (in.lightCount == 0 || in.entryCount == 0) return 0;
    /* the build kernels take the light from the second grid dimension; scenes with more lights than that allows walk */
    if (in.lightCount > 65535u) return 0;
    /* an entry of a list keeps 24 bits of primitive number or leaf position */
    if (in.entryCount >= GRID_REF_MAX_ENTRIES) return 0;
    const bool verbose = std::getenv("RT_LIGHT_GRID_VERBOSE") != nullptr;
    /* RT_GRID_DENSITY buckets per primitive over the whole cube */
    double density = RT_GRID_DENSITY;
    if (const char* env = std::getenv("RT_LIGHT_GRID_DENSITY")) density = std::atof(env);
    uint32_t resolution = (uint32_t)std::min((double)RT_GRID_MAX_RESOLUTION, std::max((double)RT_GRID_MIN_RESOLUTION,
                                                                                    std::ceil(std::sqrt(density * (double)in.entryCount / 6.0))));
    /* what the grids may take: a share of the memory that is free now */
    size_t freeBytes = 0, deviceBytes = 0;
    if (hipMemGetInfo(&freeBytes, &deviceBytes) != hipSuccess) { (void)hipGetLastError(); freeBytes = 0; }
    size_t budget = std::min((size_t)(RT_GRID_MEMORY_SHARE * (double)freeBytes), (size_t)(RT_GRID_SCENE_MULTIPLE * (double)in.sceneBytes) + ((size_t)1 << 30));
    if (const char* env = std::getenv("RT_LIGHT_GRID_MAX_BYTES")) budget = std::min(budget, (size_t)std::strtoull(env, nullptr, 10));
    if (const char* env = std::getenv("RT_LIGHT_GRID_BUDGET_BYTES")) budget = (size_t)std::strtoull(env, nullptr, 10);   /* experiments: instead of the rule above */
    /* the longest lists worth having (the length of the list an average reference sits in) */
    double maxExpected = RT_GRID_MAX_EXPECTED;
    if (const char* env = std::getenv("RT_LIGHT_GRID_MAX_EXPECTED")) maxExpected = std::atof(env);

    hipError_t failure = hipSuccess;
    uint32_t *counts = nullptr, *blocks = nullptr, *refs = nullptr, *always = nullptr, *alwaysCount = nullptr, *lights = nullptr, *longestOfLight = nullptr;
    uint32_t* farthest = nullptr;
    std::vector<uint32_t> hostFarthest(in.lightCount);
    uint32_t blocksPerLight = 0;
    std::vector<uint32_t> hostLongestOfLight;
    unsigned long long* totals = nullptr;
    void* temp = nullptr;
    size_t tempBytes = 0, perLight = 0, slots = 0;
    unsigned long long all = 0;
    std::vector<uint32_t> hostAlwaysCount(in.lightCount);
    std::vector<unsigned long long> hostTotals(2 * (size_t)in.lightCount);   /* per light: references, sum of squared list lengths */
    std::vector<uint32_t> info;
    BuildParams b{};
    const uint64_t items = in.nodes ? (uint64_t)in.nodeCount * 8u : (uint64_t)in.sphereCount + in.triangleCount;
    const dim3 grid((unsigned)((items + RT_GRID_BLOCK - 1) / RT_GRID_BLOCK), in.lightCount);
    b.nodes = in.nodes; b.nodeCount = in.nodeCount; b.sphereCount = in.sphereCount; b.triangleCount = in.triangleCount; b.leafIndex = in.leafIndex; b.triangleCorner = in.triangleCorner;
    b.sphere = in.sphere; b.light = in.light; b.lightCount = in.lightCount;
    b.sphereGrow = in.sphereGrow; b.nearLimit = RT_GRID_NEAR * in.extent; b.extent = in.extent;

    GRID_TRY(hipMalloc(&always, (size_t)in.lightCount * RT_GRID_ALWAYS * sizeof(uint32_t)));
    GRID_TRY(hipMalloc(&alwaysCount, in.lightCount * sizeof(uint32_t)));
    GRID_TRY(hipMalloc(&totals, 2 * in.lightCount * sizeof(unsigned long long)));
    GRID_TRY(hipMalloc(&farthest, in.lightCount * sizeof(uint32_t)));
    b.farthest = farthest;
    /* what a ray's arithmetic can add to a distance: its start is moved 1e-3 along it and it is followed as far as the
     * light's distance from the unmoved start, so 1e-3 beyond the light; plus twice the grids' own allowance */
    b.levelSlop = 2e-3f + 2.0f * (float)RT_GRID_SLOP * in.extent;
    /* pass 0 at falling resolutions (three quarters each time) until the lists fit the budget and every light's fit 32-bit offsets */
    for (;; resolution = std::max<uint32_t>(RT_GRID_MIN_RESOLUTION, resolution * 3u / 4u)) {
        perLight = 6u * (size_t)resolution * resolution;
        slots = (perLight + 1) * in.lightCount;          /* every light's run of starts ends with its total */
        blocksPerLight = (uint32_t)((perLight + 7) / 8);
        /* while building: a count (then start, then write cursor) per bucket and the packed index, 12 bytes per eight buckets */
        const size_t indexBytes = slots * sizeof(uint32_t) + (size_t)in.lightCount * blocksPerLight * 3 * sizeof(uint32_t);
        /* the kernels number a block with 32 bits, from 24-bit factors (light, blocks per light) */
        if (indexBytes <= budget && (unsigned long long)in.lightCount * blocksPerLight < 0xFFFFFFFFull && blocksPerLight < (1u << 24) && in.lightCount < (1u << 24)) {
            GRID_TRY(hipMalloc(&counts, slots * sizeof(uint32_t)));
            {
                /* A few lights alone (the first, the middle one, the last), to see what the lists would be like before the
                 * whole pass is paid for.  Coarser buckets only make lists longer, so where the lists these lights' rays
                 * would meet are too long on average no grid is built at all; where their references, scaled to all the
                 * lights, would not fit the budget, the next coarser resolution is looked at straight away.  One light
                 * in an awkward place neither decides for the others nor is decided for: which lights get a grid is
                 * settled light by light further down. */
                const uint32_t probes[3] = {0u, in.lightCount / 2u, in.lightCount - 1u};
                double meanExpected = 0.0, meanReferences = 0.0;
                int looked = 0;
                for (int k = 0; k < 3; k++) {
                    if (k > 0 && probes[k] == probes[k - 1]) continue;
                    uint32_t* const row = counts + (size_t)probes[k] * (perLight + 1);
                    GRID_TRY(hipMemset(row, 0, (perLight + 1) * sizeof(uint32_t)));
                    GRID_TRY(hipMemset(alwaysCount, 0, in.lightCount * sizeof(uint32_t)));
                    GRID_TRY(hipMemset(totals, 0, 2 * sizeof(unsigned long long)));
                    b.resolution = resolution; b.counts = counts; b.refs = nullptr; b.always = always; b.alwaysCount = alwaysCount;
                    b.firstLight = probes[k];
                    hipLaunchKernelGGL(grid_kernel<0>, dim3(grid.x, 1), dim3(RT_GRID_BLOCK), 0, nullptr, b);
                    b.firstLight = 0;
                    GRID_TRY(hipGetLastError());
                    hipLaunchKernelGGL(sum_kernel, dim3(256, 1), dim3(256), 0, nullptr, row, perLight + 1, totals);
                    GRID_TRY(hipGetLastError());
                    GRID_TRY(hipMemcpy(hostTotals.data(), totals, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                    const double expected = hostTotals[0] ? (double)hostTotals[1] / (double)hostTotals[0] : 0.0;
                    if (verbose) std::fprintf(stderr, "light grid: resolution %u, light %u alone: list length met by a reference %.1f, %llu references\n",
                                              resolution, probes[k], expected, hostTotals[0]);
                    meanExpected += expected;
                    meanReferences += (double)hostTotals[0];
                    looked++;
                }
                meanExpected /= looked;
                meanReferences /= looked;
                if (meanExpected > 1.25 * maxExpected) {
                    (void)hipFree(counts); (void)hipFree(always); (void)hipFree(alwaysCount); (void)hipFree(totals); (void)hipFree(farthest);
                    return 0;
                }
                if ((double)indexBytes + 0.9 * meanReferences * in.lightCount * sizeof(uint32_t) > (double)budget && resolution > RT_GRID_MIN_RESOLUTION) {
                    (void)hipFree(counts); counts = nullptr;
                    continue;
                }
            }
            GRID_TRY(hipMemset(counts, 0, slots * sizeof(uint32_t)));
            GRID_TRY(hipMemset(always, 0, (size_t)in.lightCount * RT_GRID_ALWAYS * sizeof(uint32_t)));
            GRID_TRY(hipMemset(alwaysCount, 0, in.lightCount * sizeof(uint32_t)));
            GRID_TRY(hipMemset(totals, 0, 2 * in.lightCount * sizeof(unsigned long long)));
            GRID_TRY(hipMemset(farthest, 0, in.lightCount * sizeof(uint32_t)));
            b.resolution = resolution; b.counts = counts; b.refs = nullptr; b.always = always; b.alwaysCount = alwaysCount;
            hipLaunchKernelGGL(grid_kernel<0>, grid, dim3(RT_GRID_BLOCK), 0, nullptr, b);
            GRID_TRY(hipGetLastError());
            hipLaunchKernelGGL(sum_kernel, dim3(256, in.lightCount), dim3(256), 0, nullptr, counts, perLight + 1, totals);
            GRID_TRY(hipGetLastError());
            GRID_TRY(hipMemcpy(hostTotals.data(), totals, 2 * in.lightCount * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            GRID_TRY(hipMemcpy(hostAlwaysCount.data(), alwaysCount, in.lightCount * sizeof(uint32_t), hipMemcpyDeviceToHost));
            GRID_TRY(hipMemcpy(hostFarthest.data(), farthest, in.lightCount * sizeof(uint32_t), hipMemcpyDeviceToHost));
            all = 0;
            unsigned long long most = 0;
            for (uint32_t l = 0; l < in.lightCount; l++) { all += hostTotals[2 * l]; most = std::max(most, hostTotals[2 * l]); }
            const unsigned long long bytes = (unsigned long long)indexBytes + (all + 1) * sizeof(uint32_t);
            if (verbose) {
                std::fprintf(stderr, "light grid: resolution %u, %zu buckets, %llu references, %llu bytes while building (budget %zu), always:",
                             resolution, perLight * in.lightCount, all, bytes, budget);
                for (uint32_t c : hostAlwaysCount) std::fprintf(stderr, " %u", c);
                std::fprintf(stderr, "; list length met by a reference:");
                for (uint32_t l = 0; l < in.lightCount; l++) std::fprintf(stderr, " %.1f", hostTotals[2 * l] ? (double)hostTotals[2 * l + 1] / (double)hostTotals[2 * l] : 0.0);
                std::fprintf(stderr, "\n");
            }
            if (verbose) {
                uint32_t* longest = nullptr;
                uint32_t hostLongest[7] = {};
                if (hipMalloc(&longest, sizeof hostLongest) == hipSuccess) {
                    (void)hipMemset(longest, 0, sizeof hostLongest);
                    hipLaunchKernelGGL(longest_kernel, dim3(1024), dim3(256), 0, nullptr, counts, slots, longest);
                    (void)hipMemcpy(hostLongest, longest, sizeof hostLongest, hipMemcpyDeviceToHost);
                    (void)hipFree(longest);
                    std::fprintf(stderr, "light grid: longest list %u entries; lists longer than 16/32/64/128/256/512: %u %u %u %u %u %u\n", hostLongest[0],
                                 hostLongest[1], hostLongest[2], hostLongest[3], hostLongest[4], hostLongest[5], hostLongest[6]);
                }
            }
            if (most <= RT_GRID_MAX_REFERENCES && bytes <= budget) break;
            (void)hipFree(counts); counts = nullptr;
        }
        if (resolution <= RT_GRID_MIN_RESOLUTION) {       /* no room for grids: the rays walk the hierarchy */
            (void)hipFree(always); (void)hipFree(alwaysCount); (void)hipFree(totals); (void)hipFree(farthest);
            return 0;
        }
    }
    /* how long each light's longest list is: the index keeps a list's length in one byte */
    GRID_TRY(hipMalloc(&longestOfLight, in.lightCount * sizeof(uint32_t)));
    GRID_TRY(hipMemset(longestOfLight, 0, in.lightCount * sizeof(uint32_t)));
    hipLaunchKernelGGL(longest_per_light_kernel, dim3(256, in.lightCount), dim3(256), 0, nullptr, counts, perLight + 1, longestOfLight);
    GRID_TRY(hipGetLastError());
    hostLongestOfLight.resize(in.lightCount);
    GRID_TRY(hipMemcpy(hostLongestOfLight.data(), longestOfLight, in.lightCount * sizeof(uint32_t), hipMemcpyDeviceToHost));
    /* each light's counts -> its starts (offsets into the light's own stretch of refs), in place */
    GRID_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tempBytes, counts, counts, perLight + 1));
    GRID_TRY(hipMalloc(&temp, tempBytes ? tempBytes : 16));
    for (uint32_t l = 0; l < in.lightCount; l++)
        GRID_TRY(hipcub::DeviceScan::ExclusiveSum(temp, tempBytes, counts + (size_t)l * (perLight + 1), counts + (size_t)l * (perLight + 1), perLight + 1));
    /* ... -> the index the kernels read, before the starts are used up as write cursors */
    GRID_TRY(hipMalloc(&blocks, (size_t)in.lightCount * blocksPerLight * 3 * sizeof(uint32_t)));
    hipLaunchKernelGGL(pack_kernel, dim3((blocksPerLight + 255u) / 256u, in.lightCount), dim3(256), 0, nullptr, counts, perLight, blocksPerLight, blocks);
    GRID_TRY(hipGetLastError());
    /* two spare entries behind the last list, both reference 0: the kernels read one entry and one record ahead */
    GRID_TRY(hipMalloc(&refs, ((size_t)all + 2) * sizeof(uint32_t)));
    GRID_TRY(hipMemset(refs + all, 0, 2 * sizeof(uint32_t)));
    info.resize((size_t)in.lightCount * RT_GRID_INFO_WORDS);
    out.enabledLights = 0;
    {
        unsigned long long at = 0;
        for (uint32_t l = 0; l < in.lightCount; l++) {
            /* a grid pays while its lists are short: the list a ray meets is as long as the one an average reference
             * sits in, plus the always list */
            const double expected = (hostTotals[2 * l] ? (double)hostTotals[2 * l + 1] / (double)hostTotals[2 * l] : 0.0) + hostAlwaysCount[l];
            const bool enabled = hostAlwaysCount[l] <= RT_GRID_ALWAYS && expected <= maxExpected && hostLongestOfLight[l] <= 255u;
            info[RT_GRID_INFO_WORDS * l] = enabled ? 1u : 0u;
            info[RT_GRID_INFO_WORDS * l + 1] = enabled ? hostAlwaysCount[l] : 0u;
            info[RT_GRID_INFO_WORDS * l + 2] = (uint32_t)at;
            info[RT_GRID_INFO_WORDS * l + 3] = (uint32_t)(at >> 32);
            /* the light's distance levels: the farthest listed primitive lies in the last one */
            float reach = 0.0f, perUnit = 0.0f;
            std::memcpy(&reach, &hostFarthest[l], sizeof reach);
            perUnit = (float)(GRID_REF_LEVELS - 1u) / std::max(reach, 1e-3f);
            std::memcpy(&info[RT_GRID_INFO_WORDS * l + 4], &perUnit, sizeof perUnit);
            at += hostTotals[2 * l];
            out.enabledLights += enabled ? 1u : 0u;
        }
    }
    if (out.enabledLights == 0) {       /* nothing gained: every ray would walk the hierarchy anyway */
        (void)hipFree(temp); (void)hipFree(counts); (void)hipFree(blocks); (void)hipFree(refs); (void)hipFree(always); (void)hipFree(alwaysCount); (void)hipFree(totals);
        (void)hipFree(longestOfLight); (void)hipFree(farthest);
        out = LightGridBuild{};
        return 0;
    }
    GRID_TRY(hipMalloc(&lights, info.size() * sizeof(uint32_t)));
    GRID_TRY(hipMemcpy(lights, info.data(), info.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    /* the starts are the write cursors of the filling pass */
    b.refs = refs; b.info = lights;
    hipLaunchKernelGGL(grid_kernel<1>, grid, dim3(RT_GRID_BLOCK), 0, nullptr, b);
    GRID_TRY(hipGetLastError());
    /* every list nearest first (the cursors now stand at the lists' ends) */
    hipLaunchKernelGGL(sort_lists_kernel, dim3((unsigned)((perLight + 255u) / 256u), in.lightCount), dim3(256), 0, nullptr, counts, perLight, lights, refs);
    GRID_TRY(hipGetLastError());
    GRID_TRY(hipDeviceSynchronize());
    if (std::getenv("RT_LIGHT_GRID_CHECK")) {
        /* diagnosis: every list within its light's stretch, sorted, and every entry a primitive of the scene */
        std::vector<uint32_t> ends(slots), words((size_t)all + 2);
        GRID_TRY(hipMemcpy(ends.data(), counts, slots * sizeof(uint32_t), hipMemcpyDeviceToHost));
        GRID_TRY(hipMemcpy(words.data(), refs, ((size_t)all + 2) * sizeof(uint32_t), hipMemcpyDeviceToHost));
        unsigned long long at = 0, unsorted = 0, outside = 0, wrongEnd = 0, levels[4] = {0, 0, 0, 0};
        for (uint32_t l = 0; l < in.lightCount; l++) {
            const uint32_t* run = ends.data() + (size_t)l * (perLight + 1);
            uint32_t begin = 0;
            for (size_t g = 0; g < perLight; g++) {
                const uint32_t end = run[g];
                if (end < begin || end > hostTotals[2 * l]) { wrongEnd++; begin = end; continue; }
                for (uint32_t i = begin; i < end; i++) {
                    const uint32_t word = words[at + i];
                    if (info[RT_GRID_INFO_WORDS * l] != 0u && i > begin && words[at + i - 1] > word) unsorted++;
                    if ((word & GRID_REF_INDEX) >= in.entryCount) outside++;
                    levels[(word >> GRID_REF_LEVEL_SHIFT) / 32u]++;
                }
                begin = end;
            }
            if (run[perLight] != hostTotals[2 * l] || begin != hostTotals[2 * l]) wrongEnd++;
            at += hostTotals[2 * l];
        }
        std::fprintf(stderr, "light grid check: %llu references, %llu out of order, %llu outside the scene, %llu lists out of place; levels by quarter: %llu %llu %llu %llu\n",
                     all, unsorted, outside, wrongEnd, levels[0], levels[1], levels[2], levels[3]);
        for (uint32_t l = 0; l < in.lightCount; l++) {
            float perUnit; std::memcpy(&perUnit, &info[RT_GRID_INFO_WORDS * l + 4], sizeof perUnit);
            std::fprintf(stderr, "light grid check: light %u: %g levels per unit length\n", l, perUnit);
        }
    }
    (void)hipFree(farthest); farthest = nullptr;
    (void)hipFree(temp); temp = nullptr;
    (void)hipFree(counts); counts = nullptr;
    (void)hipFree(alwaysCount); alwaysCount = nullptr;
    (void)hipFree(totals); totals = nullptr;
    (void)hipFree(longestOfLight); longestOfLight = nullptr;

    out.view.blocks = blocks;
    out.view.refs = refs;
    out.view.always = always;
    out.view.info = lights;
    out.view.resolution = resolution;
    out.view.halfResolution = 0.5f * (float)resolution;
    out.view.blocksPerLight = blocksPerLight;
    out.view.reach = RT_GRID_REACH * in.extent;
    out.referenceCount = all;
    out.bytes = (size_t)in.lightCount * blocksPerLight * 3 * sizeof(uint32_t) + ((size_t)all + 2) * sizeof(uint32_t) +
                (size_t)in.lightCount * (RT_GRID_ALWAYS + RT_GRID_INFO_WORDS) * sizeof(uint32_t);
    out.peakBytes = out.bytes + slots * sizeof(uint32_t) + tempBytes;
    return 0;

failed:
    if (temp) (void)hipFree(temp);
    if (counts) (void)hipFree(counts);
    if (blocks) (void)hipFree(blocks);
    if (longestOfLight) (void)hipFree(longestOfLight);
    if (refs) (void)hipFree(refs);
    if (always) (void)hipFree(always);
    if (alwaysCount) (void)hipFree(alwaysCount);
    if (totals) (void)hipFree(totals);
    if (lights) (void)hipFree(lights);
    if (farthest) (void)hipFree(farthest);
    out = LightGridBuild{};
    if (failure == hipErrorOutOfMemory) {       /* someone else took the memory meanwhile: the scene works without grids */
        (void)hipGetLastError();
        error.clear();
        return 0;
    }
    return -1;
}

void rt_light_grid_release(LightGridBuild& grid) {
    if (grid.view.blocks) (void)hipFree(const_cast<uint32_t*>(grid.view.blocks));
    if (grid.view.refs) (void)hipFree(const_cast<uint32_t*>(grid.view.refs));
    if (grid.view.always) (void)hipFree(const_cast<uint32_t*>(grid.view.always));
    if (grid.view.info) (void)hipFree(const_cast<uint32_t*>(grid.view.info));
    grid = LightGridBuild{};
}
