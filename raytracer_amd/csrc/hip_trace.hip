/*
 * hip_trace.hip -- trace kernels for gfx950 and the hip_trace C ABI (include/hip_trace.h).
 *
 * One work-item per pixel, one 16x16 screen tile per 256-thread workgroup, each of its four
 * wavefronts owning an 8x8 sub-tile so that the 64 rays of a wave start out coherent.
 *
 * Arithmetic contract.  Everything that decides WHICH primitive a ray meets, and every
 * quantity a later ray is built from (hit distance, hit point, normals, reflected and
 * refracted directions, shadow-ray set-up) evaluates the reference's fp32 expressions in
 * the reference's operand order with no fused multiply-add (the file is compiled with
 * -ffp-contract=off; hipcc's default correctly rounded fp32 divide and sqrt are relied on),
 * so those values carry the same bits as the reference's CPU tracer
 * (reference: src/raytracer.c:9-314, src/utils/vec3.c:4-106).  Fused multiply-add is used
 * where only culling is at stake: the ray/box slab tests, written with __builtin_fmaf.
 *
 * Hierarchy walk.  Per ray, a depth-first walk over 8-wide nodes of quantised child boxes
 * (device_layout.h: QNode; built from the scene's primitives or re-packed from the caller's
 * octree, see flatten.cpp): all eight child boxes of a node are tested at once, the survivors
 * are visited front to back (slot order from the ray's direction signs), and a child whose box
 * is entered beyond the best hit so far is skipped.  The walk keeps one (node,
 * pending-children) word per tree level in LDS, so its footprint is treeLevels * 4 bytes per
 * ray instead of the 200/250-entry private stacks of the reference's device walk (reference:
 * src/kernel.cl:504-601).  Triangles are first rejected by a conservative fused-arithmetic
 * filter on a 64-byte record stored in leaf order; only survivors run the exact test.
 *
 * Shadow rays of a whole tile are pooled in an LDS job queue (shadow_pass) and pulled by
 * whichever lane is free, so a wavefront does not idle behind its longest shadow walk.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "hip_trace.h"
#include "flatten.h"
#include "light_grid.h"

#define RT_EPSILON 0.00001f   /* reference: src/raytracer.h:8 */
/* How the shadow pass shares a wavefront between its three bodies depends on how many leaves a ray
 * opens.  Over the library's own hierarchy (1.7 leaves per ray) the best schedule is the plainest:
 * walkers step until every running lane holds a leaf, then all held leaves are tested together, and
 * ray set-up waits for 40 idle lanes.  Over an octree's topology (4.3 leaves per ray, many of them
 * duplicates) a lane walks on with up to three leaves in hand, leaves are tested as soon as a
 * quarter of the running lanes are stuck with them, and set-up runs for 16 idle lanes.  MANY_LEAVES
 * selects the second schedule (measured on the 1M-triangle scene: 9.6 vs 7.95 ms the wrong way round
 * for the hierarchy, 36.9 vs 33.2 ms for the octree). */
template <bool MANY_LEAVES> struct ShadowSchedule {
    static constexpr uint32_t leafSlots = MANY_LEAVES ? 3u : 1u;
#ifndef RT_NEED_BATCH
#define RT_NEED_BATCH 40u
#endif
    static constexpr uint32_t needBatch = MANY_LEAVES ? 16u : RT_NEED_BATCH;
    /* start every ray with the leaf that last hid a ray towards the same light: 32.9 -> 31.6 ms over
     * the octree, 7.99 -> 8.38 ms over the hierarchy, whose whole walk costs little more than the guess */
    static constexpr bool guessOccluder = MANY_LEAVES;
};
#define RT_BLOCK 256
#define RT_PLANES_IN_LDS 32u
/* Planes that cannot lie between a point and a light.  A shadow ray runs from a point P (moved 1/1000 along the ray) to
 * a light L and a plane hides L iff it is met at 0 < t <= |L - P| (reference: src/raytracer.c:70-86, 286-292).  When P
 * and L are on the same side of the plane n . x + d = 0, both by more than RT_SIDE_MARGIN (in units of n . x + d), the
 * crossing of the ray's line lies behind P or at least 0.004 beyond |L - P| -- ten times what the fp32 evaluation of t
 * can be off by for coordinates up to 400 and normals of length 0.5 to 2 -- so the test is skipped.  Sides are bit masks:
 * bit i = in front of plane i, bit 16 + i = behind it, for the first sixteen planes; a point's are worked out when it is
 * stored for the shadow pass, a light's at upload. */
#define RT_SIDE_PLANES 16u
#define RT_SIDE_MARGIN 1e-2f
#define RT_SIDE_REACH 400.0f      /* points with a larger |coordinate| claim no side */
#define RT_PARK_SLOTS 13u      /* words a thread parks in LDS across the shadow pass (trace_kernel) */
/* LDS of a workgroup of trace_kernel that does not depend on the scene: the parked words, shading points, occlusion
 * masks, point list, tree positions, plane sides, last occluders, ten counters and flags (padded to 16 bytes), the root's
 * rows and the planes -- in this order; the walk stacks follow. */
#define RT_FIXED_LDS_BYTES (RT_PARK_SLOTS * 256u * 4u + 256u * 24u + 256u + 48u + 80u + RT_PLANES_IN_LDS * 16u)
#define RT_DEFAULT_SLACK 1e-4f   /* DeviceScene::filterMargin of a scene whose coordinates stay below 100 */
#ifndef RT_ORDER_BELOW
#define RT_ORDER_BELOW 20000u  /* local tile count up to which tiles are launched longest first: half a 4K frame
                                  (16200 tiles) gains 2 %, a whole one (32400) loses 2.7 % */
#endif

/* Profiling build (-DRT_WALK_STATS): thirty-four more device counters describing the walk itself:
 * per-lane counts (wide nodes whose eight boxes were tested, leaves opened, triangle filter
 * tests, exact triangle tests), per-wavefront counts of the same four bodies, a histogram
 * of shadow-walk lengths in powers of two, the non-empty child boxes among the nodes tested, and
 * the shader-clock cycles each wavefront spent in closest-hit search, shadow pass, shading and
 * waiting at workgroup barriers.  hip_trace_walk_stats() returns them. */
#ifdef RT_WALK_STATS
#define RT_COUNTERS 40
#define RT_STAT(slot) (walkStats[slot]++)
#define RT_STAT_ADD(slot, n) (walkStats[slot] += (n))
enum { RT_STAT_CHILD_BOXES = 16, RT_STAT_CYCLES_CLOSEST = 17, RT_STAT_CYCLES_SHADOW = 18, RT_STAT_CYCLES_SHADING = 19, RT_STAT_CYCLES_BARRIER = 20,
       /* shadow pass, summed over its node steps: the steps, and the lanes that were walking / holding a leaf / waiting for a ray / finished */
       RT_STAT_SHADOW_STEPS = 21, RT_STAT_STEP_WALKING = 22, RT_STAT_STEP_HOLDING = 23, RT_STAT_STEP_WAITING = 24, RT_STAT_STEP_FINISHED = 25,
       /* shadow pass through the light grids: turns in which a wavefront tested list entries, and the lanes that had one */
       RT_STAT_LIST_TURNS = 26, RT_STAT_LIST_LANES = 27,
       /* the walk a wavefront shares (shared_walk): its node tests, (a spare slot), the triangles of
        * the leaves it tested together, the nodes handed to single lanes, the times the lanes went off on their own, its leaves */
       RT_STAT_SHARED_STEPS = 28, RT_STAT_SHARED_UNUSED = 29, RT_STAT_SHARED_TRIANGLES = 30, RT_STAT_SHELVED = 31, RT_STAT_DRAINS = 32,
       RT_STAT_SHARED_LEAVES = 33 };
/* time since the previous mark goes to `slot`, counted once per wavefront */
#define RT_PHASE_START() long long phaseMark = clock64()
#define RT_PHASE(slot) do { const long long now_ = clock64(); if ((threadIdx.x & 63u) == 0u) walkStats[slot] += (uint32_t)(now_ - phaseMark); phaseMark = now_; } while (0)
/* wave-level executions of a body: counted by the lowest lane that runs it */
#define RT_WAVE_STAT(slot) do { if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)__ballot(1)) - 1)) walkStats[slot]++; } while (0)
#else
#define RT_COUNTERS 4
#define RT_STAT(slot) ((void)0)
#define RT_STAT_ADD(slot, n) ((void)0)
#define RT_PHASE_START() ((void)0)
#define RT_PHASE(slot) ((void)0)
#define RT_WAVE_STAT(slot) ((void)0)
#endif

/* ================================================================= device code */

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
__device__ __forceinline__ V3 vadd(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 vsub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 vmul(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 vcross(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float vlength(V3 a) { return __builtin_sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
/* reference: src/utils/vec3.c:56-65 */
__device__ __forceinline__ V3 vnorm(V3 a) {
    float len = vlength(a);
    if (len != 0) { a.x /= len; a.y /= len; a.z /= len; }
    return a;
}
/* reference: src/utils/vec3.c:100-106 */
__device__ __forceinline__ V3 vreflect(V3 incoming, V3 normal) {
    V3 reversed = vmul(incoming, -1.0f);
    return vnorm(vsub(vmul(normal, 2.0f * vdot(normal, reversed)), reversed));
}
/* reference: src/utils/math.c:3-5 */
__device__ __forceinline__ float clampf(float in, float lo, float hi) { return in < lo ? lo : (in > hi ? hi : in); }

struct RayD { V3 o, d; };

/* Per-ray constants of the box tests. */
struct BoxWalk {
    float ix, iy, iz;        /* 1 / direction, |direction| clamped away from 0        */
    float ox, oy, oz;        /* ray origin                                            */
    uint32_t octant;         /* bit0: direction.x < 0, bit1: z < 0, bit2: y < 0 -- the XOR mask
                                turning "k-th nearest" into a child number, and the choice
                                of entry/exit plane per axis                          */
    float px, py, pz;        /* PADDED walks only: how far, in ray parameter, every box is widened per axis
                                (a distance in space times |1 / direction|)            */
};

/* 1 / d for the box tests only (culling: the hardware reciprocal's 1 ulp is far inside the
 * padding of the boxes), with |d| kept away from zero so that no 0 * inf can arise. */
__device__ __forceinline__ float safe_inverse(float d) {
    float a = fabsf(d) < 1e-30f ? copysignf(1e-30f, d) : d;
    return __builtin_amdgcn_rcpf(a);
}

/* pad: every box is treated as this much larger in every direction (0 where the padding applied at upload suffices).
 * The boxes are padded at upload for rays that start within the scene; the rounding of the fused slab arithmetic grows
 * with the distance of the ray's origin from the boxes, so frames whose camera lies far outside the scene widen the
 * boxes by a distance that follows the camera (the general kernel only). */
__device__ __forceinline__ BoxWalk make_walk(const RayD& r, float pad = 0.0f) {
    BoxWalk w;
    w.ix = safe_inverse(r.d.x); w.iy = safe_inverse(r.d.y); w.iz = safe_inverse(r.d.z);
    w.ox = r.o.x; w.oy = r.o.y; w.oz = r.o.z;
    w.octant = (w.ix < 0 ? 1u : 0u) | (w.iz < 0 ? 2u : 0u) | (w.iy < 0 ? 4u : 0u);
    w.px = pad * fabsf(w.ix); w.py = pad * fabsf(w.iy); w.pz = pad * fabsf(w.iz);
    return w;
}

/* A record of the node array by 32-bit byte offset from its (wave-uniform) base: the address is
 * then a scalar base plus one vector register, not a 64-bit sum built per lane. */
template <typename T>
__device__ __forceinline__ const T* node_bytes(const QNode* nodes, uint32_t byteOffset) {
    return reinterpret_cast<const T*>(reinterpret_cast<const char*>(nodes) + byteOffset);
}

__device__ __forceinline__ float byte_of(uint32_t word, int k) { return (float)((word >> (8 * k)) & 0xFFu); }


/* Which of the eight children of `node` can hold a hit nearer than tmax.  ORDERED: bit k of the
 * result is the k-th child in front-to-back order (child number k ^ octant); otherwise bit c is
 * child c -- an occlusion query gains nothing from the order and saves the shuffle (7.59 -> 7.41 ms).
 * Slab test on the quantised boxes: plane = origin + scale * q, so
 * t = (plane - o) / d = q * (scale / d) + (origin - o) / d, one fused multiply-add per plane. */
template <bool ORDERED, bool PADDED>
__device__ __forceinline__ uint32_t slab_tests(const uint4 q0, const uint4 q1, const uint4 q2, const uint4 grid, const uint4 tail,
                                               const BoxWalk& w, float tmax, uint32_t& firstInner, uint32_t& innerMask) {
    const float4 g0 = make_float4(__uint_as_float(grid.x), __uint_as_float(grid.y), __uint_as_float(grid.z), __uint_as_float(grid.w));
    const float2 g1 = make_float2(__uint_as_float(tail.x), __uint_as_float(tail.y));
    firstInner = tail.z;
    innerMask = tail.w;
    const bool negX = w.octant & 1u, negZ = w.octant & 2u, negY = w.octant & 4u;
    /* packed rows: children 0-3 in word 0, children 4-7 in word 1 */
    const uint32_t nearX[2] = {negX ? q1.z : q0.x, negX ? q1.w : q0.y}, farX[2] = {negX ? q0.x : q1.z, negX ? q0.y : q1.w};
    const uint32_t nearY[2] = {negY ? q2.x : q0.z, negY ? q2.y : q0.w}, farY[2] = {negY ? q0.z : q2.x, negY ? q0.w : q2.y};
    const uint32_t nearZ[2] = {negZ ? q2.z : q1.x, negZ ? q2.w : q1.y}, farZ[2] = {negZ ? q1.x : q2.z, negZ ? q1.y : q2.w};
    const float ax = g0.w * w.ix, ay = g1.x * w.iy, az = g1.y * w.iz;
    const float bx = (g0.x - w.ox) * w.ix, by = (g0.y - w.oy) * w.iy, bz = (g0.z - w.oz) * w.iz;
    /* entry planes come earlier and exit planes later by the padding */
    const float bxn = PADDED ? bx - w.px : bx, byn = PADDED ? by - w.py : by, bzn = PADDED ? bz - w.pz : bz;
    const float bxf = PADDED ? bx + w.px : bx, byf = PADDED ? by + w.py : by, bzf = PADDED ? bz + w.pz : bz;
    /* the sign bit of tf - tn is set exactly when the child is missed (all values are finite);
     * shifted in from the right, child 7 first, so that bit c ends up belonging to child c */
    uint32_t miss = 0;
#pragma unroll
    for (int c = 7; c >= 0; c--) {
        const int h = c >> 2, k = c & 3;
        const float tn = fmaxf(fmaxf(__builtin_fmaf(byte_of(nearX[h], k), ax, bxn), __builtin_fmaf(byte_of(nearY[h], k), ay, byn)),
                               fmaxf(__builtin_fmaf(byte_of(nearZ[h], k), az, bzn), 0.0f));
        const float tf = fminf(fminf(__builtin_fmaf(byte_of(farX[h], k), ax, bxf), __builtin_fmaf(byte_of(farY[h], k), ay, byf)),
                               fminf(__builtin_fmaf(byte_of(farZ[h], k), az, bzf), tmax));
        miss = __builtin_amdgcn_alignbit(miss, __float_as_uint(tf - tn), 31);
    }
    uint32_t mask = ~miss & 0xFFu;
    if (!ORDERED) return mask;
    if (w.octant & 1u) mask = ((mask & 0xAAu) >> 1) | ((mask & 0x55u) << 1);
    if (w.octant & 2u) mask = ((mask & 0xCCu) >> 2) | ((mask & 0x33u) << 2);
    if (w.octant & 4u) mask = ((mask & 0xF0u) >> 4) | ((mask & 0x0Fu) << 4);
    return mask;
}

/* How many of a node's eight children exist: an empty child has a minimum above its maximum
 * (x rows: minimum corners in q0.x/q0.y, maximum corners in q1.z/q1.w, one byte per child). */
__device__ __forceinline__ uint32_t nonempty_children(const uint4 q0, const uint4 q1) {
    uint32_t n = 0;
    for (int c = 0; c < 8; c++) {
        const uint32_t lo = ((c < 4 ? q0.x : q0.y) >> (8 * (c & 3))) & 0xFFu, hi = ((c < 4 ? q1.z : q1.w) >> (8 * (c & 3))) & 0xFFu;
        n += lo <= hi ? 1u : 0u;
    }
    return n;
}

template <bool ORDERED, bool PADDED>
__device__ __forceinline__ uint32_t children_entered(const QNode* nodes, uint32_t index, const BoxWalk& w, float tmax,
                                                     uint32_t& firstInner, uint32_t& innerMask, uint32_t* walkStats) {
    const uint32_t at = index * (uint32_t)sizeof(QNode);      /* fewer than 2^24 nodes: fits 32 bits */
    const uint4 q0 = *node_bytes<uint4>(nodes, at), q1 = *node_bytes<uint4>(nodes, at + 16u), q2 = *node_bytes<uint4>(nodes, at + 32u);
    const uint4 grid = *node_bytes<uint4>(nodes, at + 48u);     /* origin, scale.x */
    const uint4 tail = *node_bytes<uint4>(nodes, at + 64u);     /* scale.y, scale.z, firstInner, innerMask */
    RT_STAT_ADD(RT_STAT_CHILD_BOXES, nonempty_children(q0, q1));
    return slab_tests<ORDERED, PADDED>(q0, q1, q2, grid, tail, w, tmax, firstInner, innerMask);
}

/* Entry distance of one child (for re-checking a leaf against a best hit that has improved
 * since its parent's boxes were tested). */
template <bool PADDED>
__device__ __forceinline__ bool child_still_entered(const QNode* node, int c, const BoxWalk& w, float tmax) {
    const float4 g0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(node) + 48);
    const float2 g1 = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(node) + 64);
    const float qx = (float)((w.octant & 1u) ? node->qhi[0][c] : node->qlo[0][c]);
    const float qy = (float)((w.octant & 4u) ? node->qhi[1][c] : node->qlo[1][c]);
    const float qz = (float)((w.octant & 2u) ? node->qhi[2][c] : node->qlo[2][c]);
    const float tx = __builtin_fmaf(qx, g0.w * w.ix, (g0.x - w.ox) * w.ix) - (PADDED ? w.px : 0.0f);
    const float ty = __builtin_fmaf(qy, g1.x * w.iy, (g0.y - w.oy) * w.iy) - (PADDED ? w.py : 0.0f);
    const float tz = __builtin_fmaf(qz, g1.y * w.iz, (g0.z - w.oz) * w.iz) - (PADDED ? w.pz : 0.0f);
    return fmaxf(fmaxf(tx, ty), tz) <= tmax;
}

/* reference: src/raytracer.c:70-86 */
__device__ __forceinline__ bool hit_plane(const Float4& p, const RayD& r, float& t) {
    V3 n = mk(p.x, p.y, p.z);
    float denominator = vdot(n, r.d);
    if ((denominator < -RT_EPSILON) || (denominator > RT_EPSILON)) {
        float cosAngle = vdot(n, r.o);
        t = (-p.w - cosAngle) / denominator;
        return t > 0;
    }
    return false;
}

/* reference: src/raytracer.c:88-116; `a` is dot(direction, direction), the same for every
 * sphere a ray meets.  sqrt of a negative discriminant is NaN and fails the comparison. */
__device__ __forceinline__ bool hit_sphere(const Float4& s, const RayD& r, float a, float& t) {
    V3 rel = vsub(r.o, mk(s.x, s.y, s.z));
    float b = 2.0f * vdot(r.d, rel);
    float c = vdot(rel, rel) - s.w * s.w;
    float denominator = 2.0f * a;
    float root = __builtin_sqrtf(b * b - 4.0f * a * c);
    if (root > RT_EPSILON) {
        float tpos = (-b + root) / denominator;
        float tneg = (-b - root) / denominator;
        t = tpos;
        if ((tneg > 0) && (tneg < tpos)) t = tneg;
        return t > 0;
    }
    return false;
}

/* reference: src/raytracer.c:122-128 with the normal and its offset read instead of recomputed. */
__device__ __forceinline__ bool triangle_plane_distance(const Float4& pl, const RayD& r, float& t) {
    V3 n = mk(pl.x, pl.y, pl.z);
    float normalDotDir = vdot(n, r.d);
    if (fabsf(normalDotDir) < RT_EPSILON) return false;
    t = -(vdot(n, r.o) - pl.w) / normalDotDir;
    return t > 0;
}

/* reference: src/raytracer.c:130-154 -- the three inside-outside edge tests. */
__device__ __forceinline__ bool triangle_contains(const Float4* corner, const Float4& pl, const RayD& r, float t) {
    V3 n = mk(pl.x, pl.y, pl.z);
    V3 hitPoint = vadd(r.o, vmul(r.d, t));
    Float4 c0 = corner[0], c1 = corner[1], c2 = corner[2];
    V3 v0 = mk(c0.x, c0.y, c0.z), v1 = mk(c1.x, c1.y, c1.z), v2 = mk(c2.x, c2.y, c2.z);
    V3 c;
    c = vcross(vsub(v1, v0), vsub(hitPoint, v0));
    if (vdot(n, c) < 0) return false;
    c = vcross(vsub(v2, v1), vsub(hitPoint, v1));
    if (vdot(n, c) < 0) return false;
    c = vcross(vsub(v0, v2), vsub(hitPoint, v2));
    if (vdot(n, c) < 0) return false;
    return true;
}

/* Conservative test of one leaf triangle, in fused arithmetic.  Returns TRI_MISS only
 * when the exact test (triangle_plane_distance + triangle_contains, which follow the reference
 * operation by operation) cannot report a hit nearer than `limit`: every comparison leaves a
 * margin that bounds the difference between this evaluation and the exact one.
 *   num, den differ from the exact dots by a few ulp of their largest term (coordinates up to the
 *   scene's extent E: a few 1e-7 * E; `slack` = 1e-6 * E, at least 1e-4), so t differs by <= slack / |den| (+ the
 *   reciprocal's relative error), the hit point by the same amount (|d| = 1), and each edge
 *   distance by that plus rounding.
 * A ray almost parallel to the plane (|den| < 2e-5, where the reference's own cut-off at 1e-5
 * lies) is passed on unexamined. */
enum { TRI_MISS = 0, TRI_MAYBE = 1, TRI_SURE = 2 };
template <bool ANY_HIT>
__device__ __forceinline__ int triangle_may_hit(const LeafTriangle& tri, const RayD& r, float limit, float slack) {
    /* straight-line on purpose: in a wavefront some lane nearly always passes each stage, so early
     * exits only add mask bookkeeping, and they make the compiler sink the record's loads into them */
    const Float4 pl = tri.plane;
    const float den = __builtin_fmaf(pl.x, r.d.x, __builtin_fmaf(pl.y, r.d.y, pl.z * r.d.z));
    const bool examined = fabsf(den) >= 2e-5f;
    const float num = pl.w - __builtin_fmaf(pl.x, r.o.x, __builtin_fmaf(pl.y, r.o.y, pl.z * r.o.z));
    const float inv = __builtin_amdgcn_rcpf(den);
    const float t = num * inv;
    const float margin = __builtin_fmaf(slack, fabsf(inv), __builtin_fmaf(4e-6f, fabsf(t), slack));
    const bool outOfRange = (t < -margin) | (t > limit + margin);
    const float px = __builtin_fmaf(r.d.x, t, r.o.x), py = __builtin_fmaf(r.d.y, t, r.o.y), pz = __builtin_fmaf(r.d.z, t, r.o.z);
    const Float4 e0 = tri.edge[0], e1 = tri.edge[1], e2 = tri.edge[2];
    const float d0 = __builtin_fmaf(e0.x, px, __builtin_fmaf(e0.y, py, __builtin_fmaf(e0.z, pz, -e0.w)));
    const float d1 = __builtin_fmaf(e1.x, px, __builtin_fmaf(e1.y, py, __builtin_fmaf(e1.z, pz, -e1.w)));
    const float d2 = __builtin_fmaf(e2.x, px, __builtin_fmaf(e2.y, py, __builtin_fmaf(e2.z, pz, -e2.w)));
    /* fminf skips a NaN operand, and so did the comparison each distance used to get on its own */
    const float nearest = fminf(d0, fminf(d1, d2));
    const bool outside = nearest < -margin;
    /* The same bound read the other way: inside every edge by more than the margin, and the distance
     * inside (0, limit] by more than the margin, means the exact test reports a hit no farther than
     * `limit`.  An occlusion query needs nothing more than that. */
    const bool sure = ANY_HIT & (nearest > margin) & (t > margin) & (t < limit - margin);
    const int verdict = (outOfRange | outside) ? TRI_MISS : sure ? TRI_SURE : TRI_MAYBE;
    return examined ? verdict : TRI_MAYBE;
}

/* Does (t, id) beat the best so far?  The reference scans planes, spheres, triangles in index
 * order and replaces on strict < (reference: src/raytracer.c:170,186,202), so among equal
 * distances the smallest id wins; the walk meets primitives in another order, hence the
 * explicit pair comparison. */
__device__ __forceinline__ bool beats(float t, uint32_t id, float bestT, uint32_t bestId) {
    return t < bestT || (t == bestT && id < bestId);
}

#ifndef RT_ORDERED_SHADOWS
#define RT_ORDERED_SHADOWS 0   /* 1: occlusion walks visit children front to back as well (tuning aid) */
#endif
struct WalkStack {
    const uint4* root;  /* LDS copy of the first 80 bytes of node 0: every ray starts there */
    const Float4* plane; /* LDS copy of the first RT_PLANES_IN_LDS planes: every ray tests them, one after another */
    uint32_t* entry;    /* LDS, [level * RT_BLOCK + thread]: node (low 24 bits) and its children still to visit (high 8);
                           the node's firstInner and innerMask are read again from the node itself at the pop */
    uint32_t* trail;    /* LDS, [wavefront * levels + level]: the same for the walk a wavefront shares (shared_walk) */
    uint32_t* shelf;    /* LDS, [slot * RT_BLOCK + thread], RT_SHELF_SLOTS slots: nodes the shared walk handed to this lane */
    uint32_t levels;    /* entries per stack */
};
#define RT_STACK_BYTES_PER_LEVEL sizeof(uint32_t)
#define RT_MAX_WIDE_NODES (1u << 24)   /* a stack entry keeps 24 bits of node number */

struct Closest { float t; uint32_t id; uint32_t material; };

/* Position of one ray's depth-first walk over the wide nodes. */
struct Walker { uint32_t level, cur, pending, firstInner, innerMask; };

template <bool ANY_HIT, bool PADDED>
__device__ __forceinline__ void walker_start(const DeviceScene& s, const WalkStack& st, const BoxWalk& w, float limit, Walker& k, uint32_t* walkStats) {
    k.level = 0; k.cur = 0;
    k.pending = slab_tests<!ANY_HIT || RT_ORDERED_SHADOWS, PADDED>(st.root[0], st.root[1], st.root[2], st.root[3], st.root[4], w, limit, k.firstInner, k.innerMask);
    RT_STAT(0);
    RT_STAT_ADD(RT_STAT_CHILD_BOXES, nonempty_children(st.root[0], st.root[1]));
}

/* One step of the walk: pop finished levels, take the next child front to back; an inner child
 * is entered (its eight boxes tested against `limit`), a leaf child is handed back.  Returns
 * 0 = moved on, 1 = leaf in (leafLink, leafKind), 2 = nothing left.
 * The node of an inner child follows from the parent's firstInner and innerMask, which came with
 * the parent's boxes: descending costs one dependent memory access per level, not two. */
template <bool ANY_HIT, bool PADDED>
__device__ __forceinline__ int walker_step(const DeviceScene& s, const BoxWalk& w, const WalkStack& st, float limit,
                                           Walker& k, uint32_t& leafLink, uint32_t& leafKind, uint32_t* walkStats) {
    const uint32_t tid = threadIdx.x;
    /* only nodes with children still to visit are on the stack, so one pop always yields work */
    if (k.pending == 0 && k.level > 0) {
        k.level--;
        const uint32_t entry = st.entry[k.level * RT_BLOCK + tid];
        k.cur = entry & 0xFFFFFFu;
        k.pending = entry >> 24;
        const uint2 numbering = *node_bytes<uint2>(s.nodes, k.cur * (uint32_t)sizeof(QNode) + 72u);
        k.firstInner = numbering.x;
        k.innerMask = numbering.y;
    }
    if (k.pending == 0) return 2;
    const int order = __builtin_ctz(k.pending);
    k.pending &= k.pending - 1;
    const uint32_t c = (ANY_HIT && !RT_ORDERED_SHADOWS) ? (uint32_t)order : (uint32_t)order ^ w.octant;
    if ((k.innerMask >> c) & 1u) {
        if (k.pending != 0) {
            st.entry[k.level * RT_BLOCK + tid] = k.cur | (k.pending << 24);
            k.level++;
        }
        k.cur = k.firstInner + (uint32_t)__builtin_popcount(k.innerMask & ((1u << c) - 1u));
        k.pending = children_entered<!ANY_HIT || RT_ORDERED_SHADOWS, PADDED>(s.nodes, k.cur, w, limit, k.firstInner, k.innerMask, walkStats);
        RT_STAT(0);
        RT_WAVE_STAT(4);
        return 0;
    }
    const QNode* node = &s.nodes[k.cur];
    /* closest hit: the best distance may have improved since this child's box was tested */
    if (!ANY_HIT && !child_still_entered<PADDED>(node, (int)c, w, limit)) return 0;
    const uint32_t at = k.cur * (uint32_t)sizeof(QNode);
    leafLink = *node_bytes<uint32_t>(s.nodes, at + 80u + 4u * c);
    leafKind = *node_bytes<uint16_t>(s.nodes, at + 112u + 2u * c);
    return 1;
}


/* Tests one leaf's primitives.  ANY_HIT: returns true as soon as one has t <= tmax.
 * Otherwise updates `best`. */
/* INDEXED: the filter records are stored once per triangle and found through the leaf's index run (a dependent load per
 * triangle), instead of once per leaf entry in leaf order.  Used for the caller's octree, whose leaves refer to every
 * triangle dozens of times: the per-entry copy of the 1M-triangle scene is 3.5 GB, the per-triangle one 64 MB. */
template <bool ANY_HIT, bool INDEXED>
__device__ __forceinline__ bool leaf_test(const DeviceScene& s, const RayD& r, uint32_t leafLink, uint32_t leafKind,
                                          Closest& best, float tmax, uint32_t* walkStats, float slack, bool strict = false) {
    const uint32_t sphereBase = s.planeCount, triangleBase = s.planeCount + s.sphereCount;
    const uint32_t sphereCount = leafKind >> 11, triangleCount = leafKind & 0x7FFu;
    RT_STAT(1);
    RT_WAVE_STAT(5);
    for (uint32_t i = 0; i < sphereCount; i++) {
        const uint32_t idx = s.leafIndex[leafLink + i];
        float t;
        if (hit_sphere(s.sphere[idx], r, vdot(r.d, r.d), t)) {
            if (ANY_HIT) { if (strict ? t < tmax : t <= tmax) return true; }
            else if (beats(t, sphereBase + idx, best.t, best.id)) {
                best.t = t; best.id = sphereBase + idx; best.material = s.sphereMaterial[idx];
            }
        }
    }
    /* triangles: a cheap conservative filter over the whole run first (all lanes busy with the
     * same short body), then the exact test only for the survivors, which are few */
    const uint32_t first = leafLink + sphereCount;
    for (uint32_t chunk = 0; chunk < triangleCount; chunk += 32u) {
        const uint32_t n = min(32u, triangleCount - chunk);
        const LeafTriangle* tri = s.leafTriangle + (INDEXED ? 0u : first + chunk);
        uint32_t candidates = 0;
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t at = INDEXED ? s.leafIndex[first + chunk + i] : i;
            LeafTriangle rec;
            rec.plane = tri[at].plane; rec.edge[0] = tri[at].edge[0]; rec.edge[1] = tri[at].edge[1]; rec.edge[2] = tri[at].edge[2];
            RT_STAT(2);
            RT_WAVE_STAT(6);
            const int verdict = triangle_may_hit<ANY_HIT>(rec, r, ANY_HIT ? tmax : best.t, slack);
            if (ANY_HIT && verdict == TRI_SURE) return true;
            if (verdict != TRI_MISS) candidates |= 1u << i;
        }
        while (candidates) {
            const uint32_t i = (uint32_t)__builtin_ctz(candidates);
            candidates &= candidates - 1;
            const uint32_t idx = s.leafIndex[first + chunk + i];
            Float4 pl = s.trianglePlane[idx];
            /* the plane and the three corners are fetched together: the exact test is rare, runs for
             * a few lanes only, and left alone it waits for memory four times in a row */
            Float4 corner[3] = {s.triangleCorner[3 * (size_t)idx], s.triangleCorner[3 * (size_t)idx + 1], s.triangleCorner[3 * (size_t)idx + 2]};
            asm volatile("" : "+v"(pl.x), "+v"(pl.w), "+v"(corner[0].x), "+v"(corner[1].x), "+v"(corner[2].x));
            float t;
            RT_STAT(3);
            RT_WAVE_STAT(7);
            if (!triangle_plane_distance(pl, r, t)) continue;
            const uint32_t id = triangleBase + idx;
            if (ANY_HIT) { if (!(strict ? t < tmax : t <= tmax)) continue; }
            else if (!beats(t, id, best.t, best.id)) continue;
            if (!triangle_contains(corner, pl, r, t)) continue;
            if (ANY_HIT) return true;
            best.t = t; best.id = id; best.material = s.triangleMaterial[idx];
        }
    }
    return false;
}

/* ---- wave-shared walk of coherent rays -------------------------------------------------------------------------------
 * The 64 rays of a wavefront that start out coherent (primary rays of an 8x8 block of pixels) visit mostly the same
 * nodes near the root.  For them the wavefront walks ONE stack: the node number is the same in every lane, so the node's
 * boxes arrive through scalar loads as floats (device_layout.h: WideNode) and each lane tests its own ray against them
 * with one fused multiply-add per slab plane and no byte conversions, no per-lane addresses, no entry/exit selects (the
 * rays share their direction signs, so the choice of rows is the wavefront's).  A child is visited by the wavefront when
 * at least RT_SHARE lanes enter it; children entered by fewer lanes are handed to
 * those lanes, which put (node, their own mask of such children) on a shelf in LDS and walk them on their own -- the
 * per-lane walk below, started at that node -- once half the wavefront has something shelved, a shelf is full or the
 * shared walk is through.  Results cannot depend on which way a primitive is reached: every lane still reaches every
 * primitive whose boxes its ray enters, and the exact tests and the (t, id) order decide. */
#ifndef RT_SHARE
#define RT_SHARE 16u
#endif
#ifndef RT_DRAIN_LANES
#define RT_DRAIN_LANES 32u
#endif
#define RT_SHELF_SLOTS 7u      /* words of the parking area that are free while closest hits are searched */

#define RT_UNIFORM __attribute__((address_space(4)))
typedef float Float8 __attribute__((ext_vector_type(8)));
typedef float Float4v __attribute__((ext_vector_type(4)));
typedef float Float16 __attribute__((ext_vector_type(16)));
typedef uint32_t Uint2 __attribute__((ext_vector_type(2)));
/* The same memory seen as constant and wave-uniform: loads through such a pointer at a wave-uniform address are scalar. */
template <typename T, typename S>
__device__ __forceinline__ const RT_UNIFORM T* uniform_view(const S* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (const RT_UNIFORM T*)p;
#pragma clang diagnostic pop
}

/* Population count of a lane mask as a 32-bit scalar.  (Written as the instruction itself: from __popcll the compiler makes
 * a 64-bit count whose comparisons it then moves to the vector unit, which has 64-bit compares where the scalar unit has none.) */
__device__ __forceinline__ uint32_t lanes_set(unsigned long long mask) {
    uint32_t n;
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(n) : "s"(mask) : "scc");
    return n;
}

/* t = plane * i + b along each axis (b = -origin * i): the rounding of b and of the fused sum moves a plane by a few
 * 1e-6 for coordinates up to 100, far inside the padding the float boxes carry (flatten.cpp: RT_PAD_FACTOR) */
struct SharedRay { float ix, iy, iz, bx, by, bz; };

/* All eight children of `node` (wave-uniform) against every lane's ray, for rays whose direction signs are OCTANT (a
 * compile-time constant: the rows to read are then fixed offsets and the front-to-back order of the children is fixed).
 * Masks are in front-to-back order: bit k = child k ^ OCTANT.  `visit` receives the children that at least RT_SHARE lanes
 * enter nearer than tmax, where the wavefront goes together; `entered` those any lane enters; the return value is the lane's
 * own mask, worked out only when some child is left to single lanes (entered != visit), otherwise 0. */
template <uint32_t OCTANT>
__device__ __forceinline__ uint32_t shared_node_test(const WideNode* wide, uint32_t node, const SharedRay& sr, float tmax, uint32_t& visit, uint32_t& entered, uint32_t* walkStats) {
    const RT_UNIFORM float* base = uniform_view<float>(wide) + (size_t)node * (sizeof(WideNode) / sizeof(float));
    constexpr bool negX = OCTANT & 1u, negZ = OCTANT & 2u, negY = OCTANT & 4u;
    /* rows [axis][child]: minimum corners at 0, maximum corners at 24 floats */
    const Float8 nearX = *reinterpret_cast<const RT_UNIFORM Float8*>(base + (negX ? 24 : 0)), farX = *reinterpret_cast<const RT_UNIFORM Float8*>(base + (negX ? 0 : 24));
    const Float8 nearY = *reinterpret_cast<const RT_UNIFORM Float8*>(base + 8 + (negY ? 24 : 0)), farY = *reinterpret_cast<const RT_UNIFORM Float8*>(base + 8 + (negY ? 0 : 24));
    const Float8 nearZ = *reinterpret_cast<const RT_UNIFORM Float8*>(base + 16 + (negZ ? 24 : 0)), farZ = *reinterpret_cast<const RT_UNIFORM Float8*>(base + 16 + (negZ ? 0 : 24));
    bool in[8];
    uint32_t together = 0, any = 0;
#pragma unroll
    for (int k = 7; k >= 0; k--) {
        const int c = k ^ (int)OCTANT;
        /* an empty child has its minimum at +FLT_MAX and its maximum at -FLT_MAX: entry +inf, exit -inf, never entered */
        const float tn = fmaxf(fmaxf(__builtin_fmaf(nearX[c], sr.ix, sr.bx), __builtin_fmaf(nearY[c], sr.iy, sr.by)),
                               fmaxf(__builtin_fmaf(nearZ[c], sr.iz, sr.bz), 0.0f));
        const float tf = fminf(fminf(__builtin_fmaf(farX[c], sr.ix, sr.bx), __builtin_fmaf(farY[c], sr.iy, sr.by)),
                               fminf(__builtin_fmaf(farZ[c], sr.iz, sr.bz), tmax));
        in[k] = tn <= tf;
        const uint32_t lanes = lanes_set(__ballot(in[k]));
        together |= lanes >= RT_SHARE ? (1u << k) : 0u;
        any |= lanes != 0u ? (1u << k) : 0u;
    }
    visit = together;
    entered = any;
#ifdef RT_WALK_STATS
    {
        uint32_t present = 0;
        for (int c = 0; c < 8; c++) present += (negX ? farX[c] <= nearX[c] : nearX[c] <= farX[c]) ? 1u : 0u;
        RT_STAT_ADD(RT_STAT_CHILD_BOXES, present);
    }
#endif
    uint32_t mine = 0;
    if (any != together) {
#pragma unroll
        for (int k = 0; k < 8; k++) mine |= in[k] ? (1u << k) : 0u;
    }
    return mine;
}

/* One leaf (wave-uniform: link and kind as in WideNode::Ref) against every lane's ray: the records arrive through scalar
 * loads, the tests are those of leaf_test. */
template <bool INDEXED>
__device__ __forceinline__ void shared_leaf_test(const DeviceScene& s, const RayD& r, uint32_t link, uint32_t kind, Closest& best,
                                                 float slack, uint32_t* walkStats) {
    const uint32_t sphereBase = s.planeCount, triangleBase = s.planeCount + s.sphereCount;
    const uint32_t sphereCount = kind >> 20, triangleCount = kind & 0xFFFFFu;
    const RT_UNIFORM uint32_t* run = uniform_view<uint32_t>(s.leafIndex) + link;
    RT_STAT(1);
    RT_WAVE_STAT(RT_STAT_SHARED_LEAVES);
    for (uint32_t i = 0; i < sphereCount; i++) {
        const uint32_t idx = run[i];
        const Float4v raw = *uniform_view<Float4v>(s.sphere + idx);
        Float4 sp; sp.x = raw[0]; sp.y = raw[1]; sp.z = raw[2]; sp.w = raw[3];
        float t;
        if (hit_sphere(sp, r, vdot(r.d, r.d), t) && beats(t, sphereBase + idx, best.t, best.id)) {
            best.t = t; best.id = sphereBase + idx; best.material = uniform_view<uint32_t>(s.sphereMaterial)[idx];
        }
    }
    for (uint32_t i = 0; i < triangleCount; i++) {
        const uint32_t idx = run[sphereCount + i];
        const uint32_t at = INDEXED ? idx : link + sphereCount + i;
        const Float16 raw = *uniform_view<Float16>(s.leafTriangle + at);
        LeafTriangle rec;
        rec.plane = Float4{raw[0], raw[1], raw[2], raw[3]}; rec.edge[0] = Float4{raw[4], raw[5], raw[6], raw[7]};
        rec.edge[1] = Float4{raw[8], raw[9], raw[10], raw[11]}; rec.edge[2] = Float4{raw[12], raw[13], raw[14], raw[15]};
        RT_STAT(2);
        RT_WAVE_STAT(RT_STAT_SHARED_TRIANGLES);
        const bool candidate = triangle_may_hit<false>(rec, r, best.t, slack) != TRI_MISS;
        if (__ballot(candidate) == 0ull) continue;
        const RT_UNIFORM Float4v* shape = uniform_view<Float4v>(s.triangleCorner + 3 * (size_t)idx);
        const Float4v c0 = shape[0], c1 = shape[1], c2 = shape[2];
        const Float4 pl = rec.plane;     /* the filter record's plane is the triangle's own (expand_leaves_kernel) */
        if (candidate) {
            RT_STAT(3);
            const Float4 corner[3] = {Float4{c0[0], c0[1], c0[2], c0[3]}, Float4{c1[0], c1[1], c1[2], c1[3]}, Float4{c2[0], c2[1], c2[2], c2[3]}};
            float t;
            const uint32_t id = triangleBase + idx;
            if (triangle_plane_distance(pl, r, t) && beats(t, id, best.t, best.id) && triangle_contains(corner, pl, r, t)) {
                best.t = t; best.id = id; best.material = uniform_view<uint32_t>(s.triangleMaterial)[idx];
            }
        }
    }
}

/* The rest of closest_hit for a wavefront whose live lanes share `octant` (wave-uniform): see above. */
template <bool INDEXED>
__device__ __forceinline__ void shared_walk(const DeviceScene& s, const RayD& r, const BoxWalk& w, uint32_t octantOfAll, const WalkStack& st,
                                            Closest& best, float slack, uint32_t* walkStats) {
    const uint32_t tid = threadIdx.x;
    SharedRay sr;
    sr.ix = w.ix; sr.iy = w.iy; sr.iz = w.iz;
    sr.bx = -(r.o.x * w.ix); sr.by = -(r.o.y * w.iy); sr.bz = -(r.o.z * w.iz);
    uint32_t* const shelf = st.shelf + tid;                     /* [slot * RT_BLOCK] */
    uint32_t* const trail = st.trail + (tid >> 6) * st.levels;   /* the wavefront's stack: node | children still to visit << 24 */
    uint32_t shelved = 0;
    uint32_t cur = 0, level = 0, pending = 0;      /* wave-uniform */
    bool fresh = true, finished = false;
    Walker k; k.level = 0; k.cur = 0; k.pending = 0; k.firstInner = 0; k.innerMask = 0;
    bool walking = false, haveLeaf = false;
    uint32_t leafLink = 0, leafKind = 0;
    for (;;) {
        /* ---- the wavefront's own steps, until enough lanes have work of their own ---- */
        while (!finished) {
            if ((uint32_t)__popcll(__ballot(shelved != 0u)) >= RT_DRAIN_LANES || __ballot(shelved == RT_SHELF_SLOTS) != 0ull) break;
            /* the same in every lane by construction; said so explicitly, so that the node's loads are scalar */
            cur = __builtin_amdgcn_readfirstlane(cur);
            pending = __builtin_amdgcn_readfirstlane(pending);
            level = __builtin_amdgcn_readfirstlane(level);
            const uint32_t octant = __builtin_amdgcn_readfirstlane(octantOfAll);
            if (fresh) {
                uint32_t visit = 0, entered = 0, mine = 0;
                switch (octant) {
                case 0: mine = shared_node_test<0>(s.wideNodes, cur, sr, best.t, visit, entered, walkStats); break;
                case 1: mine = shared_node_test<1>(s.wideNodes, cur, sr, best.t, visit, entered, walkStats); break;
                case 2: mine = shared_node_test<2>(s.wideNodes, cur, sr, best.t, visit, entered, walkStats); break;
                case 3: mine = shared_node_test<3>(s.wideNodes, cur, sr, best.t, visit, entered, walkStats); break;
                case 4: mine = shared_node_test<4>(s.wideNodes, cur, sr, best.t, visit, entered, walkStats); break;
                case 5: mine = shared_node_test<5>(s.wideNodes, cur, sr, best.t, visit, entered, walkStats); break;
                case 6: mine = shared_node_test<6>(s.wideNodes, cur, sr, best.t, visit, entered, walkStats); break;
                default: mine = shared_node_test<7>(s.wideNodes, cur, sr, best.t, visit, entered, walkStats); break;
                }
                RT_STAT(0);
                RT_WAVE_STAT(RT_STAT_SHARED_STEPS);
                const uint32_t keep = mine & ~visit;
                if (keep != 0u) { shelf[shelved * RT_BLOCK] = cur | (keep << 24); shelved++; RT_STAT(RT_STAT_SHELVED); }
                pending = visit;
                fresh = false;
            }
            if (pending == 0u) {
                if (level == 0u) { finished = true; break; }
                level--;
                const uint32_t entry = __builtin_amdgcn_readfirstlane(trail[level]);
                cur = entry & 0xFFFFFFu;
                pending = entry >> 24;
                continue;
            }
            const uint32_t c = (uint32_t)__builtin_ctz(pending) ^ octant;
            pending &= pending - 1u;
            const Uint2 ref = *uniform_view<Uint2>(&s.wideNodes[cur].ref[c]);
            if (ref[1] & WIDE_INNER) {
                if (pending != 0u) { trail[level] = cur | (pending << 24); level++; }
                cur = ref[0];
                fresh = true;
            } else {
                shared_leaf_test<INDEXED>(s, r, ref[0], ref[1], best, slack, walkStats);
            }
        }
        /* ---- every lane works off its shelf: the per-lane walk, started at the shelved node with the lane's own children ---- */
        RT_WAVE_STAT(RT_STAT_DRAINS);
        uint32_t taken = 0;
        for (;;) {
            if (!walking && !haveLeaf && taken < shelved) {
                const uint32_t entry = shelf[taken * RT_BLOCK];
                taken++;
                k.level = 0;
                k.cur = entry & 0xFFFFFFu;
                k.pending = entry >> 24;      /* already in front-to-back order */
                const uint2 numbering = *node_bytes<uint2>(s.nodes, k.cur * (uint32_t)sizeof(QNode) + 72u);
                k.firstInner = numbering.x;
                k.innerMask = numbering.y;
                walking = true;
            }
            if (walking && !haveLeaf) {
                const int step = walker_step<false, false>(s, w, st, best.t, k, leafLink, leafKind, walkStats);
                haveLeaf = step == 1;
                if (step == 2) walking = false;
            }
            const unsigned long long holders = __ballot(haveLeaf);
            const unsigned long long walkers = __ballot(!haveLeaf && (walking || taken < shelved));
            if ((holders | walkers) == 0ull) break;
            if (walkers != 0ull && __popcll(holders) < __popcll(walkers)) continue;
            if (!haveLeaf) continue;
            haveLeaf = false;
            leaf_test<false, INDEXED>(s, r, leafLink, leafKind, best, 0.0f, walkStats, slack);
        }
        shelved = 0;
        if (finished) break;
    }
}

/* Closest primitive along r: planes by loop, spheres and triangles through the hierarchy.
 * Each lane is either WALKING or HOLDING a leaf; the wavefront schedules the two bodies with
 * ballots: walkers step every iteration, holders wait until they are at least as many as the
 * walkers (or nobody walks any more) and then test their leaves together, so neither body runs
 * for a handful of lanes while the rest of the wavefront idles.
 * coherent (the same in every lane): the wavefront's rays started out together (primary rays); if their
 * direction signs agree as well, they walk the top of the hierarchy as one (shared_walk). */
template <bool INDEXED, bool PADDED>
__device__ __forceinline__ Closest closest_hit(const DeviceScene& s, const RayD& r, const WalkStack& st, float slack, bool coherent, uint32_t* walkStats) {
    Closest best;
    best.t = FLT_MAX;
    best.id = HIP_TRACE_NO_PRIMITIVE;
    best.material = 0;
    for (uint32_t i = 0; i < s.planeCount; i++) {
        float t;
        if (hit_plane(i < RT_PLANES_IN_LDS ? st.plane[i] : s.plane[i], r, t) && beats(t, i, best.t, best.id)) { best.t = t; best.id = i; best.material = s.planeMaterial[i]; }
    }
    const BoxWalk w = make_walk(r, PADDED ? s.boxPad : 0.0f);
    if (!PADDED && coherent && s.wideNodes != nullptr) {
        const uint32_t octant = __builtin_amdgcn_readfirstlane(w.octant);
        if (__ballot(w.octant != octant) == 0ull) {
            shared_walk<INDEXED>(s, r, w, octant, st, best, slack, walkStats);
            return best;
        }
    }
    Walker k;
    walker_start<false, PADDED>(s, st, w, best.t, k, walkStats);
    uint32_t leafLink = 0, leafKind = 0;
    bool haveLeaf = false, done = false;
    for (;;) {
        if (!haveLeaf && !done) {
            const int step = walker_step<false, PADDED>(s, w, st, best.t, k, leafLink, leafKind, walkStats);
            haveLeaf = step == 1;
            done = step == 2;
        }
        const unsigned long long holders = __ballot(haveLeaf);
        const unsigned long long walkers = __ballot(!haveLeaf && !done);
        if ((holders | walkers) == 0) break;
        if (walkers != 0 && __popcll(holders) < __popcll(walkers)) continue;
        if (!haveLeaf) continue;
        haveLeaf = false;
        leaf_test<false, INDEXED>(s, r, leafLink, leafKind, best, 0.0f, walkStats, slack);
    }
    return best;
}

/* reference: src/raytracer.c:34-56.  The reference passes (-1, 1, cos) to a clamp whose
 * signature is (in, min, max), which yields 1 whatever the direction; kept. */
__device__ __forceinline__ float fresnel(V3 dir, V3 normal, float refractionIndex) {
    float cosi = clampf(-1.0f, 1.0f, vdot(dir, normal));
    float etai = 1.0f, etat = refractionIndex;
    if (cosi > 0) { float tmp = etat; etat = etai; etai = tmp; }
    float a = 1.0f - cosi * cosi;
    float sint = etai / etat * __builtin_sqrtf(0.f > a ? 0.f : a);
    if (sint >= 1) return 1.0f;
    float b = 1.0f - sint * sint;
    float cost = __builtin_sqrtf(0.f > b ? 0.f : b);
    cosi = fabsf(cosi);
    float Rs = ((etat * cosi) - (etai * cost)) / ((etat * cosi) + (etai * cost));
    float Rp = ((etai * cosi) - (etat * cost)) / ((etai * cosi) + (etat * cost));
    return (Rs * Rs + Rp * Rp) / 2;
}

/* reference: src/raytracer.c:9-32, same clamp quirk. */
__device__ __forceinline__ V3 refract(V3 direction, V3 normal, float refractionIndex) {
    float cosi = clampf(-1.0f, 1.0f, vdot(direction, normal));
    float etai = 1.0f, etat = refractionIndex;
    V3 n = normal;
    if (cosi < 0) {
        cosi = -cosi;
    } else {
        float tmp = etai; etai = etat; etat = tmp;
        n = vmul(n, -1.0f);
    }
    float eta = etai / etat;
    float k = 1 - eta * eta * (1 - cosi * cosi);
    if (k < 0) return mk(0, 0, 0);
    return vnorm(vadd(vmul(direction, eta), vmul(n, eta * cosi - __builtin_sqrtf(k))));
}

/* reference: src/raytracer.c:66-68 */
__device__ __forceinline__ void nudge_forward(RayD& r) { r.o = vadd(r.o, vmul(r.d, 1.0f / 1000.0f)); }

struct FrameParams {
    DeviceScene scene;
    V3 camPosition, camX, camY, targetCenter;
    float targetWidth, targetHeight, focalLength, apertureSize;
    uint32_t width, height, maxDepth, jitterMode, shadingModel;
    uint32_t raysX, raysY;            /* sub-pixel grid (reference: src/gpu.c:446-465) */
    float deltaX, deltaY, contribution;
    uint32_t tilesX, tileCount, rank, worldSize, rotate, localTiles, xcdRun;
    uint32_t* out;                 /* row-major frame (worldSize 1) or this rank's tiles, tile-major */
    uint8_t* hitMask;              /* optional, laid out like `out` */
    uint32_t* primitiveId;         /* optional, laid out like `out` */
    float* hitDistance;            /* optional, laid out like `out` */
    unsigned long long* counters;  /* RT_COUNTERS: primary, shadow, reflection, refraction[, walk stats] */
    const uint32_t* tileOrder;     /* optional: workgroup b renders local tile tileOrder[b] */
    uint32_t* tileCost;            /* optional: receives the cycles each local tile took */
};

/* reference: src/kernel.cl:804-838 with one sample per pixel and the aperture draws equal to 0. */
/* i, j: position in the sub-pixel grid (0, 0 with one ray per pixel); lens: the three aperture
 * draws already halved (zero unless the device shading model draws them). */
__device__ __forceinline__ RayD primary_ray(const FrameParams& p, uint32_t x, uint32_t y, uint32_t i, uint32_t j,
                                            float deltaX, float deltaY, V3 lens) {
    float pixelWidth = 1.0f / (float)p.width;
    float pixelHeight = 1.0f / (float)p.height;
    float posX = -1.0f + 2.0f * ((float)x / (float)p.width);
    float posY = -1.0f + 2.0f * ((float)y / (float)p.height);
    V3 offsetY = vmul(p.camY, (posY - pixelHeight + (float)j * deltaY) * p.targetHeight / 2.0f);
    V3 offsetX = vmul(p.camX, (posX - pixelWidth + (float)i * deltaX) * p.targetWidth / 2.0f);
    V3 target = vsub(vadd(p.targetCenter, offsetX), offsetY);
    RayD r;
    r.o = p.camPosition;
    r.d = vnorm(vsub(target, p.camPosition));
    V3 focalPoint = vadd(r.o, vmul(r.d, p.focalLength));
    r.o = vadd(r.o, vmul(lens, p.apertureSize));
    r.d = vnorm(vsub(focalPoint, r.o));
    return r;
}

/* reference: src/kernel.cl:850-856 + 8-bit normalised texture (src/gpu.c:586). */
__device__ __forceinline__ uint32_t to_unorm8(float c) {
    c = clampf(c, 0.0f, 1.0f);
    if (c != c) return 0;
    return (uint32_t)__builtin_rintf(c * 255.0f);
}

struct Task { V3 o, d, weight; uint32_t depth; uint32_t kind; };
enum { RAY_PRIMARY = 0, RAY_SHADOW = 1, RAY_REFLECTION = 2, RAY_REFRACTION = 3 };

__device__ __forceinline__ uint32_t tile_of_block(const FrameParams& p, uint32_t local) {
    if (p.worldSize == 1) return local;
    if (!p.rotate) return local * p.worldSize + p.rank;
    const uint32_t perRow = p.tilesX / p.worldSize;
    const uint32_t ty = local / perRow, q = local % perRow;
    const uint32_t r = (p.rank + p.worldSize - ty % p.worldSize) % p.worldSize;
    return ty * p.tilesX + q * p.worldSize + r;
}

/* Workgroup-shared state of the shadow pass (LDS, after the walk stacks). */
struct ShadowShared {
    float* hitX; float* hitY; float* hitZ;   /* [RT_BLOCK] shading point of each thread            */
    uint32_t* occluded;                      /* [RT_BLOCK] bit l = light (base + l) is hidden        */
    uint16_t* pointList;                     /* [RT_BLOCK] threads that have a point to light        */
    uint16_t* treePosition;                  /* [RT_BLOCK] position of the thread's point in its ray tree */
    uint32_t* planeSides;                    /* [RT_BLOCK] sides of the first planes the thread's point is clearly on */
    unsigned long long* lastOccluder;        /* [32] per light slot: leaf (link | kind << 32) that last hid a ray, 0 = none */
    uint32_t* pointCount;                    /* how many                                             */
    uint32_t* farPoints;                     /* non-zero: one of them lies beyond the reach of the light grids */
    uint32_t* nextJob;                       /* shadow-ray queue head                                */
    uint32_t* rayCount;                      /* [4] rays traced by the workgroup so far, by kind           */
    uint32_t* anyRay;                        /* [3] "some thread has a ray for the next level", one word per level in turn */
    uint32_t* rayCountOfLevel;               /* rays of the tile's threads gathered for this level (see trace_kernel, step 1) */
};

/* One draw of the hashed light jitter: see HIP_TRACE_JITTER_HASHED. */
__host__ __device__ __forceinline__ uint32_t jitter_hash(uint32_t pixel, uint32_t path, uint32_t light, uint32_t component) {
    uint32_t v = (pixel * 0x9E3779B1u) ^ (path * 0x85EBCA77u) ^ (light * 0xC2B2AE3Du) ^ (component * 0x27D4EB2Fu);
    v ^= v >> 16; v *= 0x85EBCA6Bu;
    v ^= v >> 13; v *= 0xC2B2AE35u;
    v ^= v >> 16;
    return v;
}
/* reference: src/utils/random.c:5-11 with (h >> 8) / 2^24 in place of rand() / RAND_MAX */
__device__ __forceinline__ float jitter_bilateral(uint32_t pixel, uint32_t path, uint32_t light, uint32_t component) {
    const float unilateral = (float)(jitter_hash(pixel, path, light, component) >> 8) / 16777216.0f;
    return -1.0f + 2.0f * unilateral;
}

/* Draw of the device shading model: the hash keyed additionally by sub-pixel sample and shadow
 * sample, mapped to [0,1) in 2^17 steps (reference: src/kernel.cl:85-91). */
__device__ __forceinline__ float kernel_bilateral(uint32_t pixel, uint32_t path, uint32_t sample, uint32_t light,
                                                  uint32_t shadowSample, uint32_t component) {
    const uint32_t h = jitter_hash(pixel, path | (sample << 16), light | (shadowSample << 16), component);
    const float unilateral = (float)(h % (1u << 17)) / (float)(1 << 17);
    return -1.0f + 2.0f * unilateral;
}

/* The random unit vector added to the direction towards a light (reference:
 * src/raytracer.c:278, src/kernel.cl:682-686), or zero without jitter. */
__device__ __forceinline__ V3 light_offset(bool hashed, bool kernelModel, uint32_t pixel, uint32_t path, uint32_t sample,
                                           uint32_t light, uint32_t shadowSample) {
    if (!hashed) return mk(0.0f, 0.0f, 0.0f);
    if (kernelModel)
        return vnorm(mk(kernel_bilateral(pixel, path, sample, light, shadowSample, 0),
                        kernel_bilateral(pixel, path, sample, light, shadowSample, 1),
                        kernel_bilateral(pixel, path, sample, light, shadowSample, 2)));
    return vnorm(mk(jitter_bilateral(pixel, path, light, 0), jitter_bilateral(pixel, path, light, 1),
                    jitter_bilateral(pixel, path, light, 2)));
}

/* reference: src/raytracer.c:277-284 -- the shadow ray from a shading point towards a light. */
__device__ __forceinline__ RayD shadow_ray(V3 hitPoint, const Float4& light, V3 offset, float& distanceToLight) {
    V3 hitToLight = vsub(mk(light.x, light.y, light.z), hitPoint);
    hitToLight = vadd(hitToLight, offset);
    distanceToLight = vlength(hitToLight);
    RayD shadow;
    shadow.o = hitPoint;
    shadow.d = vnorm(hitToLight);
    nudge_forward(shadow);
    return shadow;
}

/* What the general kernel needs to know about the frame's options inside the shadow pass. */
enum { MODE_PLAIN = 0, MODE_JITTER = 1, MODE_GENERAL = 2 };
struct ShadeMode {
    bool hashed, kernelModel;
    uint32_t samplesPerLight;   /* 1, or 4 for the device shading model (reference: src/kernel.cl:673) */
    uint32_t sample;            /* sub-pixel sample being traced */
};

/* The number a list entry refers to its primitive by (device_layout.h: LightGrid), for use as an index.
 * The compiler is not to know that it has 24 bits: told so, it turns index * 3 (a triangle's first corner) into a 24-bit
 * multiply, drops the mask as redundant for that, and then widens the multiply to a full 64-bit multiply-add of the
 * unmasked word (v_mad_u64_u32, seen in the generated code): the address of an entry with distance bits set lies
 * gigabytes outside the array. */
__device__ __forceinline__ uint32_t grid_entry_number(uint32_t entry) {
    uint32_t number = entry & GRID_REF_INDEX;
    asm volatile("" : "+v"(number));
    return number;
}

/* One entry of a light's direction grid (light_grid.hip) against an occlusion ray: a sphere by the reference's test, a
 * triangle by the conservative filter and, when that cannot decide, the exact test -- what leaf_test<true> does for the
 * entries of a leaf.  entry: a word of a grid's list (device_layout.h: LightGrid). */
template <bool INDEXED>
__device__ __forceinline__ bool grid_entry_hides(const DeviceScene& s, const RayD& r, float tmax, uint32_t entry, float slack, uint32_t* walkStats) {
    const uint32_t ref = grid_entry_number(entry);
    if (entry & GRID_REF_SPHERE) {
        float t;
        return hit_sphere(s.sphere[ref], r, vdot(r.d, r.d), t) && t <= tmax;
    }
    LeafTriangle rec;
    rec.plane = s.leafTriangle[ref].plane; rec.edge[0] = s.leafTriangle[ref].edge[0]; rec.edge[1] = s.leafTriangle[ref].edge[1]; rec.edge[2] = s.leafTriangle[ref].edge[2];
    RT_STAT(2);
    RT_WAVE_STAT(6);
    const int verdict = triangle_may_hit<true>(rec, r, tmax, slack);
    if (verdict == TRI_SURE) return true;
    if (verdict == TRI_MISS) return false;
    RT_WAVE_STAT(7);
    const uint32_t idx = INDEXED ? ref : s.leafIndex[ref];
    Float4 pl = s.trianglePlane[idx];
    Float4 corner[3] = {s.triangleCorner[3 * (size_t)idx], s.triangleCorner[3 * (size_t)idx + 1], s.triangleCorner[3 * (size_t)idx + 2]};
    asm volatile("" : "+v"(pl.x), "+v"(pl.w), "+v"(corner[0].x), "+v"(corner[1].x), "+v"(corner[2].x));
    RT_STAT(3);
    float t;
    if (!triangle_plane_distance(pl, r, t) || !(t <= tmax)) return false;
    return triangle_contains(corner, pl, r, t);
}

/* Bucket of a light's grid in which the light sees `point` (the same face and axis conventions as
 * light_grid.hip: face = 2 * major axis + (negative side), u and w the next two axes in cyclic order over the major one). */
__device__ __forceinline__ uint32_t grid_bucket(const LightGrid& grid, V3 point, const Float4& light) {
    const float v[3] = {point.x - light.x, point.y - light.y, point.z - light.z};
    const float ax = fabsf(v[0]), ay = fabsf(v[1]), az = fabsf(v[2]);
    const int axis = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
    const float major = axis == 0 ? v[0] : axis == 1 ? v[1] : v[2];
    const float u0 = axis == 0 ? v[1] : axis == 1 ? v[2] : v[0];
    const float w0 = axis == 0 ? v[2] : axis == 1 ? v[0] : v[1];
    const float inv = __builtin_amdgcn_rcpf(fabsf(major));
    const float half = grid.halfResolution;
    const int last = (int)grid.resolution - 1;
    const int iu = max(0, min(last, (int)floorf((u0 * inv + 1.0f) * half)));
    const int iw = max(0, min(last, (int)floorf((w0 * inv + 1.0f) * half)));
    const uint32_t face = 2u * (uint32_t)axis + (major < 0.0f ? 1u : 0u);
    /* (face * R + w) * R + u with 24-bit multiplies (R <= 4096: every factor fits), which issue at full rate where a
     * 32-bit integer multiply takes four times as long */
    return __umul24(__umul24(face, grid.resolution) + (uint32_t)iw, grid.resolution) + (uint32_t)iu;
}

/* The list of bucket `bucket` of light `light`: where it begins in the light's stretch of references, and its length
 * (device_layout.h: LightGrid -- one 12-byte block per eight buckets). */
__device__ __forceinline__ void grid_list(const LightGrid& grid, uint32_t light, uint32_t bucket, uint32_t& first, uint32_t& length) {
    /* the block's number fits 32 bits (the build refuses an index of 2^32 blocks), its factors 24 */
    const uint32_t* const block = grid.blocks + (size_t)(__umul24(light, grid.blocksPerLight) + (bucket >> 3)) * 3u;
    const uint32_t base = block[0], lo = block[1], hi = block[2];
    const uint32_t k = bucket & 7u, shift = 8u * (k & 3u);
    const uint32_t word = k < 4u ? lo : hi;
    length = (word >> shift) & 0xFFu;
    /* the lengths before it in its block: the bytes of `word` below byte k, plus all of `lo` for the upper half */
    const uint32_t before = __builtin_amdgcn_sad_u8(word & ((1u << shift) - 1u), 0u, 0u);
    first = base + before + (k < 4u ? 0u : __builtin_amdgcn_sad_u8(lo, 0u, 0u));
}

/* The shadow pass when every ray of it goes through a light grid (all lights of the scene have one and no point of the
 * tile lies beyond their reach): the same job queue and batches as shadow_pass below, without anything a walk needs.
 * Lane states: NEED a ray, running down a LIST, EXITed. */
/* The record of a grid entry: a triangle's filter record, or a sphere in its first row. */
__device__ __forceinline__ void fetch_entry(const DeviceScene& s, uint32_t entry, LeafTriangle& rec) {
    const uint32_t ref = grid_entry_number(entry);
    /* a scene without spheres (the same for every lane: a scalar branch) has no such entries */
    if (s.sphereCount != 0u && (entry & GRID_REF_SPHERE)) { rec.plane = s.sphere[ref]; return; }
    rec.plane = s.leafTriangle[ref].plane; rec.edge[0] = s.leafTriangle[ref].edge[0]; rec.edge[1] = s.leafTriangle[ref].edge[1]; rec.edge[2] = s.leafTriangle[ref].edge[2];
}

template <bool INDEXED>
__device__ __forceinline__ void shadow_pass_lists(const DeviceScene& s, const WalkStack& st, const ShadowShared& sh,
                                                  uint32_t lightBase, uint32_t lightsInPass, uint32_t* walkStats) {
    enum { NEED = 0, LIST = 1, EXIT = 2 };
    const float slack = RT_DEFAULT_SLACK;
    const uint32_t points = *sh.pointCount;
    const uint32_t jobs = points * lightsInPass;
    const uint32_t pointsReciprocal = points ? (uint32_t)((0x100000000ull + points - 1u) / points) : 0u;
    uint32_t state = jobs ? NEED : EXIT;
    RayD r; r.o = mk(0, 0, 0); r.d = mk(0, 0, 1);
    float tmax = 0.0f;
    uint32_t ownerAndSlot = 0;          /* thread that owns the point (low 8 bits), light slot of the job (from bit 8) */
    const uint32_t* list = nullptr;     /* the entry after nextRef in the lane's bucket */
    uint32_t listLeft = 0;              /* entries still to test, nextRef included */
    uint32_t nextRef = 0;               /* read one turn ahead of its record */
    uint32_t ref = 0;                   /* the entry whose record is in rec */
    uint32_t rayLevel = 0;              /* the distance level of the ray's start, seen from its light: entries beyond it cannot hide the light */
    LeafTriangle rec;                   /* read one turn ahead; a sphere's (centre, radius) sits in rec.plane */
    rec.plane = Float4{0, 0, 0, 0}; rec.edge[0] = rec.plane; rec.edge[1] = rec.plane; rec.edge[2] = rec.plane;
    for (;;) {
        const uint32_t nNeed = (uint32_t)__popcll(__ballot(state == NEED));
        const uint32_t nList = (uint32_t)__popcll(__ballot(state == LIST));
        if (nNeed + nList == 0) break;
        if (state == NEED && (nNeed >= ShadowSchedule<false>::needBatch || nList == 0u)) {
            const uint32_t job = atomicAdd(sh.nextJob, 1u);
            if (job >= jobs) {
                state = EXIT;
            } else {
                const uint32_t slot = points == 1u ? job : __umulhi(job, pointsReciprocal);
                const uint32_t owner = sh.pointList[job - __umul24(slot, points)];
                ownerAndSlot = owner | (slot << 8);
                const uint32_t lightNumber = lightBase + slot;
                const Float4 light = s.light[2 * lightNumber];
                const V3 point = mk(sh.hitX[owner], sh.hitY[owner], sh.hitZ[owner]);
                /* the bucket first: its two reads in a row are under way while the ray is set up and the planes are tested */
                const uint32_t* const info = s.grid.info + RT_GRID_INFO_WORDS * lightNumber;
                const uint32_t alwaysCount = info[1];
                const uint32_t* const lightRefs = s.grid.refs + ((size_t)info[2] | ((size_t)info[3] << 32));
                const float levelsPerUnit = __uint_as_float(info[4]);
                uint32_t first;
                grid_list(s.grid, lightNumber, grid_bucket(s.grid, point, light), first, listLeft);
                r = shadow_ray(point, light, mk(0.0f, 0.0f, 0.0f), tmax);
                /* rounded up a little and then down to a whole level: an entry of a higher level lies farther from the
                 * light than any point of the ray (the build rounds a primitive's distance down, with a margin) */
                rayLevel = (uint32_t)(tmax * levelsPerUnit * 1.00001f);
                list = lightRefs + first;
                nextRef = *list;      /* one word past the last list is allocated: in bounds even for an empty bucket */
                const uint32_t both = sh.planeSides[owner] & __float_as_uint(s.light[2 * lightNumber + 1].w);
                const uint32_t clear = (both | (both >> 16)) & 0xFFFFu;
                bool hidden = false;
                /* does plane i hide the light?  A fused estimate first: a plane met well beyond the light (or behind the ray)
                 * cannot; the margin covers the estimate's rounding, anything closer gets the exact test */
                auto plane_hides = [&](uint32_t i) {
                    const Float4 pl = i < RT_PLANES_IN_LDS ? st.plane[i] : s.plane[i];
                    const float den = __builtin_fmaf(pl.x, r.d.x, __builtin_fmaf(pl.y, r.d.y, pl.z * r.d.z));
                    if (fabsf(den) >= 2e-5f) {
                        const float num = -pl.w - __builtin_fmaf(pl.x, r.o.x, __builtin_fmaf(pl.y, r.o.y, pl.z * r.o.z));
                        const float inv = __builtin_amdgcn_rcpf(den);
                        const float est = num * inv;
                        const float margin = __builtin_fmaf(slack, fabsf(inv), __builtin_fmaf(4e-6f, fabsf(est), slack));
                        if (est < -margin || est > tmax + margin) return false;
                    }
                    float t;
                    return hit_plane(pl, r, t) && t <= tmax;
                };
                /* the first planes still to test, as a bit mask: in a closed room none is left and no loop runs at all */
                uint32_t todo = ~clear & (s.planeCount >= RT_SIDE_PLANES ? 0xFFFFu : (1u << s.planeCount) - 1u);
                while (todo != 0u && !hidden) {
                    hidden = plane_hides((uint32_t)__builtin_ctz(todo));
                    todo &= todo - 1u;
                }
                for (uint32_t i = RT_SIDE_PLANES; i < s.planeCount && !hidden; i++) hidden = plane_hides(i);
                for (uint32_t i = 0; i < alwaysCount && !hidden; i++)
                    hidden = grid_entry_hides<INDEXED>(s, r, tmax, s.grid.always[lightNumber * RT_GRID_ALWAYS + i], slack, walkStats);
                if (hidden) atomicOr(&sh.occluded[owner], 1u << slot);
                else if (listLeft != 0u && (nextRef >> GRID_REF_LEVEL_SHIFT) <= rayLevel) {
                    state = LIST;
                    ref = nextRef;
                    nextRef = *++list;
                    fetch_entry(s, ref, rec);
                }      /* else nothing in the bucket between the point and the light */
            }
        }
        if (__ballot(state == LIST) != 0ull) {
#ifdef RT_WALK_STATS
            if ((threadIdx.x & 63u) == 0u) walkStats[RT_STAT_LIST_TURNS]++;
            if (state == LIST) walkStats[RT_STAT_LIST_LANES]++;
#endif
            if (state == LIST) {
                /* the filter uses up the record, then the next entry's record is read into the same registers: under way
                 * during the rest of this turn and the start of the next, with no second set of registers and no copy
                 * between two (reading it before the filter, into registers of its own: 2.26 ms where this takes 2.08).
                 * Behind a bucket's last entry lies the next bucket's first, or the spare entry the build puts behind the
                 * last list of all (a valid reference), so the read is harmless where it is not needed. */
                const uint32_t entry = ref;
                listLeft--;
                int verdict;
                if (s.sphereCount != 0u && (entry & GRID_REF_SPHERE)) {
                    float t;
                    verdict = hit_sphere(rec.plane, r, vdot(r.d, r.d), t) && t <= tmax ? TRI_SURE : TRI_MISS;
                } else {
                    RT_STAT(2);
                    RT_WAVE_STAT(6);
                    verdict = triangle_may_hit<true>(rec, r, tmax, slack);
                }
                ref = nextRef;
                /* the lists are sorted by level: behind an entry too far from the light to matter come only more of them */
                if ((ref >> GRID_REF_LEVEL_SHIFT) > rayLevel) listLeft = 0u;
                asm volatile("" : "+v"(ref) : "v"(verdict));      /* the reads below stay below the filter */
                fetch_entry(s, ref, rec);
                nextRef = *++list;
                bool hides = verdict == TRI_SURE;
                if (verdict == TRI_MAYBE) {
                    RT_WAVE_STAT(7);
                    const uint32_t idx = INDEXED ? grid_entry_number(entry) : s.leafIndex[grid_entry_number(entry)];
                    Float4 pl = s.trianglePlane[idx];
                    Float4 corner[3] = {s.triangleCorner[3 * (size_t)idx], s.triangleCorner[3 * (size_t)idx + 1], s.triangleCorner[3 * (size_t)idx + 2]};
                    asm volatile("" : "+v"(pl.x), "+v"(pl.w), "+v"(corner[0].x), "+v"(corner[1].x), "+v"(corner[2].x));
                    RT_STAT(3);
                    float t;
                    hides = triangle_plane_distance(pl, r, t) && t <= tmax && triangle_contains(corner, pl, r, t);
                }
                if (hides) {
                    atomicOr(&sh.occluded[ownerAndSlot & 0xFFu], 1u << (ownerAndSlot >> 8));
                    state = NEED;
                } else if (listLeft == 0u) {
                    state = NEED;
                }
            }
        }
    }
}

/* Shadow pass of one workgroup: the (point, light) pairs of the tile form one queue in LDS and
 * every lane of every wavefront pulls its next ray from it as soon as it has finished the
 * previous one, so a lane with a short walk does not wait for the longest walk of its
 * wavefront -- only the queue's last few rays run under-occupied.  The reference lights a point
 * iff distanceToLight < closest hit distance of the shadow ray (reference:
 * src/raytracer.c:286-292): hidden iff some primitive has t <= distanceToLight.
 * Lane states: NEED a ray, RUNning one (walking, or holding a leaf), EXITed.  Ray set-up is the
 * most expensive body, so lanes queue up in NEED until a batch of them waits (or
 * nothing else is left to do). */
template <int MODE, bool MANY_LEAVES, bool GRID>
__device__ __forceinline__ void shadow_pass(const DeviceScene& s, const WalkStack& st, const ShadowShared& sh,
                                            uint32_t lightBase, uint32_t lightsInPass, const ShadeMode& mode, uint32_t tileX0,
                                            uint32_t tileY0, uint32_t width, uint32_t* walkStats) {
    enum { NEED = 0, RUN = 1, EXIT = 2, LIST = 3 };
    /* LIST: the ray goes through its light's direction grid (light_grid.hip) instead of walking the hierarchy: the lane
     * tests the entries of one bucket, one per turn.  Only the plain kernel does this, and where every ray of a tile can,
     * shadow_pass_lists serves the pass instead of this loop. */
    static_assert(!GRID || MODE == MODE_PLAIN, "the light grids serve the plain kernel");
    if (GRID && s.grid.allLights != 0u && *sh.farPoints == 0u) {      /* the same for the whole workgroup */
        shadow_pass_lists<MANY_LEAVES>(s, st, sh, lightBase, lightsInPass, walkStats);
        return;
    }
    const uint32_t* list = nullptr;     /* the entry after nextRef in the lane's bucket */
    uint32_t listLeft = 0;              /* entries still to test, nextRef included */
    uint32_t nextRef = 0;               /* read one turn ahead, so that a turn waits for one load, not for two in a row */
    const uint32_t points = *sh.pointCount;
    /* one job per (point, light, shadow sample); bit (light * samplesPerLight + sample) of the
     * owner's mask records a hidden sample */
    const uint32_t perLight = MODE == MODE_GENERAL ? mode.samplesPerLight : 1u;
    const bool strict = MODE == MODE_GENERAL && mode.kernelModel;   /* that model hides for t < distance, not <= */
    const float slack = MODE == MODE_GENERAL ? s.filterMargin : RT_DEFAULT_SLACK;
    const uint32_t jobs = points * lightsInPass * perLight;
    const uint32_t pointsReciprocal = points ? (uint32_t)((0x100000000ull + points - 1u) / points) : 0u;
    uint32_t state = jobs ? NEED : EXIT;
    RayD r; r.o = mk(0, 0, 0); r.d = mk(0, 0, 1);
    BoxWalk w = make_walk(r);
    Walker k; k.level = 0; k.cur = 0; k.pending = 0; k.firstInner = 0; k.innerMask = 0;
    float tmax = 0.0f;
    uint32_t ownerAndSlot = 0;     /* thread that owns the point (low 8 bits), light slot of the job (from bit 8) */
    /* leaves this lane has found and not yet tested, oldest first */
    uint32_t queued = 0, link0 = 0, kind0 = 0, link1 = 0, kind1 = 0, link2 = 0, kind2 = 0;
    bool walkDone = false;
#ifdef RT_WALK_STATS
    uint32_t raySteps = 0;
#define RT_RAY_DONE() do { walkStats[8 + min(7u, 31u - (uint32_t)__builtin_clz(raySteps | 1u))]++; } while (0)
#else
#define RT_RAY_DONE() ((void)0)
#endif
    for (;;) {
        const uint32_t nNeed = (uint32_t)__popcll(__ballot(state == NEED));
        const uint32_t nRun = (uint32_t)__popcll(__ballot(state == RUN || (GRID && state == LIST)));
        if (nNeed + nRun == 0) break;
        /* rays that go through a grid start together in large batches whatever the hierarchy: their turns are short */
        if (state == NEED && (nNeed >= (GRID ? ShadowSchedule<false>::needBatch : ShadowSchedule<MANY_LEAVES>::needBatch) || nRun == 0u)) {
            const uint32_t job = atomicAdd(sh.nextJob, 1u);
            if (job >= jobs) {
                state = EXIT;
            } else {
                /* job / points by multiplication: exact because job * points < 2^32 (at most 8192 jobs);
                 * a single point has no 32-bit reciprocal (2^32), and its quotient is the job itself */
                const uint32_t slot = points == 1u ? job : __umulhi(job, pointsReciprocal);   /* light * perLight + shadow sample */
                const uint32_t l = MODE == MODE_GENERAL ? slot / perLight : slot;
                const uint32_t owner = sh.pointList[job - __umul24(slot, points)];
                ownerAndSlot = owner | (slot << 8);
                const Float4 light = s.light[2 * (lightBase + l)];
                V3 offset = mk(0.0f, 0.0f, 0.0f);
                if (MODE != MODE_PLAIN) {
                    /* the owner's pixel, from its place in the tile (same mapping as in trace_kernel) */
                    const uint32_t ownerX = tileX0 + ((owner >> 6) & 1u) * 8u + (owner & 7u);
                    const uint32_t ownerY = tileY0 + (owner >> 7) * 8u + ((owner & 63u) >> 3);
                    offset = light_offset(mode.hashed, mode.kernelModel, ownerY * width + ownerX, sh.treePosition[owner],
                                          mode.sample, lightBase + l, slot - l * perLight);
                }
                const V3 point = mk(sh.hitX[owner], sh.hitY[owner], sh.hitZ[owner]);
                /* the ray's bucket in its light's grid, if it has one, before anything else: the two reads in a row that
                 * lead to the bucket's first entry are under way while the ray is set up and the planes are tested */
                bool viaGrid = false;
                uint32_t alwaysCount = 0, first = 0;
                const uint32_t* lightRefs = nullptr;
                if (GRID && s.grid.info[RT_GRID_INFO_WORDS * (lightBase + l)] != 0u &&
                    fmaxf(fabsf(point.x), fmaxf(fabsf(point.y), fabsf(point.z))) <= s.grid.reach) {
                    viaGrid = true;
                    const uint32_t lightNumber = lightBase + l;
                    const uint32_t* const info = s.grid.info + RT_GRID_INFO_WORDS * lightNumber;
                    alwaysCount = info[1];
                    lightRefs = s.grid.refs + ((size_t)info[2] | ((size_t)info[3] << 32));
                    grid_list(s.grid, lightNumber, grid_bucket(s.grid, point, light), first, listLeft);
                }
                r = shadow_ray(point, light, offset, tmax);
                if (GRID && viaGrid) {
                    list = lightRefs + first;
                    nextRef = *list;      /* one word past the last list is allocated: in bounds even for an empty bucket */
                }
                bool hidden = false;
                uint32_t clear = 0;     /* bit i: the point and the light are clearly on the same side of plane i */
                if (MODE == MODE_PLAIN) {
                    const uint32_t both = sh.planeSides[owner] & __float_as_uint(s.light[2 * (lightBase + l) + 1].w);
                    clear = (both | (both >> 16)) & 0xFFFFu;
                }
                /* does plane i hide the light?  A fused estimate first: a plane met well beyond the light (or behind the
                 * ray) cannot; the margin covers the estimate's rounding, anything closer gets the exact test
                 * (reference: src/raytracer.c:70-86) */
                auto plane_hides = [&](uint32_t i) {
                    const Float4 pl = i < RT_PLANES_IN_LDS ? st.plane[i] : s.plane[i];
                    const float den = __builtin_fmaf(pl.x, r.d.x, __builtin_fmaf(pl.y, r.d.y, pl.z * r.d.z));
                    if (fabsf(den) >= 2e-5f) {
                        const float num = -pl.w - __builtin_fmaf(pl.x, r.o.x, __builtin_fmaf(pl.y, r.o.y, pl.z * r.o.z));
                        const float inv = __builtin_amdgcn_rcpf(den);
                        const float est = num * inv;
                        const float margin = __builtin_fmaf(slack, fabsf(inv), __builtin_fmaf(4e-6f, fabsf(est), slack));
                        if (est < -margin || est > tmax + margin) return false;
                    }
                    float t;
                    return hit_plane(pl, r, t) && (strict ? t < tmax : t <= tmax);
                };
                /* the first planes still to test, as a bit mask (all of them where no sides are known) */
                uint32_t todo = ~clear & (s.planeCount >= RT_SIDE_PLANES ? 0xFFFFu : (1u << s.planeCount) - 1u);
                while (todo != 0u && !hidden) {
                    hidden = plane_hides((uint32_t)__builtin_ctz(todo));
                    todo &= todo - 1u;
                }
                for (uint32_t i = RT_SIDE_PLANES; i < s.planeCount && !hidden; i++) hidden = plane_hides(i);
                if (GRID && viaGrid && !hidden) {
                    /* the few primitives too close to this light, or too large, to be listed by direction */
                    for (uint32_t i = 0; i < alwaysCount && !hidden; i++)
                        hidden = grid_entry_hides<MANY_LEAVES>(s, r, tmax, s.grid.always[(lightBase + l) * RT_GRID_ALWAYS + i], slack, walkStats);
                    if (!hidden) state = listLeft != 0u ? LIST : NEED;      /* an empty bucket: nothing between the point and the light */
                }
                if (hidden) {
                    atomicOr(&sh.occluded[owner], 1u << slot);
                } else if (!viaGrid) {
                    w = make_walk(r, MODE == MODE_GENERAL ? s.boxPad : 0.0f);
                    walker_start<true, MODE == MODE_GENERAL>(s, st, w, tmax, k, walkStats);
                    /* the leaf that last hid a ray towards this light goes first: neighbouring
                     * points are mostly hidden by the same few primitives, and whichever occluder
                     * is found first, the answer is the same */
                    const unsigned long long guess = ShadowSchedule<MANY_LEAVES>::guessOccluder ? sh.lastOccluder[slot] : 0ull;
                    queued = 0;
                    if ((uint32_t)(guess >> 32) != 0u) { link0 = (uint32_t)guess; kind0 = (uint32_t)(guess >> 32); queued = 1; }
                    walkDone = false;
                    state = RUN;
#ifdef RT_WALK_STATS
                    raySteps = 0;
#endif
                }
            }
        }
        if (GRID && __ballot(state == LIST) != 0ull) {
#ifdef RT_WALK_STATS
            if ((threadIdx.x & 63u) == 0u) walkStats[RT_STAT_LIST_TURNS]++;
            if (state == LIST) walkStats[RT_STAT_LIST_LANES]++;
#endif
            if (state == LIST) {
                const uint32_t ref = nextRef;
                listLeft--;
                nextRef = *++list;      /* under way during the test; after the last entry it reads the next bucket's first, or the spare word */
                if (grid_entry_hides<MANY_LEAVES>(s, r, tmax, ref, slack, walkStats)) {
                    atomicOr(&sh.occluded[ownerAndSlot & 0xFFu], 1u << (ownerAndSlot >> 8));
                    state = NEED;
                } else if (listLeft == 0u) {
                    state = NEED;
                }
            }
        }
        /* walk, or test the oldest leaf of every lane that has one: see ShadowSchedule */
        const bool canWalk = state == RUN && !walkDone && queued < ShadowSchedule<MANY_LEAVES>::leafSlots;
        bool walkNow = __ballot(canWalk) != 0;
        if (MANY_LEAVES && walkNow) {
            const bool stuck = state == RUN && queued > 0u && !canWalk;
            walkNow = (uint32_t)__popcll(__ballot(stuck)) * 4u < (uint32_t)__popcll(__ballot(state == RUN));
        }
        if (walkNow) {
#ifdef RT_WALK_STATS
            if ((threadIdx.x & 63u) == 0u) walkStats[RT_STAT_SHADOW_STEPS]++;
            if (canWalk) walkStats[RT_STAT_STEP_WALKING]++;
            else if (state == RUN) walkStats[RT_STAT_STEP_HOLDING]++;
            else if (state == NEED) walkStats[RT_STAT_STEP_WAITING]++;
            else walkStats[RT_STAT_STEP_FINISHED]++;
#endif
            if (canWalk) {
                uint32_t link = 0, kind = 0;
                const int step = walker_step<true, MODE == MODE_GENERAL>(s, w, st, tmax, k, link, kind, walkStats);
#ifdef RT_WALK_STATS
                raySteps++;
#endif
                if (step == 1) {
                    if (ShadowSchedule<MANY_LEAVES>::leafSlots == 1u || queued == 0u) { link0 = link; kind0 = kind; }
                    else if (queued == 1u) { link1 = link; kind1 = kind; }
                    else { link2 = link; kind2 = kind; }
                    queued++;
                } else if (step == 2) {
                    walkDone = true;
                    if (queued == 0u) { state = NEED; RT_RAY_DONE(); }     /* reached the light */
                }
            }
        } else if (state == RUN && queued > 0u) {
            const uint32_t link = link0, kind = kind0;
            if (ShadowSchedule<MANY_LEAVES>::leafSlots > 1u) { link0 = link1; kind0 = kind1; link1 = link2; kind1 = kind2; }
            queued--;
            Closest unused;
            unused.t = 0.0f; unused.id = 0; unused.material = 0;
            if (leaf_test<true, MANY_LEAVES>(s, r, link, kind, unused, tmax, walkStats, slack, strict)) {
                atomicOr(&sh.occluded[ownerAndSlot & 0xFFu], 1u << (ownerAndSlot >> 8));
                if (ShadowSchedule<MANY_LEAVES>::guessOccluder)
                    sh.lastOccluder[ownerAndSlot >> 8] = (unsigned long long)link | ((unsigned long long)kind << 32);
                state = NEED;
                RT_RAY_DONE();
            } else if (walkDone && queued == 0u) {
                state = NEED;
                RT_RAY_DONE();
            }
        }
    }
}

/* Walking an octree's topology keeps ten and more levels of stack per ray in LDS, so a CU holds five workgroups at most
 * and a sixth wave's worth of registers given up would buy nothing (21.1 vs 22.7 ms); the instantiation for scenes with
 * glass carries its stack of parked rays and likewise does better with the larger register budget. */
#ifndef RT_MIN_WAVES_OCTREE
#define RT_MIN_WAVES_OCTREE 5
#endif
#ifndef RT_MIN_WAVES_GLASS
#define RT_MIN_WAVES_GLASS 5
#endif
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 6   /* waves per SIMD the register allocation must allow (80 VGPRs).  The kernel is bound by vector
                            instruction issue and by how many lanes each instruction finds alive; more resident waves give the
                            scheduler more of them to choose from: 2.53 ms against 2.75 at 5 (96 VGPRs, no scratch at all) on
                            the 1M-triangle frame, with the walk stacks and the parked state small enough for six
                            workgroups' LDS to fit a CU.  The instantiation with light grids fits the 80 registers without
                            scratch memory; the one whose shadow rays walk spills three words per lane. */
#endif
/* One workgroup renders one 16x16 tile, level by level of the ray tree:
 *   1. every thread traces its current ray (primary, reflected or refracted) to its closest hit;
 *   2. the tile's shading points x lights are traced as one shadow-ray queue (shadow_pass);
 *   3. every thread shades its point with the lights found visible and spawns the next ray(s).
 * A glass surface has two children: one is continued, the other parked on a small private stack. */
/* MODE_PLAIN is the measured path: hard shadows, the CPU shading model, one ray per pixel, scenes
 * within the default slack.  MODE_JITTER is the same with the hashed light jitter (the reference's
 * soft shadows).  MODE_GENERAL adds, by run-time switches, the device shading model (reference:
 * src/kernel.cl:671-714), the sub-pixel sample grid (src/kernel.cl:817-843) and larger scenes. */
/* GLASS: the scene has a material that refracts.  Only then can a surface have two children, and only that
 * instantiation carries the private stack of parked rays (the kernel's one use of scratch memory beyond a few spilled words). */
/* GRID: the scene has direction grids for (some of) its lights and shadow rays towards those go through them. */
template <int MODE, bool MANY_LEAVES, bool GLASS, bool GRID>
__global__ __launch_bounds__(RT_BLOCK, (MANY_LEAVES ? RT_MIN_WAVES_OCTREE : GLASS ? RT_MIN_WAVES_GLASS : RT_MIN_WAVES)) void trace_kernel(const FrameParams p) {
    extern __shared__ unsigned char lds[];
    const DeviceScene& s = p.scene;
    WalkStack st;
    ShadowShared sh;
    /* What a thread knows about its shading point is parked here while the tile's shadow rays are traced:
     * during that pass a lane works on other threads' rays, and values kept in registers across it would
     * only be spilled to scratch memory.  [slot * RT_BLOCK + thread]. */
    float* park;
    {
        /* Everything of a fixed size comes first, at offsets known when the kernel is compiled: a thread's word in any
         * of these arrays is then one address register (4 * thread) plus a constant in the instruction.  The walk stacks,
         * whose size follows the depth of the hierarchy, come behind them, the wavefronts' shared stacks last. */
        size_t at = 0;
        park = reinterpret_cast<float*>(lds + at); at += RT_PARK_SLOTS * RT_BLOCK * sizeof(float);
        sh.hitX = reinterpret_cast<float*>(lds + at); at += RT_BLOCK * sizeof(float);
        sh.hitY = reinterpret_cast<float*>(lds + at); at += RT_BLOCK * sizeof(float);
        sh.hitZ = reinterpret_cast<float*>(lds + at); at += RT_BLOCK * sizeof(float);
        sh.occluded = reinterpret_cast<uint32_t*>(lds + at); at += RT_BLOCK * sizeof(uint32_t);
        sh.pointList = reinterpret_cast<uint16_t*>(lds + at); at += RT_BLOCK * sizeof(uint16_t);
        sh.treePosition = reinterpret_cast<uint16_t*>(lds + at); at += RT_BLOCK * sizeof(uint16_t);
        sh.planeSides = reinterpret_cast<uint32_t*>(lds + at); at += RT_BLOCK * sizeof(uint32_t);
        sh.lastOccluder = reinterpret_cast<unsigned long long*>(lds + at); at += 32 * sizeof(unsigned long long);
        sh.pointCount = reinterpret_cast<uint32_t*>(lds + at); at += sizeof(uint32_t);
        sh.farPoints = reinterpret_cast<uint32_t*>(lds + at); at += sizeof(uint32_t);
        sh.nextJob = reinterpret_cast<uint32_t*>(lds + at); at += sizeof(uint32_t);
        sh.rayCount = reinterpret_cast<uint32_t*>(lds + at); at += 4 * sizeof(uint32_t);
        sh.anyRay = reinterpret_cast<uint32_t*>(lds + at); at += 3 * sizeof(uint32_t);
        sh.rayCountOfLevel = reinterpret_cast<uint32_t*>(lds + at);      /* in the padding up to the next 16 bytes */
        at = (at + 15) & ~(size_t)15;
        uint4* root = reinterpret_cast<uint4*>(lds + at);
        if (threadIdx.x < 5u) root[threadIdx.x] = reinterpret_cast<const uint4*>(s.nodes)[threadIdx.x];   /* visible after the first barrier below */
        st.root = root;
        at += 5 * sizeof(uint4);
        Float4* planes = reinterpret_cast<Float4*>(lds + at);
        if (threadIdx.x < min(s.planeCount, RT_PLANES_IN_LDS)) planes[threadIdx.x] = s.plane[threadIdx.x];
        st.plane = planes;
        at += RT_PLANES_IN_LDS * sizeof(Float4);
        static_assert(RT_FIXED_LDS_BYTES == RT_PARK_SLOTS * RT_BLOCK * 4 + RT_BLOCK * (3 * 4 + 4 + 2 + 2 + 4) + 32 * 8 + 10 * 4 + 8 + 5 * 16 + RT_PLANES_IN_LDS * 16,
                      "the host sizes a workgroup's LDS from RT_FIXED_LDS_BYTES");
        st.entry = reinterpret_cast<uint32_t*>(lds + RT_FIXED_LDS_BYTES);
        /* a node is pushed only while one of its inner children is walked, so the deepest inner level never is */
        st.levels = s.treeLevels > 1u ? s.treeLevels - 1u : 1u;
        st.trail = st.entry + st.levels * RT_BLOCK;      /* (RT_BLOCK / 64) * levels words */
        /* the shelves lie in the parked words that are only written once the closest hits are found (from PARK_DIRECTION on) */
        st.shelf = reinterpret_cast<uint32_t*>(park) + 6u * RT_BLOCK;
    }
    if (threadIdx.x < 32u) sh.lastOccluder[threadIdx.x] = 0ull;   /* visible after the first barrier below */

    /* Workgroups are dealt round-robin over the 8 XCDs (workgroup b runs on XCD b % 8, each with
     * its own L2).  Tiles are handed out in runs of `xcdRun` consecutive tiles per XCD, round
     * robin: neighbouring tiles share an L2 inside a run, while every XCD still gets the same mix
     * of cheap and expensive screen regions (one contiguous eighth of the screen per XCD leaves
     * most XCDs idle while the one that owns the mesh finishes). */
    const long long startedAt = p.tileCost ? clock64() : 0;
    uint32_t local = blockIdx.x;
    if (p.tileOrder) {
        local = p.tileOrder[blockIdx.x];
    } else if (p.xcdRun > 0) {
        const uint32_t span = 8u * p.xcdRun;
        const uint32_t full = (gridDim.x / span) * span;   /* the ragged tail keeps identity order */
        if (local < full) {
            const uint32_t xcd = local % 8u, j = local / 8u;
            local = (j / p.xcdRun) * span + xcd * p.xcdRun + (j % p.xcdRun);
        }
    }
    const uint32_t tile = tile_of_block(p, local);
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    /* The thread's place in the tile and its pixel are worked out again wherever they are needed (a few integer
     * instructions from the thread number, which passes through an empty asm so that the results are not carried
     * across the whole kernel in registers of their own). */
#define RT_PIXEL()                                                                                                   \
    uint32_t tidHere = threadIdx.x;                                                                                  \
    asm volatile("" : "+v"(tidHere));                                                                                \
    const uint32_t inTileX = ((tidHere >> 6) & 1u) * 8u + (tidHere & 7u), inTileY = (tidHere >> 7) * 8u + ((tidHere & 63u) >> 3); \
    const uint32_t x = (tile % p.tilesX) * HIP_TRACE_TILE + inTileX, y = (tile / p.tilesX) * HIP_TRACE_TILE + inTileY
#define RT_LIVE() (tile < p.tileCount && x < p.width && y < p.height)

    /* ray counters live in LDS (one atomic per wavefront and level), walk statistics of the profiling build in registers */
    uint32_t walkStatsOfLane[RT_COUNTERS > 4 ? RT_COUNTERS - 4 : 1] = {};
    uint32_t* const walkStats = walkStatsOfLane;
    (void)walkStats;
    if (threadIdx.x < 4u) sh.rayCount[threadIdx.x] = 0u;   /* visible after the first barrier below */
    if (threadIdx.x == 0u) { *sh.pointCount = 0u; *sh.farPoints = 0u; sh.anyRay[0] = 0u; sh.anyRay[1] = 0u; sh.anyRay[2] = 0u; *sh.rayCountOfLevel = 0u; }
    __syncthreads();
    Task pendingTasks[GLASS ? HIP_TRACE_MAX_DEPTH : 1];
    /* position of the current ray's hit in the pixel's ray tree; only the general kernel keeps it */
    uint32_t pendingPath[GLASS && MODE != MODE_PLAIN ? HIP_TRACE_MAX_DEPTH : 1];
    const uint32_t tileX0 = (tile % p.tilesX) * HIP_TRACE_TILE, tileY0 = (tile / p.tilesX) * HIP_TRACE_TILE;
    ShadeMode mode;
    mode.hashed = MODE == MODE_JITTER || (MODE == MODE_GENERAL && p.jitterMode == HIP_TRACE_JITTER_HASHED);
    mode.kernelModel = MODE == MODE_GENERAL && p.shadingModel == HIP_TRACE_SHADING_KERNEL_MODEL;
    mode.samplesPerLight = mode.kernelModel ? 4u : 1u;
    mode.sample = 0;
    const uint32_t lightsPerPass = 32u / mode.samplesPerLight;
    /* the measured instantiation has the slack of the conservative tests as a literal; scenes larger
     * than that literal covers are sent to the general one by the host */
    const float slack = MODE == MODE_GENERAL ? s.filterMargin : RT_DEFAULT_SLACK;
    const uint32_t sampleCount = MODE == MODE_GENERAL ? p.raysX * p.raysY : 1u;
    /* parked words: see the store below */
    enum { PARK_PIXEL = 0, PARK_THROUGH = 3, PARK_DIRECTION = 6, PARK_REFLECTION = 9, PARK_REFRACTION = 10, PARK_PRIMITIVE = 11, PARK_STATE = 12 };
    enum { STATE_DEPTH_MASK = 0xFu, STATE_KIND_SHIFT = 4, STATE_LIT = 0x40u, STATE_HAD = 0x80u, STATE_MATERIAL_SHIFT = 8 };
    static_assert(PARK_STATE + 1 == RT_PARK_SLOTS, "the host sizes the parking area by RT_PARK_SLOTS");
    static_assert(PARK_DIRECTION == 6 && PARK_DIRECTION + RT_SHELF_SLOTS <= RT_PARK_SLOTS, "the shelves of shared_walk lie in the words from PARK_DIRECTION on");
#define RT_PARK(slot) park[(slot) * RT_BLOCK + tid]
#define RT_PARK_V3(slot, v) do { RT_PARK(slot) = (v).x; RT_PARK((slot) + 1) = (v).y; RT_PARK((slot) + 2) = (v).z; } while (0)
#define RT_UNPARK_V3(slot) mk(RT_PARK(slot), RT_PARK((slot) + 1), RT_PARK((slot) + 2))
    /* the pixel's running sum lives in LDS for the whole kernel: only its owner touches it */
    RT_PARK_V3(PARK_PIXEL, mk(0, 0, 0));

    uint32_t turn = 0;      /* which of the three "any ray" words the next level uses: goes on from sample to sample */
    for (uint32_t sample = 0; sample < sampleCount; sample++) {
    mode.sample = sample;
    uint32_t pendingCount = 0, path = 1;
    uint32_t levelOfSample = 0;      /* levels of the ray trees traced so far for this sample */
    /* A thread's next ray waits in LDS between the levels of its ray tree, not in registers: origin in the thread's
     * shading-point words, direction, weight and depth/kind in the parked words of the same meaning.  Kept in registers
     * across the loop they are eleven more values alive through both walks, and were spilled. */
#define RT_POST_TASK(o_, d_, weight_, depth_, kind_)                                                              \
    do {                                                                                                          \
        sh.hitX[tid] = (o_).x; sh.hitY[tid] = (o_).y; sh.hitZ[tid] = (o_).z;                                      \
        RT_PARK_V3(PARK_DIRECTION, (d_));                                                                         \
        RT_PARK_V3(PARK_THROUGH, (weight_));                                                                      \
        RT_PARK(PARK_STATE) = __uint_as_float((depth_) | ((kind_) << STATE_KIND_SHIFT));                          \
    } while (0)
    bool have = false;
    {
        RT_PIXEL();
        have = RT_LIVE();
    if (have) {
        V3 lens = mk(0.0f / 2.0f, 0.0f / 2.0f, 0.0f / 2.0f);
        if (MODE == MODE_GENERAL && mode.kernelModel && mode.hashed)   /* depth of field: reference: src/kernel.cl:832-838 */
            lens = mk(kernel_bilateral(y * p.width + x, 0, sample, 0, 0, 0) / 2.0f, kernel_bilateral(y * p.width + x, 0, sample, 0, 0, 1) / 2.0f,
                      kernel_bilateral(y * p.width + x, 0, sample, 0, 0, 2) / 2.0f);
        RayD r = MODE == MODE_GENERAL ? primary_ray(p, x, y, sample % p.raysX, sample / p.raysX, p.deltaX, p.deltaY, lens)
                         : primary_ray(p, x, y, 0, 0, 1.0f / (float)p.width, 1.0f / (float)p.height, lens);
        const float share = MODE == MODE_GENERAL ? p.contribution : 1.0f;
        RT_POST_TASK(r.o, r.d, mk(share, share, share), 0u, (uint32_t)RAY_PRIMARY);
    }
    }

    RT_PHASE_START();
    /* "Does any thread of the workgroup have a ray for this level?" with one barrier: a wavefront that has one sets the
     * level's flag word, everybody reads it behind the barrier, and the first thread clears the word of the level after
     * next -- last read a level ago by threads that have all passed this barrier since, and not written again before
     * the next barrier, which the first thread joins only after clearing.  (The library's workgroup-wide "or" does
     * the same with the threads' three-dimensional ids and more barriers.)  The level that finds no ray leaves its word
     * zero and the word before it set: the next sample of the pixel starts with the same word again, behind a barrier
     * of its own so that no thread sets it while another still reads it. */
    for (;; turn = turn == 2u ? 0u : turn + 1u, levelOfSample++) {
        if (__ballot(have) != 0ull && lane == 0u) sh.anyRay[turn] = 1u;
        __syncthreads();
        if (sh.anyRay[turn] == 0u) break;
        if (tid == 0u) sh.anyRay[turn == 0u ? 2u : turn - 1u] = 0u;
        RT_PHASE(RT_STAT_CYCLES_BARRIER);
        /* ---- 1. closest hit of this thread's current ray ---- */
        /* the point count is zero here (set at the start, and again by the level before once its passes were through),
         * and a thread's mask is written by others only during the shadow pass, behind the next barrier */
        sh.occluded[tid] = 0;
        {
            /* count this level's rays by kind: one LDS atomic per wavefront and kind present */
            {
                const uint32_t posted = have ? __float_as_uint(RT_PARK(PARK_STATE)) : 0u;
                const uint32_t taskKind = (posted >> STATE_KIND_SHIFT) & 3u;
                const uint32_t kinds[3] = {RAY_PRIMARY, RAY_REFLECTION, RAY_REFRACTION};
                for (uint32_t kind : kinds) {
                    if (!GLASS && kind == RAY_REFRACTION) continue;
                    const uint32_t n = (uint32_t)__popcll(__ballot(have && taskKind == kind));
                    if (n != 0u && lane == 0u) atomicAdd(&sh.rayCount[kind], n);
                }
            }
            /* Who traces whose ray.  On the first level every thread has one and traces its own: the primary rays of a
             * wavefront's 8x8 pixels go together.  On the later levels only the threads whose last hit reflects or refracts
             * have one -- scattered over the tile, a quarter of them on the 1M-triangle frame -- and a walk for a quarter
             * of a wavefront's lanes costs what a walk for all of them costs.  So the tile's rays are gathered: the
             * threads that have one put their numbers on a list, thread j traces the ray of the j-th thread on it, out of
             * that thread's words in LDS and back into them, and the wavefronts beyond the list's end go straight to the
             * next barrier.  (The list lies in the words for the points' tree positions, which the plain kernel does not
             * use; the other kernels trace every ray where it is.) */
            const bool gather = MODE == MODE_PLAIN && levelOfSample != 0u;      /* the same for the whole workgroup */
            bool trace = have;
            if (gather) {
                const unsigned long long mask = __ballot(have);
                uint32_t base = 0;
                if (lane == 0u && mask != 0ull) base = atomicAdd(sh.rayCountOfLevel, (uint32_t)__popcll(mask));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                if (have) sh.treePosition[base + rank] = (uint16_t)tid;
                else park[PARK_STATE * RT_BLOCK + tid] = __uint_as_float(0u);      /* nobody traces for this thread */
                __syncthreads();
                trace = tid < *sh.rayCountOfLevel;
            }
            /* whose ray this thread traces: looked up again behind the walk rather than kept through it in a register
             * (the thread number passes through an empty asm, so that neither the number nor the address of its list
             * entry is carried across the walk, where they were spilled) */
            auto owner_of_ray = [&]() {
                uint32_t here = threadIdx.x;
                asm volatile("" : "+v"(here));
                return gather ? (uint32_t)sh.treePosition[here] : here;
            };
#define RT_PARK_OF(slot) park[(slot) * RT_BLOCK + owner]
            uint32_t state = 0;
            if (trace) {
                uint32_t owner = owner_of_ray();
                const uint32_t posted = __float_as_uint(RT_PARK_OF(PARK_STATE));
                const uint32_t taskDepth = posted & STATE_DEPTH_MASK, taskKind = (posted >> STATE_KIND_SHIFT) & 3u;
                RayD r;
                r.o = mk(sh.hitX[owner], sh.hitY[owner], sh.hitZ[owner]);
                r.d = mk(RT_PARK_OF(PARK_DIRECTION), RT_PARK_OF(PARK_DIRECTION + 1), RT_PARK_OF(PARK_DIRECTION + 2));
                /* the primary rays of a wavefront's 8x8 pixels start out together.  Never a gathered ray: the walk a wavefront
                 * shares keeps its shelves in the executing thread's own parked words (from PARK_DIRECTION on), which at this
                 * point hold nothing only if the thread traces its own ray -- on a later level they are another ray's direction
                 * and state, not yet read by whoever traces it */
                const bool coherent = MODE != MODE_GENERAL && !gather && __ballot(taskKind != RAY_PRIMARY) == 0ull;
                const Closest hit = closest_hit<MANY_LEAVES, MODE == MODE_GENERAL>(s, r, st, slack, coherent, walkStats);
                owner = owner_of_ray();
                const float t = hit.t;
                const uint32_t id = hit.id, materialIndex = hit.material;
                bool lit = false;
                if (materialIndex != 0) {
                    const Float4 m0 = s.material[3 * materialIndex], m1 = s.material[3 * materialIndex + 1];
                    const V3 hitPoint = vadd(r.o, vmul(r.d, t));
                    /* weight of everything this surface sends towards the pixel: product of the material
                     * colours and mixing factors up the ray tree */
                    const V3 weight = mk(RT_PARK_OF(PARK_THROUGH), RT_PARK_OF(PARK_THROUGH + 1), RT_PARK_OF(PARK_THROUGH + 2));
                    const V3 through = mk(weight.x * m1.x, weight.y * m1.y, weight.z * m1.z);
                    /* a perfect mirror scales direct light by (1 - reflectionIndex) = 0
                     * (reference: src/raytracer.c:308), so its shadow rays are not traced */
                    lit = (1 - m0.x) != 0.0f && s.lightCount > 0;
                    sh.hitX[owner] = hitPoint.x; sh.hitY[owner] = hitPoint.y; sh.hitZ[owner] = hitPoint.z;
                    RT_PARK_OF(PARK_PRIMITIVE) = __uint_as_float(id);      /* the normal is looked up again from it after the pass */
                    RT_PARK_OF(PARK_THROUGH) = through.x; RT_PARK_OF(PARK_THROUGH + 1) = through.y; RT_PARK_OF(PARK_THROUGH + 2) = through.z;
                    RT_PARK_OF(PARK_DIRECTION) = r.d.x; RT_PARK_OF(PARK_DIRECTION + 1) = r.d.y; RT_PARK_OF(PARK_DIRECTION + 2) = r.d.z;
                    RT_PARK_OF(PARK_REFLECTION) = m0.x;
                    RT_PARK_OF(PARK_REFRACTION) = m0.y;
                }
                if (taskKind == RAY_PRIMARY && (MODE != MODE_GENERAL || sample == 0)) {      /* never a gathered ray */
                        RT_PIXEL();
                    /* where the pixel's colour goes: row-major, or among this rank's tiles */
                    const size_t rowMajor = p.worldSize == 1 ? (size_t)y * p.width + x
                                                             : (size_t)local * (HIP_TRACE_TILE * HIP_TRACE_TILE) + inTileY * HIP_TRACE_TILE + inTileX;
                    if (p.hitMask) p.hitMask[rowMajor] = materialIndex != 0 ? 1 : 0;
                    if (p.primitiveId) p.primitiveId[rowMajor] = id;
                    if (p.hitDistance) p.hitDistance[rowMajor] = t;
                }
                if (lit) {
                    if (MODE == MODE_PLAIN) {
                        const V3 at = mk(sh.hitX[owner], sh.hitY[owner], sh.hitZ[owner]);
                        uint32_t sides = 0;
                        if (fmaxf(fabsf(at.x), fmaxf(fabsf(at.y), fabsf(at.z))) <= RT_SIDE_REACH) {
                            for (uint32_t i = 0; i < min(s.planeCount, RT_SIDE_PLANES); i++) {
                                const Float4 pl = st.plane[i];
                                const float side = __builtin_fmaf(pl.x, at.x, __builtin_fmaf(pl.y, at.y, __builtin_fmaf(pl.z, at.z, pl.w)));
                                sides |= (side > RT_SIDE_MARGIN ? 1u : 0u) << i;
                                sides |= (side < -RT_SIDE_MARGIN ? 0x10000u : 0u) << i;
                            }
                        }
                        sh.planeSides[owner] = sides;
                        if (GRID && !(fmaxf(fabsf(at.x), fmaxf(fabsf(at.y), fabsf(at.z))) <= s.grid.reach)) *sh.farPoints = 1u;
                    }
                    if (MODE != MODE_PLAIN) sh.treePosition[tid] = (uint16_t)path;
                    sh.pointList[atomicAdd(sh.pointCount, 1u)] = (uint16_t)owner;
                }
                state = taskDepth | (taskKind << STATE_KIND_SHIFT) | (lit ? STATE_LIT : 0u) | STATE_HAD | (materialIndex << STATE_MATERIAL_SHIFT);
            }
            if (trace || !gather) { const uint32_t owner = owner_of_ray(); RT_PARK_OF(PARK_STATE) = __uint_as_float(state); }
#undef RT_PARK_OF
            /* the threads read what was traced for them behind the next barrier: that of the first shadow pass, or this one */
            if (gather && s.lightCount == 0u) __syncthreads();
        }

        RT_PHASE(RT_STAT_CYCLES_CLOSEST);
        /* the surface normal of this thread's hit, from the parked primitive number and the hit point (for a sphere the
         * expression of src/raytracer.c:109-110 on the same hit point: the same bits as when the hit was found) */
        auto parked_normal = [&](V3 hitPoint) {
            const uint32_t id = __float_as_uint(RT_PARK(PARK_PRIMITIVE));
            if (id < s.planeCount) {
                const Float4 pl = s.plane[id];
                return mk(pl.x, pl.y, pl.z);
            }
            if (id < s.planeCount + s.sphereCount) {
                const Float4 sp = s.sphere[id - s.planeCount];
                return vnorm(vsub(hitPoint, mk(sp.x, sp.y, sp.z)));
            }
            const Float4 pl = s.trianglePlane[id - s.planeCount - s.sphereCount];
            return mk(pl.x, pl.y, pl.z);
        };
        /* ---- 2 + 3. shadow rays of the tile, then direct light (reference: src/raytracer.c:274-310) ---- */
        V3 direct = mk(0, 0, 0);
        for (uint32_t lightBase = 0; lightBase < s.lightCount; lightBase += lightsPerPass) {
            const uint32_t lightsInPass = min(lightsPerPass, s.lightCount - lightBase);
            if (lightBase > 0) sh.occluded[tid] = 0;
            if (tid == 0) *sh.nextJob = 0;
            __syncthreads();
            RT_PHASE(RT_STAT_CYCLES_BARRIER);
            /* every (point, light, shadow sample) of the pass is traced exactly once */
            if (tid == 0) atomicAdd(&sh.rayCount[RAY_SHADOW], *sh.pointCount * lightsInPass * mode.samplesPerLight);
            shadow_pass<MODE, MANY_LEAVES, GRID>(s, st, sh, lightBase, lightsInPass, mode, tileX0, tileY0, p.width, walkStats);
            RT_PHASE(RT_STAT_CYCLES_SHADOW);
            __syncthreads();
            RT_PHASE(RT_STAT_CYCLES_BARRIER);
            const uint32_t state = __float_as_uint(park[PARK_STATE * RT_BLOCK + tid]);
            if (!(state & STATE_LIT)) continue;
            RT_PIXEL();
            (void)x; (void)y; (void)inTileX; (void)inTileY;
            const V3 hitPoint = mk(sh.hitX[tid], sh.hitY[tid], sh.hitZ[tid]), normal = parked_normal(hitPoint);
            const float keep = 1 - RT_PARK(PARK_REFLECTION);
            if (!(MODE == MODE_GENERAL && mode.kernelModel)) {
                /* What follows only colours the pixel (reference: src/raytracer.c:293-309); nothing of it feeds a later ray or a
                 * decision.  So it is worked out with the hardware's reciprocal and reciprocal square root (one unit in the
                 * last place each) instead of the exact square roots and divisions every geometric quantity gets: a relative
                 * 1e-6 on a light's contribution, against the 1/255 a channel is held to.  (The mirrored direction has to be
                 * normalised as the reference does: a plane's normal is whatever the scene gives, not a unit vector.) */
                const uint32_t hidden = sh.occluded[tid];
                const V3 eye = vsub(p.camPosition, hitPoint);
                const V3 toView = vmul(eye, __builtin_amdgcn_rsqf(vdot(eye, eye)));
                for (uint32_t l = 0; l < lightsInPass; l++) {
                    if (hidden & (1u << l)) continue;
                    const Float4 l0 = s.light[2 * (lightBase + l)], l1 = s.light[2 * (lightBase + l) + 1];
                    V3 offset = mk(0.0f, 0.0f, 0.0f);
                    if (MODE != MODE_PLAIN) offset = light_offset(mode.hashed, false, y * p.width + x, path, sample, lightBase + l, 0);
                    const V3 hitToLight = vadd(vsub(mk(l0.x, l0.y, l0.z), hitPoint), offset);
                    const float distanceSquared = vdot(hitToLight, hitToLight);
                    const V3 toLight = vmul(hitToLight, __builtin_amdgcn_rsqf(distanceSquared));
                    const float towards = vdot(toLight, normal);
                    const float cosAngle = clampf(towards, 0.0f, 1.0f);
                    const float lightStrength = l0.w * __builtin_amdgcn_rcpf(4 * 3.14159265358979323846f * distanceSquared);
                    const V3 emission = mk(l1.x, l1.y, l1.z);
                    const V3 mirrored = vsub(vmul(normal, 2.0f * towards), toLight);
                    float c = vdot(toView, mirrored) * __builtin_amdgcn_rsqf(vdot(mirrored, mirrored));
                    /* powf(c, 64) (reference: src/raytracer.c:304) as six squarings */
                    c = c * c; c = c * c; c = c * c; c = c * c; c = c * c; c = c * c;
                    direct = vadd(direct, vmul(emission, (cosAngle + c) * lightStrength * keep));
                }
            } else {
                /* reference: src/kernel.cl:671-714 -- four samples per light; the running sum of a
                 * light is divided by four after EVERY sample and added each time, as written there */
                const uint32_t hidden = sh.occluded[tid];
                const uint32_t materialIndex = state >> STATE_MATERIAL_SHIFT;
                const Float4 m0 = s.material[3 * materialIndex], m1 = s.material[3 * materialIndex + 1],
                             m2 = s.material[3 * materialIndex + 2];
                for (uint32_t l = 0; l < lightsInPass; l++) {
                    const Float4 l0 = s.light[2 * (lightBase + l)], l1 = s.light[2 * (lightBase + l) + 1];
                    const V3 emission = mk(l1.x, l1.y, l1.z);
                    V3 directLighting = mk(0.0f, 0.0f, 0.0f);
                    for (uint32_t k = 0; k < 4u; k++) {
                        if (!(hidden & (1u << (l * 4u + k)))) {
                            const V3 offset = light_offset(mode.hashed, true, y * p.width + x, path, sample, lightBase + l, k);
                            V3 hitToLight = vadd(vsub(mk(l0.x, l0.y, l0.z), hitPoint), offset);
                            const float distanceSquared = hitToLight.x * hitToLight.x + hitToLight.y * hitToLight.y + hitToLight.z * hitToLight.z;
                            const V3 toLightDir = vnorm(hitToLight);
                            float cosAngle = clampf(vdot(toLightDir, normal), 0.0f, 1.0f);
                            const float attenuation = 1.0f / (1.0f + 4 * 3.14159265358979323846f * distanceSquared);
                            const float lightStrength = l0.w * attenuation;
                            const V3 ambient = vmul(emission, m0.z * lightStrength);
                            const V3 diffuse = vmul(emission, m0.w * cosAngle * lightStrength);
                            const V3 toView = vnorm(vsub(p.camPosition, hitPoint));
                            const V3 mirrored = vreflect(vmul(toLightDir, -1.0f), normal);
                            cosAngle = powf(vdot(toView, mirrored), m1.w);
                            const V3 specular = vmul(emission, m2.x * cosAngle * lightStrength);
                            directLighting = vadd(directLighting, vmul(vadd(ambient, vadd(diffuse, specular)), keep));
                        }
                        directLighting = mk(directLighting.x / 4.0f, directLighting.y / 4.0f, directLighting.z / 4.0f);
                        direct = vadd(direct, directLighting);
                    }
                }
            }
        }

        /* every pass of this level has read the point count (each ends in a barrier): ready for the next level, whose
         * first addition to it comes behind the barrier at the top of the loop */
        if (tid == 0) { *sh.pointCount = 0; *sh.farPoints = 0; *sh.rayCountOfLevel = 0; }
        /* ---- the thread's own state comes back from LDS; secondary rays (reference: src/raytracer.c:235-271) ---- */
        const uint32_t state = __float_as_uint(park[PARK_STATE * RT_BLOCK + tid]);
        const uint32_t materialIndex = state >> STATE_MATERIAL_SHIFT, depth = state & STATE_DEPTH_MASK;
        const bool had = (state & STATE_HAD) != 0;
        have = false;
        if (had && materialIndex != 0) {
            const V3 through = RT_UNPARK_V3(PARK_THROUGH);
            if (state & STATE_LIT) {
                const V3 pixel = RT_UNPARK_V3(PARK_PIXEL);
                RT_PARK_V3(PARK_PIXEL, vadd(pixel, mk(direct.x * through.x, direct.y * through.y, direct.z * through.z)));
            }
            if (depth + 1 < p.maxDepth) {
                const float reflectionIndex = RT_PARK(PARK_REFLECTION), refractionIndex = GLASS ? RT_PARK(PARK_REFRACTION) : 0.0f;
                if (refractionIndex > 0 || reflectionIndex > 0) {
                    const V3 hitPoint = mk(sh.hitX[tid], sh.hitY[tid], sh.hitZ[tid]);
                    const V3 normal = parked_normal(hitPoint), rayDir = RT_UNPARK_V3(PARK_DIRECTION);
                    RayD reflected;
                    reflected.o = hitPoint;
                    reflected.d = vreflect(rayDir, normal);
                    nudge_forward(reflected);
                    if (GLASS && refractionIndex > 0) {
                        const float kr = fresnel(rayDir, normal, refractionIndex);
                        if (kr < 1) {
                            RayD refracted;
                            refracted.o = hitPoint;
                            refracted.d = refract(rayDir, normal, refractionIndex);
                            nudge_forward(refracted);
                            if (MODE != MODE_PLAIN) pendingPath[pendingCount] = 2u * path;
                            Task waiting;      /* the reflected ray waits */
                            waiting.o = reflected.o; waiting.d = reflected.d; waiting.weight = vmul(through, kr);
                            waiting.depth = depth + 1; waiting.kind = RAY_REFLECTION;
                            pendingTasks[pendingCount++] = waiting;
                            RT_POST_TASK(refracted.o, refracted.d, vmul(through, 1 - kr), depth + 1u, (uint32_t)RAY_REFRACTION);
                            if (MODE != MODE_PLAIN) path = 2u * path + 1u;
                        } else {
                            RT_POST_TASK(reflected.o, reflected.d, vmul(through, kr), depth + 1u, (uint32_t)RAY_REFLECTION);
                            if (MODE != MODE_PLAIN) path = 2u * path;
                        }
                    } else {
                        RT_POST_TASK(reflected.o, reflected.d, vmul(through, reflectionIndex), depth + 1u, (uint32_t)RAY_REFLECTION);
                        if (MODE != MODE_PLAIN) path = 2u * path;
                    }
                    have = true;
                }
            }
        }
        if (GLASS && had && !have && pendingCount > 0) {
            const Task waiting = pendingTasks[--pendingCount];
            RT_POST_TASK(waiting.o, waiting.d, waiting.weight, waiting.depth, waiting.kind);
            if (MODE != MODE_PLAIN) path = pendingPath[pendingCount];
            have = true;
        }
        RT_PHASE(RT_STAT_CYCLES_SHADING);
    }
    if (sampleCount > 1u) __syncthreads();
    }   /* sub-pixel samples */
#undef RT_POST_TASK
#undef RT_PARK
#undef RT_PARK_V3
#undef RT_UNPARK_V3

    {
        RT_PIXEL();
        if (RT_LIVE()) {
        const V3 pixel = mk(park[PARK_PIXEL * RT_BLOCK + tid], park[(PARK_PIXEL + 1) * RT_BLOCK + tid], park[(PARK_PIXEL + 2) * RT_BLOCK + tid]);
        const uint32_t rgba = to_unorm8(pixel.x) | (to_unorm8(pixel.y) << 8) | (to_unorm8(pixel.z) << 16) | 0xFF000000u;
        if (p.worldSize == 1) {
            p.out[(size_t)y * p.width + x] = rgba;
        } else {
            p.out[(size_t)local * (HIP_TRACE_TILE * HIP_TRACE_TILE) + inTileY * HIP_TRACE_TILE + inTileX] = rgba;
        }
        }
    }
#undef RT_PIXEL
#undef RT_LIVE

    if (p.tileCost && tid == 0) {
        const long long took = clock64() - startedAt;
        p.tileCost[local] = took > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)took;
    }

    __syncthreads();
    if (tid < 4u && sh.rayCount[tid] != 0u) atomicAdd(&p.counters[tid], (unsigned long long)sh.rayCount[tid]);
    /* walk statistics: one atomic per counter per wavefront */
#pragma unroll
    for (int k = 4; k < RT_COUNTERS; k++) {
        uint32_t v = walkStats[k - 4];
#pragma unroll
        for (int offset = 32; offset > 0; offset >>= 1) v += __shfl_down(v, offset, 64);
        if (lane == 0 && v) atomicAdd(&p.counters[k], (unsigned long long)v);
    }
}

/* Longest tiles first: orders the local tiles by the cycles they took in the previous frame,
 * descending (counting sort on 1024 cost classes; the order inside a class is arbitrary, which
 * cannot change a pixel).  A tile takes 0.3-0.6 ms whatever the frame size, so a frame of few tiles
 * -- a 4K frame split over eight devices lasts about four tile durations -- ends with a long
 * tail of under-filled CUs unless the slow tiles start first.  With many tiles per device the
 * tail is negligible and starting all expensive tiles together costs more than it saves, so
 * the host only uses this below a tile count.  One workgroup of 1024 threads. */
__global__ __launch_bounds__(1024) void order_tiles_kernel(const uint32_t* cost, uint32_t* order, uint32_t count) {
    __shared__ uint32_t bin[1024];
    __shared__ uint32_t highest;
    const uint32_t tid = threadIdx.x;
    bin[tid] = 0;
    if (tid == 0) highest = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t i = tid; i < count; i += 1024u) mine = max(mine, cost[i]);
    atomicMax(&highest, mine);
    __syncthreads();
    uint32_t shift = 0;
    while ((highest >> shift) >= 1024u) shift++;
    for (uint32_t i = tid; i < count; i += 1024u) atomicAdd(&bin[1023u - (cost[i] >> shift)], 1u);
    __syncthreads();
    if (tid == 0) {
        uint32_t running = 0;
        for (uint32_t b = 0; b < 1024u; b++) { const uint32_t n = bin[b]; bin[b] = running; running += n; }
    }
    __syncthreads();
    for (uint32_t i = tid; i < count; i += 1024u) order[atomicAdd(&bin[1023u - (cost[i] >> shift)], 1u)] = i;
}

/* Scatter the gathered tile-major buffers (rank-major, then local tile, then 16x16 pixels)
 * into the row-major frame.  One workgroup per tile.  T: the colour word, or a side output's element. */
template <typename T>
__global__ __launch_bounds__(RT_BLOCK) void untile_kernel(const T* gathered, T* frame, uint32_t width,
                                                         uint32_t height, uint32_t tilesX, uint32_t tileCount,
                                                         uint32_t worldSize, uint32_t rotate, uint32_t localTiles) {
    const uint32_t tile = blockIdx.x;
    if (tile >= tileCount) return;
    const uint32_t tx = tile % tilesX, ty = tile / tilesX;
    const uint32_t owner = (tile + (rotate ? ty : 0u)) % worldSize;
    const uint32_t local = tile / worldSize;
    const uint32_t px = threadIdx.x % HIP_TRACE_TILE, py = threadIdx.x / HIP_TRACE_TILE;
    const uint32_t x = tx * HIP_TRACE_TILE + px, y = ty * HIP_TRACE_TILE + py;
    if (x < width && y < height) {
        frame[(size_t)y * width + x] =
            gathered[((size_t)owner * localTiles + local) * (HIP_TRACE_TILE * HIP_TRACE_TILE) + threadIdx.x];
    }
}

/* Upload-time expansion of the leaf triangle records (device_layout.h: LeafTriangle): one
 * work-item per (wide node, child); a leaf child writes the filter record of each of its
 * triangles into its run.  The edge planes are derived in fp64. */
__global__ __launch_bounds__(RT_BLOCK) void expand_leaves_kernel(const QNode* nodes, uint32_t nodeCount,
                                                               const uint32_t* leafIndex, const Float4* trianglePlane,
                                                               const Float4* corner, LeafTriangle* out) {
    const uint64_t item = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x;
    if (item >= (uint64_t)nodeCount * 8u) return;
    const uint32_t kind = nodes[item >> 3].kind[item & 7u], link = nodes[item >> 3].link[item & 7u];
    if (kind & QNODE_INNER) return;
    const uint32_t sphereCount = kind >> 11, triangleCount = kind & 0x7FFu;
    for (uint32_t i = 0; i < triangleCount; i++) {
        const uint32_t at = link + sphereCount + i;
        const uint32_t k = leafIndex[at];
        LeafTriangle rec;
        rec.plane = trianglePlane[k];
        const double n[3] = {rec.plane.x, rec.plane.y, rec.plane.z};
        double v[3][3];
        for (int c = 0; c < 3; c++) { const Float4 p = corner[3 * (size_t)k + c]; v[c][0] = p.x; v[c][1] = p.y; v[c][2] = p.z; }
        for (int e = 0; e < 3; e++) {
            const double* a = v[e];
            const double* b = v[(e + 1) % 3];
            const double ed[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
            double m[3] = {n[1] * ed[2] - n[2] * ed[1], n[2] * ed[0] - n[0] * ed[2], n[0] * ed[1] - n[1] * ed[0]};
            const double len = sqrt(ed[0] * ed[0] + ed[1] * ed[1] + ed[2] * ed[2]);
            if (len > 0) { m[0] /= len; m[1] /= len; m[2] /= len; }
            rec.edge[e].x = (float)m[0]; rec.edge[e].y = (float)m[1]; rec.edge[e].z = (float)m[2];
            rec.edge[e].w = (float)(m[0] * a[0] + m[1] * a[1] + m[2] * a[2]);
        }
        out[at] = rec;
    }
}

/* The same records, one per triangle (for hierarchies whose leaves find them through the index run). */
__global__ __launch_bounds__(RT_BLOCK) void expand_triangles_kernel(uint32_t triangleCount, const Float4* trianglePlane,
                                                                  const Float4* corner, LeafTriangle* out) {
    const uint32_t k = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (k >= triangleCount) return;
    LeafTriangle rec;
    rec.plane = trianglePlane[k];
    const double n[3] = {rec.plane.x, rec.plane.y, rec.plane.z};
    double v[3][3];
    for (int c = 0; c < 3; c++) { const Float4 p = corner[3 * (size_t)k + c]; v[c][0] = p.x; v[c][1] = p.y; v[c][2] = p.z; }
    for (int e = 0; e < 3; e++) {
        const double* a = v[e];
        const double* b = v[(e + 1) % 3];
        const double ed[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
        double m[3] = {n[1] * ed[2] - n[2] * ed[1], n[2] * ed[0] - n[0] * ed[2], n[0] * ed[1] - n[1] * ed[0]};
        const double len = sqrt(ed[0] * ed[0] + ed[1] * ed[1] + ed[2] * ed[2]);
        if (len > 0) { m[0] /= len; m[1] /= len; m[2] /= len; }
        rec.edge[e].x = (float)m[0]; rec.edge[e].y = (float)m[1]; rec.edge[e].z = (float)m[2];
        rec.edge[e].w = (float)(m[0] * a[0] + m[1] * a[1] + m[2] * a[2]);
    }
    out[k] = rec;
}

/* ================================================================== host code */

/* Scratch of hip_trace_frame_static: the ray counters, the per-tile cycle counts and the tile launch
 * order that hip_trace_frame_device keeps in buffers allocated by hip_trace_init, here in one array
 * that is part of the loaded code object, so that rendering through it allocates nothing.  Every
 * device holds its own copy at its own address; host code never uses the array's name as an address,
 * only the pointer hipGetSymbolAddress returned for that device (scratch_of_device).
 * Layout: [0, 512) counters; then RT_STATIC_MAX_TILES cycle counts; then as many order entries. */
#define RT_STATIC_HEADER 512u
#define RT_STATIC_MAX_TILES HIP_TRACE_STATIC_MAX_TILES   /* 7680 x 4320 is 129600 tiles */
#define RT_STATIC_SCRATCH_BYTES HIP_TRACE_STATIC_SCRATCH_BYTES
static_assert(RT_COUNTERS * sizeof(unsigned long long) <= RT_STATIC_HEADER, "the counters take the first 512 bytes of the scratch");
static_assert(RT_STATIC_HEADER + 8u * HIP_TRACE_STATIC_MAX_TILES == HIP_TRACE_STATIC_SCRATCH_BYTES, "layout of the static scratch");
static __device__ __align__(256) unsigned char scratch[RT_STATIC_SCRATCH_BYTES];

#define RT_MAX_DEVICES 64
struct StaticScratch {
    unsigned char* base = nullptr;      /* device address of `scratch` on this device */
    const HipTrace* lastUser = nullptr; /* handle whose frame left its tile costs there */
};
static StaticScratch g_static[RT_MAX_DEVICES];
static std::mutex g_staticMutex;

struct HipTrace {
    HipTraceOptions opt;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t evStart = nullptr, evKernelStart = nullptr, evKernelStop = nullptr, evStop = nullptr;
    bool haveScene = false;
    LightGridBuild lightGrid;         /* per-light direction grids of the uploaded scene (may be empty) */
    float sceneExtent = 32.0f;        /* FlatScene::extent of the uploaded scene */
    bool sceneRefracts = false;       /* some material has refractionIndex > 0: surfaces can have two children */
    bool frameRecorded = false;
    /* scene */
    void* sceneBlock = nullptr;       /* one allocation holding every scene array */
    size_t sceneBytes = 0;
    DeviceScene dscene{};
    HipTraceSceneInfo info{};
    /* frame */
    uint32_t tilesX = 0, tilesY = 0, tileCount = 0, localTiles = 0, rotate = 0;
    uint32_t xcdRun = 16;              /* tiles per XCD run; RT_XCD_RUN overrides (tuning aid) */
    uint32_t raysX = 1, raysY = 1;     /* sub-pixel grid */
    float deltaX = 0.0f, deltaY = 0.0f, contribution = 1.0f;
    uint32_t* dOut = nullptr;          /* internal output buffer */
    uint8_t* dHitMask = nullptr;
    uint32_t* dPrimitiveId = nullptr;
    float* dHitDistance = nullptr;
    unsigned long long* dCounters = nullptr;
    unsigned long long* hCounters = nullptr;   /* pinned */
    uint32_t* dTileCost = nullptr;     /* cycles per local tile in the last frame (longest-first launch order) */
    uint32_t* dTileOrder = nullptr;
    bool orderTiles = false, haveCosts = false;
    /* a handle made by hip_trace_init_multi: one member handle per device entry (rank r of `members.size()`),
     * each with a stream of its own; this handle owns the gathered tiles and the frame on members[0]'s device */
    std::vector<HipTrace*> members;
    std::vector<hipEvent_t> memberDone;
    hipStream_t ownStream = nullptr;          /* stream created by the library (members of a multi-device handle) */
    void* dGathered = nullptr;
    void* dGatheredAux = nullptr;             /* the same for one side output at a time (allocated when first asked for) */
    hipEvent_t gatherDone = nullptr;          /* recorded on `stream` once a frame's parts have been copied out of the members' buffers */
    bool gatherRecorded = false;
    bool ownFrame = false;                    /* the most recent frame went into dOut (hip_trace_last_frame) */
    StaticScratch* staticScratch = nullptr;   /* this device's entry of the static scratch table (hip_trace_frame_static) */
    uint64_t sceneSerial = 0, staticSceneSerial = ~0ull;   /* uploads so far; the one the static scratch's tile costs belong to */
    std::string error;
};

static thread_local std::string g_initError;

/* Every entry point makes the handle's device current; the caller's own current device is put back on the way out. */
struct DeviceGuard {
    int saved = 0;
    bool have = false;
    DeviceGuard() { have = hipGetDevice(&saved) == hipSuccess; }
    ~DeviceGuard() { if (have) (void)hipSetDevice(saved); }
};

static int fail(HipTrace* h, int code, const std::string& message) {
    if (h) h->error = message; else g_initError = message;
    return code;
}

#define HIP_TRY(h, call)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail((h), e_ == hipErrorOutOfMemory ? HIP_TRACE_ERR_MEMORY : HIP_TRACE_ERR_DEVICE, \
                        std::string(#call) + ": " + hipGetErrorString(e_));                      \
    } while (0)

extern "C" uint32_t hip_trace_tile_owner(uint32_t tx, uint32_t ty, uint32_t tilesX, uint32_t worldSize) {
    if (worldSize <= 1 || tilesX == 0) return 0;
    const uint32_t rotate = (tilesX % worldSize == 0) ? 1u : 0u;
    return (ty * tilesX + tx + (rotate ? ty : 0u)) % worldSize;
}

static void release_scene(HipTrace* h) {
    rt_light_grid_release(h->lightGrid);
    if (h->sceneBlock) (void)hipFree(h->sceneBlock);
    h->sceneBlock = nullptr;
    h->sceneBytes = 0;
    h->haveScene = false;
    h->dscene = DeviceScene{};
    h->info = HipTraceSceneInfo{};
}

static void release_all(HipTrace* h) {
    release_scene(h);
    if (h->dOut) (void)hipFree(h->dOut);
    if (h->dHitMask) (void)hipFree(h->dHitMask);
    if (h->dPrimitiveId) (void)hipFree(h->dPrimitiveId);
    if (h->dHitDistance) (void)hipFree(h->dHitDistance);
    if (h->dCounters) (void)hipFree(h->dCounters);
    if (h->dTileCost) (void)hipFree(h->dTileCost);
    if (h->dTileOrder) (void)hipFree(h->dTileOrder);
    if (h->hCounters) (void)hipHostFree(h->hCounters);
    if (h->evStart) (void)hipEventDestroy(h->evStart);
    if (h->evKernelStart) (void)hipEventDestroy(h->evKernelStart);
    if (h->evKernelStop) (void)hipEventDestroy(h->evKernelStop);
    if (h->evStop) (void)hipEventDestroy(h->evStop);
    h->dOut = nullptr; h->dHitMask = nullptr; h->dPrimitiveId = nullptr; h->dHitDistance = nullptr;
    h->dCounters = nullptr; h->hCounters = nullptr; h->dTileCost = nullptr; h->dTileOrder = nullptr;
    h->evStart = h->evKernelStart = h->evKernelStop = h->evStop = nullptr;
}

/* Address of this device's copy of `scratch`, looked up once per device with that device current. */
static int scratch_of_device(HipTrace* h, int device, StaticScratch** out) {
    if (device < 0 || device >= RT_MAX_DEVICES) return fail(h, HIP_TRACE_ERR_ARGUMENT, "device ordinal beyond the static scratch table");
    std::lock_guard<std::mutex> lock(g_staticMutex);
    StaticScratch& entry = g_static[device];
    if (!entry.base) {
        void* address = nullptr;
        HIP_TRY(h, hipGetSymbolAddress(&address, HIP_SYMBOL(scratch)));
        if (!address) return fail(h, HIP_TRACE_ERR_DEVICE, "hipGetSymbolAddress(scratch) returned no address");
        entry.base = static_cast<unsigned char*>(address);
    }
    *out = &entry;
    return HIP_TRACE_OK;
}

static int init_body(HipTrace* h) {
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipEventCreate(&h->evStart));
    HIP_TRY(h, hipEventCreate(&h->evKernelStart));
    HIP_TRY(h, hipEventCreate(&h->evKernelStop));
    HIP_TRY(h, hipEventCreate(&h->evStop));
    HIP_TRY(h, hipMalloc(&h->dOut, (size_t)h->localTiles * HIP_TRACE_TILE * HIP_TRACE_TILE * 4 >
                                           (size_t)h->opt.width * h->opt.height * 4
                                       ? (size_t)h->localTiles * HIP_TRACE_TILE * HIP_TRACE_TILE * 4
                                       : (size_t)h->opt.width * h->opt.height * 4));
    HIP_TRY(h, hipMalloc(&h->dCounters, RT_COUNTERS * sizeof(unsigned long long)));
    HIP_TRY(h, hipHostMalloc(&h->hCounters, RT_COUNTERS * sizeof(unsigned long long)));
    if (h->orderTiles) {
        HIP_TRY(h, hipMalloc(&h->dTileCost, (size_t)h->localTiles * sizeof(uint32_t)));
        HIP_TRY(h, hipMalloc(&h->dTileOrder, (size_t)h->localTiles * sizeof(uint32_t)));
    }
    return scratch_of_device(h, h->device, &h->staticScratch);
}

extern "C" int hip_trace_init(HipTrace** out, const HipTraceOptions* options) {
    if (out) *out = nullptr;
    if (!out || !options) return fail(nullptr, HIP_TRACE_ERR_ARGUMENT, "hip_trace_init: NULL argument");
    const HipTraceOptions& o = *options;
    if (o.width == 0 || o.height == 0 || o.width > 65536 || o.height > 65536)
        return fail(nullptr, HIP_TRACE_ERR_ARGUMENT, "hip_trace_init: width/height must be in [1, 65536]");
    if (o.maxDepth == 0 || o.maxDepth > HIP_TRACE_MAX_DEPTH)
        return fail(nullptr, HIP_TRACE_ERR_ARGUMENT, "hip_trace_init: maxDepth must be in [1, HIP_TRACE_MAX_DEPTH]");
    if (o.raysPerPixel == 0 || o.raysPerPixel > 64)
        return fail(nullptr, HIP_TRACE_ERR_ARGUMENT, "hip_trace_init: raysPerPixel must be in [1, 64]");
    if (o.shadingModel != HIP_TRACE_SHADING_CPU_MODEL && o.shadingModel != HIP_TRACE_SHADING_KERNEL_MODEL)
        return fail(nullptr, HIP_TRACE_ERR_UNSUPPORTED, "hip_trace_init: unknown shadingModel");
    if (o.jitterMode != HIP_TRACE_JITTER_ZERO && o.jitterMode != HIP_TRACE_JITTER_HASHED)
        return fail(nullptr, HIP_TRACE_ERR_UNSUPPORTED, "hip_trace_init: unknown jitterMode");
    if (o.worldSize == 0 || o.rank >= o.worldSize || o.worldSize > 1024)
        return fail(nullptr, HIP_TRACE_ERR_ARGUMENT, "hip_trace_init: need 0 <= rank < worldSize <= 1024");
    if (o.accel != HIP_TRACE_ACCEL_CALLER_OCTREE && o.accel != HIP_TRACE_ACCEL_PRIVATE_BVH)
        return fail(nullptr, HIP_TRACE_ERR_UNSUPPORTED, "hip_trace_init: unknown accel");
    if (o.lightGrid != HIP_TRACE_LIGHT_GRID_AUTO && o.lightGrid != HIP_TRACE_LIGHT_GRID_OFF)
        return fail(nullptr, HIP_TRACE_ERR_UNSUPPORTED, "hip_trace_init: unknown lightGrid");
    for (uint32_t r : o.reserved)
        if (r != 0) return fail(nullptr, HIP_TRACE_ERR_ARGUMENT, "hip_trace_init: reserved fields must be zero");

    int deviceCount = 0;
    hipError_t e = hipGetDeviceCount(&deviceCount);
    if (e != hipSuccess || deviceCount == 0)
        return fail(nullptr, HIP_TRACE_ERR_DEVICE, std::string("hip_trace_init: no HIP device (") + hipGetErrorString(e) + ")");
    int device = o.device;
    if (device < 0) {
        e = hipGetDevice(&device);
        if (e != hipSuccess) return fail(nullptr, HIP_TRACE_ERR_DEVICE, std::string("hipGetDevice: ") + hipGetErrorString(e));
    }
    if (device >= deviceCount) return fail(nullptr, HIP_TRACE_ERR_ARGUMENT, "hip_trace_init: device ordinal out of range");

    DeviceGuard restoreDevice;
    HipTrace* h = new (std::nothrow) HipTrace();
    if (!h) return fail(nullptr, HIP_TRACE_ERR_MEMORY, "hip_trace_init: out of host memory");
    h->opt = o;
    h->device = device;
    h->tilesX = (o.width + HIP_TRACE_TILE - 1) / HIP_TRACE_TILE;
    h->tilesY = (o.height + HIP_TRACE_TILE - 1) / HIP_TRACE_TILE;
    h->tileCount = h->tilesX * h->tilesY;
    h->localTiles = (h->tileCount + o.worldSize - 1) / o.worldSize;
    h->rotate = (o.worldSize > 1 && h->tilesX % o.worldSize == 0) ? 1u : 0u;
    if (const char* run = std::getenv("RT_XCD_RUN")) h->xcdRun = (uint32_t)std::strtoul(run, nullptr, 10);
    /* longest-first launch order pays below about six tiles per resident workgroup slot (measured:
     * -5 % kernel time at 4050 tiles, -1.5 % at 8100, +5 % at 32400); RT_TILE_ORDER=0/1 overrides */
    h->orderTiles = h->localTiles <= RT_ORDER_BELOW;
    if (const char* env = std::getenv("RT_TILE_ORDER")) h->orderTiles = std::atoi(env) != 0;
    {
        /* the sub-pixel grid, as the reference derives it from raysPerPixel (src/gpu.c:446-465) */
        const uint32_t raysPerPixel = o.raysPerPixel;
        h->contribution = 1.0f / (float)raysPerPixel;
        const float pixelWidth = 1.0f / (float)o.width, pixelHeight = 1.0f / (float)o.height;
        const float rootTerm = sqrtf(pixelWidth / pixelHeight * raysPerPixel + powf(pixelWidth - pixelHeight, 2) / 4 * powf(pixelHeight, 2));
        h->raysX = h->raysY = 1;
        h->deltaX = pixelWidth;
        h->deltaY = pixelHeight;
        if (raysPerPixel > 1) {
            h->raysX = (uint32_t)(rootTerm - (pixelWidth - pixelHeight / 2 * pixelHeight));
            if (h->raysX == 0) h->raysX = 1;     /* the reference would divide by zero here */
            h->raysY = (uint32_t)(raysPerPixel / h->raysX);
            h->deltaX = pixelWidth / h->raysX;
            h->deltaY = pixelHeight / h->raysY;
        }
    }
    int rc = init_body(h);
    if (rc != HIP_TRACE_OK) {
        g_initError = h->error;
        release_all(h);
        delete h;
        return rc;
    }
    *out = h;
    return HIP_TRACE_OK;
}

template <typename T>
static size_t place(size_t& cursor, size_t count) {
    cursor = (cursor + 255) & ~(size_t)255;
    size_t at = cursor;
    cursor += count * sizeof(T);
    return at;
}

extern "C" uint32_t hip_trace_jitter_hash(uint32_t pixel, uint32_t treePosition, uint32_t light, uint32_t component) {
    return jitter_hash(pixel, treePosition, light, component);
}

extern "C" int hip_trace_check_hierarchy(const Scene* scene, const Octree* octree, uint32_t accel,
                                         HipTraceHierarchyReport* report, char* message, size_t capacity) {
    auto say = [&](const std::string& text) {
        if (message && capacity) { std::strncpy(message, text.c_str(), capacity - 1); message[capacity - 1] = 0; }
    };
    say("");
    if (!scene || !report) { say("hip_trace_check_hierarchy: NULL argument"); return HIP_TRACE_ERR_ARGUMENT; }
    if (accel != HIP_TRACE_ACCEL_CALLER_OCTREE && accel != HIP_TRACE_ACCEL_PRIVATE_BVH) {
        say("hip_trace_check_hierarchy: unknown accel");
        return HIP_TRACE_ERR_UNSUPPORTED;
    }
    FlatScene flat;
    std::string text;
    int rc;
    try {
        rc = rt_flatten(scene, octree, accel, flat, text);
        if (rc == 0) rt_check_hierarchy(scene, flat, accel, *report);
    } catch (const std::bad_alloc&) {
        say("hip_trace_check_hierarchy: out of host memory");
        return HIP_TRACE_ERR_MEMORY;
    }
    if (rc != 0) { say("hip_trace_check_hierarchy: " + text); return rc == -5 ? HIP_TRACE_ERR_TREE : HIP_TRACE_ERR_ARGUMENT; }
    return HIP_TRACE_OK;
}

/* LDS of one workgroup of trace_kernel: the fixed part, then the walk stacks (4 B per level per thread) and the
 * wavefronts' shared stacks (the kernel lays them out in this order). */
static size_t workgroup_lds_bytes(uint32_t treeLevels) {
    const size_t levels = treeLevels > 1u ? treeLevels - 1u : 1u;
    return RT_FIXED_LDS_BYTES + levels * RT_BLOCK * RT_STACK_BYTES_PER_LEVEL + (RT_BLOCK / 64u) * levels * sizeof(uint32_t);
}
#define RT_MAX_WORKGROUP_LDS 65536u   /* what a launch may ask for without opting in to more */

static int group_upload(HipTrace* h, const Scene* scene, const Octree* octree);

extern "C" int hip_trace_upload_scene(HipTrace* h, const Scene* scene, const Octree* octree) {
    if (!h) return HIP_TRACE_ERR_ARGUMENT;
    if (!h->members.empty()) {
        if (!scene || (!octree && h->opt.accel == HIP_TRACE_ACCEL_CALLER_OCTREE))
            return fail(h, HIP_TRACE_ERR_ARGUMENT, "hip_trace_upload_scene: NULL argument");
        DeviceGuard restoreDevice;
        return group_upload(h, scene, octree);
    }
    if (!scene || (!octree && h->opt.accel == HIP_TRACE_ACCEL_CALLER_OCTREE))
        return fail(h, HIP_TRACE_ERR_ARGUMENT, "hip_trace_upload_scene: NULL argument");
    FlatScene flat;
    std::string message;
    int rc;
    try {
        rc = rt_flatten(scene, octree, h->opt.accel, flat, message);
    } catch (const std::bad_alloc&) {
        return fail(h, HIP_TRACE_ERR_MEMORY, "hip_trace_upload_scene: out of host memory while flattening");
    }
    if (rc != 0) return fail(h, rc == -5 ? HIP_TRACE_ERR_TREE : HIP_TRACE_ERR_ARGUMENT, "hip_trace_upload_scene: " + message);
    if (workgroup_lds_bytes(flat.treeLevels) > RT_MAX_WORKGROUP_LDS)
        return fail(h, HIP_TRACE_ERR_TREE, "hip_trace_upload_scene: a hierarchy of " + std::to_string(flat.treeLevels) +
                                           " levels needs more LDS for its walk stacks than a workgroup may have");

    DeviceGuard restoreDevice;
    HIP_TRY(h, hipSetDevice(h->device));
    size_t cursor = 0;
    const size_t atNodes = place<QNode>(cursor, flat.qnodes.size());
    /* the same nodes with float boxes, for the walk a wavefront shares; RT_SHARED_WALK=0 leaves them out (tuning aid) */
    bool sharedWalk = true;
    if (const char* env = std::getenv("RT_SHARED_WALK")) sharedWalk = std::atoi(env) != 0;
    const size_t atWide = sharedWalk ? place<WideNode>(cursor, flat.nodes.size()) : 0;
    const size_t atIndex = place<uint32_t>(cursor, flat.leafIndex.size());
    /* filter records: per leaf entry for the library's own hierarchy (each triangle is in one leaf), per triangle for
     * the caller's octree (leaf_test<.., INDEXED>) */
    const bool recordsByTriangle = h->opt.accel == HIP_TRACE_ACCEL_CALLER_OCTREE;
    const size_t atLeafTri = place<LeafTriangle>(cursor, recordsByTriangle ? flat.trianglePlane.size() : flat.leafIndex.size());
    const size_t atTriPlane = place<Float4>(cursor, flat.trianglePlane.size());
    const size_t atTriCorner = place<Float4>(cursor, flat.triangleCorner.size());
    const size_t atTriMat = place<uint32_t>(cursor, flat.triangleMaterial.size());
    const size_t atSphere = place<Float4>(cursor, flat.sphere.size());
    const size_t atSphereMat = place<uint32_t>(cursor, flat.sphereMaterial.size());
    const size_t atPlane = place<Float4>(cursor, flat.plane.size());
    const size_t atPlaneMat = place<uint32_t>(cursor, flat.planeMaterial.size());
    const size_t atMaterial = place<Float4>(cursor, flat.material.size());
    const size_t atLight = place<Float4>(cursor, flat.light.size());
    const size_t total = (cursor + 255) & ~(size_t)255;

    /* which side of each of the first planes every light is clearly on (RT_SIDE_MARGIN), in the spare word of its record;
     * a plane whose normal is far from unit length, or a light far outside, claims none */
    for (uint32_t l = 0; l < flat.lightCount; l++) {
        const Float4 at = flat.light[2 * (size_t)l];
        uint32_t sides = 0;
        if (std::max(std::fabs(at.x), std::max(std::fabs(at.y), std::fabs(at.z))) <= RT_SIDE_REACH) {
            for (uint32_t i = 0; i < std::min<uint32_t>(flat.planeCount, RT_SIDE_PLANES); i++) {
                const Float4 pl = flat.plane[i];
                const double length = std::sqrt((double)pl.x * pl.x + (double)pl.y * pl.y + (double)pl.z * pl.z);
                if (!(length >= 0.5 && length <= 2.0) || !std::isfinite(pl.w)) continue;
                const double side = (double)pl.x * at.x + (double)pl.y * at.y + (double)pl.z * at.z + (double)pl.w;
                if (side > 1.01 * RT_SIDE_MARGIN) sides |= 1u << i;
                if (side < -1.01 * RT_SIDE_MARGIN) sides |= 0x10000u << i;
            }
        }
        std::memcpy(&flat.light[2 * (size_t)l + 1].w, &sides, sizeof sides);
    }

    void* block = nullptr;
    HIP_TRY(h, hipMalloc(&block, total));
    char* base = static_cast<char*>(block);
    struct Piece { size_t at; const void* src; size_t bytes; };
    const Piece pieces[] = {
        {atNodes, flat.qnodes.data(), flat.qnodes.size() * sizeof(QNode)},
        {atWide, flat.nodes.data(), sharedWalk ? flat.nodes.size() * sizeof(WideNode) : 0},
        {atIndex, flat.leafIndex.data(), flat.leafIndex.size() * sizeof(uint32_t)},
        {atTriPlane, flat.trianglePlane.data(), flat.trianglePlane.size() * sizeof(Float4)},
        {atTriCorner, flat.triangleCorner.data(), flat.triangleCorner.size() * sizeof(Float4)},
        {atTriMat, flat.triangleMaterial.data(), flat.triangleMaterial.size() * sizeof(uint32_t)},
        {atSphere, flat.sphere.data(), flat.sphere.size() * sizeof(Float4)},
        {atSphereMat, flat.sphereMaterial.data(), flat.sphereMaterial.size() * sizeof(uint32_t)},
        {atPlane, flat.plane.data(), flat.plane.size() * sizeof(Float4)},
        {atPlaneMat, flat.planeMaterial.data(), flat.planeMaterial.size() * sizeof(uint32_t)},
        {atMaterial, flat.material.data(), flat.material.size() * sizeof(Float4)},
        {atLight, flat.light.data(), flat.light.size() * sizeof(Float4)},
    };
    for (const Piece& piece : pieces) {
        hipError_t e = hipMemcpy(base + piece.at, piece.src, piece.bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(block);
            return fail(h, HIP_TRACE_ERR_DEVICE, std::string("hipMemcpy(scene): ") + hipGetErrorString(e));
        }
    }
    /* expand the per-leaf filter records on the device */
    {
        const uint64_t items = (uint64_t)flat.nodes.size() * 8u;
        if (recordsByTriangle) {
            const uint32_t triangles = (uint32_t)flat.trianglePlane.size();
            if (triangles)
                hipLaunchKernelGGL(expand_triangles_kernel, dim3((triangles + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, nullptr, triangles,
                                   reinterpret_cast<const Float4*>(base + atTriPlane), reinterpret_cast<const Float4*>(base + atTriCorner),
                                   reinterpret_cast<LeafTriangle*>(base + atLeafTri));
        } else
        hipLaunchKernelGGL(expand_leaves_kernel, dim3((unsigned)((items + RT_BLOCK - 1) / RT_BLOCK)), dim3(RT_BLOCK), 0, nullptr,
                           reinterpret_cast<const QNode*>(base + atNodes), (uint32_t)flat.qnodes.size(),
                           reinterpret_cast<const uint32_t*>(base + atIndex), reinterpret_cast<const Float4*>(base + atTriPlane),
                           reinterpret_cast<const Float4*>(base + atTriCorner), reinterpret_cast<LeafTriangle*>(base + atLeafTri));
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            (void)hipFree(block);
            return fail(h, HIP_TRACE_ERR_DEVICE, std::string("expanding leaf triangles: ") + hipGetErrorString(e));
        }
    }
    /* per-light direction grids for the shadow rays of the plain kernel (light_grid.hip) */
    LightGridBuild lightGrid;
    {
        bool wanted = h->opt.lightGrid == HIP_TRACE_LIGHT_GRID_AUTO &&
                      h->opt.jitterMode == HIP_TRACE_JITTER_ZERO && h->opt.shadingModel == HIP_TRACE_SHADING_CPU_MODEL &&
                      h->opt.raysPerPixel == 1;
        if (const char* env = std::getenv("RT_LIGHT_GRID")) wanted = wanted && std::atoi(env) != 0;
        /* a scene too large for the plain kernel's slack is always rendered by the general one, which does not use
         * grids; and an entry of a list keeps 24 bits for a primitive's number or its position in the leaf arrays */
        wanted = wanted && 1e-6f * flat.extent <= RT_DEFAULT_SLACK &&
                 (recordsByTriangle || flat.leafIndex.size() < GRID_REF_MAX_ENTRIES) &&
                 flat.trianglePlane.size() < GRID_REF_MAX_ENTRIES && flat.sphereCount < GRID_REF_MAX_ENTRIES;
        if (wanted) {
            LightGridInput in{};
            /* over the caller's octree a triangle sits in many leaves and the filter records are per triangle */
            in.nodes = recordsByTriangle ? nullptr : reinterpret_cast<const QNode*>(base + atNodes);
            in.sphereCount = flat.sphereCount;
            in.triangleCount = flat.triangleCount;
            in.nodeCount = (uint32_t)flat.qnodes.size();
            in.leafIndex = reinterpret_cast<const uint32_t*>(base + atIndex);
            in.entryCount = flat.sphereCount + flat.triangleCount;
            in.triangleCorner = reinterpret_cast<const Float4*>(base + atTriCorner);
            in.sphere = reinterpret_cast<const Float4*>(base + atSphere);
            in.light = reinterpret_cast<const Float4*>(base + atLight);
            in.lightCount = flat.lightCount;
            in.extent = flat.extent;
            /* the reference's sphere test can say "hit" beside the sphere (flatten.cpp, bounds_of): the same allowance here */
            in.sphereGrow = flat.sphereGrow > 0.0f ? flat.sphereGrow : 4e-7f * (RT_SPHERE_DISTANCE * flat.extent) * (RT_SPHERE_DISTANCE * flat.extent);
            in.sceneBytes = total;
            std::string gridError;
            if (rt_light_grid_build(in, lightGrid, gridError) != 0) {
                /* the grids are an optimisation: whatever went wrong building them, the scene renders by walking */
                if (std::getenv("RT_LIGHT_GRID_VERBOSE")) std::fprintf(stderr, "light grid: not built (%s)\n", gridError.c_str());
                (void)hipGetLastError();
                lightGrid = LightGridBuild{};
            }
        }
    }
    /* the previous scene stays valid until the new one is completely on the device */
    (void)hipStreamSynchronize(h->stream);
    release_scene(h);
    h->lightGrid = lightGrid;
    h->sceneBlock = block;
    h->sceneBytes = total;
    DeviceScene& d = h->dscene;
    d.nodes = reinterpret_cast<const QNode*>(base + atNodes);
    d.wideNodes = sharedWalk ? reinterpret_cast<const WideNode*>(base + atWide) : nullptr;
    d.leafIndex = reinterpret_cast<const uint32_t*>(base + atIndex);
    d.leafTriangle = reinterpret_cast<const LeafTriangle*>(base + atLeafTri);
    d.trianglePlane = reinterpret_cast<const Float4*>(base + atTriPlane);
    d.triangleCorner = reinterpret_cast<const Float4*>(base + atTriCorner);
    d.triangleMaterial = reinterpret_cast<const uint32_t*>(base + atTriMat);
    d.sphere = reinterpret_cast<const Float4*>(base + atSphere);
    d.sphereMaterial = reinterpret_cast<const uint32_t*>(base + atSphereMat);
    d.plane = reinterpret_cast<const Float4*>(base + atPlane);
    d.planeMaterial = reinterpret_cast<const uint32_t*>(base + atPlaneMat);
    d.material = reinterpret_cast<const Float4*>(base + atMaterial);
    d.light = reinterpret_cast<const Float4*>(base + atLight);
    d.planeCount = flat.planeCount;
    d.sphereCount = flat.sphereCount;
    d.triangleCount = flat.triangleCount;
    d.lightCount = flat.lightCount;
    d.materialCount = flat.materialCount;
    d.treeLevels = flat.treeLevels;
    d.filterMargin = std::max(RT_DEFAULT_SLACK, 1e-6f * flat.extent);
    d.boxPad = 0.0f;
    d.grid = h->lightGrid.view;
    d.grid.allLights = h->lightGrid.view.blocks != nullptr && h->lightGrid.enabledLights == flat.lightCount ? 1u : 0u;
    h->sceneExtent = flat.extent;
    h->info.wideNodeCount = (uint32_t)flat.nodes.size();
    h->info.leafCount = flat.leafCount;
    h->info.leafIndexCount = (uint32_t)flat.leafIndex.size();
    h->info.treeLevels = flat.treeLevels;
    h->info.deviceBytes = total + h->lightGrid.bytes;
    h->info.lightGridBytes = h->lightGrid.bytes;
    h->info.lightGridReferences = h->lightGrid.referenceCount;
    h->info.lightGridLights = h->lightGrid.enabledLights;
    h->info.lightGridPeakMiB = (uint32_t)((h->lightGrid.peakBytes + (1u << 20) - 1u) >> 20);
    /* does any primitive carry a material that refracts?  (unused materials do not count: the synthetic mesh scenes keep
     * the default scene's glass in their material table without using it) */
    h->sceneRefracts = false;
    {
        std::vector<uint8_t> refracts(scene->materialCount, 0);
        bool any = false;
        for (uint32_t m = 0; m < scene->materialCount; m++)
            if (scene->materials[m].refractionIndex > 0) { refracts[m] = 1; any = true; }
        if (any) {
            for (uint32_t i = 0; i < scene->planeCount && !h->sceneRefracts; i++) h->sceneRefracts = refracts[scene->planes[i].materialIndex] != 0;
            for (uint32_t i = 0; i < scene->sphereCount && !h->sceneRefracts; i++) h->sceneRefracts = refracts[scene->spheres[i].materialIndex] != 0;
            for (uint32_t i = 0; i < scene->triangleCount && !h->sceneRefracts; i++) h->sceneRefracts = refracts[scene->triangles[i].materialIndex] != 0;
        }
    }
    h->haveScene = true;
    h->sceneSerial++;
    h->haveCosts = false;
    return HIP_TRACE_OK;
}

static int check_camera(HipTrace* h, const Camera* camera) {
    if (!camera) return fail(h, HIP_TRACE_ERR_ARGUMENT, "camera is NULL");
    if (!h->haveScene) return fail(h, HIP_TRACE_ERR_NO_SCENE, "no scene uploaded");
    if (camera->width != h->opt.width || camera->height != h->opt.height)
        return fail(h, HIP_TRACE_ERR_DIMENSION, "camera width/height differ from hip_trace_init");
    return HIP_TRACE_OK;
}

static V3 v3_of(const Vec3& v) { V3 r; r.x = v.x; r.y = v.y; r.z = v.z; return r; }

/* Device memory a frame uses beyond its inputs and outputs. */
struct FrameScratch {
    unsigned long long* counters;   /* RT_COUNTERS */
    uint32_t* tileCost;             /* localTiles, or NULL: no longest-first launch order */
    uint32_t* tileOrder;            /* localTiles */
    bool haveCosts;                 /* tileCost holds the previous frame of this handle */
};

static FrameScratch own_scratch(HipTrace* h) {
    FrameScratch f;
    f.counters = h->dCounters;
    f.tileCost = h->orderTiles ? h->dTileCost : nullptr;
    f.tileOrder = h->dTileOrder;
    f.haveCosts = h->haveCosts;
    h->haveCosts = h->orderTiles;
    return f;
}

/* Enqueues counters reset, kernel and counter read-back on the handle's stream. */
static int enqueue_frame(HipTrace* h, const Camera* camera, uint32_t* dOut, uint8_t* dMask, uint32_t* dId, float* dT,
                         const FrameScratch& scratchOfFrame) {
    FrameParams p;
    std::memset(&p, 0, sizeof p);
    p.scene = h->dscene;
    p.camPosition = v3_of(camera->position);
    p.camX = v3_of(camera->x);
    p.camY = v3_of(camera->y);
    p.targetCenter = v3_of(camera->renderTargetCenter);
    p.targetWidth = camera->renderTargetWidth;
    p.targetHeight = camera->renderTargetHeight;
    p.focalLength = camera->focalLength;
    p.apertureSize = camera->apertureSize;
    p.width = h->opt.width; p.height = h->opt.height; p.maxDepth = h->opt.maxDepth;
    p.jitterMode = h->opt.jitterMode;
    p.shadingModel = h->opt.shadingModel;
    p.raysX = h->raysX; p.raysY = h->raysY; p.deltaX = h->deltaX; p.deltaY = h->deltaY; p.contribution = h->contribution;
    p.tilesX = h->tilesX; p.tileCount = h->tileCount;
    p.rank = h->opt.rank; p.worldSize = h->opt.worldSize; p.rotate = h->rotate; p.localTiles = h->localTiles;
    p.xcdRun = h->xcdRun;
    p.out = dOut;
    p.hitMask = dMask;
    p.primitiveId = dId;
    p.hitDistance = dT;
    p.counters = scratchOfFrame.counters;
    p.tileOrder = nullptr;
    p.tileCost = scratchOfFrame.tileCost;
    if (scratchOfFrame.tileCost && scratchOfFrame.haveCosts) {
        hipLaunchKernelGGL(order_tiles_kernel, dim3(1), dim3(1024), 0, h->stream, scratchOfFrame.tileCost, scratchOfFrame.tileOrder, h->localTiles);
        HIP_TRY(h, hipGetLastError());
        p.tileOrder = scratchOfFrame.tileOrder;
    }

    const size_t ldsBytes = workgroup_lds_bytes(h->dscene.treeLevels);
    HIP_TRY(h, hipMemsetAsync(scratchOfFrame.counters, 0, RT_COUNTERS * sizeof(unsigned long long), h->stream));
    HIP_TRY(h, hipEventRecord(h->evKernelStart, h->stream));
    /* The slack of the conservative tests covers rounding that grows with the size of the numbers involved: the scene's
     * own extent (set at upload) and, for primary rays, how far from the origin the camera stands.  A camera beyond what
     * the upload-time slack covers sends the frame to the general kernel with a slack and a box padding that follow it. */
    const float reach = std::max(std::fabs(camera->position.x), std::max(std::fabs(camera->position.y), std::fabs(camera->position.z)));
    p.scene.boxPad = 0.0f;
    if (1e-6f * reach > p.scene.filterMargin) {
        p.scene.filterMargin = 1e-6f * reach;
        p.scene.boxPad = 2e-6f * reach;
    }
    /* The reference's sphere test says "hit" for rays that pass a sphere at up to sqrt(r^2 + ~4e-7 D^2) from its centre,
     * D the distance of the ray's origin (cancellation in its discriminant).  The sphere boxes cover that for origins
     * within RT_SPHERE_REACH scene sizes; a camera farther out widens every box by that much, whatever it costs. */
    if (h->dscene.sphereCount > 0 && reach > RT_SPHERE_REACH * h->sceneExtent) {
        p.scene.filterMargin = std::max(p.scene.filterMargin, 1e-6f * reach);
        p.scene.boxPad = std::max(p.scene.boxPad, 1.2e-3f * reach);
    }
    const bool general = p.shadingModel != HIP_TRACE_SHADING_CPU_MODEL || p.raysX * p.raysY != 1 ||
                         p.scene.filterMargin != RT_DEFAULT_SLACK;
    const int kernelMode = general ? MODE_GENERAL : p.jitterMode == HIP_TRACE_JITTER_HASHED ? MODE_JITTER : MODE_PLAIN;
    const bool manyLeaves = h->opt.accel == HIP_TRACE_ACCEL_CALLER_OCTREE;
    const dim3 grid(h->localTiles), block(RT_BLOCK);
    const bool glass = h->sceneRefracts;
#define RT_LAUNCH_GLASS(MODE, LEAVES, GLASS, GRID) hipLaunchKernelGGL((trace_kernel<MODE, LEAVES, GLASS, GRID>), grid, block, ldsBytes, h->stream, p)
#define RT_LAUNCH(MODE, LEAVES, GRID) do { if (glass) RT_LAUNCH_GLASS(MODE, LEAVES, true, GRID); else RT_LAUNCH_GLASS(MODE, LEAVES, false, GRID); } while (0)
    if (kernelMode == MODE_GENERAL) { if (manyLeaves) RT_LAUNCH(MODE_GENERAL, true, false); else RT_LAUNCH(MODE_GENERAL, false, false); }
    else if (kernelMode == MODE_JITTER) { if (manyLeaves) RT_LAUNCH(MODE_JITTER, true, false); else RT_LAUNCH(MODE_JITTER, false, false); }
    else if (p.scene.grid.blocks != nullptr) { if (manyLeaves) RT_LAUNCH(MODE_PLAIN, true, true); else RT_LAUNCH(MODE_PLAIN, false, true); }
    else { if (manyLeaves) RT_LAUNCH(MODE_PLAIN, true, false); else RT_LAUNCH(MODE_PLAIN, false, false); }
#undef RT_LAUNCH
#undef RT_LAUNCH_GLASS
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventRecord(h->evKernelStop, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->hCounters, scratchOfFrame.counters, RT_COUNTERS * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    h->frameRecorded = true;
    return HIP_TRACE_OK;
}

/* ------------------------------------------------------------------ several devices behind one handle */

static void destroy_members(HipTrace* h) {
    for (HipTrace* member : h->members) {
        if (!member) continue;
        (void)hipSetDevice(member->device);
        (void)hipStreamSynchronize(member->stream);
        release_all(member);
        if (member->ownStream) (void)hipStreamDestroy(member->ownStream);
        delete member;
    }
    h->members.clear();
    (void)hipSetDevice(h->device);
    for (hipEvent_t e : h->memberDone) if (e) (void)hipEventDestroy(e);
    h->memberDone.clear();
    if (h->dGathered) (void)hipFree(h->dGathered);
    if (h->dGatheredAux) (void)hipFree(h->dGatheredAux);
    if (h->gatherDone) (void)hipEventDestroy(h->gatherDone);
    h->dGathered = nullptr;
    h->dGatheredAux = nullptr;
    h->gatherDone = nullptr;
    h->gatherRecorded = false;
}

extern "C" int hip_trace_init_multi(HipTrace** out, const HipTraceOptions* options, const int32_t* devices, uint32_t deviceCount) {
    if (out) *out = nullptr;
    if (!out || !options || !devices) return fail(nullptr, HIP_TRACE_ERR_ARGUMENT, "hip_trace_init_multi: NULL argument");
    if (deviceCount == 0 || deviceCount > 64) return fail(nullptr, HIP_TRACE_ERR_ARGUMENT, "hip_trace_init_multi: deviceCount must be in [1, 64]");
    if (options->worldSize > 1 || options->rank != 0)
        return fail(nullptr, HIP_TRACE_ERR_ARGUMENT, "hip_trace_init_multi: rank/worldSize describe one device of a caller-managed set; leave them 0/1 here");
    DeviceGuard restoreDevice;
    /* the handle the caller sees: the whole frame, on the first device of the list */
    HipTraceOptions whole = *options;
    whole.worldSize = 1;
    whole.rank = 0;
    whole.device = devices[0];
    HipTrace* h = nullptr;
    int rc = hip_trace_init(&h, &whole);
    if (rc != HIP_TRACE_OK) return rc;
    if (deviceCount == 1) { *out = h; return HIP_TRACE_OK; }
    auto bail = [&](int code, const std::string& message) {
        destroy_members(h);
        hip_trace_destroy(h);
        return fail(nullptr, code, message);
    };
    size_t memberBytes = 0;
    for (uint32_t r = 0; r < deviceCount; r++) {
        HipTraceOptions part = *options;
        part.worldSize = deviceCount;
        part.rank = r;
        part.device = devices[r];
        HipTrace* member = nullptr;
        rc = hip_trace_init(&member, &part);
        if (rc != HIP_TRACE_OK) return bail(rc, "hip_trace_init_multi: device entry " + std::to_string(r) + ": " + g_initError);
        h->members.push_back(member);
        memberBytes = hip_trace_output_bytes(member);
        hipError_t e = hipSetDevice(member->device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&member->ownStream, hipStreamNonBlocking);
        if (e != hipSuccess) return bail(HIP_TRACE_ERR_DEVICE, std::string("hip_trace_init_multi: stream: ") + hipGetErrorString(e));
        member->stream = member->ownStream;
        hipEvent_t done = nullptr;
        e = hipEventCreateWithFlags(&done, hipEventDisableTiming);
        if (e != hipSuccess) return bail(HIP_TRACE_ERR_DEVICE, std::string("hip_trace_init_multi: event: ") + hipGetErrorString(e));
        h->memberDone.push_back(done);
        /* direct peer copies over the link between the two devices where the platform allows them
         * (refusal is not an error: hipMemcpyPeerAsync then goes through host memory) */
        if (member->device != h->device) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, h->device, member->device) == hipSuccess && can) {
                (void)hipSetDevice(h->device);
                e = hipDeviceEnablePeerAccess(member->device, 0);
                if (e != hipSuccess) (void)hipGetLastError();   /* already enabled, or not possible */
            }
        }
    }
    hipError_t e = hipSetDevice(h->device);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->gatherDone, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(&h->dGathered, memberBytes * deviceCount);
    if (e != hipSuccess) return bail(e == hipErrorOutOfMemory ? HIP_TRACE_ERR_MEMORY : HIP_TRACE_ERR_DEVICE,
                                     std::string("hip_trace_init_multi: gather buffer: ") + hipGetErrorString(e));
    *out = h;
    return HIP_TRACE_OK;
}

static int group_upload(HipTrace* h, const Scene* scene, const Octree* octree) {
    for (size_t r = 0; r < h->members.size(); r++) {
        int rc = hip_trace_upload_scene(h->members[r], scene, octree);
        if (rc != HIP_TRACE_OK) {
            /* members before r already hold the new scene: the handle cannot render a consistent frame any more */
            if (r > 0) for (HipTrace* member : h->members) { DeviceGuard g; (void)hipSetDevice(member->device); release_scene(member); }
            h->haveScene = r > 0 ? false : h->haveScene;
            return fail(h, rc, "device entry " + std::to_string(r) + ": " + h->members[r]->error);
        }
    }
    h->info = h->members[0]->info;
    h->info.deviceBytes = 0;
    for (HipTrace* member : h->members) h->info.deviceBytes += member->info.deviceBytes;
    h->haveScene = true;
    return HIP_TRACE_OK;
}

/* Copies one tile-major buffer of every member (`bytes` each, picked by `of`) into consecutive slots of `gathered` on the
 * first device, on the handle's stream: peer copies, an all-to-one gather over as many point-to-point links as there are peers. */
template <typename Pick>
static int group_gather(HipTrace* h, void* gathered, size_t bytes, Pick of) {
    for (size_t r = 0; r < h->members.size(); r++) {
        HipTrace* member = h->members[r];
        char* slot = static_cast<char*>(gathered) + r * bytes;
        if (member->device == h->device) HIP_TRY(h, hipMemcpyAsync(slot, of(member), bytes, hipMemcpyDeviceToDevice, h->stream));
        else HIP_TRY(h, hipMemcpyPeerAsync(slot, h->device, of(member), member->device, bytes, h->stream));
    }
    return HIP_TRACE_OK;
}

/* One frame over all members: every member traces its tiles on its own device and stream, each finished part is
 * copied to its place in the gather buffer on the first device (no collective needed), and the frame is re-assembled
 * there into dFrame.  The side outputs asked for (row-major buffers on the first device, or NULL) travel the same way. */
static int group_enqueue(HipTrace* h, const Camera* camera, uint32_t* dFrame, uint8_t* dMask = nullptr, uint32_t* dId = nullptr,
                         float* dT = nullptr) {
    const size_t memberBytes = hip_trace_output_bytes(h->members[0]);
    const size_t memberPixels = memberBytes / 4;
    for (size_t r = 0; r < h->members.size(); r++) {
        HipTrace* member = h->members[r];
        HIP_TRY(h, hipSetDevice(member->device));
        if (dMask && !member->dHitMask) HIP_TRY(h, hipMalloc(&member->dHitMask, memberPixels));
        if (dId && !member->dPrimitiveId) HIP_TRY(h, hipMalloc(&member->dPrimitiveId, memberPixels * 4));
        if (dT && !member->dHitDistance) HIP_TRY(h, hipMalloc(&member->dHitDistance, memberPixels * 4));
        /* the member's buffers are read by the previous frame's copies on the handle's stream: calls return without
         * waiting, so the next frame's kernel must not start before those copies are through */
        if (h->gatherRecorded) HIP_TRY(h, hipStreamWaitEvent(member->stream, h->gatherDone, 0));
        int rc = enqueue_frame(member, camera, member->dOut, dMask ? member->dHitMask : nullptr, dId ? member->dPrimitiveId : nullptr,
                               dT ? member->dHitDistance : nullptr, own_scratch(member));
        if (rc != HIP_TRACE_OK) return fail(h, rc, "device entry " + std::to_string(r) + ": " + member->error);
        HIP_TRY(h, hipEventRecord(h->memberDone[r], member->stream));
    }
    HIP_TRY(h, hipSetDevice(h->device));
    for (size_t r = 0; r < h->members.size(); r++) HIP_TRY(h, hipStreamWaitEvent(h->stream, h->memberDone[r], 0));
    HipTrace* first = h->members[0];
    int rc = group_gather(h, h->dGathered, memberBytes, [](HipTrace* m) { return static_cast<const void*>(m->dOut); });
    if (rc != HIP_TRACE_OK) return rc;
#define RT_UNTILE(T, from, to)                                                                                                          \
    do {                                                                                                                                \
        hipLaunchKernelGGL(untile_kernel<T>, dim3(first->tileCount), dim3(RT_BLOCK), 0, h->stream, static_cast<const T*>(from), (to),   \
                           h->opt.width, h->opt.height, first->tilesX, first->tileCount, first->opt.worldSize, first->rotate,          \
                           first->localTiles);                                                                                          \
        HIP_TRY(h, hipGetLastError());                                                                                                  \
    } while (0)
    RT_UNTILE(uint32_t, h->dGathered, dFrame);
    if (dMask || dId || dT) {
        /* one staging buffer serves the three side outputs in turn: the stream keeps them in order */
        if (!h->dGatheredAux) HIP_TRY(h, hipMalloc(&h->dGatheredAux, memberBytes * h->members.size()));
        if (dMask) {
            rc = group_gather(h, h->dGatheredAux, memberPixels, [](HipTrace* m) { return static_cast<const void*>(m->dHitMask); });
            if (rc != HIP_TRACE_OK) return rc;
            RT_UNTILE(uint8_t, h->dGatheredAux, dMask);
        }
        if (dId) {
            rc = group_gather(h, h->dGatheredAux, memberBytes, [](HipTrace* m) { return static_cast<const void*>(m->dPrimitiveId); });
            if (rc != HIP_TRACE_OK) return rc;
            RT_UNTILE(uint32_t, h->dGatheredAux, dId);
        }
        if (dT) {
            rc = group_gather(h, h->dGatheredAux, memberBytes, [](HipTrace* m) { return static_cast<const void*>(m->dHitDistance); });
            if (rc != HIP_TRACE_OK) return rc;
            RT_UNTILE(float, h->dGatheredAux, dT);
        }
    }
#undef RT_UNTILE
    HIP_TRY(h, hipEventRecord(h->gatherDone, h->stream));
    h->gatherRecorded = true;
    h->frameRecorded = true;
    return HIP_TRACE_OK;
}

/* Sums the members' ray counters into h->hCounters and returns the longest kernel time; waits for every member. */
static int group_collect(HipTrace* h, float* kernelMs) {
    for (int k = 0; k < RT_COUNTERS; k++) h->hCounters[k] = 0;
    float longest = 0.0f;
    for (HipTrace* member : h->members) {
        HIP_TRY(h, hipSetDevice(member->device));
        HIP_TRY(h, hipStreamSynchronize(member->stream));
        for (int k = 0; k < RT_COUNTERS; k++) h->hCounters[k] += member->hCounters[k];
        float ms = 0.0f;
        HIP_TRY(h, hipEventElapsedTime(&ms, member->evKernelStart, member->evKernelStop));
        longest = std::max(longest, ms);
    }
    HIP_TRY(h, hipSetDevice(h->device));
    if (kernelMs) *kernelMs = longest;
    return HIP_TRACE_OK;
}

extern "C" int hip_trace_frame(HipTrace* h, const Camera* camera, uint32_t* rgba8_out, HipTraceAux* aux) {
    if (!h) return HIP_TRACE_ERR_ARGUMENT;
    int rc = check_camera(h, camera);
    if (rc != HIP_TRACE_OK) return rc;
    if (h->opt.worldSize != 1) return fail(h, HIP_TRACE_ERR_UNSUPPORTED, "hip_trace_frame needs worldSize == 1; use hip_trace_frame_device");
    DeviceGuard restoreDevice;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t pixels = (size_t)h->opt.width * h->opt.height;
    const bool wantMask = aux && aux->hitMask, wantId = aux && aux->primitiveId, wantT = aux && aux->hitDistance;
    if (wantMask && !h->dHitMask) HIP_TRY(h, hipMalloc(&h->dHitMask, pixels));
    if (wantId && !h->dPrimitiveId) HIP_TRY(h, hipMalloc(&h->dPrimitiveId, pixels * 4));
    if (wantT && !h->dHitDistance) HIP_TRY(h, hipMalloc(&h->dHitDistance, pixels * 4));
    h->ownFrame = true;
    if (!h->members.empty()) {
        HIP_TRY(h, hipEventRecord(h->evStart, h->stream));
        rc = group_enqueue(h, camera, h->dOut, wantMask ? h->dHitMask : nullptr, wantId ? h->dPrimitiveId : nullptr,
                           wantT ? h->dHitDistance : nullptr);
        if (rc != HIP_TRACE_OK) return rc;
        if (rgba8_out) HIP_TRY(h, hipMemcpyAsync(rgba8_out, h->dOut, pixels * 4, hipMemcpyDeviceToHost, h->stream));
        if (wantMask) HIP_TRY(h, hipMemcpyAsync(aux->hitMask, h->dHitMask, pixels, hipMemcpyDeviceToHost, h->stream));
        if (wantId) HIP_TRY(h, hipMemcpyAsync(aux->primitiveId, h->dPrimitiveId, pixels * 4, hipMemcpyDeviceToHost, h->stream));
        if (wantT) HIP_TRY(h, hipMemcpyAsync(aux->hitDistance, h->dHitDistance, pixels * 4, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipEventRecord(h->evStop, h->stream));
        float kernelMs = 0.0f;
        rc = group_collect(h, &kernelMs);
        if (rc != HIP_TRACE_OK) return rc;
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (aux) {
            aux->rays.primary = h->hCounters[0];
            aux->rays.shadow = h->hCounters[1];
            aux->rays.reflection = h->hCounters[2];
            aux->rays.refraction = h->hCounters[3];
            aux->kernelMs = kernelMs;
            HIP_TRY(h, hipEventElapsedTime(&aux->totalMs, h->evStart, h->evStop));
        }
        return HIP_TRACE_OK;
    }

    HIP_TRY(h, hipEventRecord(h->evStart, h->stream));
    rc = enqueue_frame(h, camera, h->dOut, wantMask ? h->dHitMask : nullptr, wantId ? h->dPrimitiveId : nullptr,
                       wantT ? h->dHitDistance : nullptr, own_scratch(h));
    if (rc != HIP_TRACE_OK) return rc;
    if (rgba8_out) HIP_TRY(h, hipMemcpyAsync(rgba8_out, h->dOut, pixels * 4, hipMemcpyDeviceToHost, h->stream));
    if (wantMask) HIP_TRY(h, hipMemcpyAsync(aux->hitMask, h->dHitMask, pixels, hipMemcpyDeviceToHost, h->stream));
    if (wantId) HIP_TRY(h, hipMemcpyAsync(aux->primitiveId, h->dPrimitiveId, pixels * 4, hipMemcpyDeviceToHost, h->stream));
    if (wantT) HIP_TRY(h, hipMemcpyAsync(aux->hitDistance, h->dHitDistance, pixels * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipEventRecord(h->evStop, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (aux) {
        aux->rays.primary = h->hCounters[0];
        aux->rays.shadow = h->hCounters[1];
        aux->rays.reflection = h->hCounters[2];
        aux->rays.refraction = h->hCounters[3];
        HIP_TRY(h, hipEventElapsedTime(&aux->kernelMs, h->evKernelStart, h->evKernelStop));
        HIP_TRY(h, hipEventElapsedTime(&aux->totalMs, h->evStart, h->evStop));
    }
    return HIP_TRACE_OK;
}

extern "C" int hip_trace_frame_device(HipTrace* h, const Camera* camera, void* d_out) {
    if (!h) return HIP_TRACE_ERR_ARGUMENT;
    int rc = check_camera(h, camera);
    if (rc != HIP_TRACE_OK) return rc;
    DeviceGuard restoreDevice;
    HIP_TRY(h, hipSetDevice(h->device));
    h->ownFrame = d_out == nullptr;
    if (!h->members.empty()) return group_enqueue(h, camera, d_out ? static_cast<uint32_t*>(d_out) : h->dOut);
    return enqueue_frame(h, camera, d_out ? static_cast<uint32_t*>(d_out) : h->dOut, nullptr, nullptr, nullptr, own_scratch(h));
}

extern "C" int hip_trace_last_frame(HipTrace* h, void** d_rgba8) {
    if (d_rgba8) *d_rgba8 = nullptr;
    if (!h || !d_rgba8) return h ? fail(h, HIP_TRACE_ERR_ARGUMENT, "hip_trace_last_frame: NULL argument") : HIP_TRACE_ERR_ARGUMENT;
    if (!h->frameRecorded || !h->ownFrame)
        return fail(h, HIP_TRACE_ERR_NO_SCENE, "hip_trace_last_frame: the most recent frame was not rendered into the handle's own buffer");
    *d_rgba8 = h->dOut;
    return HIP_TRACE_OK;
}

extern "C" int hip_trace_frame_static(HipTrace* h, const Camera* camera, void* d_out) {
    if (!h) return HIP_TRACE_ERR_ARGUMENT;
    int rc = check_camera(h, camera);
    if (rc != HIP_TRACE_OK) return rc;
    if (!d_out) return fail(h, HIP_TRACE_ERR_ARGUMENT, "hip_trace_frame_static: d_out is NULL");
    if (!h->members.empty()) return fail(h, HIP_TRACE_ERR_UNSUPPORTED, "hip_trace_frame_static: needs a single-device handle");
    if (h->localTiles > RT_STATIC_MAX_TILES)
        return fail(h, HIP_TRACE_ERR_UNSUPPORTED, "hip_trace_frame_static: " + std::to_string(h->localTiles) + " tiles of 16x16 pixels on this handle, the static scratch holds " +
                                                  std::to_string(RT_STATIC_MAX_TILES) + "; use hip_trace_frame_device");
    if (!h->staticScratch) return fail(h, HIP_TRACE_ERR_DEVICE, "hip_trace_frame_static: the static scratch was not resolved at hip_trace_init");
    DeviceGuard restoreDevice;
    HIP_TRY(h, hipSetDevice(h->device));
    StaticScratch& entry = *h->staticScratch;
    h->ownFrame = false;
    FrameScratch f;
    f.counters = reinterpret_cast<unsigned long long*>(entry.base);
    f.tileCost = h->orderTiles ? reinterpret_cast<uint32_t*>(entry.base + RT_STATIC_HEADER) : nullptr;
    f.tileOrder = reinterpret_cast<uint32_t*>(entry.base + RT_STATIC_HEADER + RT_STATIC_MAX_TILES * sizeof(uint32_t));
    {
        /* the cycle counts in the array are only this handle's if its frame was the last one through it */
        std::lock_guard<std::mutex> lock(g_staticMutex);
        f.haveCosts = entry.lastUser == h && h->staticSceneSerial == h->sceneSerial;
        entry.lastUser = f.tileCost ? h : nullptr;
        h->staticSceneSerial = h->sceneSerial;
    }
    return enqueue_frame(h, camera, static_cast<uint32_t*>(d_out), nullptr, nullptr, nullptr, f);
}

extern "C" size_t hip_trace_output_bytes(const HipTrace* h) {
    if (!h) return 0;
    if (h->opt.worldSize == 1) return (size_t)h->opt.width * h->opt.height * 4;
    return (size_t)h->localTiles * HIP_TRACE_TILE * HIP_TRACE_TILE * 4;
}

extern "C" int hip_trace_untile(HipTrace* h, const void* d_gathered, void* d_rgba8) {
    if (!h) return HIP_TRACE_ERR_ARGUMENT;
    if (!d_gathered || !d_rgba8) return fail(h, HIP_TRACE_ERR_ARGUMENT, "hip_trace_untile: NULL argument");
    if (h->opt.worldSize == 1) return fail(h, HIP_TRACE_ERR_UNSUPPORTED, "hip_trace_untile: worldSize is 1, the frame is already row-major");
    DeviceGuard restoreDevice;
    HIP_TRY(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(untile_kernel<uint32_t>, dim3(h->tileCount), dim3(RT_BLOCK), 0, h->stream,
                       static_cast<const uint32_t*>(d_gathered), static_cast<uint32_t*>(d_rgba8), h->opt.width,
                       h->opt.height, h->tilesX, h->tileCount, h->opt.worldSize, h->rotate, h->localTiles);
    HIP_TRY(h, hipGetLastError());
    return HIP_TRACE_OK;
}

extern "C" int hip_trace_last_frame_stats(HipTrace* h, HipTraceRayCounts* rays, float* kernelMs) {
    if (!h) return HIP_TRACE_ERR_ARGUMENT;
    if (!h->frameRecorded) return fail(h, HIP_TRACE_ERR_NO_SCENE, "hip_trace_last_frame_stats: no frame rendered yet");
    DeviceGuard restoreDevice;
    HIP_TRY(h, hipSetDevice(h->device));
    if (!h->members.empty()) {
        float longest = 0.0f;
        int rc = group_collect(h, &longest);
        if (rc != HIP_TRACE_OK) return rc;
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (rays) {
            rays->primary = h->hCounters[0]; rays->shadow = h->hCounters[1];
            rays->reflection = h->hCounters[2]; rays->refraction = h->hCounters[3];
        }
        if (kernelMs) *kernelMs = longest;
        return HIP_TRACE_OK;
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (rays) {
        rays->primary = h->hCounters[0];
        rays->shadow = h->hCounters[1];
        rays->reflection = h->hCounters[2];
        rays->refraction = h->hCounters[3];
    }
    if (kernelMs) HIP_TRY(h, hipEventElapsedTime(kernelMs, h->evKernelStart, h->evKernelStop));
    return HIP_TRACE_OK;
}

extern "C" int hip_trace_walk_stats(HipTrace* h, uint64_t* out, uint32_t capacity) {
    if (!h || !out) return HIP_TRACE_ERR_ARGUMENT;
    if (!h->frameRecorded) return fail(h, HIP_TRACE_ERR_NO_SCENE, "hip_trace_walk_stats: no frame rendered yet");
    DeviceGuard restoreDevice;
    HIP_TRY(h, hipSetDevice(h->device));
    if (!h->members.empty()) { int rc = group_collect(h, nullptr); if (rc != HIP_TRACE_OK) return rc; }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    uint32_t n = 0;
    for (uint32_t k = 4; k < RT_COUNTERS && n < capacity; k++) out[n++] = h->hCounters[k];
    return (int)n;
}

extern "C" int hip_trace_set_stream(HipTrace* h, void* hipStream) {
    if (!h) return HIP_TRACE_ERR_ARGUMENT;
    DeviceGuard restoreDevice;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->stream = static_cast<hipStream_t>(hipStream);
    return HIP_TRACE_OK;
}

extern "C" int hip_trace_scene_info(const HipTrace* h, HipTraceSceneInfo* info) {
    if (!h || !info) return HIP_TRACE_ERR_ARGUMENT;
    *info = h->info;
    return h->haveScene ? HIP_TRACE_OK : HIP_TRACE_ERR_NO_SCENE;
}

extern "C" void hip_trace_destroy(HipTrace* h) {
    if (!h) return;
    DeviceGuard restoreDevice;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->staticScratch) {
        std::lock_guard<std::mutex> lock(g_staticMutex);
        if (h->staticScratch->lastUser == h) h->staticScratch->lastUser = nullptr;
    }
    destroy_members(h);
    release_all(h);
    delete h;
}

extern "C" const char* hip_trace_last_error(const HipTrace* h) {
    return h ? h->error.c_str() : g_initError.c_str();
}
