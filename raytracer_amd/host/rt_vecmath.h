/*
 * rt_vecmath.h -- fp32 vector helpers for the host modules.
 *
 * Every helper evaluates in the operand order of the reference's vector
 * library (reference: src/utils/vec3.c:4-106, src/utils/math.c:3-13) so that
 * host results are bit-identical to it.  Translation units that include this
 * header must be compiled with -ffp-contract=off.
 */
#ifndef RT_VECMATH_H
#define RT_VECMATH_H

#include <math.h>
#include "rt_types.h"

#define RT_PI ((float)3.14159265358979323846)

static inline Vec3 v3(float x, float y, float z) { Vec3 v; v.x = x; v.y = y; v.z = z; return v; }
static inline Vec3 v3_add(Vec3 a, Vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline Vec3 v3_sub(Vec3 a, Vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline Vec3 v3_scale(Vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline float v3_dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline Vec3 v3_cross(Vec3 a, Vec3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float v3_length(Vec3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
/* A zero vector stays zero (reference: src/utils/vec3.c:56-65). */
static inline Vec3 v3_unit(Vec3 a) {
    float len = v3_length(a);
    if (len != 0) { a.x /= len; a.y /= len; a.z /= len; }
    return a;
}
static inline float rt_deg2rad(float deg) { return deg * (RT_PI / 180.f); }

#endif
