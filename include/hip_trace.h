/*
 * hip_trace.h -- C ABI of the MI355X (gfx950) per-pixel trace path.
 *
 * This library takes the place of the reference renderer's OpenCL host layer
 * (reference: src/gpu.h:60-66, src/gpu.c:23-58) for the trace path only: it
 * owns the device context, keeps a private device copy of the scene, and
 * renders one RGBA8 frame per call.  Presentation (window, GL texture, quad)
 * is not part of it.
 *
 * All entry points are plain C, take plain pointers and sizes, and are
 * callable from C11.  The scene model is the reference's own host records
 * (rt_types.h; include the reference headers first when building inside the
 * reference tree).  Every function returns HIP_TRACE_OK (0) or a negative
 * HipTraceStatus; hip_trace_last_error() gives the message.  Nothing here
 * aborts or exits, and a failed call leaves no allocation behind.
 *
 * Threading: one host thread per handle; a handle is not re-entrant.
 * Ownership: the library owns all device memory; the caller owns every host
 * pointer and the library never keeps one past the call that received it.
 */
#ifndef HIP_TRACE_H
#define HIP_TRACE_H

#include <stdint.h>
#include <stddef.h>
#include "rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct HipTrace HipTrace;

typedef enum {
    HIP_TRACE_OK            =  0,
    HIP_TRACE_ERR_ARGUMENT  = -1,  /* NULL / out-of-range argument                     */
    HIP_TRACE_ERR_DEVICE    = -2,  /* a HIP runtime call failed                        */
    HIP_TRACE_ERR_MEMORY    = -3,  /* host or device allocation failed                 */
    HIP_TRACE_ERR_NO_SCENE  = -4,  /* frame requested before a successful upload       */
    HIP_TRACE_ERR_TREE      = -5,  /* octree malformed or deeper than the device stack */
    HIP_TRACE_ERR_DIMENSION = -6,  /* camera width/height differ from init             */
    HIP_TRACE_ERR_UNSUPPORTED = -7 /* option value not implemented                     */
} HipTraceStatus;

/* Primitive id reported for a pixel whose primary ray meets no primitive. */
#define HIP_TRACE_NO_PRIMITIVE 0xFFFFFFFFu

/* Deepest octree (root = level 0) hip_trace_upload_scene accepts. */
#define HIP_TRACE_MAX_TREE_LEVELS 32u
/* Largest maxDepth (ray-tree height) hip_trace_init accepts. */
#define HIP_TRACE_MAX_DEPTH 8u

enum {
    HIP_TRACE_SHADING_CPU_MODEL = 0,    /* reference: src/raytracer.c:211-314 -- the parity target */
    /* The shading of the reference's device kernel (src/kernel.cl:671-714): four jittered shadow
     * rays per light hidden for t < distance, ambient/diffuse/specular weights and the material's
     * exponent, attenuation 1 / (1 + 4 pi d^2), that kernel's running division of the four samples,
     * and (with HIP_TRACE_JITTER_HASHED) its depth-of-field aperture draws.  Its random draws come
     * from the hash keyed by (pixel, sub-pixel sample, tree position, light, shadow sample,
     * component), not from the kernel's per-pixel xorshift stream.  The reference can run this model
     * only on an OpenCL device, so it is checked against this project's own CPU restatement only. */
    HIP_TRACE_SHADING_KERNEL_MODEL = 1
};
enum {
    HIP_TRACE_JITTER_ZERO = 0,    /* random_bilateral() == 0: hard point-light shadows */
    /* The reference's light jitter (a random unit vector added to the direction towards the light,
     * reference: src/raytracer.c:278-279) with each random_bilateral() draw replaced by a hash of
     * (pixel = y * width + x, position of the shading point in the pixel's ray tree, light,
     * component): -1 + 2 * ((h >> 8) / 2^24), h = hip_trace_jitter_hash(...).  Tree positions:
     * primary hit 1, hit of the ray reflected at p is 2p, refracted at p is 2p + 1.  Soft shadows
     * that do not depend on the order in which rays are traced. */
    HIP_TRACE_JITTER_HASHED = 1
};
/* Which hierarchy the device walks.  Results are identical: both reach every sphere and triangle
 * a ray can meet, and the winner is decided by the exact tests alone. */
typedef enum {
    HIP_TRACE_ACCEL_PRIVATE_BVH = 0,   /* an 8-wide bounding volume hierarchy built at upload from the
                                          scene's spheres and triangles, each stored once; the octree
                                          argument is ignored and may be NULL                      */
    HIP_TRACE_ACCEL_CALLER_OCTREE = 1  /* the topology of the octree passed to hip_trace_upload_scene
                                          (reference: src/octree.c:384-419), re-packed              */
} HipTraceAccel;

/* How occlusion queries towards the point lights are answered.  Results are identical either way: the candidates
 * are found differently, the tests that decide are the same. */
enum {
    /* Where it applies (hard shadows, the CPU shading model, one ray per pixel): at upload every light gets a direction
     * grid -- a cube around it whose buckets list the spheres and triangles the light sees in those directions -- and a
     * shadow ray tests the entries of the one bucket its direction falls in instead of walking the hierarchy.  Costs
     * device memory (some sixteen 4-byte references per primitive and light, and a byte and a half for each of as many
     * buckets; never more than a quarter of the free memory or sixteen times the scene plus 1 GiB) and upload time (75 ms for a million triangles
     * and eight lights).  A scene too dense for the grids to pay keeps walking. */
    HIP_TRACE_LIGHT_GRID_AUTO = 0,
    HIP_TRACE_LIGHT_GRID_OFF = 1       /* shadow rays always walk the hierarchy (reference: src/kernel.cl:504-550) */
};

typedef struct {
    uint32_t width, height;   /* frame size; must equal camera->width/height at every frame
                                 (reference: src/gpu.c:39 launches width x height items)       */
    uint32_t maxDepth;        /* maxRecursionDepth of raytracer_raycast (src/raytracer.c:316):
                                 bounces + 1; 1 = primary hit and direct light only           */
    uint32_t raysPerPixel;    /* reference: src/main.c:27; 1..64, laid out as the sub-pixel grid
                                 of src/gpu.c:446-465 and averaged (src/kernel.cl:817-843)     */
    uint32_t shadingModel;    /* HIP_TRACE_SHADING_*                                           */
    uint32_t jitterMode;      /* HIP_TRACE_JITTER_*                                            */
    int32_t  device;          /* HIP device ordinal; -1 = the calling thread's current device  */
    uint32_t rank;            /* this handle renders the screen tiles owned by `rank` ...      */
    uint32_t worldSize;       /* ... out of `worldSize` handles (1 = whole frame)              */
    uint32_t accel;           /* HipTraceAccel                                                 */
    uint32_t lightGrid;       /* HIP_TRACE_LIGHT_GRID_*                                        */
    uint32_t reserved[5];     /* must be zero                                                  */
} HipTraceOptions;

/* Ray counters, exact, accumulated by the device during one frame. */
typedef struct {
    uint64_t primary;      /* closest-hit queries from the camera                  */
    uint64_t shadow;       /* occlusion queries towards a light                    */
    uint64_t reflection;   /* closest-hit queries spawned by mirror/glass surfaces */
    uint64_t refraction;   /* closest-hit queries spawned through glass            */
} HipTraceRayCounts;

/* Optional per-frame side outputs.  Host pointers may be NULL; each non-NULL
 * one receives width*height elements, row 0 = top. */
typedef struct {
    uint8_t*  hitMask;     /* 1 where the primary ray's closest primitive has material != 0
                              (reference: src/raytracer.c:227), else 0                        */
    uint32_t* primitiveId; /* closest primitive of the primary ray: planes [0,P), spheres
                              [P,P+S), triangles [P+S,P+S+T); HIP_TRACE_NO_PRIMITIVE if none;
                              ties in distance go to the smallest id, which is the winner of
                              the reference's loop order (src/raytracer.c:223-225)            */
    float*    hitDistance; /* that primitive's t, bit-exact; FLT_MAX if none                  */
    HipTraceRayCounts rays;/* out */
    float     kernelMs;    /* out: device time of the trace kernel, from HIP events           */
    float     totalMs;     /* out: device time of the whole call (uploads, kernel, copies)    */
} HipTraceAux;

/* Sizes of the private device copy, for reporting. */
typedef struct {
    uint32_t wideNodeCount;    /* 8-wide inner nodes of the walked hierarchy (128 B) */
    uint32_t leafCount;        /* non-empty leaves                                  */
    uint32_t leafIndexCount;   /* entries of the leaf primitive-index array         */
    uint32_t treeLevels;       /* depth of the deepest leaf + 1                     */
    uint64_t deviceBytes;      /* total device memory held for the scene            */
    uint64_t lightGridBytes;   /* of which: the lights' direction grids (0 = none)  */
    uint64_t lightGridReferences; /* entries of all their buckets together          */
    uint32_t lightGridLights;  /* lights that have a grid                           */
    uint32_t lightGridPeakMiB; /* device memory the grids' build held at its largest,
                                  in MiB (rounded up): the grids plus four bytes per
                                  bucket, released again before the upload returns   */
} HipTraceSceneInfo;

/* Replaces gpu_initContext's device/context/queue set-up (reference: src/gpu.c:173-199).
 * Creates a handle bound to one device.  *out is NULL on failure. */
int hip_trace_init(HipTrace** out, const HipTraceOptions* options);

/* One handle over several devices of this process -- no counterpart in the reference, which drives a
 * single device (reference: src/gpu.c:176).  `devices` lists deviceCount HIP ordinals (1..64 entries; an
 * ordinal may be repeated, which puts several parts of the frame on one device).  The handle behaves like
 * a single-device handle on devices[0]: hip_trace_upload_scene copies the scene to every listed device,
 * hip_trace_frame / hip_trace_frame_device render part r of the frame (the interleaved 16x16 tiles of rank
 * r of deviceCount, see hip_trace_tile_owner) on devices[r], each on a stream of its own, copy the parts to
 * devices[0] (peer-to-peer copies: an all-to-one gather over as many direct links as there are peers) and
 * re-assemble the row-major frame there; ray counters are summed, kernelMs is the slowest part's.
 * The per-pixel side outputs of HipTraceAux travel the same way.  Calls that return without waiting may follow one
 * another: a part's buffer is not overwritten before the previous frame's copy out of it has finished.
 * options->rank/worldSize must be 0/1 and options->device is ignored.  Not available through such a handle:
 * hip_trace_frame_static, hip_trace_untile. */
int hip_trace_init_multi(HipTrace** out, const HipTraceOptions* options, const int32_t* devices, uint32_t deviceCount);

/* Replaces gpu_allocateCLMemory's ten buffer uploads (reference: src/gpu.c:83-153,201-276).
 * Copies materials, planes, spheres, triangles and lights, and prepares the hierarchy the
 * device walks: with HIP_TRACE_ACCEL_PRIVATE_BVH one built here from the spheres and triangles
 * (octree may be NULL), with HIP_TRACE_ACCEL_CALLER_OCTREE the octree (reference:
 * src/octree.h:15-34 as built by src/octree.c:384-419) re-packed into the device layout.
 * May be called again to replace the scene; on failure the previous scene stays in place.
 * scene->camera is not read. */
int hip_trace_upload_scene(HipTrace* handle, const Scene* scene, const Octree* octree);

/* Replaces gpu_renderScene's camera write, kernel launch and image read-back
 * (reference: src/gpu.c:35-54).  Renders one frame for `camera` and, when rgba8_out is not
 * NULL, copies width*height pixels to it: 0xAABBGGRR, alpha 255, row 0 = top
 * (reference: src/gpu.c:46-52, src/utils/image.c:55-59).  Synchronous: the outputs are
 * complete on return.  Requires worldSize == 1. */
int hip_trace_frame(HipTrace* handle, const Camera* camera, uint32_t* rgba8_out, HipTraceAux* aux);

/* Same frame, output left in device memory (no counterpart in the reference, whose frame
 * stays in a GL texture: src/gpu.c:40-41).  d_out is a device pointer to
 * hip_trace_output_bytes() bytes: the row-major frame at worldSize 1, this rank's tiles in
 * tile-major order otherwise.  Enqueued on the handle's stream; returns without waiting. */
int hip_trace_frame_device(HipTrace* handle, const Camera* camera, void* d_out);

/* hip_trace_frame_device for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the
 * reference.  The same frame into d_out (required, hip_trace_output_bytes() bytes of device
 * memory), but the memory the kernels need beyond inputs and outputs -- ray counters, per-tile
 * cycle counts, tile launch order -- is a fixed array inside the library's code object
 * (HIP_TRACE_STATIC_SCRATCH_BYTES per device the library has been used on, held until the
 * process ends) instead of the buffers hip_trace_init allocated.  The call itself allocates
 * and frees nothing.  Limit: handles of at most HIP_TRACE_STATIC_MAX_TILES local 16x16 tiles
 * (a whole 7680x4320 frame fits); larger ones get HIP_TRACE_ERR_UNSUPPORTED.  All handles on
 * one device share that array: calls that overlap on one device (different streams or
 * threads) race on it, and hip_trace_last_frame_stats reports the counters of the handle's
 * most recent frame through either entry point.  Enqueued on the handle's stream; returns
 * without waiting. */
#define HIP_TRACE_STATIC_MAX_TILES 131072u
#define HIP_TRACE_STATIC_SCRATCH_BYTES (512u + 8u * HIP_TRACE_STATIC_MAX_TILES)
int hip_trace_frame_static(HipTrace* handle, const Camera* camera, void* d_out);

/* Where the most recent frame lies when it was rendered into the handle's own buffer -- by hip_trace_frame, or by
 * hip_trace_frame_device with d_out == NULL: *d_rgba8 receives a device pointer (on the handle's device; devices[0]
 * of a multi-device handle) to hip_trace_output_bytes() bytes laid out as hip_trace_frame_device lays them out.
 * This is the frame of the reference's gpu_renderScene(context, scene, NULL) (reference: src/main.c:181), which that
 * renderer leaves in a GL texture (src/gpu.c:40-41,56-57): a presenter copies or maps it from here instead of paying
 * for a read-back to the host.  The pointer stays valid until the handle is destroyed; its contents are complete once
 * the work enqueued so far on the handle's stream has finished (always, after hip_trace_frame) and are overwritten by
 * the next frame.  HIP_TRACE_ERR_NO_SCENE if no frame has been rendered yet or the last one went to a caller's buffer. */
int hip_trace_last_frame(HipTrace* handle, void** d_rgba8);

/* Bytes hip_trace_frame_device writes for this handle (equal on every rank). */
size_t hip_trace_output_bytes(const HipTrace* handle);

/* Reassembles the row-major frame from the concatenation, in rank order, of every rank's
 * hip_trace_frame_device output (device pointers; d_gathered holds worldSize *
 * hip_trace_output_bytes() bytes, d_rgba8 width*height*4).  No counterpart in the
 * reference (single device).  Enqueued on the handle's stream. */
int hip_trace_untile(HipTrace* handle, const void* d_gathered, void* d_rgba8);

/* Ray counters and kernel time of the most recent frame; waits for it to finish. */
int hip_trace_last_frame_stats(HipTrace* handle, HipTraceRayCounts* rays, float* kernelMs);

/* Profiling aid, no counterpart in the reference: copies the walk counters of the most recent
 * frame (wide nodes tested, leaves opened, triangle distance tests, triangle edge tests) into
 * out[0..capacity) and returns how many were written -- 0 unless the library was built with
 * -DRT_WALK_STATS. */
int hip_trace_walk_stats(HipTrace* handle, uint64_t* out, uint32_t capacity);

/* Use `hipStream` (a hipStream_t, NULL = the default stream) for all later work. */
int hip_trace_set_stream(HipTrace* handle, void* hipStream);

int hip_trace_scene_info(const HipTrace* handle, HipTraceSceneInfo* info);

/* Replaces gpu_destroyContext (reference: src/gpu.c:60-70).  NULL is accepted. */
void hip_trace_destroy(HipTrace* handle);

/* Message of the last failed call on this handle ("" if none); NULL handle gives the
 * message of the last failed hip_trace_init on this thread. */
const char* hip_trace_last_error(const HipTrace* handle);

/* Screen-tile partition shared by the kernel, hip_trace_untile and host code.
 * The frame is cut into HIP_TRACE_TILE x HIP_TRACE_TILE pixel tiles numbered row-major
 * (k = ty * tilesX + tx).  Tile k belongs to rank (k + s*ty) % worldSize where s = 1 when
 * tilesX is a multiple of worldSize (so that ownership is not constant down a column) and
 * 0 otherwise; it is that rank's local tile k / worldSize. */
#define HIP_TRACE_TILE 16u
uint32_t hip_trace_tile_owner(uint32_t tx, uint32_t ty, uint32_t tilesX, uint32_t worldSize);

/* The 32 bits behind one draw of HIP_TRACE_JITTER_HASHED (host function; the kernels evaluate
 * the same integer expression). */
uint32_t hip_trace_jitter_hash(uint32_t pixel, uint32_t treePosition, uint32_t light, uint32_t component);

/* Host-only diagnostic, no counterpart in the reference: builds the hierarchy that
 * hip_trace_upload_scene would build for (scene, octree, accel) without touching a device, and
 * checks what the walk relies on.  Returns HIP_TRACE_OK with *report filled, or the status the
 * upload would return (message, if not NULL, receives the text; capacity includes the NUL). */
typedef struct {
    uint32_t wideNodeCount, leafCount, leafIndexCount, treeLevels;
    uint32_t childCount;             /* non-empty children over all wide nodes                       */
    uint32_t unreachedSpheres;       /* spheres no leaf refers to: must be 0                          */
    uint32_t unreachedTriangles;     /* triangles no leaf refers to: must be 0                        */
    uint32_t duplicateReferences;    /* leaf entries beyond the first per primitive (0 for the BVH)   */
    uint32_t containmentChecked;     /* 1 if the next field was evaluated (HIP_TRACE_ACCEL_PRIVATE_BVH) */
    uint32_t escapedPrimitives;      /* (ancestor child box, primitive) pairs where the primitive's
                                        bounds leave the quantised box: must be 0                     */
    uint32_t badLinks;               /* child links outside the node / leaf arrays, or a node reached
                                        twice: must be 0                                              */
    uint32_t deepestLevel;           /* levels actually met walking from the root (<= treeLevels)     */
    uint32_t checksum;               /* FNV-1a over the packed nodes and the leaf index array         */
} HipTraceHierarchyReport;
int hip_trace_check_hierarchy(const Scene* scene, const Octree* octree, uint32_t accel,
                              HipTraceHierarchyReport* report, char* message, size_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* HIP_TRACE_H */
