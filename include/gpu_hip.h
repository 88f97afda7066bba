/*
 * gpu_hip.h -- the reference's device entry points (reference: src/gpu.h:60-66) implemented
 * over the hip_trace C ABI, so that the reference's main loop (reference: src/main.c:68,110,
 * 171,181,188) links against this library instead of its OpenCL/OpenGL layer.
 *
 * Same names, same arguments, same nullable `image`.  What differs, on purpose:
 *   - GPUContext is opaque (main.c never looks inside it);
 *   - no OpenGL: gpu_renderScene with image == NULL renders and leaves the frame in device
 *     memory, gpu_updateDimensions does nothing.  Presentation (texture upload, quad) stays with
 *     the application (reference: src/gpu.c:521-658);
 *   - a failed gpu_initContext returns NULL without leaking (the reference leaks its context:
 *     src/gpu.c:29-31) and the reason is available from gpu_hip_lastError().
 *
 * Include the reference's scene.h / octree.h / utils/image.h BEFORE this header to use their
 * definitions; otherwise identical ones are declared here.
 */
#ifndef GPU_HIP_H
#define GPU_HIP_H

#include "hip_trace.h"

#ifdef __cplusplus
extern "C" {
#endif

#if !defined(RAYTRACER_IMAGE_H) && !defined(RT_IMAGE_DECLARED)
#define RT_IMAGE_DECLARED
/* reference: src/utils/image.h:8-12 -- pixels 0xAABBGGRR, top row first */
typedef struct {
    uint32_t width, height;
    uint32_t* buffer;
    uint32_t bufferSize;
} Image;
#endif

typedef struct GPUContext GPUContext;

/* reference: src/gpu.c:23-33.  Uploads the scene; `octree` may be NULL (the library builds its own
 * hierarchy unless gpu_hip_configure selected HIP_TRACE_ACCEL_CALLER_OCTREE).  Frame size is
 * scene->camera->width x height.  raysPerPixel: 1..64. */
GPUContext* gpu_initContext(Scene* scene, Octree* octree, uint32_t raysPerPixel);

/* reference: src/gpu.c:35-58.  Renders scene->camera's view; with `image` copies the pixels into
 * image->buffer (image->width/height must equal the frame size).  Synchronous.  Errors are
 * printed to stdout as the reference does ("Couldn't enqueue kernel.", src/gpu.c:42-45) and kept
 * for gpu_hip_lastError(). */
void gpu_renderScene(GPUContext* context, Scene* scene, Image* image);

/* reference: src/gpu.c:60-70.  NULL is accepted. */
void gpu_destroyContext(GPUContext* context);

/* reference: src/gpu.c:645-658 (GL viewport and letter-box scale only): nothing to do here. */
void gpu_updateDimensions(GPUContext* context, uint32_t viewWidth, uint32_t viewHeight, uint32_t renderWidth,
                          uint32_t renderHeight);

/* Not in the reference: options for contexts created afterwards.  maxDepth 0 keeps the default
 * of 5, the fixed depth of the reference kernel (src/kernel.cl:730); device -1 = current device. */
void gpu_hip_configure(uint32_t maxDepth, uint32_t accel, int32_t device);

/* Not in the reference (single device: src/gpu.c:176): contexts created afterwards split every frame over
 * these HIP devices (hip_trace_init_multi: replicated scene, interleaved tiles, parts gathered to devices[0]).
 * deviceCount 0 goes back to the single device of gpu_hip_configure.  Like gpu_hip_configure this sets
 * process-wide defaults read by gpu_initContext; configure before creating contexts from several threads. */
int gpu_hip_configureDevices(const int32_t* devices, uint32_t deviceCount);

/* Not in the reference: text of the last failure on this thread ("" if none). */
const char* gpu_hip_lastError(void);

/* Not in the reference, where the frame of gpu_renderScene(context, scene, NULL) lies in a GL texture
 * (src/gpu.c:40-41) and is drawn from there (:56-57): the same frame here, as a device pointer to width * height
 * 0xAABBGGRR pixels, top row first, for the application's presentation code (hip_trace_last_frame).  width and
 * height may be NULL. */
int gpu_hip_lastFrame(GPUContext* context, void** d_rgba8, uint32_t* width, uint32_t* height);

/* Not in the reference: ray counters and kernel time of the context's last frame. */
int gpu_hip_lastFrameStats(GPUContext* context, HipTraceRayCounts* rays, float* kernelMs);

#ifdef __cplusplus
}
#endif
#endif /* GPU_HIP_H */
