/*
 * gpu_hip.c -- gpu_initContext / gpu_renderScene / gpu_destroyContext / gpu_updateDimensions
 * (reference: src/gpu.h:60-66) over libhiptrace.  Plain C11, no GPU code here.
 */
#include "gpu_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct GPUContext {
    HipTrace* trace;
    uint32_t width, height;
};

static _Thread_local char lastError[512];
static uint32_t configuredDepth = 5;      /* reference: src/kernel.cl:730 */
static uint32_t configuredAccel = HIP_TRACE_ACCEL_PRIVATE_BVH;
static int32_t configuredDevice = -1;
static int32_t configuredDevices[64];     /* gpu_hip_configureDevices: the frame is split over these */
static uint32_t configuredDeviceCount = 0;

static void remember(const char* what, const char* detail) {
    snprintf(lastError, sizeof lastError, "%s%s%s", what, detail && detail[0] ? ": " : "", detail ? detail : "");
}

void gpu_hip_configure(uint32_t maxDepth, uint32_t accel, int32_t device) {
    configuredDepth = maxDepth ? maxDepth : 5;
    configuredAccel = accel;
    configuredDevice = device;
}

int gpu_hip_configureDevices(const int32_t* devices, uint32_t deviceCount) {
    if (deviceCount > 64 || (deviceCount && !devices)) return HIP_TRACE_ERR_ARGUMENT;
    for (uint32_t i = 0; i < deviceCount; i++) configuredDevices[i] = devices[i];
    configuredDeviceCount = deviceCount;
    return HIP_TRACE_OK;
}

const char* gpu_hip_lastError(void) { return lastError; }

GPUContext* gpu_initContext(Scene* scene, Octree* octree, uint32_t raysPerPixel) {
    lastError[0] = 0;
    if (!scene || !scene->camera) { remember("gpu_initContext", "scene or scene->camera is NULL"); return NULL; }
    GPUContext* context = calloc(1, sizeof *context);
    if (!context) { remember("gpu_initContext", "out of memory"); return NULL; }
    HipTraceOptions options;
    memset(&options, 0, sizeof options);
    options.width = scene->camera->width;
    options.height = scene->camera->height;
    options.maxDepth = configuredDepth;
    options.raysPerPixel = raysPerPixel;
    options.device = configuredDevice;
    options.worldSize = 1;
    options.accel = configuredAccel;
    int rc = configuredDeviceCount > 0 ? hip_trace_init_multi(&context->trace, &options, configuredDevices, configuredDeviceCount)
                                       : hip_trace_init(&context->trace, &options);
    if (rc != HIP_TRACE_OK) {
        remember("gpu_initContext", hip_trace_last_error(NULL));
        free(context);
        return NULL;
    }
    rc = hip_trace_upload_scene(context->trace, scene, octree);
    if (rc != HIP_TRACE_OK) {
        remember("gpu_initContext", hip_trace_last_error(context->trace));
        hip_trace_destroy(context->trace);
        free(context);
        return NULL;
    }
    context->width = options.width;
    context->height = options.height;
    return context;
}

void gpu_renderScene(GPUContext* context, Scene* scene, Image* image) {
    lastError[0] = 0;
    if (!context || !scene || !scene->camera) {
        remember("gpu_renderScene", "NULL argument");
        printf("Couldn't render: %s\n", lastError);
        return;
    }
    if (image && (image->width != context->width || image->height != context->height || !image->buffer)) {
        remember("gpu_renderScene", "image size differs from the frame size");
        printf("Couldn't render: %s\n", lastError);
        return;
    }
    int rc = hip_trace_frame(context->trace, scene->camera, image ? image->buffer : NULL, NULL);
    if (rc != HIP_TRACE_OK) {
        remember("gpu_renderScene", hip_trace_last_error(context->trace));
        printf("Couldn't render: %s\n", lastError);
    }
}

void gpu_destroyContext(GPUContext* context) {
    if (!context) return;
    hip_trace_destroy(context->trace);
    free(context);
}

void gpu_updateDimensions(GPUContext* context, uint32_t viewWidth, uint32_t viewHeight, uint32_t renderWidth,
                          uint32_t renderHeight) {
    (void)context; (void)viewWidth; (void)viewHeight; (void)renderWidth; (void)renderHeight;
}

int gpu_hip_lastFrameStats(GPUContext* context, HipTraceRayCounts* rays, float* kernelMs) {
    if (!context) return HIP_TRACE_ERR_ARGUMENT;
    return hip_trace_last_frame_stats(context->trace, rays, kernelMs);
}

int gpu_hip_lastFrame(GPUContext* context, void** d_rgba8, uint32_t* width, uint32_t* height) {
    if (d_rgba8) *d_rgba8 = NULL;
    if (!context || !d_rgba8) return HIP_TRACE_ERR_ARGUMENT;
    if (width) *width = context->width;
    if (height) *height = context->height;
    int rc = hip_trace_last_frame(context->trace, d_rgba8);
    if (rc != HIP_TRACE_OK) remember("gpu_hip_lastFrame", hip_trace_last_error(context->trace));
    return rc;
}
