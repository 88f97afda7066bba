/*
 * main_sequence.c -- the reference program's device call sequence, in C, linked against libgpuhip.so
 * and librthost.so: scene, octree, context, one frame into an Image, bitmap, teardown
 * (reference: src/main.c:64-68 set-up, :171-178 render + screenshot, :188-192 teardown).
 * No window: presentation is not part of the library.  In its place the frame of the interactive call
 * gpu_renderScene(context, scene, NULL) (reference: src/main.c:181) is taken from device memory where the library left
 * it (gpu_hip_lastFrame) -- what a presenter would map or copy into its texture -- and written to <out.bmp>.next.bmp.
 *
 * usage: main_sequence <out.bmp> <width> <height> [device ordinal ...]
 *        (ordinals given: the frame is split over them with gpu_hip_configureDevices)
 */
#include <stdio.h>
#include <stdlib.h>

#include <hip/hip_runtime_api.h>

#include "rt_host.h"
#include "gpu_hip.h"

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s <out.bmp> <width> <height> [device ...]\n", argv[0]); return 2; }
    const uint32_t width = (uint32_t)atoi(argv[2]), height = (uint32_t)atoi(argv[3]);
    int32_t devices[64];
    uint32_t deviceCount = 0;
    for (int i = 4; i < argc && deviceCount < 64; i++) devices[deviceCount++] = atoi(argv[i]);

    Scene* scene = rt_scene_init(width, height);                 /* src/main.c:64 */
    if (!scene) { fprintf(stderr, "scene\n"); return 1; }
    Octree* octree = rt_octree_buildFromScene(scene);            /* src/main.c:65 */
    Image* image = rt_image_create(width, height);               /* src/main.c:66 */
    if (!octree || !image) { fprintf(stderr, "octree or image\n"); return 1; }

    if (deviceCount && gpu_hip_configureDevices(devices, deviceCount) != HIP_TRACE_OK) { fprintf(stderr, "devices\n"); return 1; }
    GPUContext* context = gpu_initContext(scene, octree, 1);     /* src/main.c:68 */
    if (!context) { fprintf(stderr, "gpu_initContext: %s\n", gpu_hip_lastError()); return 1; }

    gpu_renderScene(context, scene, image);                      /* src/main.c:171 */
    if (gpu_hip_lastError()[0]) { fprintf(stderr, "gpu_renderScene: %s\n", gpu_hip_lastError()); return 1; }
    if (!rt_bitmap_save_image(argv[1], image)) { fprintf(stderr, "bitmap\n"); return 1; }   /* src/main.c:172-178 */

    /* the camera moves, the next frame follows it (src/main.c:138-160, 181) */
    rt_move_camera(scene->camera, 0, 1, 0);
    gpu_renderScene(context, scene, NULL);
    if (gpu_hip_lastError()[0]) { fprintf(stderr, "gpu_renderScene: %s\n", gpu_hip_lastError()); return 1; }
    void* deviceFrame = NULL;
    uint32_t frameWidth = 0, frameHeight = 0;
    if (gpu_hip_lastFrame(context, &deviceFrame, &frameWidth, &frameHeight) != HIP_TRACE_OK || !deviceFrame ||
        frameWidth != width || frameHeight != height) { fprintf(stderr, "gpu_hip_lastFrame: %s\n", gpu_hip_lastError()); return 1; }
    if (hipMemcpy(image->buffer, deviceFrame, (size_t)width * height * 4, hipMemcpyDeviceToHost) != hipSuccess) { fprintf(stderr, "hipMemcpy\n"); return 1; }
    char nextName[4096];
    snprintf(nextName, sizeof nextName, "%s.next.bmp", argv[1]);
    if (!rt_bitmap_save_image(nextName, image)) { fprintf(stderr, "bitmap\n"); return 1; }
    HipTraceRayCounts rays;
    float kernelMs = 0.0f;
    if (gpu_hip_lastFrameStats(context, &rays, &kernelMs) != HIP_TRACE_OK) { fprintf(stderr, "stats\n"); return 1; }
    printf("primary %llu shadow %llu reflection %llu refraction %llu\n", (unsigned long long)rays.primary,
           (unsigned long long)rays.shadow, (unsigned long long)rays.reflection, (unsigned long long)rays.refraction);

    gpu_destroyContext(context);                                 /* src/main.c:188 */
    rt_image_destroy(image);
    rt_octree_destroy(octree);
    rt_scene_destroy(scene);
    return 0;
}
