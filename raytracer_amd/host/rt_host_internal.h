/* rt_host_internal.h -- shared by the host modules, not installed. */
#ifndef RT_HOST_INTERNAL_H
#define RT_HOST_INTERNAL_H

#include "rt_host.h"

/* Material ids of the default scene, in the order it registers them
 * (reference: src/scene.c:43-105). */
enum {
    RT_MAT_BACKGROUND = 0,
    RT_MAT_GREY = 1,
    RT_MAT_RED_MIRROR = 2,
    RT_MAT_MIRROR = 3,
    RT_MAT_GLASS = 4,
    RT_MAT_YELLOW = 5,
    RT_MAT_METAL = 6
};

void rt_scene_add_default_materials(Scene* scene);
void rt_scene_add_default_planes(Scene* scene);
Camera* rt_default_camera(uint32_t width, uint32_t height);

#endif
