#include "rt_host.h"
#include <stdio.h>
#include <stdlib.h>
int main(int argc, char** argv) {
    Scene* s = rt_scene_init(64, 36);
    Octree* t = rt_octree_buildFromScene(s);
    printf("default nodes %u levels %u\n", t->nodeCount, rt_octree_levels(t));
    rt_octree_destroy(t); rt_scene_destroy(s);
    s = rt_scene_synth_mesh(64, 36, 120, 80, 0, 9, 200);
    t = rt_octree_buildFromScene(s);
    printf("mesh nodes %u indexes %u levels %u\n", t->nodeCount, t->indexCount, rt_octree_levels(t));
    Camera c = *s->camera; rt_camera_orbit(&c, 7, 60); rt_move_camera(&c, 1, -1, 2); rt_camera_setup(&c);
    rt_octree_destroy(t);
    for (int i = 1; i < argc; i++) {
        Object* o = rt_object_loadFromFile(argv[i]);
        if (!o) { printf("%s: %s\n", argv[i], rt_object_lastError()); continue; }
        printf("%s: %u triangles\n", argv[i], o->triangleCount);
        Vec3 d = {1, 2, 3};
        rt_object_scale(o, 2.0f); rt_object_translate(o, d); rt_object_materialIndex(o, 2);
        rt_scene_addObject(s, *o);
        rt_object_destroy(o);
    }
    rt_scene_shrinkToFit(s);
    t = rt_octree_buildFromScene(s);
    printf("with objects nodes %u\n", t->nodeCount);
    rt_octree_destroy(t);
    Image* im = rt_image_create(33, 2);
    for (uint32_t i = 0; i < 66; i++) im->buffer[i] = i * 2654435761u;
    printf("bmp %d\n", (int)rt_bitmap_save_image("/tmp/rt_sanitize_out.bmp", im));
    rt_image_destroy(im);
    rt_scene_destroy(s);
    s = rt_scene_create(); t = rt_octree_buildFromScene(s); rt_octree_destroy(t); rt_scene_destroy(s);
    return 0;
}
