/*
 * rt_host.h -- host-side scene model helpers (plain C, no GPU).
 *
 * Stand-alone equivalents of the reference renderer's host modules, for
 * programs that do not link the reference itself (the benchmark, the tests):
 *   scene container and default scene   reference: src/scene.c:7-290
 *   camera                              reference: src/camera.c:9-58
 *   octree builder                      reference: src/octree.c:7-427
 *   OBJ ingestion                       reference: src/object.c:145-217
 *   frame container, bitmap writer      reference: src/utils/image.c:5-66
 * Names carry an rt_ prefix so both can be linked into one program.  Given the
 * same input, each function produces the same bytes as its counterpart (the
 * octree: same node numbering, boxes and index runs).
 *
 * The synthetic scene generators have no counterpart in the reference.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include <stdbool.h>

#include "rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scene container (reference: src/scene.h:38-48) ---- */
Scene*   rt_scene_create(void);
Scene*   rt_scene_init(uint32_t width, uint32_t height);   /* the reference's hard-coded scene */
uint32_t rt_scene_addMaterial(Scene* scene, Material material);
void     rt_scene_addPlane(Scene* scene, Plane plane);
void     rt_scene_addSphere(Scene* scene, Sphere sphere);
void     rt_scene_addTriangle(Scene* scene, Triangle triangle);
void     rt_scene_addObject(Scene* scene, Object object);
void     rt_scene_addPointLight(Scene* scene, PointLight pointLight);
/* Not in the reference: how many elements the adders above have had to drop since the process started because memory
 * ran out (they return nothing, like the reference's).  A caller that builds a large scene compares the value before
 * and after; anything but equal means the scene is incomplete and should not be uploaded. */
uint32_t rt_scene_droppedElements(void);
void     rt_scene_shrinkToFit(Scene* scene);
void     rt_scene_destroy(Scene* scene);

/* ---- Wavefront OBJ ingestion (reference: src/object.h:14-18) ---- */
/* The triangles of a file's `v` / `f` records (faces fanned, material index 0), identical to the
 * reference loader's for the same file; NULL for an unreadable file or a face that names a vertex
 * that does not exist (the reference asserts there), with the reason in rt_object_lastError(). */
Object* rt_object_loadFromFile(const char* filepath);
void    rt_object_scale(Object* object, float factor);
void    rt_object_translate(Object* object, Vec3 translation);
void    rt_object_materialIndex(Object* object, uint32_t materialIndex);
void    rt_object_destroy(Object* object);
const char* rt_object_lastError(void);

/* ---- frame container and bitmap writer (reference: src/utils/image.h:14-15,40) ---- */
Image* rt_image_create(uint32_t width, uint32_t height);
void   rt_image_destroy(Image* image);
/* Writes `image` as a 32-bit Windows bitmap, byte for byte the file the reference writes. */
bool   rt_bitmap_save_image(const char* path, const Image* image);

/* ---- camera (reference: src/camera.h:17-20) ---- */
Camera* rt_camera_create(Vec3 position, Vec3 lookAt, uint32_t width, uint32_t height,
                         float FOV, float apertureSize);
void    rt_camera_setup(Camera* camera);
void    rt_move_camera(Camera* camera, int upDown, int side, int frontal);
void    rt_camera_destroy(Camera* camera);

/* ---- octree (reference: src/octree.h:36-37) ---- */
Octree* rt_octree_buildFromScene(Scene* scene);
void    rt_octree_destroy(Octree* octree);
/* Deepest leaf level + 1 (root alone = 1); 0 for a malformed tree. */
uint32_t rt_octree_levels(const Octree* octree);

/* ---- synthetic scenes ---- */

/* The default scene's materials, five walls/floor planes and camera, plus a height-field
 * mesh of 2*quadsX*quadsZ triangles over x,z in [-25,25]:
 *   y(ix,iz) = 1.5 + 0.75*sin(0.37x)*cos(0.29z) + 0.25*h(ix,iz),  h = 32-bit hash in [0,1)
 * truncated to maxTriangles (0 = keep all); material cycles metal / yellow / redMirror by
 * (ix+iz)%3; lightCount lights of strength 10000 on a ring r=30, y=20;
 * sphereCount spheres r in [0.2,0.6] on a jittered grid at y in [3,8] (PCG32 seed 42),
 * materials cycling mirror / glass / metal. */
Scene* rt_scene_synth_mesh(uint32_t width, uint32_t height, uint32_t quadsX, uint32_t quadsZ,
                           uint32_t maxTriangles, uint32_t lightCount, uint32_t sphereCount);

/* 1,000,000 triangles (1000 x 500 quads), 8 lights, no spheres. */
Scene* rt_scene_synth_s1(uint32_t width, uint32_t height);
/* 10,000,000 triangles (3163 x 1581 quads, truncated), 64 lights on two rings, 1000 spheres. */
Scene* rt_scene_synth_s2(uint32_t width, uint32_t height);

/* Camera for frame `frame` of `frameCount` on an orbit of radius 40 around the origin, rising
 * from y=2 to y=12, looking at the origin; rebuilds the basis with rt_camera_setup. */
void rt_camera_orbit(Camera* camera, uint32_t frame, uint32_t frameCount);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
