"""ctypes mirrors of the plain-C scene records in include/rt_types.h.

Field order and sizes follow the reference's host records (reference:
src/utils/vec3.h:10-17, src/material.h:6-14, src/plane.h:8-12, src/sphere.h:8-12,
src/triangle.h:8-11, src/pointlight.h:6-10, src/ray.h:7-10, src/camera.h:7-15,
src/scene.h:14-36, src/octree.h:10-34).
"""
import ctypes as C


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def __init__(self, x=0.0, y=0.0, z=0.0):
        super().__init__(x, y, z)

    def tuple(self):
        return (self.x, self.y, self.z)


class Ray(C.Structure):
    _fields_ = [("origin", Vec3), ("direction", Vec3)]


class Material(C.Structure):
    _fields_ = [("reflectionIndex", C.c_float), ("refractionIndex", C.c_float),
                ("ambientWeight", C.c_float), ("diffuseWeight", C.c_float),
                ("specularWeight", C.c_float), ("specularExponent", C.c_float),
                ("color", Vec3)]


class Plane(C.Structure):
    _fields_ = [("materialIndex", C.c_uint32), ("normal", Vec3), ("distanceFromOrigin", C.c_float)]


class Sphere(C.Structure):
    _fields_ = [("materialIndex", C.c_uint32), ("position", Vec3), ("radius", C.c_float)]


class Triangle(C.Structure):
    _fields_ = [("materialIndex", C.c_uint32), ("v0", Vec3), ("v1", Vec3), ("v2", Vec3)]


class Object(C.Structure):
    """reference: src/object.h:8-12"""
    _fields_ = [("capacity", C.c_uint32), ("triangleCount", C.c_uint32), ("triangles", C.POINTER(Triangle))]


class Image(C.Structure):
    """reference: src/utils/image.h:8-12"""
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("buffer", C.POINTER(C.c_uint32)), ("bufferSize", C.c_uint32)]


class PointLight(C.Structure):
    _fields_ = [("position", Vec3), ("emissionColor", Vec3), ("strength", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("position", Vec3), ("x", Vec3), ("y", Vec3), ("z", Vec3), ("lookAt", Vec3),
                ("width", C.c_uint32), ("height", C.c_uint32),
                ("FOV", C.c_float), ("aspectRatio", C.c_float),
                ("focalLength", C.c_float), ("apertureSize", C.c_float),
                ("renderTargetCenter", Vec3),
                ("renderTargetWidth", C.c_float), ("renderTargetHeight", C.c_float),
                ("renderTargetDistance", C.c_float)]


class Scene(C.Structure):
    _fields_ = [("camera", C.POINTER(Camera)),
                ("materialCapacity", C.c_uint32), ("materialCount", C.c_uint32),
                ("materials", C.POINTER(Material)),
                ("planeCapacity", C.c_uint32), ("planeCount", C.c_uint32),
                ("planes", C.POINTER(Plane)),
                ("sphereCapacity", C.c_uint32), ("sphereCount", C.c_uint32),
                ("spheres", C.POINTER(Sphere)),
                ("triangleCapacity", C.c_uint32), ("triangleCount", C.c_uint32),
                ("triangles", C.POINTER(Triangle)),
                ("pointLightCapacity", C.c_uint32), ("pointLightCount", C.c_uint32),
                ("pointLights", C.POINTER(PointLight))]


class BoundingBox(C.Structure):
    _fields_ = [("bottomLeftFrontCorner", Vec3), ("topRightBackCorner", Vec3)]


class OctreeNode(C.Structure):
    _fields_ = [("boundingBox", BoundingBox),
                ("sphereIndexOffset", C.c_uint32), ("sphereIndexCount", C.c_uint32),
                ("triangleIndexOffset", C.c_uint32), ("triangleIndexCount", C.c_uint32),
                ("childNodeIndexes", C.c_int32 * 8)]


class Octree(C.Structure):
    _fields_ = [("nodes", C.POINTER(OctreeNode)), ("nodeCount", C.c_uint32), ("nodeCapacity", C.c_uint32),
                ("indexes", C.POINTER(C.c_uint32)), ("indexCount", C.c_uint32), ("indexCapacity", C.c_uint32)]


assert C.sizeof(Vec3) == 12 and C.sizeof(Material) == 36 and C.sizeof(Plane) == 20
assert C.sizeof(Sphere) == 20 and C.sizeof(Triangle) == 40 and C.sizeof(PointLight) == 28
assert C.sizeof(Camera) == 108 and C.sizeof(OctreeNode) == 72 and C.sizeof(Scene) == 88
assert C.sizeof(Octree) == 32 and C.sizeof(Ray) == 24
