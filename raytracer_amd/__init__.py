"""MI355X-native per-pixel trace path of the raytracer, behind the hip_trace C ABI.

`raytracer_amd.host` binds the plain-C scene model (librthost.so); `raytracer_amd.hiptrace`
binds the HIP library (libhiptrace.so) and raises if it has not been built.
"""
from . import types  # noqa: F401
