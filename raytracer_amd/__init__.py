"""MI355X-native path-tracing hot path behind the RayTracer scene API.

`raytracer_amd.scene` / `raytracer_amd.scenes` describe scenes (host only);
`raytracer_amd.renderer` drives the native library and needs the built
raytracer_amd/lib/libraytracer_amd.so and a GPU - there is no CPU fallback.
"""
from . import abi, scene, scenes  # noqa: F401

__all__ = ["abi", "scene", "scenes"]
