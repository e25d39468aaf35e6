"""nrays_amd — MI355X-native replacement for nrays' per-pixel trace loop (scene::render).

The package holds only what that path needs: the HIP kernels + C ABI (csrc/), the C++ host
front-end (host/), and this thin ctypes mirror of the reference's scene-model surface.
"""
from . import abi  # noqa: F401
from .scene import (Ball, Capsule, Cone, Cuboid, Cylinder, ImageData, Interpolation, Isometry3, Light,  # noqa: F401
                    NormalMaterial, Overflow, PhongMaterial, Plane, Scene, SceneDescriptor, SceneNode, Texture2d,
                    TriMesh, UVMaterial, cast_rays, get_stats, make_params, render, shadow_rays)
