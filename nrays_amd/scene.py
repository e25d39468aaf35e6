"""Host-side mirror of the reference's scene-model surface, flattened into the C-ABI descriptors.

Same names, argument order and meaning as the reference constructors:
  Light.new            src/light.rs:16-23
  PhongMaterial.new    src/phong_material.rs:19-35;  NormalMaterial / UVMaterial  src/{normal,uv}_material.rs
  Texture2d / ImageData src/texture2d.rs:10-76
  SceneNode.new        src/scene_node.rs:22-47
  Scene.new / lights / set_background   src/scene.rs:119-145
  render               src/scene.rs:29-36
The geometry classes stand in for the ncollide3d shapes the loader constructs
(examples/loader3d.rs:601-695).  Nothing here computes pixels: `render` calls the HIP library
through the C ABI (include/nrays_abi.h) and fails loudly if it is missing.
"""
import ctypes as C
import math

import numpy as np

from . import abi


# ----------------------------------------------------------------------------- geometry ----
class Ball:
    kind = abi.SHAPE_BALL

    def __init__(self, radius):
        self.params = (float(radius), 0.0, 0.0)


class Cuboid:
    kind = abi.SHAPE_CUBOID

    def __init__(self, half_extents):
        self.params = tuple(float(x) for x in half_extents)


class Cylinder:
    kind = abi.SHAPE_CYLINDER

    def __init__(self, half_height, radius):
        self.params = (float(half_height), float(radius), 0.0)


class Capsule:
    kind = abi.SHAPE_CAPSULE

    def __init__(self, half_height, radius):
        self.params = (float(half_height), float(radius), 0.0)


class Cone:
    kind = abi.SHAPE_CONE

    def __init__(self, half_height, radius):
        self.params = (float(half_height), float(radius), 0.0)


class Plane:
    """Plane::new(Unit::new_normalize(n)) — loader3d.rs:656 (parse_plane normalises, :863-867)."""
    kind = abi.SHAPE_PLANE

    def __init__(self, normal):
        n = np.asarray(normal, dtype=np.float64)
        n = n / np.linalg.norm(n)
        self.params = tuple(float(x) for x in n)


class TriMesh:
    """TriMesh::new(points, indices, uvs) — loader3d.rs:695."""
    kind = abi.SHAPE_TRIMESH

    def __init__(self, points, indices, uvs=None):
        self.points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        self.uvs = None if uvs is None else np.ascontiguousarray(uvs, dtype=np.float64).reshape(-1, 2)
        if self.uvs is not None and len(self.uvs) != len(self.points):
            raise ValueError("uvs must have one entry per vertex")
        if len(self.indices) and int(self.indices.max()) >= len(self.points):
            raise ValueError("triangle index out of range")
        self.params = (0.0, 0.0, 0.0)


class Isometry3:
    """Isometry3::new(translation, axis_angle): `axis_angle` is a scaled-axis rotation in radians."""

    def __init__(self, translation=(0.0, 0.0, 0.0), axis_angle=(0.0, 0.0, 0.0)):
        self.translation = tuple(float(x) for x in translation)
        self.axis_angle = tuple(float(x) for x in axis_angle)

    @staticmethod
    def new(translation, axis_angle):
        return Isometry3(translation, axis_angle)

    @staticmethod
    def identity():
        return Isometry3()


# ----------------------------------------------------------------------------- textures ----
class ImageData:
    """RGBA texels, row 0 = bottom row (the Y flip of texture2d.rs:99-107 already applied)."""

    def __init__(self, pixels, dims=None):
        arr = np.ascontiguousarray(pixels)
        if arr.dtype == np.uint8:
            self.format = abi.TEXEL_RGBA8
        else:
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            self.format = abi.TEXEL_RGBA32F
        if arr.ndim != 3 or arr.shape[2] != 4:
            raise ValueError("texels must be H x W x 4")
        if dims is not None and (int(dims[0]), int(dims[1])) != (arr.shape[1], arr.shape[0]):
            raise ValueError("dims mismatch")
        if arr.shape[0] < 1 or arr.shape[1] < 1:
            raise ValueError("empty texture")
        self.pixels = arr
        self.dims = (arr.shape[1], arr.shape[0])

    @staticmethod
    def from_image_rows(rows_top_first, opacity=False):
        """Decode of texture2d.rs:99-177 for an 8-bit image given top row first: flips Y, expands depth
        1/3/4 to RGBA with the reference's opaque / opacity conventions.  (Depth 2 multiplies two
        channels in f32 and therefore yields an RGBA32F image.)"""
        img = np.asarray(rows_top_first)
        if img.dtype != np.uint8:
            raise ValueError("8-bit image expected")
        if img.ndim == 2:
            img = img[:, :, None]
        img = img[::-1]
        h, w, d = img.shape
        if d == 2:
            r = img[:, :, 0].astype(np.float32) / np.float32(255.0)
            g = img[:, :, 1].astype(np.float32) / np.float32(255.0)
            out = np.ones((h, w, 4), dtype=np.float32)
            if opacity:
                out[:, :, 3] = g * r
            else:
                out[:, :, 0] = out[:, :, 1] = out[:, :, 2] = r * g
            return ImageData(out)
        out = np.full((h, w, 4), 255, dtype=np.uint8)
        if opacity:
            out[:, :, 3] = img[:, :, 3] if d == 4 else img[:, :, 0]
        elif d == 1:
            out[:, :, 0] = out[:, :, 1] = out[:, :, 2] = img[:, :, 0]
        else:
            out[:, :, :3] = img[:, :, :3]
        return ImageData(out)


class Interpolation:
    Bilinear = abi.INTERP_BILINEAR
    Nearest = abi.INTERP_NEAREST


class Overflow:
    Wrap = abi.OVERFLOW_WRAP
    ClampToEdges = abi.OVERFLOW_CLAMP


class Texture2d:
    def __init__(self, data, interpolation=Interpolation.Bilinear, overflow=Overflow.Wrap):
        self.data = data
        self.interpol = interpolation
        self.overflow = overflow

    @staticmethod
    def new(data, interpolation, overflow):
        return Texture2d(data, interpolation, overflow)


# ----------------------------------------------------------------------------- materials ---
class PhongMaterial:
    kind = abi.MAT_PHONG

    def __init__(self, ambiant_color, diffuse_color, specular_color, texture=None, alpha=None, shininess=100.0):
        self.ambiant_color = tuple(float(x) for x in ambiant_color)
        self.diffuse_color = tuple(float(x) for x in diffuse_color)
        self.specular_color = tuple(float(x) for x in specular_color)
        self.texture = texture
        self.alpha = alpha
        self.shininess = float(shininess)

    @staticmethod
    def new(ambiant_color, diffuse_color, specular_color, texture, alpha, shininess):
        return PhongMaterial(ambiant_color, diffuse_color, specular_color, texture, alpha, shininess)


class NormalMaterial:
    kind = abi.MAT_NORMAL
    texture = None
    alpha = None

    @staticmethod
    def new():
        return NormalMaterial()


class UVMaterial:
    kind = abi.MAT_UV
    texture = None
    alpha = None

    @staticmethod
    def new():
        return UVMaterial()


# ----------------------------------------------------------------------------- lights ------
class Light:
    def __init__(self, pos, radius, nsample, color):
        self.pos = tuple(float(x) for x in pos)
        self.radius = float(radius)
        self.racsample = int(np.sqrt(np.float32(nsample)))  # light.rs:20
        self.color = tuple(float(x) for x in color)

    @staticmethod
    def new(pos, radius, nsample, color):
        return Light(pos, radius, nsample, color)


# ----------------------------------------------------------------------------- scene node --
class SceneNode:
    def __init__(self, material, refl_mix, refl_atenuation, alpha, refr_coeff, transform, geometry, nmap=None,
                 solid=False):
        if nmap is not None:
            # scene_node.rs:60-74 is dead code in the reference (the loader always passes None).
            raise NotImplementedError("nmap is never constructed by the reference loader and is out of scope")
        self.material = material
        self.refl_mix = float(refl_mix)
        self.refl_atenuation = float(refl_atenuation)
        self.alpha = float(alpha)
        self.refr_coeff = float(refr_coeff)
        self.transform = transform
        self.geometry = geometry
        self.solid = bool(solid)

    @staticmethod
    def new(material, refl_mix, refl_atenuation, alpha, refr_coeff, transform, geometry, nmap, solid):
        return SceneNode(material, refl_mix, refl_atenuation, alpha, refr_coeff, transform, geometry, nmap, solid)


# ----------------------------------------------------------------------------- flattening --
class SceneDescriptor:
    """Owns the ctypes arrays behind an NraysSceneDesc (and keeps the numpy buffers alive)."""

    def __init__(self, nodes, lights, background):
        self._keep = []
        tex_index, textures = {}, []
        mat_index, materials = {}, []
        mesh_index, meshes = {}, []

        def tex_id(tex):
            if tex is None:
                return -1
            key = (id(tex.data), tex.interpol, tex.overflow)
            if key not in tex_index:
                t = abi.NraysTexture()
                t.width, t.height = tex.data.dims
                t.format, t.interp, t.overflow = tex.data.format, tex.interpol, tex.overflow
                t.texels = tex.data.pixels.ctypes.data
                self._keep.append(tex.data.pixels)
                tex_index[key] = len(textures)
                textures.append(t)
            return tex_index[key]

        def mat_id(mat):
            if id(mat) not in mat_index:
                m = abi.NraysMaterial()
                m.kind = mat.kind
                if mat.kind == abi.MAT_PHONG:
                    m.ambiant[:] = mat.ambiant_color
                    m.diffuse[:] = mat.diffuse_color
                    m.specular[:] = mat.specular_color
                    m.shininess = mat.shininess
                m.texture_id = tex_id(mat.texture)
                m.alpha_texture_id = tex_id(mat.alpha)
                mat_index[id(mat)] = len(materials)
                materials.append(m)
                self._keep.append(mat)
            return mat_index[id(mat)]

        def mesh_id(geom):
            if id(geom) not in mesh_index:
                m = abi.NraysMesh()
                m.num_vertices, m.num_triangles = len(geom.points), len(geom.indices)
                m.vertices = geom.points.ctypes.data_as(C.POINTER(C.c_double))
                m.uvs = geom.uvs.ctypes.data_as(C.POINTER(C.c_double)) if geom.uvs is not None else None
                m.indices = geom.indices.ctypes.data_as(C.POINTER(C.c_uint32))
                mesh_index[id(geom)] = len(meshes)
                meshes.append(m)
                self._keep.append(geom)
            return mesh_index[id(geom)]

        cnodes = (abi.NraysNode * max(1, len(nodes)))()
        for i, n in enumerate(nodes):
            c = cnodes[i]
            c.shape_kind = n.geometry.kind
            c.solid = 1 if n.solid else 0
            c.params[:] = n.geometry.params
            c.translation[:] = n.transform.translation
            c.axis_angle[:] = n.transform.axis_angle
            c.refl_mix, c.refl_atenuation, c.alpha = n.refl_mix, n.refl_atenuation, n.alpha
            c.refr_coeff = n.refr_coeff
            c.material_id = mat_id(n.material)
            c.mesh_id = mesh_id(n.geometry) if n.geometry.kind == abi.SHAPE_TRIMESH else -1
        clights = (abi.NraysLight * max(1, len(lights)))()
        for i, l in enumerate(lights):
            clights[i].pos[:] = l.pos
            clights[i].radius = l.radius
            clights[i].racsample = l.racsample
            clights[i].color[:] = l.color
        self._nodes, self._lights = cnodes, clights
        self._materials = (abi.NraysMaterial * max(1, len(materials)))(*materials)
        self._textures = (abi.NraysTexture * max(1, len(textures)))(*textures)
        self._meshes = (abi.NraysMesh * max(1, len(meshes)))(*meshes)
        d = abi.NraysSceneDesc()
        d.background[:] = tuple(float(x) for x in background)
        d.num_lights, d.lights = len(lights), self._lights
        d.num_materials, d.materials = len(materials), self._materials
        d.num_textures, d.textures = len(textures), self._textures
        d.num_meshes, d.meshes = len(meshes), self._meshes
        d.num_nodes, d.nodes = len(nodes), self._nodes
        self.desc = d

    def pointer(self):
        return C.byref(self.desc)


class Scene:
    """Scene::new(nodes, lights, background) — src/scene.rs:119-133.  The device-resident scene (BVHs
    included) is created on first render on the current HIP device and reused afterwards."""

    def __init__(self, nodes, lights, background=(1.0, 1.0, 1.0)):
        self._nodes = list(nodes)
        self._lights = list(lights)
        self._background = tuple(float(x) for x in background)
        self._descriptor = None
        self._handle = None

    @staticmethod
    def new(nodes, lights, background):
        return Scene(nodes, lights, background)

    def lights(self):
        return self._lights

    def set_background(self, background):
        self._background = tuple(float(x) for x in background)
        self._release()
        self._descriptor = None

    @property
    def descriptor(self):
        if self._descriptor is None:
            self._descriptor = SceneDescriptor(self._nodes, self._lights, self._background)
        return self._descriptor

    def device_handle(self):
        if self._handle is None:
            lib = abi.load_hip_lib()
            h = C.c_void_p()
            abi.check(lib.nrays_scene_create(self.descriptor.pointer(), C.byref(h)))
            self._handle = h
        return self._handle

    def _release(self):
        if self._handle is not None:
            abi.load_hip_lib().nrays_scene_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass


def make_params(resolution, ray_per_pixel, window_width, camera_eye, projection, max_depth=0, seed=0,
                band_rows=0, band_owner=0, band_owners=1):
    """Packs the arguments of scene::render (src/scene.rs:29-36) into NraysRenderParams."""
    from .math3d import column_major16
    p = abi.NraysRenderParams()
    p.width, p.height = int(resolution[0]), int(resolution[1])
    p.ray_per_pixel = int(ray_per_pixel)
    p.max_depth = int(max_depth)
    p.window_width = float(window_width)
    p.camera_eye[:] = tuple(float(x) for x in camera_eye)
    p.inv_proj_view[:] = tuple(column_major16(projection))
    p.seed = int(seed)
    p.band_rows, p.band_owner, p.band_owners = int(band_rows), int(band_owner), int(band_owners)
    return p


def render(scene, resolution, ray_per_pixel, window_width, camera_eye, projection, max_depth=0, seed=0):
    """scene::render (src/scene.rs:29-116): returns the image as an (H, W, 3) float32 array, row 0 = top.
    Blocking; pixels are computed by the HIP kernels behind nrays_render."""
    if ray_per_pixel <= 0:
        raise ValueError("ray_per_pixel must be > 0")  # assert!(ray_per_pixel > 0), scene.rs:37
    lib = abi.load_hip_lib()
    p = make_params(resolution, ray_per_pixel, window_width, camera_eye, projection, max_depth, seed)
    out = np.empty((p.height, p.width, 3), dtype=np.float32)
    abi.check(lib.nrays_render(scene.device_handle(), C.byref(p), out.ctypes.data_as(C.POINTER(C.c_float))))
    return out


def cast_rays(scene, origins, dirs):
    """The closest-hit query of Scene::trace (src/scene.rs:164-166) on caller-supplied rays, by the HIP traversal and
    intersectors alone (nrays_debug_cast_batch).  Returns (hit mask, (n, 8) array of toi, nx, ny, nz, has_uv, u, v, node) —
    the layout of oracle.cast."""
    o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
    d = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
    n = len(o)
    res = (abi.NraysCastResult * max(n, 1))()
    abi.check(abi.load_hip_lib().nrays_debug_cast_batch(scene.device_handle(), 0, n, o.ctypes.data_as(C.POINTER(C.c_double)),
                                                        d.ctypes.data_as(C.POINTER(C.c_double)), None, res))
    out = np.zeros((n, 8), dtype=np.float64)
    hit = np.zeros(n, dtype=bool)
    for i in range(n):
        r = res[i]
        hit[i] = bool(r.flags & 1)
        out[i] = (r.toi, r.normal[0], r.normal[1], r.normal[2], 1.0 if r.flags & 2 else 0.0, r.uv[0], r.uv[1], r.node_id)
    return hit, out


def shadow_rays(scene, origins, dirs, max_toi):
    """Scene::intersects_ray (src/scene.rs:147-161) on caller-supplied rays by the HIP shadow traversal: returns
    (blocked mask, (n, 3) colour filters)."""
    o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
    d = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
    t = np.ascontiguousarray(max_toi, dtype=np.float64).reshape(-1)
    n = len(o)
    res = (abi.NraysCastResult * max(n, 1))()
    abi.check(abi.load_hip_lib().nrays_debug_cast_batch(scene.device_handle(), 1, n, o.ctypes.data_as(C.POINTER(C.c_double)),
                                                        d.ctypes.data_as(C.POINTER(C.c_double)), t.ctypes.data_as(C.POINTER(C.c_double)), res))
    blocked = np.array([bool(res[i].flags & 1) for i in range(n)])
    filt = np.array([[res[i].normal[0], res[i].normal[1], res[i].normal[2]] for i in range(n)], dtype=np.float64).reshape(n, 3)
    return blocked, filt


def get_stats(scene):
    st = abi.NraysStats()
    abi.check(abi.load_hip_lib().nrays_get_stats(scene.device_handle(), C.byref(st)))
    return st
