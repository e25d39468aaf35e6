"""Camera set-up math of examples/loader3d.rs:68-79 (nalgebra 0.15 Perspective3 / Isometry3::look_at_rh /
Matrix4::try_inverse), in float64 numpy.  Host-side only; the per-ray unprojection runs on the GPU."""
import math

import numpy as np


def perspective(aspect, fovy_rad, znear, zfar):
    """Perspective3::new(aspect, fovy, znear, zfar) as a 4x4."""
    t = math.tan(fovy_rad / 2.0)
    m = np.zeros((4, 4), dtype=np.float64)
    m[0, 0] = 1.0 / (aspect * t)
    m[1, 1] = 1.0 / t
    m[2, 2] = (zfar + znear) / (znear - zfar)
    m[2, 3] = 2.0 * zfar * znear / (znear - zfar)
    m[3, 2] = -1.0
    return m


def look_at_rh(eye, at, up=(0.0, 1.0, 0.0)):
    """Isometry3::look_at_rh(eye, at, up).to_homogeneous()."""
    eye = np.asarray(eye, dtype=np.float64)
    at = np.asarray(at, dtype=np.float64)
    up = np.asarray(up, dtype=np.float64)
    z = eye - at
    z = z / np.linalg.norm(z)
    x = np.cross(up, z)
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    m = np.eye(4, dtype=np.float64)
    m[0, :3], m[1, :3], m[2, :3] = x, y, z
    m[:3, 3] = -(m[:3, :3] @ eye)
    return m


def inverse_projection(eye, at, fovy_deg, resx, resy, znear=1.0, zfar=100000.0):
    """(P * V)^-1 exactly as loader3d.rs:68-79 builds the `projection` argument of scene::render."""
    p = perspective(float(resx) / float(resy), math.radians(fovy_deg), znear, zfar)
    v = look_at_rh(eye, at)
    return np.linalg.inv(p @ v)


def column_major16(m):
    """nalgebra stores Matrix4 column-major; the ABI takes that layout."""
    return np.ascontiguousarray(np.asarray(m, dtype=np.float64).T).reshape(16)
