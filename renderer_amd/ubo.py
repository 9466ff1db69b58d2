"""Host-side mirrors of the reference's uniform blocks and camera.

Byte layouts follow /root/reference/renderer/src/render/ubo.rs:28-63 (std140,
``repr(C, align(16))``) and the GLSL blocks in
renderer/shaders/fs_deferred_lighting.glsl:12-27.  The matrix builders restate
cgmath 0.18's closed forms used by renderer/src/camera.rs:75-88 (cgmath is a
crates.io dependency that is not vendored in the reference: parity unpinned).
Matrices are kept as ``M[row, col]`` float arrays and flattened column-major
(cgmath ``Matrix4`` = 4 column vectors) when packed.
"""
import math
import numpy as np

FRAME_UBO_BYTES = 272       # FrameMatrixData  (ubo.rs:28-41)
LIGHT_STRIDE_BYTES = 32     # DirectionalLight (ubo.rs:54-63), std140 array stride
PUSH_CONSTANT_BYTES = 12    # vec2 resolution + uint light_count (fs_deferred_lighting.glsl:24-27)
MAX_LIGHTS = 100            # inc_structs.glsl:1 ; pool size render/pbr/mod.rs:36


def perspective(fovy_rad, aspect, near, far):
    """cgmath ``PerspectiveFov{fovy,aspect,near,far}.into()`` (camera.rs:76-84):
    OpenGL-style clip space, ``m[2][3] = -1``."""
    f = 1.0 / math.tan(fovy_rad / 2.0)
    m = np.zeros((4, 4), dtype=np.float64)
    m[0, 0] = f / aspect
    m[1, 1] = f
    m[2, 2] = (far + near) / (near - far)
    m[3, 2] = -1.0
    m[2, 3] = (2.0 * far * near) / (near - far)
    return m.astype(np.float32)


def look_to_rh(eye, direction, up):
    """cgmath ``Matrix4::look_to_rh(eye, dir, up)`` (camera.rs:86-88)."""
    eye = np.asarray(eye, np.float64)
    f = np.asarray(direction, np.float64)
    f = f / np.linalg.norm(f)
    s = np.cross(f, np.asarray(up, np.float64))
    s = s / np.linalg.norm(s)
    u = np.cross(s, f)
    m = np.eye(4, dtype=np.float64)
    m[0, :3], m[1, :3], m[2, :3] = s, u, -f
    m[0, 3], m[1, 3], m[2, 3] = -np.dot(eye, s), -np.dot(eye, u), np.dot(eye, f)
    return m.astype(np.float32)


def ortho_zo(half_extent, near, far):
    """Orthographic projection with depth 0..1 (used only by the shadow extension)."""
    m = np.eye(4, dtype=np.float64)
    m[0, 0] = 1.0 / half_extent
    m[1, 1] = 1.0 / half_extent
    m[2, 2] = -1.0 / (far - near)
    m[2, 3] = -near / (far - near)
    return m.astype(np.float32)


def invert(m):
    """``SquareMatrix::invert`` (render/mod.rs:100-101), evaluated in float64."""
    return np.linalg.inv(np.asarray(m, np.float64)).astype(np.float32)


def columns(m):
    """M[row, col] -> 16 floats column-major."""
    return np.ascontiguousarray(np.asarray(m, np.float32).T).reshape(16)


class PerspectiveCamera:
    """camera.rs:17-25 / main.rs:70-78 defaults."""

    def __init__(self, position=(0.0, 3.0, 0.0), forward=(1.0, 0.0, 0.0), up=(0.0, -1.0, 0.0),
                 fov_deg=90.0, aspect_ratio=16.0 / 9.0, near=0.05, far=100.0):
        self.position = np.asarray(position, np.float32)
        self.forward = np.asarray(forward, np.float32)
        self.up = np.asarray(up, np.float32)
        self.fov = math.radians(fov_deg)
        self.aspect_ratio = aspect_ratio
        self.near, self.far = near, far

    def projection_matrix(self):
        return perspective(self.fov, self.aspect_ratio, self.near, self.far)

    def view_matrix(self):
        return look_to_rh(self.position, self.forward, self.up)

    def frame_matrix_data(self):
        """What ``Frame::build`` fills per frame (render/mod.rs:96-104)."""
        view, proj = self.view_matrix(), self.projection_matrix()
        return frame_matrix_data(view, proj, self.position)


def frame_matrix_data(view, projection, camera_position):
    """68 float32 = 272 B: view, projection, invProjection, invView, cameraPosition, pad."""
    out = np.zeros(68, dtype=np.float32)
    out[0:16] = columns(view)
    out[16:32] = columns(projection)
    out[32:48] = columns(invert(projection))
    out[48:64] = columns(invert(view))
    out[64:67] = np.asarray(camera_position, np.float32)
    return out


def directional_lights(entries):
    """[(direction, intensity, color), ...] -> (n, 8) float32; direction is
    normalised on the CPU as main.rs:82,87 does (the shader does not)."""
    out = np.zeros((len(entries), 8), dtype=np.float32)
    for i, (d, inten, col) in enumerate(entries):
        d = np.asarray(d, np.float64)
        out[i, 0:3] = d / np.linalg.norm(d)
        out[i, 3] = inten
        out[i, 4:7] = col
    return out


def default_directional_lights():
    """main.rs:80-91."""
    return directional_lights([((5.0, 5.0, 1.0), 2.5, (1.0, 1.0, 0.8)),
                               ((-5.0, 5.0, 1.0), 2.5, (0.8, 1.0, 1.0))])


def point_lights(position, radius, color, intensity):
    """EXTENSION E1 record, same 32-byte stride: {position.xyz, radius, color.xyz, intensity}."""
    position = np.asarray(position, np.float32).reshape(-1, 3)
    n = position.shape[0]
    out = np.zeros((n, 8), dtype=np.float32)
    out[:, 0:3] = position
    out[:, 3] = radius
    out[:, 4:7] = color
    out[:, 7] = intensity
    return out


def push_constants(width, height, light_count):
    """12 bytes: vec2 resolution, uint light_count (render/mod.rs:211-214)."""
    b = np.zeros(3, dtype=np.uint32)
    b[0:2] = np.array([width, height], dtype=np.float32).view(np.uint32)
    b[2] = light_count
    return b
