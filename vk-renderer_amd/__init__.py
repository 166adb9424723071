"""MI355X-native deferred lighting resolve + frustum cull (the two compute passes of vk-renderer).

This package is a thin ctypes binding of the C ABI in ``include/vkr_c.h`` (built in-tree as
``lib/libvkr_hip.so`` from the HIP sources in ``csrc/``) plus host-side helpers.  There is no CPU
fallback: creating a :class:`Context` without a gfx950 GPU raises.

The directory name contains a hyphen, so import it with
``importlib.import_module("vk-renderer_amd")`` (``tests/conftest.py`` and ``bench.py`` do that).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvkr_hip.so")

# vkr_status
OK, ERR_INVALID_ARGUMENT, ERR_OUT_OF_MEMORY, ERR_DEVICE, ERR_NOT_FOUND, ERR_NOT_READY, ERR_BAD_STATE, ERR_UNSUPPORTED = range(8)
# vkr_format (VkFormat values)
FORMAT_R8G8B8A8_UNORM = 37
FORMAT_B8G8R8A8_UNORM = 44
FORMAT_R16G16B16A16_SFLOAT = 97
FORMAT_R32_SFLOAT = 100
FORMAT_R32G32B32A32_SFLOAT = 109
FORMAT_D32_SFLOAT = 126
BUFFER_DEVICE, BUFFER_HOST_VISIBLE = 0, 1
DEFERRED_RADIUS_CUTOFF = 1
DEFERRED_NO_TILE_LISTS = 2
DEFERRED_WHOLE_TILE_WORKGROUPS = 4
CULL_DETERMINISTIC = 1
CULL_CMDS_PREZEROED = 2
NOALLOC_MAX_INSTANCES = 1 << 22
LIGHT_STRIDE = 48

DRAW_CMD_DTYPE = np.dtype([("indexCount", "<u4"), ("instanceCount", "<u4"), ("firstIndex", "<u4"),
                           ("vertexOffset", "<i4"), ("firstInstance", "<u4")])

_TEXEL_BYTES = {FORMAT_R8G8B8A8_UNORM: 4, FORMAT_B8G8R8A8_UNORM: 4, FORMAT_R16G16B16A16_SFLOAT: 8,
                FORMAT_R32_SFLOAT: 4, FORMAT_R32G32B32A32_SFLOAT: 16, FORMAT_D32_SFLOAT: 4}


class VkrError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"vkr status {status}: {message}")
        self.status = status


class Image2D(C.Structure):
    _fields_ = [("device", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32),
                ("pitch_bytes", C.c_uint32), ("format", C.c_uint32)]


class DeferredArgs(C.Structure):
    _fields_ = [("albedo", Image2D), ("material", Image2D), ("normal", Image2D), ("depth", Image2D), ("out", Image2D),
                ("lights", C.c_void_p), ("light_count", C.c_uint32), ("push", C.c_float * 20),
                ("row_offset", C.c_uint32), ("full_height", C.c_uint32), ("flags", C.c_uint32), ("reserved", C.c_uint32)]


class TileStats(C.Structure):
    _fields_ = [("tiles", C.c_uint32), ("dark_tiles", C.c_uint32), ("listed_tiles", C.c_uint32), ("overflow_tiles", C.c_uint32),
                ("max_list", C.c_uint32), ("list_cap", C.c_uint32), ("light_pixels", C.c_uint64), ("pixels", C.c_uint64),
                ("deep_tiles", C.c_uint32), ("reserved", C.c_uint32)]


class CullArgs(C.Structure):
    _fields_ = [("aabb_min", C.c_void_p), ("aabb_max", C.c_void_p), ("instance_count", C.c_uint64),
                ("planes", C.c_float * 24), ("mesh_base", C.c_void_p), ("mesh_count", C.c_uint32),
                ("first_mesh", C.c_uint32), ("first_instance", C.c_uint32), ("flags", C.c_uint32),
                ("cmd_template", C.c_void_p), ("cmd_out", C.c_void_p), ("visible", C.c_void_p),
                ("error_flag", C.c_void_p)]


class Camera(C.Structure):
    """vkr_camera: CameraInputHandler state of /root/reference/inc/vkcamera.hpp:62-63."""
    _fields_ = [("pos", C.c_float * 3), ("forward", C.c_float * 3), ("up", C.c_float * 3)]


class FrameConstants(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("view_proj", C.c_float * 16), ("inv_view_proj", C.c_float * 16),
                ("planes", C.c_float * 24), ("deferred_push", C.c_float * 20)]


LIGHT_INFO_DTYPE = np.dtype([("pos", "<f4", 3), ("opening_angle", "<f4"), ("color", "<f4", 3), ("penumbra_angle", "<f4"),
                             ("dir", "<f4", 3), ("_temp", "<f4")])
assert LIGHT_INFO_DTYPE.itemsize == LIGHT_STRIDE
MESH_INFO_DTYPE = np.dtype([("indexCount", "<u4"), ("firstIndex", "<u4"), ("vertexOffset", "<u4"), ("materialIndex", "<u4")])
PER_MESH_DATA_DTYPE = np.dtype([("materialIdx", "<u4"), ("modelMatrixBaseIndex", "<u4")])
SPHERE_DTYPE = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("albedo", "<f4", 3), ("roughness", "<f4"),
                         ("metalness", "<f4"), ("occlusion", "<f4"), ("pad", "<f4", 2)])
assert SPHERE_DTYPE.itemsize == 48
MESH_HAS_FACES, MESH_HAS_UV0, MESH_HAS_TRIANGLES = 1, 2, 4
KEY_W, KEY_A, KEY_S, KEY_D, KEY_SPACE, KEY_LEFT_SHIFT = range(6)


class SceneDesc(C.Structure):
    _fields_ = [("node_count", C.c_uint32), ("node_transform", C.c_void_p), ("node_child_begin", C.c_void_p),
                ("node_children", C.c_void_p), ("node_mesh_begin", C.c_void_p), ("node_meshes", C.c_void_p),
                ("mesh_count", C.c_uint32), ("mesh_vertex_count", C.c_void_p), ("mesh_face_count", C.c_void_p),
                ("mesh_material", C.c_void_p), ("mesh_flags", C.c_void_p)]


class SceneCounts(C.Structure):
    _fields_ = [("kept_meshes", C.c_uint32), ("kept_instances", C.c_uint32), ("total_instances", C.c_uint32),
                ("total_vertices", C.c_uint32), ("total_indices", C.c_uint32)]


class SceneTables(C.Structure):
    _fields_ = [("mesh_info", C.c_void_p), ("draw_commands", C.c_void_p), ("per_mesh_data", C.c_void_p),
                ("mesh_matrices", C.c_void_p), ("mesh_base", C.c_void_p), ("vertex_base", C.c_void_p), ("source_mesh", C.c_void_p)]


class GBufferEncodeArgs(C.Structure):
    _fields_ = [("albedo_f32", C.c_void_p), ("material_f32", C.c_void_p), ("normal_f32", C.c_void_p), ("depth_f32", C.c_void_p),
                ("albedo", Image2D), ("material", Image2D), ("normal", Image2D), ("depth", Image2D)]


class GBufferSurfaceArgs(C.Structure):
    _fields_ = [("albedo_texel", C.c_void_p), ("specular_texel", C.c_void_p), ("normal_texel", C.c_void_p), ("normal_obj", C.c_void_p),
                ("tangent_obj", C.c_void_p), ("depth_f32", C.c_void_p), ("model_matrices", C.c_void_p), ("matrix_count", C.c_uint32),
                ("reserved", C.c_uint32), ("albedo", Image2D), ("material", Image2D), ("normal", Image2D), ("depth", Image2D)]


class GBufferScene(C.Structure):
    _fields_ = [("view_proj", C.c_float * 16), ("inv_view_proj", C.c_float * 16), ("camera_pos", C.c_float * 3),
                ("sphere_count", C.c_uint32), ("ground_plane", C.c_float * 4), ("ground_albedo_a", C.c_float * 3),
                ("ground_tile", C.c_float), ("ground_albedo_b", C.c_float * 3), ("ground_roughness", C.c_float),
                ("ground_metalness", C.c_float), ("pad", C.c_float * 3)]


class GBufferFillArgs(C.Structure):
    _fields_ = [("albedo", Image2D), ("material", Image2D), ("normal", Image2D), ("depth", Image2D),
                ("spheres", C.c_void_p), ("scene", GBufferScene), ("row_offset", C.c_uint32), ("full_height", C.c_uint32)]


# Every symbol include/vkr_c.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
EXPORTS = {
    "vkr_context_create": (C.c_int32, [C.c_int32, _PP]),
    "vkr_context_destroy": (C.c_int32, [_P]),
    "vkr_last_error": (C.c_char_p, []),
    "vkr_context_device_info": (C.c_int32, [_P, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "vkr_context_self_test": (C.c_int32, [_P, C.c_uint32, C.POINTER(C.c_uint32)]),
    "vkr_context_stream": (C.c_int32, [_P, _PP]),
    "vkr_buffer_create": (C.c_int32, [_P, C.c_uint64, C.c_uint32, _PP]),
    "vkr_buffer_wrap": (C.c_int32, [_P, _P, C.c_uint64, _PP]),
    "vkr_buffer_destroy": (C.c_int32, [_P]),
    "vkr_buffer_size": (C.c_int32, [_P, C.POINTER(C.c_uint64)]),
    "vkr_buffer_map": (C.c_int32, [_P, _PP]),
    "vkr_buffer_device_ptr": (C.c_int32, [_P, _PP]),
    "vkr_buffer_upload": (C.c_int32, [_P, C.c_uint64, _P, C.c_uint64]),
    "vkr_buffer_download": (C.c_int32, [_P, C.c_uint64, _P, C.c_uint64]),
    "vkr_buffer_fill_u32": (C.c_int32, [_P, C.c_uint64, C.c_uint32, C.c_uint64]),
    "vkr_deferred_resolve": (C.c_int32, [_P, C.POINTER(DeferredArgs)]),
    "vkr_deferred_tile_stats": (C.c_int32, [_P, C.POINTER(TileStats)]),
    "vkr_prep_indirect": (C.c_int32, [_P, C.POINTER(CullArgs)]),
    "vkr_prep_indirect_noalloc": (C.c_int32, [_P, C.POINTER(CullArgs)]),
    "vkr_prep_indirect_reset": (C.c_int32, [_P, C.POINTER(CullArgs)]),
    "vkr_kernel_get": (C.c_int32, [_P, C.c_char_p, _PP]),
    "vkr_kernel_push_constant_size": (C.c_int32, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "vkr_cmd_begin": (C.c_int32, [_P, _PP]),
    "vkr_cmd_bind_kernel": (C.c_int32, [_P, _P]),
    "vkr_cmd_push_constants": (C.c_int32, [_P, C.c_uint32, C.c_uint32, _P]),
    "vkr_cmd_bind_buffer": (C.c_int32, [_P, C.c_uint32, C.c_uint32, _P, C.c_uint64, C.c_uint64]),
    "vkr_cmd_bind_image2d": (C.c_int32, [_P, C.c_uint32, C.c_uint32, C.c_uint32, _P, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "vkr_cmd_dispatch": (C.c_int32, [_P, C.c_uint32, C.c_uint32, C.c_uint32]),
    "vkr_cmd_end": (C.c_int32, [_P]),
    "vkr_cmd_destroy": (C.c_int32, [_P]),
    "vkr_queue_submit": (C.c_int32, [_P, _P, _PP]),
    "vkr_queue_marker": (C.c_int32, [_P, _PP]),
    "vkr_queue_wait_idle": (C.c_int32, [_P]),
    "vkr_fence_status": (C.c_int32, [_P]),
    "vkr_fence_wait": (C.c_int32, [_P, C.c_uint64]),
    "vkr_fence_elapsed_ms": (C.c_int32, [_P, _P, C.POINTER(C.c_float)]),
    "vkr_fence_destroy": (C.c_int32, [_P]),
    "vkr_perspective": (None, [C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    "vkr_look_at": (None, [_P, _P, _P, _P]),
    "vkr_mat4_mul": (None, [_P, _P, _P]),
    "vkr_mat4_inverse": (C.c_int32, [_P, _P]),
    "vkr_frustum_planes": (None, [_P, _P]),
    "vkr_camera_default": (None, [C.POINTER(Camera)]),
    "vkr_camera_view": (None, [C.POINTER(Camera), _P]),
    "vkr_camera_key": (None, [C.POINTER(Camera), C.c_int32, C.c_float]),
    "vkr_camera_cursor_delta": (None, [C.POINTER(Camera), C.c_float, C.c_float]),
    "vkr_frame_setup": (C.c_int32, [C.POINTER(Camera), _P, C.POINTER(FrameConstants)]),
    "vkr_pack_point_lights": (None, [C.c_uint32, _P, _P, _P, _P, _P, _P]),
    "vkr_fscene_point_lights": (C.c_int32, [C.c_char_p, C.c_uint64, _P, C.c_uint32, C.POINTER(C.c_uint32)]),
    "vkr_scene_count": (C.c_int32, [C.POINTER(SceneDesc), C.POINTER(SceneCounts)]),
    "vkr_scene_build_tables": (C.c_int32, [C.POINTER(SceneDesc), C.POINTER(SceneCounts), C.POINTER(SceneTables)]),
    "vkr_mesh_aabbs": (C.c_int32, [_P, _P, _P, C.c_uint32, _P]),
    "vkr_instance_aabbs": (C.c_int32, [_P, _P, _P, _P, C.c_uint32, C.c_uint64, _P, _P]),
    "vkr_gbuffer_encode": (C.c_int32, [_P, C.POINTER(GBufferEncodeArgs)]),
    "vkr_gbuffer_encode_surfaces": (C.c_int32, [_P, C.POINTER(GBufferSurfaceArgs)]),
    "vkr_gbuffer_fill": (C.c_int32, [_P, C.POINTER(GBufferFillArgs)]),
}

_lib = None


def lib():
    """Loads libvkr_hip.so; raises if the extension has not been built (there is no fallback)."""
    global _lib
    if _lib is None:
        path = os.environ.get("VKR_LIB_PATH", LIB_PATH)  # tuning builds only
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: build it with `python vk-renderer_amd/build.py` "
                              "(hipcc, gfx950). There is no CPU fallback.")
        handle = C.CDLL(path)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def _check(status: int):
    if status != OK:
        raise VkrError(status, lib().vkr_last_error().decode("utf-8", "replace"))


def _np_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


# --------------------------------------------------------------------------------------- host math

def perspective(fovy: float, aspect: float, z_near: float, z_far: float) -> np.ndarray:
    m = np.zeros(16, np.float32)
    lib().vkr_perspective(fovy, aspect, z_near, z_far, _np_ptr(m))
    return m


def look_at(eye, center, up) -> np.ndarray:
    e, c, u = (np.ascontiguousarray(v, np.float32) for v in (eye, center, up))
    m = np.zeros(16, np.float32)
    lib().vkr_look_at(_np_ptr(e), _np_ptr(c), _np_ptr(u), _np_ptr(m))
    return m


def mat4_mul(a, b) -> np.ndarray:
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    m = np.zeros(16, np.float32)
    lib().vkr_mat4_mul(_np_ptr(a), _np_ptr(b), _np_ptr(m))
    return m


def mat4_inverse(a) -> np.ndarray:
    a = np.ascontiguousarray(a, np.float32)
    m = np.zeros(16, np.float32)
    _check(lib().vkr_mat4_inverse(_np_ptr(a), _np_ptr(m)))
    return m


def frustum_planes(view_proj) -> np.ndarray:
    vp = np.ascontiguousarray(view_proj, np.float32)
    p = np.zeros(24, np.float32)
    lib().vkr_frustum_planes(_np_ptr(vp), _np_ptr(p))
    return p.reshape(6, 4)


def deferred_push(camera_pos, inverse_vp, point_light_radius: float = 1.0) -> np.ndarray:
    """The 80-byte DeferredPushData block (/root/reference/src/main.cpp:76-80,209)."""
    push = np.zeros(20, np.float32)
    push[0:3] = camera_pos
    push[3] = point_light_radius
    push[4:20] = np.asarray(inverse_vp, np.float32).reshape(16)
    return push


# --------------------------------------------------------------------------------------- objects

class Buffer:
    """hipMalloc'd (or pinned, host-visible) memory; counterpart of vw::Buffer."""

    def __init__(self, ctx: "Context", nbytes: int, flags: int = BUFFER_DEVICE, _wrap_ptr: int | None = None):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        h = C.c_void_p()
        if _wrap_ptr is None:
            _check(lib().vkr_buffer_create(ctx.handle, self.nbytes, flags, C.byref(h)))
        else:
            _check(lib().vkr_buffer_wrap(ctx.handle, C.c_void_p(_wrap_ptr), self.nbytes, C.byref(h)))
        self.handle = h
        p = C.c_void_p()
        _check(lib().vkr_buffer_device_ptr(self.handle, C.byref(p)))
        self.ptr = p.value or 0

    def upload(self, array: np.ndarray, offset: int = 0) -> "Buffer":
        a = np.ascontiguousarray(array)
        _check(lib().vkr_buffer_upload(self.handle, offset, _np_ptr(a), a.nbytes))
        self.ctx.wait_idle()  # the numpy temporary may die right after this call
        return self

    def download(self, dtype, count: int | None = None, offset: int = 0) -> np.ndarray:
        dtype = np.dtype(dtype)
        if count is None:
            count = (self.nbytes - offset) // dtype.itemsize
        out = np.empty(count, dtype)
        _check(lib().vkr_buffer_download(self.handle, offset, _np_ptr(out), out.nbytes))
        return out

    def fill_u32(self, value: int, offset: int = 0, nbytes: int | None = None):
        _check(lib().vkr_buffer_fill_u32(self.handle, offset, value, self.nbytes - offset if nbytes is None else nbytes))

    def mapped(self) -> np.ndarray:
        p = C.c_void_p()
        _check(lib().vkr_buffer_map(self.handle, C.byref(p)))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(self.nbytes,))

    def destroy(self):
        if self.handle:
            lib().vkr_buffer_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def align_up(v: int, a: int) -> int:
    return (v + a - 1) // a * a


class Image:
    """A linear, row-major 2D image in a Buffer (counterpart of vw::Image for the G-buffer planes and
    the output).  Rows are padded to whole 4-texel groups and 16-byte alignment."""

    def __init__(self, ctx: "Context", fmt: int, width: int, height: int, pitch_bytes: int | None = None):
        self.format, self.width, self.height = fmt, int(width), int(height)
        texel = _TEXEL_BYTES[fmt]
        self.texel_bytes = texel
        self.pitch = int(pitch_bytes) if pitch_bytes is not None else align_up(align_up(self.width, 4) * texel, 16)
        self.buffer = Buffer(ctx, max(self.pitch * self.height, 16))

    def upload(self, array: np.ndarray) -> "Image":
        """array: (H, W, ...) with W*itemsize-per-texel == texel size; copied into the pitched layout."""
        a = np.ascontiguousarray(array)
        row_bytes = self.width * self.texel_bytes
        assert a.nbytes == row_bytes * self.height, (a.shape, a.dtype, self.width, self.height)
        staged = np.zeros((self.height, self.pitch), np.uint8)
        staged[:, :row_bytes] = a.view(np.uint8).reshape(self.height, row_bytes)
        self.buffer.upload(staged)
        return self

    def download(self, dtype, channels: int) -> np.ndarray:
        raw = self.buffer.download(np.uint8, self.pitch * self.height).reshape(self.height, self.pitch)
        row_bytes = self.width * self.texel_bytes
        out = np.ascontiguousarray(raw[:, :row_bytes]).view(dtype)
        return out.reshape(self.height, self.width, channels) if channels > 1 else out.reshape(self.height, self.width)

    def desc(self, row0: int = 0, rows: int | None = None) -> Image2D:
        rows = self.height - row0 if rows is None else rows
        return Image2D(self.buffer.ptr + row0 * self.pitch, self.width, rows, self.pitch, self.format)


class Fence:
    def __init__(self, handle):
        self.handle = handle

    def signaled(self) -> bool:
        st = lib().vkr_fence_status(self.handle)
        if st == OK:
            return True
        if st == ERR_NOT_READY:
            return False
        _check(st)
        return False

    def wait(self, timeout_ns: int = 2**64 - 1) -> bool:
        st = lib().vkr_fence_wait(self.handle, timeout_ns)
        if st == ERR_NOT_READY:
            return False
        _check(st)
        return True

    def elapsed_ms_since(self, begin: "Fence") -> float:
        ms = C.c_float()
        _check(lib().vkr_fence_elapsed_ms(begin.handle, self.handle, C.byref(ms)))
        return float(ms.value)

    def __del__(self):
        try:
            if self.handle:
                lib().vkr_fence_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class Context:
    """One GPU + one HIP stream (counterpart of vw::Device + its compute vw::Queue)."""

    def __init__(self, device_ordinal: int = 0):
        h = C.c_void_p()
        _check(lib().vkr_context_create(device_ordinal, C.byref(h)))
        self.handle = h
        self.device_ordinal = device_ordinal

    def device_info(self) -> dict:
        buf = C.create_string_buffer(256)
        cu, mhz = C.c_int32(), C.c_int32()
        _check(lib().vkr_context_device_info(self.handle, buf, 256, C.byref(cu), C.byref(mhz)))
        return {"name": buf.value.decode(), "cu_count": cu.value, "clock_mhz": mhz.value}

    def self_test(self, samples: int = 1 << 22) -> int:
        """Number of failed device self-checks (0 = all good)."""
        n = C.c_uint32()
        _check(lib().vkr_context_self_test(self.handle, samples, C.byref(n)))
        return int(n.value)

    def stream(self) -> int:
        p = C.c_void_p()
        _check(lib().vkr_context_stream(self.handle, C.byref(p)))
        return p.value or 0

    def buffer(self, nbytes: int, flags: int = BUFFER_DEVICE) -> Buffer:
        return Buffer(self, nbytes, flags)

    def buffer_from(self, array: np.ndarray) -> Buffer:
        a = np.ascontiguousarray(array)
        return Buffer(self, max(a.nbytes, 16)).upload(a) if a.nbytes else Buffer(self, 16)

    def wrap(self, device_ptr: int, nbytes: int) -> Buffer:
        return Buffer(self, nbytes, _wrap_ptr=device_ptr)

    def image(self, fmt: int, width: int, height: int, pitch_bytes: int | None = None) -> Image:
        return Image(self, fmt, width, height, pitch_bytes)

    def marker(self) -> Fence:
        h = C.c_void_p()
        _check(lib().vkr_queue_marker(self.handle, C.byref(h)))
        return Fence(h)

    def wait_idle(self):
        _check(lib().vkr_queue_wait_idle(self.handle))

    # --- typed entry points -----------------------------------------------------------------
    def deferred_resolve(self, args: DeferredArgs):
        _check(lib().vkr_deferred_resolve(self.handle, C.byref(args)))

    def tile_stats(self) -> dict:
        """Light-list statistics of the most recent radius-cutoff resolve (waits for the stream)."""
        st = TileStats()
        _check(lib().vkr_deferred_tile_stats(self.handle, C.byref(st)))
        d = {name: int(getattr(st, name)) for name, _ in TileStats._fields_}
        d["mean_lights_per_pixel"] = d["light_pixels"] / max(d["pixels"], 1)
        return d

    def prep_indirect(self, args: CullArgs):
        _check(lib().vkr_prep_indirect(self.handle, C.byref(args)))

    def prep_indirect_reset(self, args: CullArgs):
        """cmd_out := the templates with zero counts: what a cull with CULL_CMDS_PREZEROED starts from."""
        _check(lib().vkr_prep_indirect_reset(self.handle, C.byref(args)))

    def prep_indirect_noalloc(self, args: CullArgs):
        """The deterministic cull with its state in the library's static per-GPU array: no allocation inside the call."""
        _check(lib().vkr_prep_indirect_noalloc(self.handle, C.byref(args)))

    def destroy(self):
        if self.handle:
            lib().vkr_context_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class GBuffer:
    """The four G-buffer planes of /root/reference/src/main.cpp:111-117 as linear device images."""

    def __init__(self, ctx: Context, width: int, height: int):
        self.width, self.height = width, height
        self.albedo = ctx.image(FORMAT_R8G8B8A8_UNORM, width, height)
        self.material = ctx.image(FORMAT_R8G8B8A8_UNORM, width, height)
        self.normal = ctx.image(FORMAT_R16G16B16A16_SFLOAT, width, height)
        self.depth = ctx.image(FORMAT_D32_SFLOAT, width, height)

    def upload(self, host: dict) -> "GBuffer":
        self.albedo.upload(host["albedo"])
        self.material.upload(host["material"])
        self.normal.upload(host["normal"])
        self.depth.upload(host["depth"])
        return self


def make_deferred_args(gbuf: GBuffer, out: Image, lights: Buffer, light_count: int, push: np.ndarray, *, flags: int = 0,
                       row0: int = 0, rows: int | None = None) -> DeferredArgs:
    """Argument block for rows [row0, row0+rows) of the frame (whole frame by default)."""
    a = DeferredArgs()
    a.albedo, a.material = gbuf.albedo.desc(row0, rows), gbuf.material.desc(row0, rows)
    a.normal, a.depth = gbuf.normal.desc(row0, rows), gbuf.depth.desc(row0, rows)
    a.out = out.desc(row0, rows)
    a.lights = lights.ptr
    a.light_count = light_count
    a.push = (C.c_float * 20)(*np.asarray(push, np.float32).reshape(20).tolist())
    a.row_offset = row0
    a.full_height = gbuf.height
    a.flags = flags
    return a


class CullScene:
    """Device-resident inputs and outputs of one cull call."""

    def __init__(self, ctx: Context, aabb_min: np.ndarray, aabb_max: np.ndarray, mesh_base: np.ndarray,
                 cmd_template: np.ndarray, *, first_mesh: int = 0, first_instance: int = 0):
        self.ctx = ctx
        self.n = int(np.asarray(aabb_min).reshape(-1, 4).shape[0])
        self.m = int(np.asarray(mesh_base).shape[0])
        self.first_mesh, self.first_instance = first_mesh, first_instance
        self.aabb_min = ctx.buffer_from(np.asarray(aabb_min, np.float32))
        self.aabb_max = ctx.buffer_from(np.asarray(aabb_max, np.float32))
        self.mesh_base = ctx.buffer_from(np.asarray(mesh_base, np.uint32))
        self.cmd_template = ctx.buffer_from(np.asarray(cmd_template, DRAW_CMD_DTYPE))
        self.cmd_out = ctx.buffer(max(self.m * 20, 16))
        self.visible = ctx.buffer(max(self.n * 4, 16))
        self.error_flag = ctx.buffer(16)
        self.error_flag.fill_u32(0)

    def args(self, planes: np.ndarray, flags: int = 0) -> CullArgs:
        a = CullArgs()
        a.aabb_min, a.aabb_max, a.instance_count = self.aabb_min.ptr, self.aabb_max.ptr, self.n
        a.planes = (C.c_float * 24)(*np.asarray(planes, np.float32).reshape(24).tolist())
        a.mesh_base, a.mesh_count = self.mesh_base.ptr, self.m
        a.first_mesh, a.first_instance, a.flags = self.first_mesh, self.first_instance, flags
        a.cmd_template, a.cmd_out, a.visible = self.cmd_template.ptr, self.cmd_out.ptr, self.visible.ptr
        a.error_flag = self.error_flag.ptr
        return a

    def run(self, planes: np.ndarray, flags: int = 0):
        self.ctx.prep_indirect(self.args(planes, flags))

    def results(self) -> dict:
        cmds = self.cmd_out.download(DRAW_CMD_DTYPE, self.m) if self.m else np.zeros(0, DRAW_CMD_DTYPE)
        vis = self.visible.download(np.uint32, self.n) if self.n else np.zeros(0, np.uint32)
        err = int(self.error_flag.download(np.uint32, 1)[0])
        return {"cmds": cmds, "visible": vis, "error": err}
