"""The caller either side of the two passes: camera, per-frame constants, light table, G-buffer
producers and the frame loop of /root/reference/src/main.cpp:198-231 (at most `frames_in_flight` frames
submitted and not yet finished, the back-pressure of vw::Queue::hasReadyBuffer,
/root/reference/inc/vkcore.hpp:132-152).  Everything goes through the C ABI.
"""
from __future__ import annotations

import ctypes as C
import importlib
import math

import numpy as np

vkr = importlib.import_module("vk-renderer_amd")


class Camera:
    """CameraInputHandler (/root/reference/inc/vkcamera.hpp) without GLFW."""

    def __init__(self):
        self.c = vkr.Camera()
        vkr.lib().vkr_camera_default(C.byref(self.c))

    @property
    def pos(self):
        return np.array(self.c.pos[:], np.float32)

    @property
    def forward(self):
        return np.array(self.c.forward[:], np.float32)

    def view(self) -> np.ndarray:
        m = np.zeros(16, np.float32)
        vkr.lib().vkr_camera_view(C.byref(self.c), m.ctypes.data)
        return m

    def key(self, key: int, delta_time: float = 1.0):
        vkr.lib().vkr_camera_key(C.byref(self.c), key, delta_time)

    def cursor_delta(self, dx: float, dy: float):
        vkr.lib().vkr_camera_cursor_delta(C.byref(self.c), dx, dy)

    def step_path(self, forward_units: float = 0.5, yaw_degrees: float = 1.0):
        """One frame of the camera-path workload: move along +forward, then yaw."""
        self.key(vkr.KEY_W, forward_units / 0.2)
        self.cursor_delta(yaw_degrees / 60.0, 0.0)


def reference_projection(width: int, height: int) -> np.ndarray:
    """perspective(radians(70), aspect, 0.1, 10000) of /root/reference/src/main.cpp:95, with the true aspect
    ratio (the reference's integer division yields 1.0 for its 1000x800 window)."""
    return vkr.perspective(math.radians(70.0), float(width) / float(height), 0.1, 10000.0)


def frame_setup(camera: Camera, proj: np.ndarray) -> dict:
    fc = vkr.FrameConstants()
    p = np.ascontiguousarray(proj, np.float32)
    vkr._check(vkr.lib().vkr_frame_setup(C.byref(camera.c), p.ctypes.data, C.byref(fc)))
    return {"view": np.array(fc.view[:], np.float32), "view_proj": np.array(fc.view_proj[:], np.float32),
            "inv_view_proj": np.array(fc.inv_view_proj[:], np.float32), "planes": np.array(fc.planes[:], np.float32).reshape(6, 4),
            "deferred_push": np.array(fc.deferred_push[:], np.float32)}


def pack_point_lights(pos, intensity, direction, opening_angle_deg, penumbra_angle_deg) -> np.ndarray:
    pos, intensity, direction = (np.ascontiguousarray(a, np.float32).reshape(-1, 3) for a in (pos, intensity, direction))
    oa, pa = (np.ascontiguousarray(a, np.float32).reshape(-1) for a in (opening_angle_deg, penumbra_angle_deg))
    out = np.zeros(pos.shape[0], vkr.LIGHT_INFO_DTYPE)
    vkr.lib().vkr_pack_point_lights(pos.shape[0], pos.ctypes.data, intensity.ctypes.data, direction.ctypes.data, oa.ctypes.data, pa.ctypes.data,
                                    out.ctypes.data)
    return out


def fscene_point_lights(text: str | bytes) -> np.ndarray:
    """Point lights of a .fscene file's text as 48-byte records (/root/reference/src/main.cpp:48-70)."""
    data = text.encode("utf-8") if isinstance(text, str) else bytes(text)
    n = C.c_uint32()
    vkr._check(vkr.lib().vkr_fscene_point_lights(data, len(data), None, 0, C.byref(n)))
    out = np.zeros(n.value, vkr.LIGHT_INFO_DTYPE)
    if n.value:
        vkr._check(vkr.lib().vkr_fscene_point_lights(data, len(data), out.ctypes.data, n.value, C.byref(n)))
    return out


def make_gbuffer_scene(view_proj, inv_view_proj, camera_pos, sphere_count: int, *, ground_plane=(0.0, 1.0, 0.0, 0.0),
                       ground_albedo_a=(0.8, 0.8, 0.8), ground_albedo_b=(0.2, 0.2, 0.25), ground_tile: float = 4.0,
                       ground_roughness: float = 0.6, ground_metalness: float = 0.0) -> "vkr.GBufferScene":
    s = vkr.GBufferScene()
    s.view_proj = (C.c_float * 16)(*np.asarray(view_proj, np.float32).reshape(16).tolist())
    s.inv_view_proj = (C.c_float * 16)(*np.asarray(inv_view_proj, np.float32).reshape(16).tolist())
    s.camera_pos = (C.c_float * 3)(*np.asarray(camera_pos, np.float32).reshape(3).tolist())
    s.sphere_count = sphere_count
    s.ground_plane = (C.c_float * 4)(*ground_plane)
    s.ground_albedo_a = (C.c_float * 3)(*ground_albedo_a)
    s.ground_albedo_b = (C.c_float * 3)(*ground_albedo_b)
    s.ground_tile = ground_tile
    s.ground_roughness = ground_roughness
    s.ground_metalness = ground_metalness
    return s


def gbuffer_fill(ctx: "vkr.Context", gb: "vkr.GBuffer", spheres: "vkr.Buffer", scene: "vkr.GBufferScene", *, row0: int = 0, rows: int | None = None):
    a = vkr.GBufferFillArgs()
    a.albedo, a.material = gb.albedo.desc(row0, rows), gb.material.desc(row0, rows)
    a.normal, a.depth = gb.normal.desc(row0, rows), gb.depth.desc(row0, rows)
    a.spheres = spheres.ptr if spheres is not None else None
    a.scene = scene
    a.row_offset = row0
    a.full_height = gb.height
    vkr._check(vkr.lib().vkr_gbuffer_fill(ctx.handle, C.byref(a)))


def gbuffer_encode(ctx: "vkr.Context", gb: "vkr.GBuffer", albedo_f32: "vkr.Buffer", material_f32: "vkr.Buffer", normal_f32: "vkr.Buffer",
                   depth_f32: "vkr.Buffer"):
    a = vkr.GBufferEncodeArgs()
    a.albedo_f32, a.material_f32, a.normal_f32, a.depth_f32 = albedo_f32.ptr, material_f32.ptr, normal_f32.ptr, depth_f32.ptr
    a.albedo, a.material, a.normal, a.depth = gb.albedo.desc(), gb.material.desc(), gb.normal.desc(), gb.depth.desc()
    vkr._check(vkr.lib().vkr_gbuffer_encode(ctx.handle, C.byref(a)))


def gbuffer_encode_surfaces(ctx: "vkr.Context", gb: "vkr.GBuffer", albedo_texel: "vkr.Buffer", specular_texel: "vkr.Buffer",
                            normal_texel: "vkr.Buffer", normal_obj: "vkr.Buffer", tangent_obj: "vkr.Buffer", depth_f32: "vkr.Buffer",
                            model_matrices: "vkr.Buffer", matrix_count: int):
    """offscreen.vert:28-35 + offscreen.frag:13-16 per pixel into the four attachments (vkr_gbuffer_encode_surfaces)."""
    a = vkr.GBufferSurfaceArgs()
    a.albedo_texel, a.specular_texel, a.normal_texel = albedo_texel.ptr, specular_texel.ptr, normal_texel.ptr
    a.normal_obj, a.tangent_obj, a.depth_f32 = normal_obj.ptr, tangent_obj.ptr, depth_f32.ptr
    a.model_matrices, a.matrix_count = model_matrices.ptr, matrix_count
    a.albedo, a.material, a.normal, a.depth = gb.albedo.desc(), gb.material.desc(), gb.normal.desc(), gb.depth.desc()
    vkr._check(vkr.lib().vkr_gbuffer_encode_surfaces(ctx.handle, C.byref(a)))


class FrameLoop:
    """cull -> G-buffer fill -> deferred resolve per frame, `frames_in_flight` output images in rotation.

    Like the reference's loop, a frame is only recorded when one of the in-flight slots has finished
    (`has_ready_slot`); `run_path` blocks on the oldest fence instead of spinning."""

    def __init__(self, ctx: "vkr.Context", width: int, height: int, lights: np.ndarray, spheres: np.ndarray, cull_scene: "vkr.CullScene | None" = None,
                 *, frames_in_flight: int = 3, cull_flags: int = 0, deferred_flags: int = 0, point_light_radius: float = 1.0):
        self.ctx, self.width, self.height = ctx, width, height
        self.proj = reference_projection(width, height)
        self.gbuf = vkr.GBuffer(ctx, width, height)
        self.outputs = [ctx.image(vkr.FORMAT_R8G8B8A8_UNORM, width, height) for _ in range(frames_in_flight)]
        self.fences: list = [None] * frames_in_flight
        lights = np.ascontiguousarray(lights)
        self.light_count = lights.shape[0]
        # The reference rewrites the light table every frame into a host-visible UBO (main.cpp:109,202).  Here
        # each in-flight slot has a pinned staging copy and the kernel reads a device-resident table (one
        # stream-ordered 48*L-byte copy per frame) instead of every workgroup pulling it over PCIe.
        self.light_staging = [ctx.buffer(max(lights.nbytes, 16), vkr.BUFFER_HOST_VISIBLE) for _ in range(frames_in_flight)]
        self.light_ubo = ctx.buffer(max(lights.nbytes, 16))
        self.lights_host = lights.view(np.uint8).reshape(-1)
        self.spheres = ctx.buffer_from(np.ascontiguousarray(spheres, vkr.SPHERE_DTYPE))
        self.sphere_count = int(np.asarray(spheres).shape[0])
        self.cull_scene = cull_scene
        self.cull_flags, self.deferred_flags, self.radius = cull_flags, deferred_flags, point_light_radius
        self.submitted = 0

    def has_ready_slot(self) -> bool:
        f = self.fences[self.submitted % len(self.fences)]
        return f is None or f.signaled()

    def submit(self, camera: Camera) -> int:
        """Records and submits one frame; returns the output slot it renders into."""
        slot = self.submitted % len(self.fences)
        if self.fences[slot] is not None:
            self.fences[slot].wait()
        if self.lights_host.size:
            stage = self.light_staging[slot].mapped()
            stage[: self.lights_host.size] = self.lights_host  # ubo.copyToMapped(lightInfos)
            vkr._check(vkr.lib().vkr_buffer_upload(self.light_ubo.handle, 0, stage.ctypes.data, self.lights_host.size))
        fc = frame_setup(camera, self.proj)
        if self.cull_scene is not None:
            self.cull_scene.run(fc["planes"], self.cull_flags)
        scene = make_gbuffer_scene(fc["view_proj"], fc["inv_view_proj"], camera.pos, self.sphere_count)
        gbuffer_fill(self.ctx, self.gbuf, self.spheres, scene)
        push = fc["deferred_push"].copy()
        push[3] = self.radius
        self.ctx.deferred_resolve(vkr.make_deferred_args(self.gbuf, self.outputs[slot], self.light_ubo, self.light_count, push, flags=self.deferred_flags))
        self.fences[slot] = self.ctx.marker()
        self.submitted += 1
        return slot

    def run_path(self, camera: Camera, frames: int, forward_units: float = 0.5, yaw_degrees: float = 1.0, only=None):
        """Walks the camera over `frames` frames.  only: the frame indices this process renders (frame-batch
        sharding: every rank walks the same path and submits its own frames); None = all."""
        for f in range(frames):
            if only is None or f in only:
                self.submit(camera)
            camera.step_path(forward_units, yaw_degrees)
        self.ctx.wait_idle()
