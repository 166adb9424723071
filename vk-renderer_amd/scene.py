"""Scene tables: the step before the cull pass.

Host side mirrors what ``vw::Scene::Scene`` builds after import
(/root/reference/src/vkmodel.cpp:55-111): per-mesh MeshInfo, template draw commands, PerMeshData and the
contiguous model-matrix array.  Device side turns vertex positions and those matrices into the per-instance
world AABBs ``vkr_prep_indirect`` reads (/root/reference/src/vkmodel.cpp:166-174 is the only hint of the intended
bounds in the reference).  Everything goes through the C ABI.
"""
from __future__ import annotations

import ctypes as C
import importlib

import numpy as np

vkr = importlib.import_module("vk-renderer_amd")


class SceneDescription:
    """Flat arrays standing in for the aiScene fields the reference reads.

    nodes: list of (local 4x4 as aiMatrix4x4 rows a1..d4, [child node indices], [mesh indices]); node 0 = root.
    meshes: list of (vertex_count, face_count, material_index, flags)."""

    def __init__(self, nodes, meshes):
        self.node_transform = np.ascontiguousarray([np.asarray(n[0], np.float32).reshape(16) for n in nodes], np.float32).reshape(-1, 16)
        child_begin, children, mesh_begin, node_meshes = [0], [], [0], []
        for _, ch, ms in nodes:
            children += list(ch)
            child_begin.append(len(children))
            node_meshes += list(ms)
            mesh_begin.append(len(node_meshes))
        self.node_child_begin = np.asarray(child_begin, np.uint32)
        self.node_children = np.asarray(children, np.uint32)
        self.node_mesh_begin = np.asarray(mesh_begin, np.uint32)
        self.node_meshes = np.asarray(node_meshes, np.uint32)
        m = np.asarray(meshes, np.uint32).reshape(-1, 4)
        self.mesh_vertex_count = np.ascontiguousarray(m[:, 0])
        self.mesh_face_count = np.ascontiguousarray(m[:, 1])
        self.mesh_material = np.ascontiguousarray(m[:, 2])
        self.mesh_flags = np.ascontiguousarray(m[:, 3])

    def c_desc(self) -> "vkr.SceneDesc":
        d = vkr.SceneDesc()
        d.node_count = self.node_transform.shape[0]
        d.node_transform = self.node_transform.ctypes.data
        d.node_child_begin = self.node_child_begin.ctypes.data
        d.node_children = self.node_children.ctypes.data if self.node_children.size else None
        d.node_mesh_begin = self.node_mesh_begin.ctypes.data
        d.node_meshes = self.node_meshes.ctypes.data if self.node_meshes.size else None
        d.mesh_count = self.mesh_vertex_count.shape[0]
        d.mesh_vertex_count = self.mesh_vertex_count.ctypes.data
        d.mesh_face_count = self.mesh_face_count.ctypes.data
        d.mesh_material = self.mesh_material.ctypes.data
        d.mesh_flags = self.mesh_flags.ctypes.data
        return d


def build_tables(desc: SceneDescription) -> dict:
    """vkr_scene_count + vkr_scene_build_tables.  Returns numpy tables plus the counts."""
    d = desc.c_desc()
    counts = vkr.SceneCounts()
    vkr._check(vkr.lib().vkr_scene_count(C.byref(d), C.byref(counts)))
    k, n = counts.kept_meshes, counts.kept_instances
    out = {
        "mesh_info": np.zeros(k, vkr.MESH_INFO_DTYPE),
        "draw_commands": np.zeros(k, vkr.DRAW_CMD_DTYPE),
        "per_mesh_data": np.zeros(k, vkr.PER_MESH_DATA_DTYPE),
        "mesh_matrices": np.zeros((n, 16), np.float32),
        "mesh_base": np.zeros(k + 1, np.uint32),
        "vertex_base": np.zeros(k + 1, np.uint32),
        "source_mesh": np.zeros(k, np.uint32),
    }
    t = vkr.SceneTables()
    for name in out:
        setattr(t, name, out[name].ctypes.data)
    vkr._check(vkr.lib().vkr_scene_build_tables(C.byref(d), C.byref(counts), C.byref(t)))
    out["counts"] = {f: getattr(counts, f) for f, _ in vkr.SceneCounts._fields_}
    return out


class DeviceScene:
    """Tables resident on one GPU + the bounds kernels; `cull_scene()` hands them to the cull pass."""

    def __init__(self, ctx: "vkr.Context", tables: dict, positions: np.ndarray):
        self.ctx = ctx
        self.tables = tables
        self.mesh_count = int(tables["counts"]["kept_meshes"])
        self.instance_count = int(tables["counts"]["kept_instances"])
        self.positions = ctx.buffer_from(np.ascontiguousarray(positions, np.float32).reshape(-1, 3))
        self.vertex_base = ctx.buffer_from(tables["vertex_base"])
        self.mesh_base = ctx.buffer_from(tables["mesh_base"])
        self.mesh_matrices = ctx.buffer_from(tables["mesh_matrices"])
        self.mesh_aabb = ctx.buffer(max(self.mesh_count * 32, 16))
        self.aabb_min = ctx.buffer(max(self.instance_count * 16, 16))
        self.aabb_max = ctx.buffer(max(self.instance_count * 16, 16))

    def compute_mesh_aabbs(self):
        vkr._check(vkr.lib().vkr_mesh_aabbs(self.ctx.handle, self.positions.ptr, self.vertex_base.ptr, self.mesh_count, self.mesh_aabb.ptr))

    def compute_instance_aabbs(self):
        vkr._check(vkr.lib().vkr_instance_aabbs(self.ctx.handle, self.mesh_matrices.ptr, self.mesh_base.ptr, self.mesh_aabb.ptr,
                                                self.mesh_count, self.instance_count, self.aabb_min.ptr, self.aabb_max.ptr))

    def download(self) -> dict:
        self.ctx.wait_idle()
        return {"mesh_aabb": self.mesh_aabb.download(np.float32, self.mesh_count * 8).reshape(-1, 8),
                "aabb_min": self.aabb_min.download(np.float32, self.instance_count * 4).reshape(-1, 4),
                "aabb_max": self.aabb_max.download(np.float32, self.instance_count * 4).reshape(-1, 4)}
