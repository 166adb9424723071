"""Regenerates the golden fixtures from the synthetic generators and the CPU oracle.

    python tests/golden/make_golden.py

The reference holds no golden vectors for either pass (it has no tests at all), so these pin the
build's own oracle: inputs + expected outputs, small enough to commit.  A fixture is data only.
"""
import hashlib
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402

synth = importlib.import_module("vk-renderer_amd.synth")


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    oracle.build()
    manifest = {}

    for (w, h, L, idx) in ((64, 36, 13, 0), (32, 18, 32, 1), (32, 18, 256, 2)):
        case = synth.deferred_case(w, h, L, config_index=idx)
        o32 = oracle.deferred(case["gbuf"], case["lights"], case["push"], mode="f32")
        omx = oracle.deferred(case["gbuf"], case["lights"], case["push"], mode="mix", want=("f64", "rgba8"))
        name = f"deferred_{w}x{h}_L{L}.npz"
        arrays = dict(albedo=case["gbuf"]["albedo"], material=case["gbuf"]["material"], normal=case["gbuf"]["normal"],
                      depth=case["gbuf"]["depth"], lights=case["lights"], push=case["push"],
                      out_f32=o32["f32"], out_f32_rgba8=o32["rgba8"], out_mix_f64=omx["f64"], out_mix_rgba8=omx["rgba8"],
                      near_threshold=omx["near_threshold"])
        np.savez_compressed(os.path.join(HERE, name), **arrays)
        manifest[name] = {k: sha(v) for k, v in arrays.items()}

    cam = synth.reference_camera(3840, 2160)
    scene = synth.make_cull_scene(4096, 16, synth.SEED_CULL, cam)
    res = oracle.cull(scene["aabb_min"], scene["aabb_max"], cam["planes"], scene["mesh_base"], scene["cmd_template"])
    arrays = dict(aabb_min=scene["aabb_min"], aabb_max=scene["aabb_max"], planes=cam["planes"], mesh_base=scene["mesh_base"],
                  cmd_template=scene["cmd_template"], cmds=res["cmds"], visible=res["visible"], mask=res["mask"])
    np.savez_compressed(os.path.join(HERE, "cull_4096_M16.npz"), **arrays)
    manifest["cull_4096_M16.npz"] = {k: sha(v) for k, v in arrays.items()}

    adv = synth.make_adversarial_cull(65536, synth.SEED_CULL + 99, cam, 16)
    res = oracle.cull(adv["aabb_min"], adv["aabb_max"], cam["planes"], adv["mesh_base"], adv["cmd_template"])
    # inputs are regenerated from the seed in the test; only the expected outputs are stored
    arrays = dict(planes=cam["planes"], cmds=res["cmds"], mask=np.packbits(res["mask"]),
                  in_sha=np.frombuffer(bytes.fromhex(sha(adv["aabb_min"]) + sha(adv["aabb_max"])), np.uint8))
    np.savez_compressed(os.path.join(HERE, "cull_adversarial_65536.npz"), **arrays)
    manifest["cull_adversarial_65536.npz"] = {k: sha(v) for k, v in arrays.items()}

    # ---- callers either side of the passes (scene_oracle.c) ------------------------------------------
    frame = importlib.import_module("vk-renderer_amd.frame")
    vkr = importlib.import_module("vk-renderer_amd")
    tree = synth.make_scene_tree(40, 9, synth.SEED_SCENE)
    t = oracle.scene_tables(tree["nodes"], [(v, f, m, 1 if fl == 7 else 0) for v, f, m, fl in tree["meshes"]])
    positions = synth.kept_positions(tree, t["source_mesh"])
    mesh = oracle.mesh_aabbs(positions, t["vertex_base"])
    mn, mx = oracle.instance_aabbs(t["mesh_matrices"], t["mesh_base"], mesh)
    arrays = dict(node_transform=np.stack([n[0] for n in tree["nodes"]]), node_children=np.array([len(n[1]) for n in tree["nodes"]], np.uint32),
                  positions=positions, mesh_info=t["mesh_info"], draw_commands=t["draw_commands"], per_mesh_data=t["per_mesh_data"],
                  mesh_matrices=t["mesh_matrices"], mesh_base=t["mesh_base"], vertex_base=t["vertex_base"], source_mesh=t["source_mesh"],
                  mesh_aabb=mesh, aabb_min=mn, aabb_max=mx)
    np.savez_compressed(os.path.join(HERE, "scene_40nodes_9meshes.npz"), **arrays)
    manifest["scene_40nodes_9meshes.npz"] = {k: sha(v) for k, v in arrays.items()}

    w, h = 96, 54
    cam = synth.reference_camera(w, h)
    spheres = synth.make_spheres(12, synth.SEED_SCENE + 1)
    sc = frame.make_gbuffer_scene(cam["vp"], cam["inv_vp"], cam["pos"], 12)
    g = oracle.gbuffer_fill(bytes(sc), spheres, w, h)
    arrays = dict(scene=np.frombuffer(bytes(sc), np.uint8), spheres=spheres.view(np.uint8), albedo=g["albedo"], material=g["material"],
                  normal=g["normal"], depth=g["depth"])
    np.savez_compressed(os.path.join(HERE, "gbuffer_fill_96x54_12spheres.npz"), **arrays)
    manifest["gbuffer_fill_96x54_12spheres.npz"] = {k: sha(v) for k, v in arrays.items()}

    text = synth.fscene_text(9, synth.SEED_SCENE + 2)
    with open(os.path.join(HERE, "lights_9.fscene"), "w") as f:
        f.write(text)
    recs = oracle.fscene_point_lights(text)
    ocam = oracle.Camera.default()
    states = []
    for _ in range(16):
        ocam.key(0, 2.5)
        ocam.cursor_delta(np.float32(1.0 / 60.0), 0.0)
        states.append(np.concatenate([ocam.state()[:6], ocam.view()]))
    arrays = dict(light_records=recs, camera_path=np.stack(states).astype(np.float32))
    np.savez_compressed(os.path.join(HERE, "lights_camera.npz"), **arrays)
    manifest["lights_camera.npz"] = {k: sha(v) for k, v in arrays.items()}
    manifest["lights_9.fscene"] = {"text": hashlib.sha256(text.encode()).hexdigest()}

    # fixtures made by other scripts (make_shader_eval.py) keep their entries
    path = os.path.join(HERE, "MANIFEST.json")
    if os.path.exists(path):
        for name, entry in json.load(open(path)).items():
            if name not in manifest and os.path.exists(os.path.join(HERE, name)):
                manifest[name] = entry
    with open(path, "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    for k in manifest:
        print(k, os.path.getsize(os.path.join(HERE, k)))
    del vkr


if __name__ == "__main__":
    main()
