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

    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    for k in manifest:
        print(k, os.path.getsize(os.path.join(HERE, k)))


if __name__ == "__main__":
    main()
