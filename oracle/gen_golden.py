#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the C oracle (TEST INFRASTRUCTURE).

The reference has no tests, fixtures or golden vectors for this path and cannot be built
or run here (nightly Rust, un-vendored crates), so these vectors are produced by the
oracle itself: they pin the oracle against regressions / toolchain differences and give the
GPU tests committed expected outputs, but they are NOT outputs of the reference (parity
unpinned — see oracle/mip_oracle.h).

    python oracle/gen_golden.py          # rewrites tests/golden/
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from renderer_amd import scene  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def special_scene(n=513):
    """Hand-built edge cases followed by random filler."""
    s = scene.make_scene(3, n=n, all_visible=True)
    pos, rot, scl, mid = s["pos"], s["rot"], s["scale"], s["mesh_id"]
    m = s["meshes"]
    m["index_len"][5, 0] = 0  # a mesh whose LOD0 is empty: dropped by compaction when near
    r = np.float32(np.sqrt(0.5))
    cases = [
        # pos, rot[i,j,k,w], scale, mesh
        ((1, 2, 30), (0, 0, 0, 1), 1.0, 0),                 # identity: M = T
        ((0, 1, 40), (0, r, 0, r), 2.0, 1),                 # 90 degrees about +Y, scale 2
        ((30, 20, -40.1), (0, 0, 0, 1), 1.0, 2),            # a reference light position (main.rs:369)
        ((0.1, 17, 0.1), (0, 0, 0, 1), 1.0, 3),             # the other one (main.rs:378)
        ((0, 1, 50), (0, 0, 0, 1), 0.0, 4),                 # zero scale
        ((0, 1, 50), (0, 0, 0, 2), 1.0, 4),                 # non-unit quaternion (not renormalised)
        ((np.nan, 1, 50), (0, 0, 0, 1), 1.0, 4),            # NaN position => visible
        ((0, 1, 50), (np.inf, 0, 0, 1), 1.0, 4),            # inf in the quaternion
        ((0, 1, 50), (0, 0, 0, 1), np.inf, 4),              # inf scale
        ((0, 1, 50), (0, 0, 0, 1), np.nan, 4),              # NaN scale
        ((np.inf, 1, 50), (0, 0, 0, 1), 1.0, 4),            # inf position
        ((0, 1, 5), (0, 0, 0, 1), 1.0, 5),                  # near (LOD 0) with an empty LOD 0 => no command
        ((0, 1, 12), (0, 0, 0, 1), 1.0, 0),                 # distance exactly 10 from the camera => LOD 0
        ((0, 1, np.nextafter(np.float32(12), np.float32(13))), (0, 0, 0, 1), 1.0, 0),  # just beyond => LOD 1
        ((0, 1, -50), (0, 0, 0, 1), 1.0, 0),                # behind the camera: culled
        ((0, 1, 1e-42), (1e-20, 1e-20, 0, 1), 1e-38, 0),    # denormals
    ]
    for k, (p, q, sc, me) in enumerate(cases):
        pos[k] = p
        rot[k] = q
        scl[k] = sc
        mid[k] = me
    return s


def dump(name, s, **extra):
    r = oracle.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"], **extra)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        pos=s["pos"], rot=s["rot"], scale=s["scale"], mesh_id=s["mesh_id"], meshes=s["meshes"],
        planes=s["planes"], cam_pos=s["cam_pos"],
        first_instance_base=np.uint32(extra.get("first_instance_base", 0)),
        first_index_base=np.uint32(extra.get("first_index_base", 0)),
        model=r["model"], world_aabb=r["world_aabb"], visible_bitmap=r["visible_bitmap"],
        coarse_culled=r["coarse_culled"], draw_cmds=r["draw_cmds"],
        draw_count=np.uint32(r["draw_count"]), draw_index_total=np.uint32(r["draw_index_total"]),
    )
    print(f"{name}: n={s['n']} visible={int((1 - r['coarse_culled']).sum())} cmds={r['draw_count']}")


def dump_skinned(name, s):
    """Extension fixtures (tests/golden/ext/): the skinned frame, inputs + oracle outputs."""
    sk = s["skeleton"]
    r = oracle.run_skinned(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], sk, s["poses"], s["planes"], s["cam_pos"])
    np.savez_compressed(
        os.path.join(OUT, "ext", name + ".npz"),
        pos=s["pos"], rot=s["rot"], scale=s["scale"], mesh_id=s["mesh_id"], meshes=s["meshes"], planes=s["planes"], cam_pos=s["cam_pos"],
        parent=sk["parent"], inverse_bind=sk["inverse_bind"], joint_box=sk["joint_box"], poses=s["poses"],
        palette=r["palette"], local_box=r["local_box"], model=r["model"], world_aabb=r["world_aabb"],
        visible_bitmap=r["visible_bitmap"], draw_cmds=r["draw_cmds"], draw_count=np.uint32(r["draw_count"]),
        draw_index_total=np.uint32(r["draw_index_total"]),
    )
    print(f"ext/{name}: n={s['n']} cmds={r['draw_count']}")


def dump_lights(name, s, lights, first_instance_base):
    """Extension fixtures: the shadow pass's per-light draw lists (row f-4)."""
    lists = oracle.light_draw_lists(s["pos"], s["mesh_id"], s["meshes"], lights, first_instance_base=first_instance_base)
    np.savez_compressed(os.path.join(OUT, "ext", name + ".npz"), pos=s["pos"], rot=s["rot"], scale=s["scale"], mesh_id=s["mesh_id"],
                        meshes=s["meshes"], lights=lights, first_instance_base=np.uint32(first_instance_base), lists=lists)
    print(f"ext/{name}: n={s['n']} lights={len(lights)}")


def dump_wire(name, golden_name):
    """Extension fixture: the WIRE form (include/mi_instance_pipeline.h, MIP_OUT_WIRE) of a committed golden frame's draw
    list, from the numpy statement of the format in tests/cpu_pipeline.py — pins the byte layout the kernels speak."""
    sys.path.insert(0, os.path.join(os.path.dirname(OUT)))
    from cpu_pipeline import encode_wire, encode_wire_packed

    g = np.load(os.path.join(OUT, golden_name + ".npz"))
    cmds = g["draw_cmds"]
    inst = (cmds["firstInstance"] - np.uint32(g["first_instance_base"])).astype(np.int64)
    far = np.array([oracle.pick_lod(2, g["cam_pos"], g["pos"][i]) for i in inst], np.uint32)
    body = encode_wire(cmds, g["mesh_id"][inst], far)
    body_packed = encode_wire_packed(cmds, g["mesh_id"][inst], far, g["first_instance_base"], len(g["meshes"]))  # MIP_OUT_WIRE_PACKED
    np.savez_compressed(os.path.join(OUT, "ext", name + ".npz"), source=np.array(golden_name), body=body, body_packed=body_packed,
                        draw_count=np.uint32(len(cmds)), draw_index_total=g["draw_index_total"])
    print(f"ext/{name}: {len(cmds)} commands -> {body.nbytes} wire bytes ({body.nbytes / max(len(cmds), 1):.2f} B per command)")


def main():
    os.makedirs(os.path.join(OUT, "ext"), exist_ok=True)
    oracle.build()
    sk_scene = scene.make_skinned_scene(301)
    sk_scene["poses"][7, 3, 0] = np.nan      # a NaN joint translation
    sk_scene["poses"][9, :, 7:10] = 0.0      # zero joint scales
    sk_scene["pos"][11] = np.nan             # NaN instance position
    dump_skinned("skinned_301", sk_scene)
    lit = scene.make_scene(3, n=1001)
    lit["pos"][5] = np.nan
    dump_lights("lights_1001", lit, np.array([[30, 20, -40.1], [0.1, 17, 0.1], [0, 0, 0], lit["pos"][17]], np.float32), 4)
    dump("box_1024", scene.make_scene(1))  # BASELINE config 1, full outputs
    for n in (1, 63, 64, 65, 257):
        dump(f"mixed_{n}", scene.make_scene(3, n=n, all_visible=(n < 100)))
    dump("mixed_4097_bases", scene.make_scene(3, n=4097), first_instance_base=1000, first_index_base=0xFFFFF000)
    dump("helmet_2000", scene.make_scene(2, n=2000))
    dump("special_513", special_scene())
    np.save(os.path.join(OUT, "default_planes.npy"), scene.default_planes())
    dump_wire("wire_4097_bases", "mixed_4097_bases")


if __name__ == "__main__":
    main()
