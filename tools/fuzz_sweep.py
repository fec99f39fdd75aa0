#!/usr/bin/env python3
"""A longer fuzz sweep than the test suite runs: random scenes with special floating-point values through every
GPU path (instance frame, light lists, skinned frame, per-triangle stage) against the oracle.
usage: tools/fuzz_sweep.py [seeds] [first_seed]   — prints one line per 25 seeds, exits non-zero on a mismatch."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402  (checker only)
import renderer_amd as ra  # noqa: E402
from fuzz_scenes import SPECIAL, random_scene  # noqa: E402
from helpers import float_mismatches  # noqa: E402
from renderer_amd.pipeline import make_frame  # noqa: E402

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda", 0)


def same(a, b):
    return len(float_mismatches(np.asarray(a).reshape(np.asarray(b).shape), b)) == 0


def fail(seed, what):
    print(f"MISMATCH seed {seed}: {what}", flush=True)
    sys.exit(1)


def random_skeleton(rng, j):
    parent = np.array([-1] + [int(rng.integers(-1 if k % 5 == 4 else 0, k)) for k in range(1, j)], np.int32)
    ibm = np.tile(np.eye(4, dtype=np.float32).reshape(16), (j, 1))
    ibm[:, 12:15] = rng.uniform(-1, 1, (j, 3))
    ibm[:, [0, 5, 10]] = rng.uniform(0.8, 1.2, (j, 3))
    lo = rng.uniform(-1, 0, (j, 3)).astype(np.float32)
    box = np.concatenate([lo, lo + rng.uniform(-0.2, 1.0, (j, 3)).astype(np.float32)], axis=1)  # some boxes are empty
    if rng.random() < 0.15:  # a joint box with a special value: the kernel's separable box fold must step aside (SkinArgs.box_bound)
        box[rng.integers(0, j), rng.integers(0, 6)] = rng.choice(SPECIAL)
    if rng.random() < 0.1:   # huge boxes: products overflow in some poses only
        box *= np.float32(1e37)
    return dict(parent=parent, inverse_bind=ibm, joint_box=box)


done = 0
for seed in range(first, first + seeds):
    rng = np.random.default_rng(50_000 + seed)
    s = random_scene(rng, oracle, n_max=int(rng.choice([300, 3000, 40_000])), special_rate=float(rng.choice([0.0, 0.01, 0.05])))
    n = s["n"]
    fib, fxb = s["first_instance_base"], s["first_index_base"]
    want = oracle.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"],
                      first_instance_base=fib, first_index_base=fxb, threads=8)
    # round 3: a quarter of the seeds in ordered-tiles mode, half of those with the three wait-free launches forced onto every size
    ordered = seed % 4 == 1
    os.environ.pop("MIP_TUNE_THREE_PASS_MIN_TILES", None)
    if ordered and seed % 8 == 1:
        os.environ["MIP_TUNE_THREE_PASS_MIN_TILES"] = "0"
    # round 5: a third of the seeds with the frame kernel's first-mover instantiation for every launch (read at context creation)
    os.environ.pop("MIP_TUNE_FIRST_MOVER", None)
    if seed % 3 == 2:
        os.environ["MIP_TUNE_FIRST_MOVER"] = "always"
    with ra.InstancePipeline(max_instances=max(n, 1), max_meshes=64, frames_in_flight=int(rng.integers(1, 4)), ordered_tiles=ordered) as p:
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        got = p.run_host(s["planes"], s["cam_pos"], first_instance_base=fib, first_index_base=fxb)
        if n:  # the same frame in both wire forms, expanded by the merge kernel (one chunk) and by the numpy statement
            from cpu_pipeline import decode_wire, unpack_wire
            from renderer_amd.pipeline import SHARD_HEADER_BYTES
            from renderer_amd.sharded import chunk_stride_bytes
            for form in (True, "packed"):
                stride = chunk_stride_bytes(n, wire=form)
                chunk = torch.zeros(stride // 4, dtype=torch.int32, device=dev)
                merged = torch.full((n, 5), -1, dtype=torch.int32, device=dev)
                mc = torch.zeros(2, dtype=torch.int32, device=dev)
                torch.cuda.synchronize()
                p.run_device(make_frame(s["planes"], s["cam_pos"], first_instance_base=fib, first_index_base=fxb), draw_cmds=chunk.data_ptr() + SHARD_HEADER_BYTES,
                             draw_count=chunk.data_ptr(), draw_index_total=chunk.data_ptr() + 4, wire=form)
                host = chunk.cpu().numpy().view(np.uint32)
                p.merge_wire_lists(chunk.data_ptr(), 1, stride, merged.data_ptr(), mc.data_ptr(), chunk_capacity=n, packed=form == "packed")
                body = host[SHARD_HEADER_BYTES // 4:]
                if form == "packed":
                    body = unpack_wire(body, int(host[0]))
                if not (int(host[0]) == want["draw_count"] and int(host[1]) == want["draw_index_total"]
                        and decode_wire(body, int(host[0]), s["meshes"]).tobytes() == want["draw_cmds"].tobytes()
                        and merged[: int(host[0])].cpu().numpy().tobytes() == want["draw_cmds"].tobytes()):
                    fail(seed, f"wire form {form} n={n} ordered={ordered}")
        if not (np.array_equal(got["visible_bitmap"], want["visible_bitmap"]) and got["draw_count"] == want["draw_count"]
                and got["draw_cmds"].tobytes() == want["draw_cmds"].tobytes() and got["draw_index_total"] == want["draw_index_total"]
                and same(got["model"], want["model"]) and same(got["world_aabb"], want["world_aabb"])):
            fail(seed, f"instance frame n={n}")
        if n:
            # light lists
            lights = rng.normal(0, 30, (int(rng.integers(1, 17)), 3)).astype(np.float32)
            lights[rng.random(lights.shape) < 0.05] = rng.choice(SPECIAL)
            out = torch.full((len(lights) * n + 1, 5), -1, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            p.light_draw_lists(lights, out.data_ptr(), first_instance_base=fib)
            wl = oracle.light_draw_lists(s["pos"], s["mesh_id"], s["meshes"], lights, first_instance_base=fib)
            if out.cpu().numpy()[:-1].tobytes() != wl.tobytes():
                fail(seed, f"light lists n={n} lights={len(lights)}")
            # several culled views in one launch
            k = int(rng.integers(1, 5))
            vw = []
            for _ in range(k):
                lp = rng.normal(0, 20, 3).astype(np.float32)
                qq = rng.normal(size=4)
                qq /= np.linalg.norm(qq)
                pl = oracle.project_camera(lp, qq.astype(np.float32), aspect=float(rng.uniform(0.5, 3)), fovy_degrees=float(rng.uniform(20, 120)),
                                           near=float(rng.uniform(0.01, 1)), far=float(rng.uniform(10, 1000)))
                if rng.random() < 0.2:
                    pl[rng.integers(0, 24, 2)] = rng.choice(SPECIAL, 2)
                vw.append((lp, pl, int(rng.integers(0, 2 ** 32)), int(rng.integers(0, 2 ** 32))))
            vb = [(torch.zeros((n, 5), dtype=torch.int32, device=dev), torch.zeros(8, dtype=torch.int32, device=dev),
                   torch.zeros((n + 31) // 32, dtype=torch.int32, device=dev)) for _ in vw]
            torch.cuda.synchronize()
            p.run_views([make_frame(pl, lp, first_instance_base=a, first_index_base=b) for lp, pl, a, b in vw],
                        [p.prepare_outputs(draw_cmds=c.data_ptr(), draw_count=sc.data_ptr(), draw_index_total=sc.data_ptr() + 4,
                                           visible_bitmap=bm.data_ptr(), async_=False) for c, sc, bm in vb])
            for (lp, pl, a, b), (c, sc, bm) in zip(vw, vb):
                wv = oracle.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], pl, lp, first_instance_base=a, first_index_base=b,
                                threads=8, want=("draw_cmds", "visible_bitmap"))
                cnt, tot = (int(x) & 0xFFFFFFFF for x in sc[:2].cpu().tolist())
                if not (cnt == wv["draw_count"] and tot == wv["draw_index_total"] and c[:cnt].cpu().numpy().tobytes() == wv["draw_cmds"].tobytes()
                        and np.array_equal(bm.cpu().numpy().view(np.uint32), wv["visible_bitmap"])):
                    fail(seed, f"run_views n={n} views={k}")
            # skinned frame
            j = int(rng.integers(1, 33))
            sk = random_skeleton(rng, j)
            poses = np.empty((n, j, 10), np.float32)
            poses[:, :, 0:3] = rng.uniform(-0.5, 0.5, (n, j, 3))
            q = rng.normal(size=(n, j, 4))
            poses[:, :, 3:7] = q / np.linalg.norm(q, axis=2, keepdims=True)
            poses[:, :, 7:10] = rng.uniform(0.7, 1.3, (n, j, 3))
            hit = rng.random(poses.shape) < 0.002
            poses[hit] = rng.choice(SPECIAL, int(hit.sum()))
            ws = oracle.run_skinned(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], sk, poses, s["planes"], s["cam_pos"],
                                    first_instance_base=fib, first_index_base=fxb)
            p.set_skeleton(sk["parent"], sk["inverse_bind"], sk["joint_box"])
            p.set_poses(poses)
            model = torch.zeros((n, 16), dtype=torch.float32, device=dev)
            palette = torch.zeros((n, j, 16), dtype=torch.float32, device=dev)
            aabb = torch.zeros((n, 6), dtype=torch.float32, device=dev)
            bitmap = torch.zeros((n + 31) // 32, dtype=torch.int32, device=dev)
            cmds = torch.zeros((n, 5), dtype=torch.int32, device=dev)
            scal = torch.zeros(8, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            p.run_skinned(make_frame(s["planes"], s["cam_pos"], first_instance_base=fib, first_index_base=fxb), palette=palette.data_ptr(),
                          model=model.data_ptr(), visible_bitmap=bitmap.data_ptr(), draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(),
                          draw_index_total=scal.data_ptr() + 4, world_aabb=aabb.data_ptr(), async_=False)
            count, total = (int(x) & 0xFFFFFFFF for x in scal[:2].cpu().tolist())
            if not (count == ws["draw_count"] and total == ws["draw_index_total"]
                    and cmds[:count].cpu().numpy().tobytes() == ws["draw_cmds"].tobytes()
                    and np.array_equal(bitmap.cpu().numpy().view(np.uint32), ws["visible_bitmap"])
                    and same(palette.cpu().numpy(), ws["palette"]) and same(aabb.cpu().numpy(), ws["world_aabb"])
                    and same(model.cpu().numpy(), ws["model"])):
                fail(seed, f"skinned frame n={n} joints={j}")
    # per-triangle stage on a generated scene with real geometry (every third seed: it is the slow one)
    if seed % 3 == 0:
        cfg = int(rng.choice([2, 3]))
        nt = int(rng.choice([40, 900, 5000, 70_000 if seed % 12 == 0 else 2000]))
        t = ra.scene.make_scene(cfg, n=nt, all_visible=bool(rng.random() < 0.5))
        hit = rng.random(nt) < 0.01
        t["scale"][hit] = rng.choice(SPECIAL, int(hit.sum()))
        vertices, indices = ra.scene.make_geometry(t["meshes"], ordering=str(rng.choice(["rows", "strips", "shuffled"])))
        pv = oracle.camera_pv(cam_pos=tuple(float(x) for x in rng.normal(0, 3, 3)), aspect=float(rng.uniform(0.7, 2.5)),
                              fovy_degrees=float(rng.uniform(30, 110)))
        # round 5: which kernel of the stage takes the frame is part of the rotation (read by mip_create): the range kernel with one range
        # per wave or with short ranges pulled from the counter, the large-frame pairing (sort kernels, both grids) chosen on the device or
        # forced either way, the round-4 kernels; an index base that is no multiple of 3; index counts that are no multiple of 3
        for k in ("MIP_TUNE_TRI_RANGE_SLOTS", "MIP_TUNE_TRI_BLOCK_MAX", "MIP_TUNE_TRI_CHOICE", "MIP_TUNE_TRI_CHUNKS_FROM"):
            os.environ.pop(k, None)
        mode = int(rng.integers(0, 8))
        if mode in (1, 2, 5):
            os.environ["MIP_TUNE_TRI_RANGE_SLOTS"] = str(int(rng.choice([256, 512, 1024])))
        if mode in (3, 4, 5, 6):
            os.environ["MIP_TUNE_TRI_BLOCK_MAX"] = "0"
            if mode in (4, 5):
                os.environ["MIP_TUNE_TRI_CHOICE"] = "block"   # the range kernel's grid
            elif mode == 6:
                os.environ["MIP_TUNE_TRI_CHOICE"] = "waves"
        if mode == 7:
            os.environ["MIP_TUNE_TRI_CHUNKS_FROM"] = "4294967295"
        index_base = int(rng.choice([0, 0, 5, 7, 3000001]))
        if rng.random() < 0.3:
            t["meshes"] = t["meshes"].copy()
            for k in range(len(t["meshes"])):
                for lod in range(int(t["meshes"]["n_lods"][k])):
                    if rng.random() < 0.3 and t["meshes"]["index_len"][k][lod] > 4:
                        t["meshes"]["index_len"][k][lod] -= int(rng.integers(1, 3))
        r = oracle.run(t["pos"], t["rot"], t["scale"], t["mesh_id"], t["meshes"], t["planes"], t["cam_pos"], first_index_base=index_base, threads=8)
        cap = index_base + r["draw_index_total"] + 3
        wc, wo, _ = oracle.cull_all_triangles(r, t["pos"], t["mesh_id"], t["meshes"], t["cam_pos"], pv, vertices, indices, out_capacity=cap)
        with ra.InstancePipeline(max_instances=nt, max_meshes=64) as p:
            p.set_mesh_table(t["meshes"])
            p.set_geometry(vertices, indices)
            p.set_instances(t["pos"], t["rot"], t["scale"], t["mesh_id"])
            model = torch.zeros((nt, 16), dtype=torch.float32, device=dev)
            cmds = torch.zeros((nt, 5), dtype=torch.int32, device=dev)
            scal = torch.zeros(8, dtype=torch.int32, device=dev)
            out = torch.full((cap,), -1, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            p.run_device(make_frame(t["planes"], t["cam_pos"], first_index_base=index_base, pv=pv), model=model.data_ptr(), draw_cmds=cmds.data_ptr(),
                         draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4, culled_index_buffer=out.data_ptr(),
                         culled_index_capacity=cap)
            count = int(scal[0].item())
            if not (count == len(wc) and cmds[:count].cpu().numpy().tobytes() == wc.tobytes()
                    and np.array_equal(out.cpu().numpy().view(np.uint32), wo)):
                fail(seed, f"triangle stage cfg={cfg} n={nt} mode={mode} index_base={index_base} env={ {k: v for k, v in os.environ.items() if k.startswith('MIP_TUNE_TRI')} }")
    done += 1
    if done % 25 == 0:
        print(f"{done} seeds ok (last: n={n})", flush=True)
print(f"FUZZ_SWEEP_OK {done} seeds from {first}", flush=True)
