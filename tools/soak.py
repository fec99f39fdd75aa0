#!/usr/bin/env python3
"""Soak: many frames back to back (several frames in flight), checking the draw count / index total
of every output set at intervals — the cross-tile prefix protocol must never time out or drift."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd
from renderer_amd import scene
from renderer_amd.pipeline import make_frame

for config, launches in ((2, 400_000), (3, 40_000), (1, 200_000)):
    s = scene.make_scene(config)
    n = s["n"]
    dev = torch.device("cuda", 0)
    F = 3
    p = renderer_amd.InstancePipeline(n, len(s["meshes"]), frames_in_flight=F)
    p.set_mesh_table(s["meshes"])
    p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
    sets = []
    for _ in range(F):
        model = torch.empty((n, 16), dtype=torch.float32, device=dev)
        bitmap = torch.zeros(((n + 31) // 32 + 1,), dtype=torch.int32, device=dev)
        cmds = torch.empty((n, 5), dtype=torch.int32, device=dev)
        scal = torch.zeros(8, dtype=torch.int32, device=dev)
        sets.append((p.prepare_outputs(model=model.data_ptr(), visible_bitmap=bitmap.data_ptr(), draw_cmds=cmds.data_ptr(),
                                       draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4), scal, cmds, model, bitmap))
    frame = make_frame(s["planes"], s["cam_pos"])
    torch.cuda.synchronize()
    p.run_many(frame, [x[0] for x in sets], F)
    p.wait()
    ref = [tuple(x[1][:2].cpu().tolist()) for x in sets]
    ref_cmds = sets[0][2][: ref[0][0]].clone()
    assert len(set(ref)) == 1 and ref[0][0] > 0
    t0 = time.time()
    done = 0
    batch = 20_000
    while done < launches:
        p.run_many(frame, [x[0] for x in sets], batch)
        p.wait()   # raises MipError on a timeout flag
        done += batch
        got = [tuple(x[1][:2].cpu().tolist()) for x in sets]
        assert got == ref, (done, got, ref)
    assert torch.equal(sets[0][2][: ref[0][0]], ref_cmds)
    dt = time.time() - t0
    print(f"config {config}: {done} frames ok in {dt:.1f} s ({dt/done*1e6:.2f} us/frame), count {ref[0][0]}", flush=True)
    p.close()

# ---- the other launch paths, repeated: culled views, skinned frames, light lists ----
s = scene.make_scene(3, n=200_000)
n = s["n"]
dev = torch.device("cuda", 0)
p = renderer_amd.InstancePipeline(n, len(s["meshes"]), frames_in_flight=2)
p.set_mesh_table(s["meshes"])
p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
import numpy as np
eyes = np.array([[0, 1, 2], [30, 20, -40.1], [0.1, 17, -0.1], [-30, 20, 40.1], [5, 5, 5]], np.float32)
bufs, frames, outs = [], [], []
for e in eyes:
    cmds = torch.empty((n, 5), dtype=torch.int32, device=dev)
    scal = torch.zeros(8, dtype=torch.int32, device=dev)
    bufs.append((cmds, scal))
    frames.append(make_frame(s["planes"], e))
    outs.append(p.prepare_outputs(draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4))
torch.cuda.synchronize()
p.run_views(frames, outs)
p.wait()
ref = [tuple(b[1][:2].cpu().tolist()) for b in bufs]
t0 = time.time()
for it in range(20):
    for _ in range(500):
        p.run_views(frames, outs)
    p.wait()
    assert [tuple(b[1][:2].cpu().tolist()) for b in bufs] == ref, it
print(f"views: 10000 launches of 5 views ok in {time.time() - t0:.1f} s, counts {[r[0] for r in ref]}", flush=True)
p.close()

sk = scene.make_skinned_scene(100_000)
n = sk["n"]
p = renderer_amd.InstancePipeline(n, 1, frames_in_flight=2)
p.set_mesh_table(sk["meshes"])
p.set_instances(sk["pos"], sk["rot"], sk["scale"], sk["mesh_id"])
p.set_skeleton(sk["skeleton"]["parent"], sk["skeleton"]["inverse_bind"], sk["skeleton"]["joint_box"])
poses = torch.from_numpy(sk["poses"]).to(dev)
cmds = [torch.empty((n, 5), dtype=torch.int32, device=dev) for _ in range(2)]
scal = [torch.zeros(8, dtype=torch.int32, device=dev) for _ in range(2)]
palette = torch.empty((n, 19, 16), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
p.set_poses_device(poses.data_ptr(), n)
frame = make_frame(sk["planes"], sk["cam_pos"])
ref = None
t0 = time.time()
for it in range(20):
    for k in range(200):
        p.run_skinned(frame, palette=palette.data_ptr(), draw_cmds=cmds[k % 2].data_ptr(), draw_count=scal[k % 2].data_ptr(),
                      draw_index_total=scal[k % 2].data_ptr() + 4, async_=True)
    p.wait()
    got = [tuple(x[:2].cpu().tolist()) for x in scal]
    ref = ref or got
    assert got == ref and got[0] == got[1], (it, got, ref)
print(f"skinned: 4000 frames ok in {time.time() - t0:.1f} s, count {ref[0][0]}", flush=True)
p.close()
