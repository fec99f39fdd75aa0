#!/usr/bin/env python3
"""Cost of the optional outputs at 1 M instances: + world AABB (24 B/instance), + TLAS rows (64 B/instance).
Measured: 22.8 / 24.4 / 30.1 us per frame."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import renderer_amd
from renderer_amd import scene
from renderer_amd.pipeline import make_frame
s = scene.make_scene(3)
n = s["n"]; dev = torch.device("cuda", 0)
p = renderer_amd.InstancePipeline(n, len(s["meshes"]))
p.set_mesh_table(s["meshes"]); p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
model = torch.empty((n, 16), dtype=torch.float32, device=dev); bitmap = torch.zeros((n + 31) // 32 + 1, dtype=torch.int32, device=dev)
cmds = torch.empty((n, 5), dtype=torch.int32, device=dev); scal = torch.zeros(8, dtype=torch.int32, device=dev)
aabb = torch.empty((n, 6), dtype=torch.float32, device=dev); tlas = torch.empty((n, 16), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
frame = make_frame(s["planes"], s["cam_pos"])
base = dict(model=model.data_ptr(), visible_bitmap=bitmap.data_ptr(), draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4)
for name, extra in (("base", {}), ("+world_aabb", dict(world_aabb=aabb.data_ptr())), ("+tlas", dict(tlas_instances=tlas.data_ptr()))):
    kw = dict(base, **extra)
    for _ in range(20): p.run_device(frame, async_=True, **kw)
    p.wait()
    t0 = time.perf_counter()
    for _ in range(300): p.run_device(frame, async_=True, **kw)
    p.wait()
    print(name, f"{(time.perf_counter() - t0) / 300 * 1e6:.1f} us", flush=True)
