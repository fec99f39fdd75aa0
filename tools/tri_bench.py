#!/usr/bin/env python3
"""Times the per-triangle stage (row f-1): mip_run with a culled index buffer."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd
from renderer_amd import scene
from renderer_amd.pipeline import make_frame

config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else None
ordering = sys.argv[3] if len(sys.argv) > 3 else "rows"  # rows | strips | shuffled (scene.make_geometry)
s = scene.make_scene(config, n=n)
n = s["n"]
if os.environ.get("TRI_BENCH_ONE_LOD") == "1":  # every command walks LOD 0 (dense vertex use)
    s["meshes"]["n_lods"][:] = 1
vertices, indices = scene.make_geometry(s["meshes"], ordering=ordering)
dev = torch.device("cuda", 0)
p = renderer_amd.InstancePipeline(n, len(s["meshes"]))
p.set_mesh_table(s["meshes"])
p.set_geometry(vertices, indices)
p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
model = torch.zeros((n, 16), dtype=torch.float32, device=dev)
bitmap = torch.zeros(((n + 31) // 32 + 1,), dtype=torch.int32, device=dev)
cmds = torch.zeros((n, 5), dtype=torch.int32, device=dev)
scal = torch.zeros(8, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
frame = make_frame(s["planes"], s["cam_pos"], pv=scene.default_pv())
p.run_device(frame, model=model.data_ptr(), draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4)
count0, total = (int(x) & 0xFFFFFFFF for x in scal[:2].cpu().tolist())
tris_in = int(cmds[:count0, 0].to(torch.int64).sum().item()) // 3
print(f"config {config} n={n}: {count0} commands, {tris_in/1e6:.1f} M triangles in, index stream capacity {total/1e6:.1f} M indices ({total*4/1e9:.2f} GB)")
out = torch.empty(total + 3, dtype=torch.int32, device=dev)
kw = dict(model=model.data_ptr(), visible_bitmap=bitmap.data_ptr(), draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(),
          draw_index_total=scal.data_ptr() + 4, culled_index_buffer=out.data_ptr(), culled_index_capacity=total + 3)
for _ in range(3):
    p.run_device(frame, **kw)
count1 = int(scal[0].item())
tris_out = int(cmds[:count1, 0].to(torch.int64).sum().item()) // 3
K = 10
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    p.run_device(frame, async_=True, **kw)
p.wait()
dt = (time.perf_counter() - t0) / K
print(f"  frame {dt*1e3:.3f} ms: {tris_in/dt/1e9:.2f} G triangles/s in, {tris_out/1e6:.1f} M survive ({tris_out/tris_in:.2%}), "
      f"{count1} commands left; index stream written {tris_out*12/dt/1e9:.0f} GB/s")
p.close()
