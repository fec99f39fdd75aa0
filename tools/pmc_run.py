#!/usr/bin/env python3
"""Launches one output variant of mip_run a few times (for rocprofv3 --pmc passes).
usage: pmc_run.py <full|no_cmds|bitmap_only|model_only> <config> [launches]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd
from renderer_amd import scene
from renderer_amd.pipeline import make_frame

variant = sys.argv[1]
config = int(sys.argv[2])
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 10
s = scene.make_scene(config)
n = s["n"]
dev = torch.device("cuda", 0)
pipe = renderer_amd.InstancePipeline(n, len(s["meshes"]))
pipe.set_mesh_table(s["meshes"])
pipe.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
model = torch.empty((n, 16), dtype=torch.float32, device=dev)
bitmap = torch.zeros(((n + 31) // 32 + 1,), dtype=torch.int32, device=dev)
cmds = torch.empty((n, 5), dtype=torch.int32, device=dev)
scal = torch.zeros(8, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
frame = make_frame(s["planes"], s["cam_pos"])
kw = {
    "full": dict(model=model.data_ptr(), visible_bitmap=bitmap.data_ptr(), draw_cmds=cmds.data_ptr(),
                 draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4),
    "no_cmds": dict(model=model.data_ptr(), visible_bitmap=bitmap.data_ptr()),
    "bitmap_only": dict(visible_bitmap=bitmap.data_ptr()),
    "model_only": dict(model=model.data_ptr()),
}[variant]
for _ in range(launches):
    pipe.run_device(frame, **kw)
print("count", int(scal[0].item()), "n", n)
pipe.close()
