#!/usr/bin/env python3
"""A few launches of mip_run_views (four views, bench.py's eyes) for rocprofv3 passes. usage: views_pmc_run.py [n] [launches]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd
from renderer_amd import scene
from renderer_amd.pipeline import make_frame

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 10
EYES = [[0, 1, 2], [30, 20, -40.1], [0.1, 17, -0.1], [-30, 20, 40.1]]
s = scene.make_scene(2 if n <= 100_000 else 3, n=n)
dev = torch.device("cuda", 0)
p = renderer_amd.InstancePipeline(n, len(s["meshes"]))
p.set_mesh_table(s["meshes"])
p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
frames, outs, keep = [], [], []
for e in np.array(EYES, np.float32):
    planes = s["planes"].copy()
    shift = e - np.asarray(s["cam_pos"], np.float32)
    planes.reshape(6, 4)[:, 3] -= planes.reshape(6, 4)[:, :3] @ shift
    cmds = torch.empty((n, 5), dtype=torch.int32, device=dev)
    scal = torch.zeros(8, dtype=torch.int32, device=dev)
    bitmap = torch.zeros((n + 31) // 32 + 1, dtype=torch.int32, device=dev)
    keep.append((cmds, scal, bitmap))
    frames.append(make_frame(planes, e))
    outs.append(p.prepare_outputs(draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4,
                                  visible_bitmap=bitmap.data_ptr()))
torch.cuda.synchronize()
for _ in range(launches):
    p.run_views(frames, outs)
p.wait()
print("commands", [int(k[1][0].item()) for k in keep])
p.close()
