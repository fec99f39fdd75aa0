#!/usr/bin/env python3
"""Experiment (library built with -DMIP_EXP_RANGE_TIMES): when every wave of the range kernel started and ended.
usage: MIP_LIBRARY=.../libmip_w5_times.so tools/r05_range_times.py <config> <n>"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd
from renderer_amd import scene
from renderer_amd.pipeline import make_frame

config, n = int(sys.argv[1]), int(sys.argv[2])
s = scene.make_scene(config, n=n)
vertices, indices = scene.make_geometry(s["meshes"])
dev = torch.device("cuda", 0)
p = renderer_amd.InstancePipeline(n, len(s["meshes"]))
p.set_mesh_table(s["meshes"]); p.set_geometry(vertices, indices); p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
model = torch.zeros((n, 16), dtype=torch.float32, device=dev)
cmds = torch.zeros((n, 5), dtype=torch.int32, device=dev)
scal = torch.zeros(8, dtype=torch.int32, device=dev)
frame = make_frame(s["planes"], s["cam_pos"], pv=scene.default_pv())
p.run_device(frame, model=model.data_ptr(), draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4)
total = int(scal[1].item()) & 0xFFFFFFFF
W = 8192
out = torch.zeros(total + 3 + 2 * W, dtype=torch.int32, device=dev)
kw = dict(model=model.data_ptr(), draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4,
          culled_index_buffer=out.data_ptr(), culled_index_capacity=total + 3)
for _ in range(3):
    p.run_device(frame, **kw)
t = out[total + 3:].cpu().numpy().view(np.uint32).reshape(W, 2).astype(np.int64)
start, end = t[:, 0], t[:, 1]
t0 = start.min()
dur = (end - start) / 100.0  # realtime counter: 100 MHz -> us
print(f"config {config} n={n}: launch {(end.max() - t0) / 100.0:.1f} us; wave start spread {(start.max() - t0) / 100.0:.1f} us")
print("wave life us: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f  mean %.1f" % (dur.min(), np.percentile(dur, 10), np.median(dur), np.percentile(dur, 90), dur.max(), dur.mean()))
# by position in the machine: workgroup index -> XCD = blockIdx % 8
wg = np.arange(W) // 4
for x in range(8):
    sel = (wg % 8) == x
    print(f"  XCD {x}: mean life {dur[sel].mean():.1f} us, max {dur[sel].max():.1f}")
order = np.argsort(dur)
print("slowest waves:", [(int(i), float(dur[i])) for i in order[-8:]])
print("fastest waves:", [(int(i), float(dur[i])) for i in order[:8]])
# correlation with the share of LOD 0 triangles is left to the reader: the ranges are equal in triangles
p.close()
