#!/usr/bin/env python3
"""Experiment (library built with -DMIP_EXP_RANGE_TIMES): life, commands and triangles of every wave of the wave-per-command kernel.
usage: MIP_LIBRARY=.../libmip_w5_times.so MIP_TUNE_TRI_CHOICE=waves tools/r05_wave_times.py <config> <n>"""
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
out = torch.zeros(total + 3 + 8 * W, dtype=torch.int32, device=dev)
kw = dict(model=model.data_ptr(), draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4,
          culled_index_buffer=out.data_ptr(), culled_index_capacity=total + 3)
for _ in range(3):
    p.run_device(frame, **kw)
t = out[total + 3:].cpu().numpy().view(np.uint32).reshape(W, 8).astype(np.int64)
start, end, ncmd, ntri, first_end = t[:, 0], t[:, 1], t[:, 2], t[:, 3], t[:, 4]
t0 = start.min()
life = (end - start) / 100.0
print(f"config {config} n={n}: launch {(end.max() - t0) / 100.0:.1f} us; wave start spread {(start.max() - t0) / 100.0:.1f} us")
print("wave life us: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f  mean %.1f" % (life.min(), np.percentile(life, 10), np.median(life), np.percentile(life, 90), life.max(), life.mean()))
print("commands per wave: min %d median %d max %d; triangles per wave: min %d median %d max %d" % (ncmd.min(), np.median(ncmd), ncmd.max(), ntri.min(), np.median(ntri), ntri.max()))
first = (first_end - start) / 100.0
print("first command done after us: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f" % (first.min(), np.percentile(first, 10), np.median(first), np.percentile(first, 90), first.max()))
rate = ntri / np.maximum(life, 1e-9)
print("triangles per us of wave life: p10 %.1f median %.1f p90 %.1f" % (np.percentile(rate, 10), np.median(rate), np.percentile(rate, 90)))
# how many waves are still alive at the end
for frac in (0.7, 0.8, 0.9, 0.95, 0.99):
    tt = t0 + frac * (end.max() - t0)
    print(f"  alive at {frac:.2f} of the launch: {(end > tt).sum()} waves")
rel = (start - t0) / 100.0
print("wave start us after the first: p50 %.1f p75 %.1f p85 %.1f p90 %.1f p95 %.1f max %.1f; waves starting later than 10 us: %d" % (
    np.percentile(rel, 50), np.percentile(rel, 75), np.percentile(rel, 85), np.percentile(rel, 90), np.percentile(rel, 95), rel.max(), (rel > 10).sum()))
late = rel > 10
wg = np.arange(W) // 4
print("late workgroups by index range:", [int(late[(wg >= lo) & (wg < lo + 256)].sum() // 4) for lo in range(0, 2048, 256)])
print("late workgroups by XCD (index % 8):", [int(late[(wg % 8) == x].sum() // 4) for x in range(8)])
print("late waves: commands median %d, triangles median %d" % (np.median(ncmd[late]) if late.any() else 0, np.median(ntri[late]) if late.any() else 0))
p.close()
