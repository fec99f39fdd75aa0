#!/usr/bin/env python3
"""Times mip_merge_draw_lists for R chunks shaped like the 10 M / 8-rank exchange."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd
from renderer_amd import scene
from renderer_amd.pipeline import SHARD_HEADER_BYTES, make_frame
from renderer_amd.sharded import chunk_stride_bytes

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = 1_250_000
s = scene.make_scene(4, n=n)
dev = torch.device("cuda", 0)
p = renderer_amd.InstancePipeline(n, len(s["meshes"]), timing=True)
p.set_mesh_table(s["meshes"])
p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
cap = 360_000
stride = chunk_stride_bytes(cap)
recv = torch.zeros(R * stride // 4, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for k in range(R):
    base = recv.data_ptr() + k * stride
    p.run_device(make_frame(s["planes"], s["cam_pos"], first_instance_base=k * n), draw_cmds=base + SHARD_HEADER_BYTES,
                 draw_count=base, draw_index_total=base + 4)
count = int(recv[0].item())
merged = torch.zeros((R * cap, 5), dtype=torch.int32, device=dev)
oc = torch.zeros(2, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for _ in range(5):
    p.merge_draw_lists(recv.data_ptr(), R, stride, merged.data_ptr(), oc.data_ptr())
p.reset_timings()
for _ in range(20):
    p.merge_draw_lists(recv.data_ptr(), R, stride, merged.data_ptr(), oc.data_ptr())
t = p.timings()
ms = t["total_merge_ms"] / t["merges"]
print(f"{R} chunks x {count} cmds ({count*20/1e6:.1f} MB each): merge {ms*1e3:.1f} us, {R*count*40/ms/1e6:.0f} GB/s (read+write); total {int(oc[0].item())}")
p.close()
