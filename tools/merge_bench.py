#!/usr/bin/env python3
"""Times the shard merge for R chunks shaped like the 10 M / 8-rank exchange: mip_merge_wire_lists_packed (`merge_bench.py R packed`,
the form the ranks exchange when it fits), mip_merge_wire_lists (default, the 8-byte wire form
the ranks exchange) or mip_merge_draw_lists (`merge_bench.py R cmds`, the 20-byte form of round 2)."""
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
WIRE = not (len(sys.argv) > 2 and sys.argv[2] == "cmds")
if len(sys.argv) > 2 and sys.argv[2] == "packed":
    WIRE = "packed"
n = 1_250_000
s = scene.make_scene(4, n=n)
dev = torch.device("cuda", 0)
p = renderer_amd.InstancePipeline(n, len(s["meshes"]), timing=True)
p.set_mesh_table(s["meshes"])
p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
cap = 360_000
stride = chunk_stride_bytes(cap, wire=WIRE)
recv = torch.zeros(R * stride // 4, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for k in range(R):
    base = recv.data_ptr() + k * stride
    p.run_device(make_frame(s["planes"], s["cam_pos"], first_instance_base=k * n), draw_cmds=base + SHARD_HEADER_BYTES,
                 draw_count=base, draw_index_total=base + 4, wire=WIRE)
count = int(recv[0].item())
merged = torch.zeros((R * cap, 5), dtype=torch.int32, device=dev)
oc = torch.zeros(2, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
if WIRE == "packed":
    def merge(*a, **kw):
        p.merge_wire_lists(*a, packed=True, **kw)
else:
    merge = p.merge_wire_lists if WIRE else p.merge_draw_lists
for _ in range(5):
    merge(recv.data_ptr(), R, stride, merged.data_ptr(), oc.data_ptr(), chunk_capacity=cap)
p.reset_timings()
for _ in range(20):
    merge(recv.data_ptr(), R, stride, merged.data_ptr(), oc.data_ptr(), chunk_capacity=cap)
t = p.timings()
ms = t["total_merge_ms"] / t["merges"]
per_cmd = (4.25 + 20) if WIRE == "packed" else ((8.0625 + 20) if WIRE else 40)
name = "packed wire" if WIRE == "packed" else ("wire" if WIRE else "20-byte")
print(f"{name} form, {R} chunks x {count} cmds: LONE launches (hipEvent pair around each, GPU idle in between) {ms*1e3:.1f} us, {R*count*per_cmd/ms/1e6:.0f} GB/s (read+write); total {int(oc[0].item())}")
# back to back on one stream, as the step issues it (behind the all-gather, the GPU already busy): median of 20 samples of 20 merges
import ctypes
st = torch.cuda.Stream()
q = renderer_amd.InstancePipeline(n, len(s["meshes"]), stream=st.cuda_stream)
q.set_mesh_table(s["meshes"])
samples = []
with torch.cuda.stream(st):
    for _ in range(20):
        merge2 = (lambda *a, **kw: q.merge_wire_lists(*a, packed=(WIRE == "packed"), **kw)) if WIRE else q.merge_draw_lists
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(20):
            merge2(recv.data_ptr(), R, stride, merged.data_ptr(), oc.data_ptr(), chunk_capacity=cap, async_=True)
        e1.record(st)
        e1.synchronize()
        samples.append(e0.elapsed_time(e1) / 20 * 1e3)
    q.wait()
us = float(np.median(samples))
print(f"{name} form, {R} chunks x {count} cmds: BACK TO BACK {us:.1f} us per merge (min {min(samples):.1f}), {R*count*per_cmd/us/1e3:.0f} GB/s (read+write) = {R*count*per_cmd/us/1e3/8000:.3f} of 8 TB/s")
q.close()
p.close()
