#!/usr/bin/env python3
"""Throughput with F frames in flight: F contexts (own prefix state, own output buffers, own
stream), launches issued round-robin — how a renderer with per-swapchain-image buffers
(DoubleBuffered<..>, renderer.rs:1225-1249) would drive the pipeline."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd
from renderer_amd import scene
from renderer_amd.pipeline import make_frame

config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
s = scene.make_scene(config)
n = s["n"]
dev = torch.device("cuda", 0)
for F in (1, 2, 3, 4):
    ctxs = []
    for f in range(F):
        st = torch.cuda.Stream()
        p = renderer_amd.InstancePipeline(n, len(s["meshes"]), stream=st.cuda_stream)
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        model = torch.empty((n, 16), dtype=torch.float32, device=dev)
        bitmap = torch.zeros(((n + 31) // 32 + 1,), dtype=torch.int32, device=dev)
        cmds = torch.empty((n, 5), dtype=torch.int32, device=dev)
        scal = torch.zeros(8, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        kw = dict(model=model.data_ptr(), visible_bitmap=bitmap.data_ptr(), draw_cmds=cmds.data_ptr(),
                  draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4)
        prepared = p.prepare_outputs(**kw)
        ctxs.append((p, prepared, (model, bitmap, cmds, scal), st))
    frame = make_frame(s["planes"], s["cam_pos"])
    fref = renderer_amd.InstancePipeline.frame_ref(frame)
    for k in range(20):
        p, kw, _, _ = ctxs[k % F]
        p.run_prepared(fref, kw)
    torch.cuda.synchronize()
    K = 400
    t0 = time.perf_counter()
    for k in range(K):
        p, kw, _, _ = ctxs[k % F]
        p.run_prepared(fref, kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"config {config} frames in flight {F}: {dt*1e6:.2f} us/step  {n/dt/1e9:.2f} G inst/s", flush=True)
    for p, *_ in ctxs:
        p.wait(); p.close()
