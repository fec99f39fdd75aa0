#!/usr/bin/env python3
"""K views (per-light culled lists) of the same instances: K async frames on K frame slots against K
serialized frames. usage: tools/views_bench.py [n]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd
from renderer_amd import scene
from renderer_amd.pipeline import make_frame

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
s = scene.make_scene(2 if n <= 100_000 else 3, n=n)
dev = torch.device("cuda", 0)
for K in (1, 2, 4, 8):
    for slots in sorted({1, K}):
        p = renderer_amd.InstancePipeline(n, len(s["meshes"]), frames_in_flight=slots)
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        outs = []
        for k in range(K):
            cmds = torch.empty((n, 5), dtype=torch.int32, device=dev)
            scal = torch.zeros(8, dtype=torch.int32, device=dev)
            bitmap = torch.zeros((n + 31) // 32 + 1, dtype=torch.int32, device=dev)
            outs.append((cmds, scal, bitmap, p.prepare_outputs(draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(),
                                                               draw_index_total=scal.data_ptr() + 4, visible_bitmap=bitmap.data_ptr())))
        torch.cuda.synchronize()
        frames = [p.frame_ref(make_frame(s["planes"], np.array([3.0 * k, 1.0, 2.0], np.float32))) for k in range(K)]
        for _ in range(50):
            for k in range(K):
                p.run_prepared(frames[k], outs[k][3])
        p.wait()
        reps = 500
        t0 = time.perf_counter()
        for _ in range(reps):
            for k in range(K):
                p.run_prepared(frames[k], outs[k][3])
        p.wait()
        dt = (time.perf_counter() - t0) / reps
        print(f"n={n} views={K} frame slots={slots}: {dt*1e6:.1f} us per set of views ({dt/K*1e6:.1f} us per view)", flush=True)
        if slots == 1 and K <= 4:  # the fused launch
            raw = [make_frame(s["planes"], np.array([3.0 * k, 1.0, 2.0], np.float32)) for k in range(K)]
            po = [o[3] for o in outs]
            for _ in range(50):
                p.run_views(raw, po)
            p.wait()
            t0 = time.perf_counter()
            for _ in range(reps):
                p.run_views(raw, po)
            p.wait()
            dt = (time.perf_counter() - t0) / reps
            print(f"n={n} views={K} mip_run_views: {dt*1e6:.1f} us per set of views ({dt/K*1e6:.1f} us per view)", flush=True)
        p.close()
