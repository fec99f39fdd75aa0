#!/usr/bin/env python3
"""Times mip_light_draw_lists (shadow-pass lists, row f-4): K back-to-back launches, wall clock."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd
from renderer_amd import scene

dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
for cfg, n, n_lights in ((2, 100_000, 4), (2, 100_000, 16), (3, 1_000_000, 4), (3, 1_000_000, 16), (3, 999_999, 16)):
    s = scene.make_scene(cfg, n=n)
    lights = rng.uniform(-40, 40, size=(n_lights, 3)).astype(np.float32)
    with renderer_amd.InstancePipeline(n, len(s["meshes"])) as p:
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        out = torch.empty((n_lights * n, 5), dtype=torch.int32, device=dev)
        for _ in range(20):
            p.light_draw_lists(lights, out.data_ptr(), async_=True)
        p.wait()
        K = 300
        t0 = time.perf_counter()
        for _ in range(K):
            p.light_draw_lists(lights, out.data_ptr(), async_=True)
        p.wait()
        dt = (time.perf_counter() - t0) / K
        nbytes = n * (16 + 20 * n_lights)
        print(f"n={n} lights={n_lights}: {dt*1e6:.1f} us/launch, {nbytes/dt/1e9:.0f} GB/s algorithmic, "
              f"{n*n_lights/dt/1e9:.2f} G commands/s", flush=True)
