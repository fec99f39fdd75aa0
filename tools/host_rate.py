#!/usr/bin/env python3
"""PCIe-inclusive rate: mip_run with host output pointers (results copied back every frame)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd
from renderer_amd import scene
for config in (2, 3):
    s = scene.make_scene(config)
    with renderer_amd.InstancePipeline(s["n"], len(s["meshes"])) as p:
        p.set_mesh_table(s["meshes"]); p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        for want in (("model", "visible_bitmap", "draw_cmds"), ("visible_bitmap", "draw_cmds")):
            p.run_host(s["planes"], s["cam_pos"], want=want)
            t0 = time.perf_counter(); K = 20
            for _ in range(K):
                p.run_host(s["planes"], s["cam_pos"], want=want)
            dt = (time.perf_counter() - t0) / K
            print(f"config {config} n={s['n']} host outputs {want}: {dt*1e6:.0f} us/frame, {s['n']/dt/1e9:.3f} G inst/s")
        t0 = time.perf_counter()
        for _ in range(5):
            p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        print(f"  full column upload (36 B/instance): {(time.perf_counter()-t0)/5*1e6:.0f} us")
