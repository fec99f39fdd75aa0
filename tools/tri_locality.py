#!/usr/bin/env python3
"""Row f-1, mixed 64-mesh scene: is the lower triangle rate (1.1e11/s against 1.9e11/s for one mesh) a cache-locality
effect? Runs the same 200 k instances with mesh ids as generated and sorted (commands of one mesh adjacent). Measured: 107 vs
110 G triangles/s - it is not."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd
from renderer_amd import scene
from renderer_amd.pipeline import make_frame
for mode in ("random", "sorted"):
    s = scene.make_scene(3, n=200000)
    if mode == "sorted":
        s["mesh_id"] = np.sort(s["mesh_id"])
    n = s["n"]
    vertices, indices = scene.make_geometry(s["meshes"])
    dev = torch.device("cuda", 0)
    p = renderer_amd.InstancePipeline(n, len(s["meshes"]))
    p.set_mesh_table(s["meshes"]); p.set_geometry(vertices, indices)
    p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
    model = torch.zeros((n, 16), dtype=torch.float32, device=dev)
    cmds = torch.zeros((n, 5), dtype=torch.int32, device=dev)
    scal = torch.zeros(8, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    frame = make_frame(s["planes"], s["cam_pos"], pv=scene.default_pv())
    p.run_device(frame, model=model.data_ptr(), draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4)
    count, total = (int(x) & 0xFFFFFFFF for x in scal[:2].cpu().tolist())
    tris = int(cmds[:count, 0].to(torch.int64).sum().item()) // 3
    out = torch.empty(total + 3, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    kw = dict(model=model.data_ptr(), draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4,
              culled_index_buffer=out.data_ptr(), culled_index_capacity=total + 3)
    for _ in range(3): p.run_device(frame, **kw)
    t0 = time.perf_counter()
    for _ in range(20): p.run_device(frame, async_=True, **kw)
    p.wait()
    dt = (time.perf_counter() - t0) / 20
    print(f"{mode}: {count} commands, {tris/1e6:.1f} M triangles, {dt*1e3:.3f} ms, {tris/dt/1e9:.1f} G tri/s; geometry {vertices.nbytes/1e6:.1f} MB vertices, {indices.nbytes/1e6:.1f} MB indices", flush=True)
    p.close()
