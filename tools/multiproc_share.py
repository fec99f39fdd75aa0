#!/usr/bin/env python3
"""P processes sharing ONE GPU, each running back-to-back frames of its own context: does the dispatch-order
assumption of the one-hop prefix hold when several processes' launches interleave on the chip?
usage: multiproc_share.py [procs] [instances] [frames]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import renderer_amd
from renderer_amd import scene
from renderer_amd.pipeline import make_frame
n, frames, tag = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
s = scene.make_scene(4, n=n)
dev = torch.device("cuda", 0)
p = renderer_amd.InstancePipeline(n, len(s["meshes"]))
p.set_mesh_table(s["meshes"]); p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
model = torch.empty((n, 16), dtype=torch.float32, device=dev); cmds = torch.empty((n, 5), dtype=torch.int32, device=dev)
scal = torch.zeros(8, dtype=torch.int32, device=dev); bitmap = torch.zeros((n + 31) // 32 + 1, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
out = p.prepare_outputs(model=model.data_ptr(), draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(), visible_bitmap=bitmap.data_ptr())
fref = p.frame_ref(make_frame(s["planes"], s["cam_pos"]))
errors, t0 = 0, time.time()
for k in range(frames):
    try:
        p.run_prepared(fref, out)
        if k % 16 == 15:
            p.wait()
    except renderer_amd.MipError as e:
        errors += 1
        print(tag, "frame", k, e, flush=True)
try:
    p.wait()
except renderer_amd.MipError as e:
    errors += 1
    print(tag, "final", e, flush=True)
print(tag, "done: errors", errors, "count", int(scal[0].item()), "general/ordered?", p.timings(), round(time.time() - t0, 2), "s", flush=True)
'''
procs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3_333_334
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 400
ps = [subprocess.Popen([sys.executable, "-c", CHILD, ROOT, str(n), str(frames), f"proc{k}"]) for k in range(procs)]
for p in ps:
    p.wait()
