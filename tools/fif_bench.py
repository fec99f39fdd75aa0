import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import renderer_amd
from renderer_amd import scene
from renderer_amd.pipeline import make_frame
import bench
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
for cfg in (3, 2):
    s = scene.make_scene(cfg)
    for frames in (1, 2, 3):
        r = bench.frames_in_flight_leg(torch, renderer_amd, make_frame, s, dev, 0, frames, 256 if cfg == 3 else 1024)
        print(cfg, frames, os.environ.get("MIP_TUNE_GRAPH_ROUND"), round(r["ms_per_step"] * 1e3, 2), "us", r["host_loop"][:60], flush=True)
