#!/usr/bin/env python3
"""Experiment (diagnostic build with -DMIP_EXP_HELP_STAMPS: make -C renderer_amd/csrc dbg TAG=_helpstamps EXTRA=-DMIP_EXP_HELP_STAMPS): where a waiting
tile's time goes in the degraded mode — ticks inside help(), helps and idle looks per tile. usage: MIP_DEBUG_TILE_ORDER=scramble tools/r05_help_stamps.py [n]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from renderer_amd import _lib

_lib._SO = os.path.join(ROOT, "renderer_amd", "lib", os.environ.get("STAMPS_LIB", "libmi_instance_pipeline_dbg_helpstamps.so"))
import renderer_amd
from renderer_amd import scene
from renderer_amd.pipeline import make_frame

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
s = scene.make_scene(3, n=n)
tiles = (n + 255) // 256
dev = torch.device("cuda", 0)
pipe = renderer_amd.InstancePipeline(n, len(s["meshes"]))
lib = pipe._lib
lib.mip_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
lib.mip_debug_write_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
pipe.set_mesh_table(s["meshes"]); pipe.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
model = torch.empty((n, 16), dtype=torch.float32, device=dev)
bitmap = torch.zeros(((n + 31) // 32 + 1,), dtype=torch.int32, device=dev)
cmds = torch.empty((n, 5), dtype=torch.int32, device=dev)
scal = torch.zeros(8, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
frame = make_frame(s["planes"], s["cam_pos"])
kw = dict(model=model.data_ptr(), visible_bitmap=bitmap.data_ptr(), draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4)
for _ in range(5):
    pipe.run_device(frame, **kw)
z = np.zeros((tiles, 8), np.uint64)
lib.mip_debug_write_stamps(pipe._ctx, z.ctypes.data, tiles)
pipe.run_device(frame, **kw)
st = np.zeros((tiles, 8), np.uint64)
lib.mip_debug_read_stamps(pipe._ctx, st.ctypes.data, tiles)
t = st[:, :6].astype(np.float64) / 100.0
t0 = t[:, 0].min()
resolve = t[:, 4] - t[:, 3]
help_us = (st[:, 6] & np.uint64(0xFFFFFFFF)).astype(np.float64) / 100.0
helps = (st[:, 6] >> np.uint64(32)).astype(np.int64)
idle = st[:, 7].astype(np.int64)
slow = resolve > 10.0
print(f"n={n} tiles={tiles}: kernel span {t[:, 5].max() - t0:.1f} us; tiles whose resolve took > 10 us: {int(slow.sum())}")
if slow.any():
    print(f"  of those: resolve mean {resolve[slow].mean():.1f} us (p90 {np.percentile(resolve[slow], 90):.1f}); inside help(): mean {help_us[slow].mean():.1f} us over {helps[slow].mean():.2f} helps "
          f"({(help_us[slow].sum() / max(helps[slow].sum(), 1)):.1f} us per help); idle looks at claimed tiles: mean {idle[slow].mean():.1f} (max {idle[slow].max()})")
    print(f"  -> per slow tile: {help_us[slow].mean():.1f} us helping, ~{idle[slow].mean() * 0.6:.1f} us looking at claimed tiles (0.6 us per look), the rest of {resolve[slow].mean():.1f} us in polls and walks")
pipe.close()
