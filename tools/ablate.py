#!/usr/bin/env python3
"""Times mip_run with subsets of the outputs enabled (null pointers switch kernel stages off)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd
from renderer_amd import scene
from renderer_amd.pipeline import make_frame


def main():
    config = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n = int(sys.argv[2]) if len(sys.argv) > 2 else None
    allvis = len(sys.argv) > 3 and sys.argv[3] == "allvis"
    s = scene.make_scene(config, n=n, all_visible=allvis)
    n = s["n"]
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        pipe = renderer_amd.InstancePipeline(n, len(s["meshes"]), stream=st.cuda_stream)
        pipe.set_mesh_table(s["meshes"])
        pipe.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        model = torch.empty((n, 16), dtype=torch.float32, device=dev)
        bitmap = torch.zeros(((n + 31) // 32 + 1,), dtype=torch.int32, device=dev)
        cmds = torch.empty((n, 5), dtype=torch.int32, device=dev)
        scal = torch.zeros(8, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        aabb = torch.empty((n, 6), dtype=torch.float32, device=dev)
        frame = make_frame(s["planes"], s["cam_pos"])
        variants = {
            "full": dict(model=model.data_ptr(), visible_bitmap=bitmap.data_ptr(), draw_cmds=cmds.data_ptr(),
                         draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4),
            "no_model": dict(visible_bitmap=bitmap.data_ptr(), draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr()),
            "no_cmds": dict(model=model.data_ptr(), visible_bitmap=bitmap.data_ptr()),
            "bitmap_only": dict(visible_bitmap=bitmap.data_ptr()),
            "model_only": dict(model=model.data_ptr()),
            "model+cmds (no bitmap)": dict(model=model.data_ptr(), draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4),
            "full+aabb": dict(model=model.data_ptr(), visible_bitmap=bitmap.data_ptr(), draw_cmds=cmds.data_ptr(),
                              draw_count=scal.data_ptr(), world_aabb=aabb.data_ptr()),
        }
        for name, kw in variants.items():
            for _ in range(10):
                pipe.run_device(frame, async_=True, **kw)
            st.synchronize()
            t0 = time.perf_counter()
            K = 200
            for _ in range(K):
                pipe.run_device(frame, async_=True, **kw)
            st.synchronize()
            wall = (time.perf_counter() - t0) / K * 1e6
            evs = []
            for _ in range(50):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(st); pipe.run_device(frame, async_=True, **kw); e1.record(st)
                evs.append((e0, e1))
            st.synchronize()
            ms = np.array([a.elapsed_time(b) for a, b in evs]) * 1e3
            print(f"{name:12s} n={n} wall/step {wall:8.2f} us   event mean {ms.mean():8.2f} med {np.median(ms):8.2f} min {ms.min():8.2f} us", flush=True)
        pipe.wait()
        pipe.close()


if __name__ == "__main__":
    main()
