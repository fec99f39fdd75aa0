#!/usr/bin/env python3
"""GPU time of mip_run_views (K = 1..4 views in one launch), HIP events on the launch stream around BATCH
back-to-back launches — tools/views_bench.py times the host loop, which at one or two views is what it measures.

  python tools/views_kbench.py [n] [--libs default,renderer_amd/lib/libmip_x.so]"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

EYES = [[0, 1, 2], [30, 20, -40.1], [0.1, 17, -0.1], [-30, 20, 40.1]]  # bench.py views_leg


def child(n):
    import numpy as np
    import torch

    import renderer_amd
    from renderer_amd import scene
    from renderer_amd.pipeline import make_frame

    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream()
    tag = os.environ.get("MIP_LIBRARY", "default")
    s = scene.make_scene(2 if n <= 100_000 else 3, n=n)
    with torch.cuda.stream(st):
        p = renderer_amd.InstancePipeline(n, len(s["meshes"]), stream=st.cuda_stream)
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        frames, outs, keep = [], [], []
        for e in np.array(EYES, np.float32):
            planes = s["planes"].copy()
            shift = e - np.asarray(s["cam_pos"], np.float32)
            planes.reshape(6, 4)[:, 3] -= planes.reshape(6, 4)[:, :3] @ shift
            cmds = torch.empty((n, 5), dtype=torch.int32, device=dev)
            scal = torch.zeros(8, dtype=torch.int32, device=dev)
            bitmap = torch.zeros((n + 31) // 32 + 1, dtype=torch.int32, device=dev)
            keep.append((cmds, scal, bitmap))
            frames.append(make_frame(planes, e))
            outs.append(p.prepare_outputs(draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4,
                                          visible_bitmap=bitmap.data_ptr()))
        torch.cuda.synchronize()
        batch = 20
        for k in (1, 2, 3, 4):
            for _ in range(3 * batch):
                p.run_views(frames[:k], outs[:k])
            p.wait()
            samples = []
            for _ in range(40):
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(batch):
                    p.run_views(frames[:k], outs[:k])
                e1.record(st)
                e1.synchronize()
                samples.append(e0.elapsed_time(e1) / batch * 1e3)
            p.wait()
            counts = [int(x[1][0].item()) for x in keep[:k]]
            us = np.array(samples)
            mb = (36 * n + sum(20 * c for c in counts) + k * n / 8) / 1e6
            print(f"{tag:40s} n={n} views={k}: median {np.median(us):6.2f} us  min {us.min():6.2f}  p90 {np.percentile(us, 90):6.2f}"
                  f"   {mb:6.1f} MB -> {mb / np.median(us):5.2f} TB/s   commands {counts}", flush=True)
        p.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n", nargs="?", type=int, default=1_000_000)
    ap.add_argument("--libs", default="default")
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        child(a.n)
        return
    for lib in a.libs.split(","):
        env = dict(os.environ)
        if lib != "default":
            env["MIP_LIBRARY"] = os.path.join(ROOT, lib) if not os.path.isabs(lib) else lib
        subprocess.run([sys.executable, os.path.abspath(__file__), str(a.n), "--child"], env=env, check=True)


if __name__ == "__main__":
    main()
