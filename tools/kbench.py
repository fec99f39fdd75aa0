#!/usr/bin/env python3
"""Serialized frame time of the instance kernel for one or more builds of the library.

  python tools/kbench.py [--configs 3,2] [--libs default,renderer_amd/lib/libmip_w8_x.so,...] [--subset full]

Each build runs in its own child process (MIP_LIBRARY is read at import). A sample is BATCH back-to-back
mip_run calls on one stream between two HIP events; the figure is the median over SAMPLES samples."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(configs, subset, ns):
    import numpy as np
    import torch

    import renderer_amd
    from renderer_amd import scene
    from renderer_amd.pipeline import make_frame

    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream()
    tag = os.environ.get("MIP_LIBRARY", "default")
    for ci, config in enumerate(configs):
        n = ns[ci] if ns and ns[ci] else None
        s = scene.make_scene(config, n=n)
        n = s["n"]
        with torch.cuda.stream(st):
            pipe = renderer_amd.InstancePipeline(n, len(s["meshes"]), stream=st.cuda_stream)
            pipe.set_mesh_table(s["meshes"])
            pipe.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
            model = torch.empty((n, 16), dtype=torch.float32, device=dev)
            bitmap = torch.zeros(((n + 31) // 32 + 1,), dtype=torch.int32, device=dev)
            cmds = torch.empty((n, 5), dtype=torch.int32, device=dev)
            scal = torch.zeros(8, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            kw = dict(model=model.data_ptr(), visible_bitmap=bitmap.data_ptr(), draw_cmds=cmds.data_ptr(),
                      draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4)
            if subset == "no_cmds":
                kw = dict(model=model.data_ptr(), visible_bitmap=bitmap.data_ptr())
            elif subset == "no_model":
                kw = dict(visible_bitmap=bitmap.data_ptr(), draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr())
            elif subset == "model_only":
                kw = dict(model=model.data_ptr())
            elif subset == "bitmap_only":
                kw = dict(visible_bitmap=bitmap.data_ptr())
            elif subset != "full":
                raise SystemExit(f"unknown --subset {subset!r}: full, no_cmds, no_model, model_only, bitmap_only")
            out = pipe.prepare_outputs(**kw)
            fref = pipe.frame_ref(make_frame(s["planes"], s["cam_pos"]))
            batch = 20 if n <= 2_000_000 else 5
            for _ in range(3 * batch):
                pipe.run_prepared(fref, out)
            st.synchronize()
            samples = []
            for _ in range(50):
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(batch):
                    pipe.run_prepared(fref, out)
                e1.record(st)
                e1.synchronize()
                samples.append(e0.elapsed_time(e1) / batch * 1e3)
            pipe.wait()
            count = int(scal[0].item()) if "draw_cmds" in kw else 0
            us = np.array(samples)
            v = count / n
            gbs = n * (100.125 + 20 * v) / (np.median(us) * 1e-6) / 1e9
            print(f"{tag:44s} cfg{config} n={n:<9d} {subset:8s} median {np.median(us):7.2f} us  min {us.min():7.2f}  p90 {np.percentile(us, 90):7.2f}"
                  f"  count {count}  {gbs:6.0f} GB/s  frac {gbs / 8000:.3f}", flush=True)
            pipe.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="3")
    ap.add_argument("--n", default="")
    ap.add_argument("--libs", default="default")
    ap.add_argument("--subset", default="full")
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    configs = [int(c) for c in a.configs.split(",")]
    ns = [int(x) if x else 0 for x in a.n.split(",")] if a.n else None
    if a.child:
        child(configs, a.subset, ns)
        return
    for lib in a.libs.split(","):
        env = dict(os.environ)
        env.pop("MIP_LIBRARY", None)
        if lib != "default":
            env["MIP_LIBRARY"] = lib
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--configs", a.configs, "--subset", a.subset]
        if a.n:
            cmd += ["--n", a.n]
        r = subprocess.run(cmd, env=env, timeout=300)
        if r.returncode:
            print(f"{lib}: exit {r.returncode}", flush=True)


if __name__ == "__main__":
    main()
