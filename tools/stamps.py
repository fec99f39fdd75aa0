#!/usr/bin/env python3
"""Diagnostic: per-tile realtime stamps of the pipeline kernel (needs `make -C renderer_amd/csrc dbg`).
Prints the distribution of each segment in microseconds (100 MHz realtime counter)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from renderer_amd import _lib

_lib._SO = os.environ.get("MIP_STAMPS_LIB") or os.path.join(ROOT, "renderer_amd", "lib", "libmi_instance_pipeline_dbg.so")
import renderer_amd
from renderer_amd import scene
from renderer_amd.pipeline import make_frame


def main():
    config = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n = int(sys.argv[2]) if len(sys.argv) > 2 else None
    s = scene.make_scene(config, n=n)
    n = s["n"]
    tiles = (n + 255) // 256
    dev = torch.device("cuda", 0)
    pipe = renderer_amd.InstancePipeline(n, len(s["meshes"]))
    lib = pipe._lib
    lib.mip_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    pipe.set_mesh_table(s["meshes"])
    pipe.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
    model = torch.empty((n, 16), dtype=torch.float32, device=dev)
    bitmap = torch.zeros(((n + 31) // 32 + 1,), dtype=torch.int32, device=dev)
    cmds = torch.empty((n, 5), dtype=torch.int32, device=dev)
    scal = torch.zeros(8, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    frame = make_frame(s["planes"], s["cam_pos"])
    kw = dict(model=model.data_ptr(), visible_bitmap=bitmap.data_ptr(), draw_cmds=cmds.data_ptr(),
              draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4)
    for _ in range(5):
        pipe.run_device(frame, **kw)
    z = np.zeros((tiles, 8), np.uint64)
    lib.mip_debug_write_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lib.mip_debug_write_stamps(pipe._ctx, z.ctypes.data, tiles)
    pipe.run_device(frame, **kw)
    st = np.zeros((tiles, 8), np.uint64)
    lib.mip_debug_read_stamps(pipe._ctx, st.ctypes.data, tiles)
    t = st.astype(np.float64) / 100.0  # us
    t0 = t[:, 0].min()
    names = ["start", "computed", "published", "stores issued", "resolved", "end"]
    print(f"n={n} tiles={tiles}; kernel span {t[:, 5].max() - t0:.2f} us")
    for k in range(6):
        rel = t[:, k] - t0
        print(f"  {names[k]:14s} abs: min {rel.min():7.2f} p50 {np.median(rel):7.2f} p90 {np.percentile(rel, 90):7.2f} max {rel.max():7.2f}")
    for k in range(1, 6):
        d = t[:, k] - t[:, k - 1]
        print(f"  seg {names[k - 1]:>14s} -> {names[k]:14s}: mean {d.mean():6.2f} p50 {np.median(d):6.2f} p90 {np.percentile(d, 90):6.2f} max {d.max():6.2f}")
    pub = t[:, 2]
    pred_pub = np.concatenate([[pub[0]], np.maximum.accumulate(pub)[:-1]])  # slowest earlier tile's publish
    earliest = np.maximum(pred_pub, t[:, 3])
    lag = t[:, 4] - earliest
    print(f"  resolve lag behind (slowest predecessor publish | own stores issued): mean {lag.mean():.2f} p50 {np.median(lag):.2f} p90 {np.percentile(lag, 90):.2f} max {lag.max():.2f}")
    wait_pred = np.maximum(pred_pub - t[:, 3], 0)
    print(f"  time spent waiting for the slowest predecessor to publish: mean {wait_pred.mean():.2f} p50 {np.median(wait_pred):.2f} p90 {np.percentile(wait_pred, 90):.2f} max {wait_pred.max():.2f}")
    print(f"  poll iterations per tile: mean {st[:, 6].mean():.1f} p50 {np.median(st[:, 6]):.0f} max {st[:, 6].max()}; unready lane-polls per tile mean {st[:, 7].mean():.1f}")
    life = t[:, 5] - t[:, 0]
    print(f"  block lifetime mean {life.mean():.2f} p50 {np.median(life):.2f} max {life.max():.2f}")
    # start time vs tile index (dispatch order)
    order = np.argsort(t[:, 0], kind="stable")
    inv = int((np.diff(order) < 0).sum())
    print(f"  dispatch inversions (start-time order vs tile index): {inv} of {tiles - 1}")
    for q in (0, tiles // 4, tiles // 2, 3 * tiles // 4, tiles - 1):
        print(f"  tile {q:5d}: " + " ".join(f"{t[q, k] - t0:7.2f}" for k in range(6)))
    pipe.close()


if __name__ == "__main__":
    main()
