#!/usr/bin/env python3
"""Round 4: the frame must come out byte-identical in ANY order the hardware starts workgroups in.

Runs the diagnostic build (libmi_instance_pipeline_dbg.so) with the tile numbering permuted — reversed, scrambled —
and with one tile that never publishes, at sizes that are / are not resident as a whole, in both kernel orders, and
compares command list, count, index total and bitmap with the oracle. Prints how many tile aggregates waiting tiles
computed themselves (MipTimings.prefix_helps) and what the frame cost.

  python tools/r04_selfhelp_check.py            (parent: one child process per mode; the env is read at launch time)
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MODES = [("normal", {}), ("skip5", {"MIP_DEBUG_SKIP_PUBLISH_TILE": "5"}), ("scramble", {"MIP_DEBUG_TILE_ORDER": "scramble"}),
         ("reverse", {"MIP_DEBUG_TILE_ORDER": "reverse"})]
SIZES = [(3, 8192), (2, 100_000), (3, 300_000), (3, 1_000_000), (3, 2_500_000)]


def child(mode):
    import numpy as np
    import torch

    import oracle
    import renderer_amd
    from renderer_amd import scene
    from renderer_amd.pipeline import make_frame

    dev = torch.device("cuda", 0)
    for config, n in SIZES:
        if mode == "reverse" and n > 1_000_000:
            continue  # every resident tile helps thousands of predecessors: minutes, and it proves nothing the 1 M case does not
        s = scene.make_scene(config, n=n)
        want = oracle.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"], threads=8,
                          want=("draw_cmds", "visible_bitmap"))
        for order in ("1", "3"):
            os.environ["MIP_TUNE_ORDER"] = order
            with renderer_amd.InstancePipeline(max_instances=n, max_meshes=len(s["meshes"])) as p:
                p.set_mesh_table(s["meshes"])
                p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
                cmds = torch.zeros((n, 5), dtype=torch.int32, device=dev)
                bitmap = torch.zeros(((n + 31) // 32,), dtype=torch.int32, device=dev)
                scal = torch.zeros(8, dtype=torch.int32, device=dev)
                model = torch.empty((n, 16), dtype=torch.float32, device=dev)
                torch.cuda.synchronize()
                frame = make_frame(s["planes"], s["cam_pos"])
                times = []
                for rep in range(3):
                    t0 = time.time()
                    p.run_device(frame, model=model.data_ptr(), visible_bitmap=bitmap.data_ptr(), draw_cmds=cmds.data_ptr(),
                                 draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4)
                    times.append((time.time() - t0) * 1e3)
                count = int(scal[0].item())
                ok = (count == want["draw_count"] and int(scal[1].item()) == want["draw_index_total"]
                      and cmds[:count].cpu().numpy().tobytes() == want["draw_cmds"].tobytes()
                      and np.array_equal(bitmap.cpu().numpy().view(np.uint32), want["visible_bitmap"]))
                helps = p.timings()["prefix_helps"]
                print(f"{mode:9s} cfg{config} n={n:<8d} order {order}: {'OK ' if ok else 'MISMATCH'} count {count:7d} helps {helps:7d} "
                      f"frame ms {min(times):8.3f} (first {times[0]:8.3f})", flush=True)
                if not ok:
                    sys.exit(1)


def main():
    if len(sys.argv) > 1:
        child(sys.argv[1])
        return
    dbg = os.path.join(ROOT, "renderer_amd", "lib", "libmi_instance_pipeline_dbg.so")
    rc = 0
    for mode, env_add in MODES:
        env = dict(os.environ, MIP_LIBRARY=dbg, **env_add)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), mode], env=env, timeout=900)
        rc = rc or r.returncode
    sys.exit(rc)


if __name__ == "__main__":
    main()
