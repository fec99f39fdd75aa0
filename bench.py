#!/usr/bin/env python3
"""bench.py — instances/sec through transform + cull + compact (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N=1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one frame: one pass of the hot path (mip_run) over the resident instance arrays,
outputs written to HBM-resident buffers. Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
ALLGATHER_MIN_INSTANCES = 1_000_000  # north star: exchange the draw list only at >= 1 M instances


def algorithmic_bytes_per_instance(v):
    """read 36 B + write 64 B matrix + 1 bit + 20 B per emitted command."""
    return 100.125 + 20.0 * v


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", type=int, default=2, help="1 Box 1k | 2 DamagedHelmet 100k | 3 mixed 1M | 4 mixed 10M")
    ap.add_argument("--instances", type=int, default=None, help="override the per-GPU instance count")
    ap.add_argument("--all-visible", action="store_true", help="every instance inside the frustum (worst-case writes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary configs reported under 'extra'")
    ap.add_argument("--cpu-seconds", type=float, default=4.0)
    ap.add_argument("--native-rccl-leg", action="store_true",
                    help="N>1 only: also time mip_run_sharded (RCCL opened by the library itself). Off by default: it "
                         "creates a second communicator beside torch's and has only been rehearsed with one rank")
    ap.add_argument("--frames-in-flight", type=int, default=2,
                    help="frames the host keeps in flight (own output buffers each), as the reference's "
                         "per-swapchain-image buffers allow; 1 = strictly serialized steps")
    return ap.parse_args()


class DeviceOutputs:
    """HBM-resident output buffers (torch is only the allocator here)."""

    def __init__(self, torch, n, device):
        self.model = torch.empty((max(n, 1), 16), dtype=torch.float32, device=device)
        self.bitmap = torch.zeros(((n + 31) // 32 + 1,), dtype=torch.int32, device=device)
        self.cmds = torch.empty((max(n, 1), 5), dtype=torch.int32, device=device)
        self.scalars = torch.zeros((8,), dtype=torch.int32, device=device)  # [0] count, [1] index total

    def kwargs(self):
        return dict(model=self.model.data_ptr(), visible_bitmap=self.bitmap.data_ptr(),
                    draw_cmds=self.cmds.data_ptr(), draw_count=self.scalars.data_ptr(),
                    draw_index_total=self.scalars.data_ptr() + 4)


def time_steps(torch, dist, step, steps, warmup, distributed, issue_many=None):
    """issue_many(k), if given, enqueues k steps from compiled code (mip_run_many) instead of k
    Python-level calls of step()."""
    if issue_many is not None:
        issue_many(warmup)
    else:
        for _ in range(warmup):
            step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if issue_many is not None:
        issue_many(steps)
    else:
        for _ in range(steps):
            step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0  # this rank's K steps, start barrier -> own work drained; MAX over ranks below
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def kernel_event_time(torch, step, steps, warmup):
    """Average launch duration of the pipeline kernel, measured live with HIP events on the
    stream the kernel is launched on (the context runs on torch's current stream): one event
    before and one after `steps` back-to-back launches. This includes the ~1 us gap between
    dependent launches, so it is an upper bound of the kernel's own duration (rocprofv3's
    per-dispatch figure, profiles/, agrees within the profiler's own slowdown)."""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    samples = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            step()
        e1.record()
        torch.cuda.synchronize()
        samples.append(e0.elapsed_time(e1) / steps)
    samples = np.array(samples)
    return {"mean": float(samples.mean()), "median": float(np.median(samples)), "min": float(samples.min())}


def pmc_traffic(config, n):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*pmc_summary.json:
    FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE), if one matches this workload."""
    import glob

    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_summary.json"))):
        try:
            for row in json.load(open(path)).get("workloads", []):
                if row.get("config") == config and row.get("instances") == n and row.get("variant") == "full":
                    best = dict(row, source=os.path.relpath(path, ROOT))
        except (OSError, ValueError):
            pass
    return best


def host_cores():
    """Threads for the CPU leg: the affinity mask, capped by the cgroup CPU quota if there is one
    and by 16 (a 1-GPU box's CPU share on the pool this runs on)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(cores, int(os.environ.get("MIP_BENCH_MAX_THREADS", "16"))))


def cpu_baseline(scene_dict, seconds):
    """The oracle (a CPU port of the reference path) on this box's host cores, bounded sample."""
    import oracle

    oracle.build()
    cores = host_cores()
    s = scene_dict
    n = s["n"]
    sample_n = min(n, 1_000_000)
    args = (s["pos"][:sample_n], s["rot"][:sample_n], s["scale"][:sample_n], s["mesh_id"][:sample_n],
            s["meshes"], s["planes"], s["cam_pos"])
    oracle.run(*args, threads=cores, want=("model", "visible_bitmap", "draw_cmds"))  # warm-up
    passes, t0 = 0, time.perf_counter()
    while True:
        oracle.run(*args, threads=cores, want=("model", "visible_bitmap", "draw_cmds"))
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or passes >= 2000:
            break
    # the reference sizes its ComputeTaskPool at physical_core_count() / 2 (src/main.rs:881-886);
    # the same port on that many threads, shorter sample
    half = max(1, cores // 2)
    h_passes, t1 = 0, time.perf_counter()
    while True:
        oracle.run(*args, threads=half, want=("model", "visible_bitmap", "draw_cmds"))
        h_passes += 1
        h_dt = time.perf_counter() - t1
        if h_dt >= seconds / 4 or h_passes >= 500:
            break
    return {
        "value": sample_n * passes / dt,
        "unit": "instances/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{passes} passes over {sample_n} instances of the same scene, {cores} threads, "
                  f"{dt:.1f} s wall (C oracle, gcc -O2 -ffp-contract=off; includes output allocation)",
        "reference_pool_size": {"threads": half, "value": sample_n * h_passes / h_dt,
                                "note": "same port on cores/2 threads, the reference's ComputeTaskPool size (src/main.rs:881-886)"},
    }


def light_leg(torch, renderer_amd, scene, make_frame, s, device, local_rank):
    """Row f-4 (shadow pass): per-light draw lists for the 4 lights the reference spawns (main.rs:368-382)."""
    n = s["n"]
    pl = renderer_amd.InstancePipeline(max_instances=n, max_meshes=len(s["meshes"]), device=local_rank)
    pl.set_mesh_table(s["meshes"])
    pl.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
    lights = np.array([[30, 20, -40.1], [0.1, 17, -0.1], [-30, 20, 40.1], [0, 30, 0]], np.float32)
    lists = torch.empty((len(lights) * n, 5), dtype=torch.int32, device=device)
    for _ in range(20):
        pl.light_draw_lists(lights, lists.data_ptr(), async_=True)
    pl.wait()
    t0 = time.perf_counter()
    for _ in range(200):
        pl.light_draw_lists(lights, lists.data_ptr(), async_=True)
    pl.wait()
    dt = (time.perf_counter() - t0) / 200
    pl.close()
    return {
        "instances": n, "lights": len(lights), "ms_per_launch": dt * 1e3,
        "algorithmic_GBps": n * (16 + 20 * len(lights)) / dt / 1e9,
        "note": "shadow_mapping.rs:405-478 as indirect lists: 16 B read + 20 B x lights written per instance",
    }


def views_leg(torch, renderer_amd, scene, make_frame, s, device, local_rank):
    """Row f-4, "per-light cull lists": four culled views (the reference's light positions as LOD reference
    points, the default frustum moved to each) of the headline scene in one launch, mip_run_views."""
    n = s["n"]
    p = renderer_amd.InstancePipeline(max_instances=n, max_meshes=len(s["meshes"]), device=local_rank)
    p.set_mesh_table(s["meshes"])
    p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
    eyes = np.array([[0, 1, 2], [30, 20, -40.1], [0.1, 17, -0.1], [-30, 20, 40.1]], np.float32)
    frames, outs, keep = [], [], []
    for e in eyes:
        planes = s["planes"].copy()
        # the default frustum translated to the eye: d' = d - n . (eye - default eye)
        shift = e - np.asarray(s["cam_pos"], np.float32)
        planes.reshape(6, 4)[:, 3] -= planes.reshape(6, 4)[:, :3] @ shift
        cmds = torch.empty((n, 5), dtype=torch.int32, device=device)
        scal = torch.zeros(8, dtype=torch.int32, device=device)
        bitmap = torch.zeros((n + 31) // 32 + 1, dtype=torch.int32, device=device)
        keep.append((cmds, scal, bitmap))
        frames.append(make_frame(planes, e))
        outs.append(p.prepare_outputs(draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4,
                                      visible_bitmap=bitmap.data_ptr()))
    torch.cuda.synchronize()
    for _ in range(20):
        p.run_views(frames, outs)
    p.wait()
    counts = [int(k[1][0].item()) for k in keep]
    t0 = time.perf_counter()
    for _ in range(300):
        p.run_views(frames, outs)
    p.wait()
    dt = (time.perf_counter() - t0) / 300
    p.close()
    return {"instances": n, "views": len(eyes), "ms_per_launch": dt * 1e3, "instance_views_per_s": n * len(eyes) / dt,
            "commands_per_view": counts,
            "note": "one launch: instance data read once, matrix + world box built once, per view plane test + compaction"}


def skinned_leg(torch, renderer_amd, scene, make_frame, s_unused, device, local_rank):
    """BASELINE config 5 (extension, no reference semantics): 256 k instances of a 19-joint figure, each
    with its own pose: palette + skinned bounds kernel, then the instance kernel."""
    s = scene.make_skinned_scene()
    n, j = s["n"], len(s["skeleton"]["parent"])
    p = renderer_amd.InstancePipeline(max_instances=n, max_meshes=len(s["meshes"]), device=local_rank)
    p.set_mesh_table(s["meshes"])
    p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
    sk = s["skeleton"]
    p.set_skeleton(sk["parent"], sk["inverse_bind"], sk["joint_box"])
    poses = torch.from_numpy(s["poses"]).to(device)
    torch.cuda.synchronize()
    p.set_poses_device(poses.data_ptr(), n)
    o = DeviceOutputs(torch, n, device)
    palette = torch.empty((n, j, 16), dtype=torch.float32, device=device)
    frame = make_frame(s["planes"], s["cam_pos"])
    torch.cuda.synchronize()
    for _ in range(5):
        p.run_skinned(frame, palette=palette.data_ptr(), async_=True, **o.kwargs())
    p.wait()
    count = int(o.scalars[0].item())
    steps = 50
    t0 = time.perf_counter()
    for _ in range(steps):
        p.run_skinned(frame, palette=palette.data_ptr(), async_=True, **o.kwargs())
    p.wait()
    dt = (time.perf_counter() - t0) / steps
    p.close()
    nbytes = n * (36 + j * 40 + j * 64 + 1) + n * (36 + 64 + 0.125 + 1) + count * 20
    return {
        "instances": n, "joints": j, "ms_per_frame": dt * 1e3, "instances_per_s": n / dt, "emitted_fraction": count / n,
        "algorithmic_GBps": nbytes / dt / 1e9,
        "note": "per instance: 19 x (40 B pose read + 64 B palette written) in the skinning kernel, then the instance kernel's "
                "100 B + 20 B per command; parity is against this repository's oracle only",
    }


def triangle_leg(torch, renderer_amd, scene, make_frame, s, device, local_rank, with_cpu):
    """generate_work.comp:68-200 for every emitted command of the headline scene (synthetic torus
    geometry with the DamagedHelmet triangle counts): frame = instance kernel + triangle kernel +
    re-compaction."""
    n = s["n"]
    vertices, indices = scene.make_geometry(s["meshes"])
    pv = scene.default_pv()
    p = renderer_amd.InstancePipeline(max_instances=n, max_meshes=len(s["meshes"]), device=local_rank)
    p.set_mesh_table(s["meshes"])
    p.set_geometry(vertices, indices)
    p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
    o = DeviceOutputs(torch, n, device)
    torch.cuda.synchronize()  # torch fills on its own stream; the library does not wait for it
    frame = make_frame(s["planes"], s["cam_pos"], pv=pv)
    p.run_device(frame, **o.kwargs())
    count0, total = (int(x) & 0xFFFFFFFF for x in o.scalars[:2].cpu().tolist())
    tris_in = int(o.cmds[:count0, 0].to(torch.int64).sum().item()) // 3
    stream_out = torch.empty(total + 3, dtype=torch.int32, device=device)
    kw = dict(o.kwargs(), culled_index_buffer=stream_out.data_ptr(), culled_index_capacity=total + 3)
    for _ in range(3):
        p.run_device(frame, **kw)
    count1 = int(o.scalars[0].item())
    tris_out = int(o.cmds[:count1, 0].to(torch.int64).sum().item()) // 3
    steps = 20
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        p.run_device(frame, async_=True, **kw)
    p.wait()
    dt = (time.perf_counter() - t0) / steps
    p.close()
    row = {
        "instances": n, "commands_in": count0, "commands_out": count1, "triangles_in": tris_in,
        "triangles_surviving": tris_out, "ms_per_frame": dt * 1e3, "triangles_per_s": tris_in / dt,
        "index_stream_write_GBps": tris_out * 12 / dt / 1e9,
        "note": "instruction-issue/latency bound, not HBM: two mat4*vec4 per vertex without FMA + 6 correctly rounded "
                "divides per triangle; geometry is L2-resident, HBM traffic is the 12 B per surviving triangle",
    }
    if with_cpu:
        import oracle

        cores = host_cores()
        sample = min(n, 4000)
        s2 = {k: (v[:sample] if k in ("pos", "rot", "scale", "mesh_id") else v) for k, v in s.items()}
        r = oracle.run(s2["pos"], s2["rot"], s2["scale"], s2["mesh_id"], s2["meshes"], s2["planes"], s2["cam_pos"], threads=cores)
        t_in = int(r["draw_cmds"]["indexCount"].astype(np.int64).sum()) // 3
        passes, t0 = 0, time.perf_counter()
        while True:
            oracle.cull_all_triangles(r, s2["pos"], s2["mesh_id"], s2["meshes"], s2["cam_pos"], pv, vertices, indices, threads=cores)
            passes += 1
            dtc = time.perf_counter() - t0
            if dtc >= 3.0 or passes >= 500:
                break
        row["cpu_baseline"] = {"value": t_in * passes / dtc, "unit": "triangles/s", "cores": cores, "kind": "port",
                               "sample": f"{passes} passes over the commands of the first {sample} instances ({t_in} triangles each), "
                                         f"{cores} threads, {dtc:.1f} s wall"}
    return row


def main():
    args = parse_args()
    args.steps = max(1, args.steps)
    args.warmup = max(0, args.warmup)
    # RCCL prints a version banner on stdout when a communicator is created; the contract is ONE
    # JSON line on stdout. Everything else this process (and the libraries it loads) prints goes
    # to stderr; the JSON line is written to the saved stdout at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    import renderer_amd
    from renderer_amd import scene
    from renderer_amd.pipeline import make_frame

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if os.environ.get("MIP_BENCH_FORCE_DIST") == "1":  # rehearsal of the N>1 code paths with one rank
        distributed = True
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the instance pipeline has no CPU path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        dist.init_process_group("nccl", device_id=device)

    if not os.path.exists(renderer_amd.library_path()):  # bare checkout: build the HIP library first
        import __graft_entry__

        __graft_entry__.build()
    renderer_amd.load_library()
    cfg = scene.CONFIGS[args.config]
    n_local = args.instances if args.instances is not None else cfg["n"]
    n_global = n_local * world
    s = scene.make_scene(args.config, n=n_local, first=rank * n_local, all_visible=args.all_visible)

    # Streams: the sharded path needs the kernels ordered with torch's NCCL ops, so there the
    # context enqueues on a real (non-null) torch stream made current here. Otherwise the
    # context owns one stream per frame in flight.
    torch_stream = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(torch_stream)
    stream = torch_stream.cuda_stream
    exchange_on = distributed and n_global >= ALLGATHER_MIN_INSTANCES
    frames = 1 if exchange_on else max(1, args.frames_in_flight)
    pipe = renderer_amd.InstancePipeline(max_instances=n_local, max_meshes=len(s["meshes"]), device=local_rank,
                                         stream=stream if exchange_on else None, frames_in_flight=frames)
    pipe.set_mesh_table(s["meshes"])
    pipe.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
    out_sets = [DeviceOutputs(torch, n_local, device) for _ in range(frames)]  # one set per frame in flight
    outs = out_sets[0]
    frame = make_frame(s["planes"], s["cam_pos"], first_instance_base=rank * n_local)
    torch.cuda.synchronize()

    exchange = None
    if exchange_on:
        from renderer_amd.sharded import DrawListExchange

        exchange = DrawListExchange(pipe, n_local, world, rank, device)

    if exchange is None:
        prepared = [pipe.prepare_outputs(**o.kwargs()) for o in out_sets]  # one foreign call per frame
        fref = pipe.frame_ref(frame)
        counter = [0]

        def step():
            k = counter[0]
            counter[0] = (k + 1) % frames
            pipe.run_prepared(fref, prepared[k])

        def issue_many(k):
            if k:
                pipe.run_many(frame, prepared, k)
    else:
        issue_many = None

        def step():
            exchange.step(frame, outs)

    # one checked run per output set: visible fraction + sanity
    for _ in range(frames):
        step()
    torch.cuda.synchronize()
    pipe.wait()
    if issue_many is not None:
        # untimed: lets mip_run_many record its launch graphs (hipGraphInstantiate, ~ms) even when
        # --warmup is shorter than one replay round; the timed region then only replays them
        issue_many(256)
        pipe.wait()
    if exchange is None:
        count = int(outs.scalars[0].item())
    else:
        count = exchange.local_count()
    bitmap = outs.bitmap[: (n_local + 31) // 32].cpu().numpy().view(np.uint32)
    visible = int(np.unpackbits(bitmap.view(np.uint8)).sum())
    v_emit = count / max(n_local, 1)

    dt = time_steps(torch, dist, step, args.steps, args.warmup, distributed, issue_many=issue_many)
    pipe.wait()
    ms_per_step = dt / args.steps * 1e3
    value = n_global * args.steps / dt

    result = {
        "metric": "instances/sec through transform+cull+compact",
        "value": value,
        "unit": "instances/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": cfg["workload"] + (f" x {world} shards" if world > 1 else ""),
            "instances_per_gpu": n_local,
            "instances_total": n_global,
            "meshes": int(len(s["meshes"])),
            "visible_fraction": visible / max(n_local, 1),
            "emitted_fraction": v_emit,
            "draw_list_exchange": "rccl all-gather + merge" if exchange is not None else "none (< 1 M instances or 1 GPU)",
            "frames_in_flight": frames,
            "host_loop": "compiled (mip_run_many: rounds of 64 launches replayed as hipGraphs, one chain per frame slot)" if exchange is None else "python",
            "outputs": "model[N] mat4 + visibility bitmap + compacted VkDrawIndexedIndirectCommand stream, HBM-resident",
        },
    }

    if rank == 0:
        # roofline of the dominant (only) kernel: one frame at a time on one stream (the kernel
        # alone on the chip, which is also what the rocprofv3 trace in profiles/ shows)
        if frames == 1 and exchange is None:
            serial = pipe
        else:
            serial = renderer_amd.InstancePipeline(max_instances=n_local, max_meshes=len(s["meshes"]),
                                                   device=local_rank, stream=stream)
            serial.set_mesh_table(s["meshes"])
            serial.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        serial_out = serial.prepare_outputs(**outs.kwargs())
        serial_frame = serial.frame_ref(frame)

        def kernel_only():
            serial.run_prepared(serial_frame, serial_out)

        kt = kernel_event_time(torch, kernel_only, max(args.steps, 50), args.warmup)
        serial.wait()
        result["serialized"] = {"ms_per_step": kt["mean"], "instances_per_s": n_local / (kt["mean"] * 1e-3),
                                "note": "one frame in flight: every step waits for the previous one"}
        if serial is not pipe:
            serial.close()
        bytes_per_launch = n_local * algorithmic_bytes_per_instance(v_emit)
        achieved = bytes_per_launch / (kt["mean"] * 1e-3) / 1e9
        pmc = pmc_traffic(args.config, n_local) if not args.all_visible else None
        result["roofline"] = {
            "bound": "hbm",
            "kernel": "mip_instance_pipeline_kernel",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": pmc["hbm_bytes_per_launch"] if pmc else None,
            "traffic_source": pmc["source"] if pmc else None,
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "bytes_per_instance": algorithmic_bytes_per_instance(v_emit),
            "kernel_ms_mean": kt["mean"],
            "kernel_ms_median": kt["median"],
            "kernel_ms_min": kt["min"],
            "read_only_frac": 36.0 * n_local / (kt["mean"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "frac_of_measured_copy_ceiling_6290": achieved / 6290.0,
        }
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(s, args.cpu_seconds)

    # Everything below is reported beside the headline. If a secondary leg stalls — a rank failing inside
    # the multi-rank leg would leave the others in the all-gather until NCCL's own 10-minute timeout — the
    # headline line must still come out: a watchdog prints it with what is there and ends the process.
    watchdog = None
    if not args.no_extra:
        import threading

        def bail():
            try:
                if rank == 0:
                    try:
                        line = json.dumps(dict(result, watchdog="secondary legs did not finish within 240 s; abandoned"))
                    except Exception:  # noqa: BLE001  (the main thread was writing into `extra`)
                        line = json.dumps({k: v for k, v in result.items() if k != "extra"})
                    os.write(json_fd, (line + "\n").encode())
            finally:
                os._exit(0)

        watchdog = threading.Timer(240.0, bail)
        watchdog.daemon = True
        watchdog.start()

    if not args.no_extra and not distributed and args.config == 2 and args.instances is None:
        # secondary regimes (not the headline): 1 M instances, HBM-bound; and all-visible
        extra = {}
        for label, conf, allvis in (("mixed_1m", 3, False), ("mixed_1m_all_visible", 3, True)):
            s2 = scene.make_scene(conf, all_visible=allvis)
            f2 = make_frame(s2["planes"], s2["cam_pos"])
            row = {"instances": s2["n"]}
            for nf in sorted({1, frames}):
                p2 = renderer_amd.InstancePipeline(max_instances=s2["n"], max_meshes=len(s2["meshes"]),
                                                   device=local_rank, stream=stream if nf == 1 else None,
                                                   frames_in_flight=nf)
                p2.set_mesh_table(s2["meshes"])
                p2.set_instances(s2["pos"], s2["rot"], s2["scale"], s2["mesh_id"])
                o2 = [DeviceOutputs(torch, s2["n"], device) for _ in range(nf)]
                torch.cuda.synchronize()  # torch fills on its own stream; the library does not wait for it
                kw2 = [p2.prepare_outputs(**o.kwargs()) for o in o2]
                f2r = p2.frame_ref(f2)
                c2 = [0]

                def step2():
                    k = c2[0]
                    c2[0] = (k + 1) % nf
                    p2.run_prepared(f2r, kw2[k])

                torch.cuda.synchronize()
                dt2 = time_steps(torch, dist, step2, 100, 10, False)
                p2.wait()
                v2 = int(o2[0].scalars[0].item()) / s2["n"]
                b2 = s2["n"] * algorithmic_bytes_per_instance(v2)
                key = "serialized" if nf == 1 else f"frames_in_flight_{nf}"
                row["emitted_fraction"] = v2
                row[key] = {"instances_per_s": s2["n"] * 100 / dt2, "ms_per_step": dt2 / 100 * 1e3,
                            "algorithmic_GBps": b2 / (dt2 / 100) / 1e9, "frac_of_8000": b2 / (dt2 / 100) / 1e9 / HBM_PEAK_GBS}
                p2.close()
                del o2
            extra[label] = row
        result["extra"] = extra

    if not args.no_extra and not distributed and args.config == 2 and args.instances is None and rank == 0:
        # row f-1 (next tier, not the headline): per-triangle cull + index-stream append on the same scene
        try:
            result.setdefault("extra", {})["triangle_cull"] = triangle_leg(torch, renderer_amd, scene, make_frame, s, device,
                                                                            local_rank, not args.no_cpu_baseline)
        except Exception as e:  # noqa: BLE001
            result.setdefault("extra", {})["triangle_cull"] = {"error": f"{type(e).__name__}: {e}"}

        for label, leg in (("light_draw_lists", light_leg), ("culled_views_x4", views_leg), ("skinned_256k", skinned_leg)):
            try:  # an extra leg must never cost the headline line
                result["extra"][label] = leg(torch, renderer_amd, scene, make_frame, s, device, local_rank)
            except Exception as e:  # noqa: BLE001
                result["extra"][label] = {"error": f"{type(e).__name__}: {e}"}

    if distributed and not args.no_extra:
        try:
            # the exchange regime (BASELINE config 4's shape): 1.25 M instances per rank, one RCCL
            # all-gather of the draw lists + merge per frame; reported beside the headline, not as it
            from renderer_amd.sharded import DrawListExchange

            n4 = 1_250_000
            s4 = scene.make_scene(4, n=n4, first=rank * n4)
            p4 = renderer_amd.InstancePipeline(max_instances=n4, max_meshes=len(s4["meshes"]), device=local_rank, stream=stream)
            p4.set_mesh_table(s4["meshes"])
            p4.set_instances(s4["pos"], s4["rot"], s4["scale"], s4["mesh_id"])
            o4 = DeviceOutputs(torch, n4, device)
            torch.cuda.synchronize()  # torch fills on its own stream; the library does not wait for it
            f4 = make_frame(s4["planes"], s4["cam_pos"], first_instance_base=rank * n4)
            ex = DrawListExchange(p4, n4, world, rank, device)
            ex.step(f4, o4)
            p4.wait()
            ex.tighten()
            dt_full = time_steps(torch, dist, lambda: ex.step(f4, o4), 50, 5, True)
            p4.wait()
            kw4 = o4.kwargs()
            dt_local = time_steps(torch, dist, lambda: p4.run_device(f4, async_=True, **kw4), 50, 5, True)
            p4.wait()
            counts, _ = ex.counts()
            row = {
                "instances_total": n4 * world, "instances_per_gpu": n4,
                "instances_per_s": n4 * world * 50 / dt_full, "ms_per_step": dt_full / 50 * 1e3,
                "ms_per_step_kernel_only": dt_local / 50 * 1e3,
                "chunk_bytes_per_rank": ex.stride, "commands_total": int(counts.sum()),
                "note": "kernel -> all_gather_into_tensor (RCCL) -> merge kernel, every frame",
            }
            # the same with two frames in flight (frame k+1's kernel under frame k's all-gather)
            from renderer_amd.sharded import PipelinedExchange

            def make_pipe(stream_handle):
                q = renderer_amd.InstancePipeline(max_instances=n4, max_meshes=len(s4["meshes"]), device=local_rank,
                                                  stream=stream_handle)
                q.set_mesh_table(s4["meshes"])
                q.set_instances(s4["pos"], s4["rot"], s4["scale"], s4["mesh_id"])
                return q

            px = PipelinedExchange(make_pipe, n4, world, rank, device, frames=2)
            o4s = [o4, DeviceOutputs(torch, n4, device)]
            torch.cuda.synchronize()
            for _ in range(2):
                px.step(f4, o4s)
            px.wait()
            px.tighten()
            dt_pipe = time_steps(torch, dist, lambda: px.step(f4, o4s), 50, 6, True)
            px.wait()
            px.close()
            row["frames_in_flight_2"] = {"instances_per_s": n4 * world * 50 / dt_pipe, "ms_per_step": dt_pipe / 50 * 1e3}
            if args.native_rccl_leg:
                # the same exchange without torch on the data path: the library opens RCCL itself
                # (mip_comm_init / mip_run_sharded), as a native host would drive it
                ids = [renderer_amd.InstancePipeline.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ids, src=0)
                pn = renderer_amd.InstancePipeline(max_instances=n4, max_meshes=len(s4["meshes"]), device=local_rank)
                pn.set_mesh_table(s4["meshes"])
                pn.set_instances(s4["pos"], s4["rot"], s4["scale"], s4["mesh_id"])
                pn.comm_init(ids[0], rank, world)
                merged = torch.empty((world * ex.capacity, 5), dtype=torch.int32, device=device)
                mcount = torch.zeros(2, dtype=torch.int32, device=device)
                torch.cuda.synchronize()

                def native_step():
                    pn.run_sharded(f4, merged.data_ptr(), mcount.data_ptr(), model=o4.model.data_ptr(),
                                   visible_bitmap=o4.bitmap.data_ptr(), chunk_capacity=ex.capacity, async_=True)

                dt_native = time_steps(torch, dist, native_step, 50, 5, True)
                pn.wait()
                row["native_rccl"] = {"instances_per_s": n4 * world * 50 / dt_native, "ms_per_step": dt_native / 50 * 1e3,
                                      "commands_total": int(mcount[0].item())}
                pn.comm_destroy()
                pn.close()
            if rank == 0:
                result.setdefault("extra", {})["sharded_exchange"] = row
            p4.close()
        except Exception as exc:  # the headline must survive a failure of the secondary leg
            if rank == 0:
                result.setdefault("extra", {})["sharded_exchange"] = {"error": repr(exc)}

    pipe.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(result) + "\n").encode())


if __name__ == "__main__":
    main()
