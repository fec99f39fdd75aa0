#!/usr/bin/env python3
"""bench.py — instances/sec through transform + cull + compact (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one frame: ONE pass of the hot path over the resident instance arrays, outputs written to
HBM-resident buffers, every step ordered behind the previous one (one frame in flight).

  N = 1   BASELINE configs[2]: the mixed 64-mesh scene, 1 M instances ("HBM-bound regime"), the largest
          single-GPU configuration. `value` is SURVEY.md §8(d)'s metric: one mip_run, device-resident in and
          out, bracketed by HIP events on the stream it is launched on — the MEDIAN over >= 50 samples after
          warm-up, where a sample is `--steps` back-to-back frames (so --steps only sets the batch per sample).
  N > 1   BASELINE configs[3]: the same generator at 10 M instances, STRONG scaling — rank r owns the
          contiguous draw_index range of ceil(10 M / N) instances; every step is shard kernel -> ONE RCCL
          all-gather of the compacted draw lists -> merge, all inside the timed region (barrier +
          synchronize on both sides, MAX over ranks per sample, median over samples).

Prints ONE JSON line on rank 0. Everything else (other configurations, frames in flight, launch graphs,
the next-tier rows) is reported under "extra" and never as `value`.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
MIN_SAMPLES = 50
MIN_WARMUP_SECONDS = 0.05  # untimed launches before the timed samples of an event-timed leg, in batches of 200
PARITY = "oracle (unpinned: the reference holds no tests, fixtures or golden vectors; SURVEY.md §8c)"


def algorithmic_bytes_per_instance(v):
    """read 36 B + write 64 B matrix + 1 bit + 20 B per emitted command."""
    return 100.125 + 20.0 * v


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="frames per timed sample")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--samples", type=int, default=MIN_SAMPLES, help=f"timed samples (at least {MIN_SAMPLES} at N=1)")
    ap.add_argument("--config", type=int, default=None,
                    help="override the workload: 1 Box 1k | 2 DamagedHelmet 100k | 3 mixed 1M (N=1 default) | 4 mixed 10M (N>1 default)")
    ap.add_argument("--instances", type=int, default=None, help="override the TOTAL instance count")
    ap.add_argument("--all-visible", action="store_true", help="every instance inside the frustum (worst-case writes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary legs reported under 'extra'")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--sharded-extras", action="store_true",
                    help="N>1 only: also time the exchange with two frames in flight (two contexts on two streams). Off by default: a "
                         "secondary leg that fails on ONE rank leaves the others inside a collective, and the N>1 headline is worth more "
                         "than the extra row")
    ap.add_argument("--native-rccl-leg", action="store_true",
                    help="N>1 only: also time mip_run_sharded (RCCL opened by the library itself); it creates a second "
                         "communicator beside torch's")
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


def event_samples(torch, step, steps, warmup, samples):
    """`samples` samples of `steps` back-to-back steps between two HIP events recorded on torch's current stream
    — the stream the context launches on. Returns the per-step times in ms (one per sample) and the wall-clock
    per-step times of the same samples (host launch + synchronize included)."""
    # untimed warm-up: the W steps the caller asked for, and in any case about 50 ms of launches for the
    # GPU's clocks to settle — the first hundred samples after an idle period carry a slow tail otherwise
    # (p90 19.1 us against 18.6 once warm, same median: profiles/r02_sizes_final.txt)
    done, t_warm = 0, time.perf_counter()
    while done < warmup or time.perf_counter() - t_warm < MIN_WARMUP_SECONDS:
        for _ in range(200):
            step()
        done += 200
        torch.cuda.synchronize()
    ev_ms, wall_ms = [], []
    for _ in range(samples):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(steps):
            step()
        e1.record()
        torch.cuda.synchronize()
        wall_ms.append((time.perf_counter() - t0) * 1e3 / steps)
        ev_ms.append(e0.elapsed_time(e1) / steps)
    return np.array(ev_ms), np.array(wall_ms)


def barrier_samples(torch, dist, step, steps, warmup, samples, distributed):
    """The contract's timed region, `samples` times: barrier + synchronize, EXACTLY `steps` steps, synchronize
    (+ barrier); per sample the MAX over ranks. Returns per-step ms, one per sample."""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    local = []
    for _ in range(samples):
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        local.append(time.perf_counter() - t0)
        if distributed:
            dist.barrier()
    t = torch.tensor(local, dtype=torch.float64, device="cuda")
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.cpu().numpy() * 1e3 / steps


def stats(ms):
    return {"median": float(np.median(ms)), "mean": float(ms.mean()), "min": float(ms.min()), "p90": float(np.percentile(ms, 90)),
            "samples": int(len(ms))}


def kernel_source_sha():
    """Hash of the source that defines the frame kernel's device code (instance_kernel.hpp): a PMC summary is this kernel's
    traffic only if it was collected from the same source. (Round 2 hashed mip_api.hip too, so every host-side edit made the
    committed summary "stale" although the kernel had not changed.)"""
    h = hashlib.sha256()
    for f in ("instance_kernel.hpp",):
        h.update(open(os.path.join(ROOT, "renderer_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(config, n):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*pmc_summary.json: FETCH_SIZE doubled
    per the gfx950 correction + WRITE_SIZE) — only if they were collected from THIS kernel source (the summary
    carries the source hash); a summary of an older kernel is reported as stale, never as a measurement."""
    import glob

    best, stale = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_summary.json"))):
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            continue
        for row in doc.get("workloads", []):
            if row.get("config") == config and row.get("instances") == n and row.get("variant") == "full":
                if doc.get("kernel_source_sha") == kernel_source_sha():
                    best = dict(row, source=os.path.relpath(path, ROOT))
                else:
                    stale = os.path.relpath(path, ROOT)
    return best, stale


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
    # outputs allocated and touched once, outside the timed passes (the reference's systems write into components that
    # exist already); round 2 timed oracle.run, which allocates ~85 MB of fresh output arrays per pass
    runner = oracle.Runner(*args, threads=cores, want=("model", "visible_bitmap", "draw_cmds"))
    runner()  # warm-up
    passes, t0 = 0, time.perf_counter()
    while True:
        runner()
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= seconds * 0.7 or passes >= 2000:
            break
    # the reference sizes its ComputeTaskPool at physical_core_count() / 2 (src/main.rs:881-886);
    # the same port on that many threads, shorter sample
    half = max(1, cores // 2)
    half_runner = oracle.Runner(*args, threads=half, want=("model", "visible_bitmap", "draw_cmds"))
    half_runner()
    h_passes, t1 = 0, time.perf_counter()
    while True:
        half_runner()
        h_passes += 1
        h_dt = time.perf_counter() - t1
        if h_dt >= seconds * 0.2 or h_passes >= 500:
            break
    # for the record, once: the same passes as round 2 timed them (fresh output arrays every pass)
    a_passes, t2 = 0, time.perf_counter()
    while True:
        oracle.run(*args, threads=cores, want=("model", "visible_bitmap", "draw_cmds"))
        a_passes += 1
        a_dt = time.perf_counter() - t2
        if a_dt >= seconds * 0.1 or a_passes >= 200:
            break
    return {
        "value": sample_n * passes / dt,
        "unit": "instances/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{passes} passes over {sample_n} instances of the same scene, {cores} threads, "
                  f"{dt:.1f} s wall (C oracle = CPU restatement of the reference path, gcc -O2 -ffp-contract=off; "
                  f"outputs pre-allocated and touched outside the timed passes)",
        "reference_pool_size": {"threads": half, "value": sample_n * h_passes / h_dt,
                                "note": "same port on cores/2 threads, the reference's ComputeTaskPool size (src/main.rs:881-886)"},
        "with_output_allocation_per_pass": {"value": sample_n * a_passes / a_dt,
                                            "note": "how round 2 timed it: every pass allocates and first-touches its output arrays"},
    }


def make_pipe(renderer_amd, s, local_rank, stream=None, frames=1, ordered_tiles=False):
    p = renderer_amd.InstancePipeline(max_instances=s["n"], max_meshes=len(s["meshes"]), device=local_rank,
                                      stream=stream, frames_in_flight=frames, ordered_tiles=ordered_tiles)
    p.set_mesh_table(s["meshes"])
    p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
    return p


def serialized_leg(torch, renderer_amd, make_frame, s, device, local_rank, stream, steps, warmup, samples, ordered_tiles=False):
    """SURVEY.md §8(d) on scene `s`: one frame in flight on `stream`, HIP-event samples."""
    n = s["n"]
    p = make_pipe(renderer_amd, s, local_rank, stream=stream, ordered_tiles=ordered_tiles)
    o = DeviceOutputs(torch, n, device)
    torch.cuda.synchronize()
    prepared = p.prepare_outputs(**o.kwargs())
    fref = p.frame_ref(make_frame(s["planes"], s["cam_pos"]))
    ev, wall = event_samples(torch, lambda: p.run_prepared(fref, prepared), steps, warmup, samples)
    p.wait()
    count = int(o.scalars[0].item())
    bitmap = o.bitmap[: (n + 31) // 32].cpu().numpy().view(np.uint32)
    visible = int(np.unpackbits(bitmap.view(np.uint8)).sum())
    p.close()
    return {"n": n, "count": count, "visible": visible, "event_ms": ev, "wall_ms": wall}


def leg_summary(r):
    v = r["count"] / max(r["n"], 1)
    ms = float(np.median(r["event_ms"]))
    b = r["n"] * algorithmic_bytes_per_instance(v)
    return {"instances": r["n"], "emitted_fraction": v, "ms_per_step": ms, "instances_per_s": r["n"] / (ms * 1e-3),
            "algorithmic_GBps": b / (ms * 1e-3) / 1e9, "frac_of_8000": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "timing": stats(r["event_ms"])}


def frames_in_flight_leg(torch, renderer_amd, make_frame, s, device, local_rank, frames, steps):
    """Throughput with `frames` frames in flight, issued from compiled code (mip_run_many): launch graphs when
    `steps` covers at least one round of 64, direct launches otherwise — the label says which."""
    n = s["n"]
    p = make_pipe(renderer_amd, s, local_rank, frames=frames)
    outs = [DeviceOutputs(torch, n, device) for _ in range(frames)]
    torch.cuda.synchronize()
    prepared = [p.prepare_outputs(**o.kwargs()) for o in outs]
    frame = make_frame(s["planes"], s["cam_pos"])
    p.run_many(frame, prepared, max(steps, 128))  # untimed: records the launch graphs (hipGraphInstantiate, ~ms)
    p.wait()
    p.reset_timings()
    times = []
    for _ in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p.run_many(frame, prepared, steps)
        p.wait()
        times.append((time.perf_counter() - t0) / steps * 1e3)
    t = p.timings()
    replayed = t["graph_frames"]
    v = int(outs[0].scalars[0].item()) / n
    p.close()
    ms = float(np.median(times))
    b = n * algorithmic_bytes_per_instance(v)
    return {"instances": n, "frames_in_flight": frames, "steps_per_sample": steps, "ms_per_step": ms, "instances_per_s": n / (ms * 1e-3),
            "algorithmic_GBps": b / (ms * 1e-3) / 1e9, "frac_of_8000": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "host_loop": (f"compiled (mip_run_many): {replayed} of {7 * steps} timed frames replayed as hipGraphs, the rest direct launches"
                          if replayed else "compiled (mip_run_many): direct launches (no replay round of 64 frames fits the steps, or launch graphs are off)"),
            "note": "wall clock around the call incl. the final wait; overlapping frames: a throughput figure, not the §8(d) metric"}


def light_leg(torch, renderer_amd, scene, make_frame, s, device, local_rank):
    """Row f-4 (shadow pass): per-light draw lists for the 4 lights the reference spawns (main.rs:368-382)."""
    n = s["n"]
    pl = make_pipe(renderer_amd, s, local_rank)
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


def views_pmc(n):
    """Counters per launch of the multi-view kernel from the committed PMC pass (tools/pmc_views.sh), if it was taken from
    this source of views_kernel.hpp."""
    import glob

    sha = hashlib.sha256(open(os.path.join(ROOT, "renderer_amd", "csrc", "views_kernel.hpp"), "rb").read()).hexdigest()[:16]
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*views*pmc_summary.json")), reverse=True):
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            continue
        if doc.get("instances") == n and doc.get("views") == 4 and doc.get("views_source_sha") == sha:
            return doc, os.path.relpath(path, ROOT)
    return None, None


def views_leg(torch, renderer_amd, scene, make_frame, s, device, local_rank, stream):
    """Row f-4, "per-light cull lists": four culled views (the reference's light positions as LOD reference
    points, the default frustum moved to each) of the scene in one launch, mip_run_views; HIP-event samples on the
    launch stream."""
    n = s["n"]
    p = make_pipe(renderer_amd, s, local_rank, stream=stream)
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
    ev, wall = event_samples(torch, lambda: p.run_views(frames, outs), 20, 20, 30)
    p.wait()
    counts = [int(k[1][0].item()) for k in keep]
    p.close()
    ms = float(np.median(ev))
    nbytes = 36 * n + sum(20 * c for c in counts) + len(eyes) * n / 8
    row = {"instances": n, "views": len(eyes), "ms_per_launch": ms, "instance_views_per_s": n * len(eyes) / (ms * 1e-3),
           "commands_per_view": counts, "algorithmic_GBps": nbytes / (ms * 1e-3) / 1e9, "timing": stats(ev),
           "wall_clock_ms_per_launch": float(np.median(wall)),
           "note": "one launch: instance data read once, matrix + world box built once, per view plane test + compaction; no matrices "
                   "written (the frame's mip_run writes them); bound by VALU issue, not HBM"}
    doc, src = views_pmc(n)
    if doc and doc["counters_per_launch"].get("SQ_INSTS_VALU"):
        valu = doc["counters_per_launch"]["SQ_INSTS_VALU"]
        row["roofline"] = {
            "bound": "valu", "kernel": "mip_cull_views_kernel", "achieved": valu / (ms * 1e-3), "peak": VALU_PEAK_WAVE_INSTR_PER_S, "peak_source": VALU_PEAK_SOURCE,
            "unit": "wave-instructions/s", "frac": valu / (ms * 1e-3) / VALU_PEAK_WAVE_INSTR_PER_S,
            "valu_wave_instructions_per_launch": valu, "kernel_ms": ms, "kernel_ms_under_pmc": doc["kernel_ns_under_pmc"] * 1e-6,
            "traffic": doc.get("hbm_bytes_per_launch"), "algorithmic_bytes_per_launch": nbytes, "source": src,
            "note": "SQ_INSTS_VALU per launch (committed PMC pass of this kernel source) / this run's launch time; peak = the MEASURED issue "
                    "ceiling of plain f32 wave64 instructions (peak_source; the kernel's translation unit is built without the SLP "
                    "vectoriser: no packed instructions)",
        }
    return row


def skinned_pmc(n, joints):
    """HBM bytes per skinned frame (skinning kernel + frame kernel) from the committed PMC passes (tools/pmc_skin.sh), if they were
    taken from these sources of skinning_kernel.hpp and instance_kernel.hpp."""
    import glob

    h = hashlib.sha256()
    for f in ("skinning_kernel.hpp", "instance_kernel.hpp"):
        h.update(open(os.path.join(ROOT, "renderer_amd", "csrc", f), "rb").read())
    sha = h.hexdigest()[:16]
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*skinned*pmc_summary.json")), reverse=True):
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            continue
        if doc.get("instances") == n and doc.get("joints") == joints and doc.get("source_sha") == sha:
            return doc, os.path.relpath(path, ROOT)
    return None, None


def skinned_leg(torch, renderer_amd, scene, make_frame, s_unused, device, local_rank):
    """BASELINE config 5 (extension, no reference semantics): 256 k instances of a 19-joint figure, each
    with its own pose: palette + skinned bounds kernel, then the instance kernel."""
    s = scene.make_skinned_scene()
    n, j = s["n"], len(s["skeleton"]["parent"])
    p = make_pipe(renderer_amd, s, local_rank)
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
    skin_bytes = n * (j * 40 + j * 64 + 32)  # the skinning kernel alone: poses in, palette + posed box out
    pmc_doc, pmc_src = skinned_pmc(n, j)
    return {
        "instances": n, "joints": j, "ms_per_frame": dt * 1e3, "instances_per_s": n / dt, "emitted_fraction": count / n,
        "algorithmic_GBps": nbytes / dt / 1e9,
        "roofline": {"bound": "hbm", "kernel": "mip_skinned_bounds_kernel + mip_instance_pipeline_kernel (one frame)", "achieved": nbytes / dt / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / dt / 1e9 / HBM_PEAK_GBS,
                     "traffic": pmc_doc["hbm_bytes_per_frame"] if pmc_doc else None, "traffic_source": pmc_src,
                     "per_kernel": pmc_doc.get("per_kernel") if pmc_doc else None,
                     "algorithmic_bytes_per_frame": nbytes, "skinning_kernel_bytes": skin_bytes,
                     "note": "wall clock per frame over 50 back-to-back frames (two kernels per frame); traffic = 2 * FETCH_SIZE + WRITE_SIZE of BOTH "
                             "kernels of a frame from the committed PMC passes of these kernel sources (tools/pmc_skin.sh), null when the sources have changed since"},
        "note": "per instance: 19 x (40 B pose read + 64 B palette written) in the skinning kernel, then the instance kernel's "
                "100 B + 20 B per command; the reference has no skinning: parity is against this repository's oracle only",
    }


# The VALU issue ceiling, MEASURED (tools/micro/valu_issue.hip -> profiles/r04_valu_issue.txt): with every CU busy and 8 waves
# per SIMD the chip retires 9.6-10.0e11 plain f32 wave64 instructions per second (v_mul / v_add / v_fma and their mix) — one
# per 2 cycles per SIMD at the 2.0-2.25 GHz the clock settles at under that load (the runtime reports 2.4 GHz); one wave
# alone issues one per 4.5 cycles, which is the constant round 3 mistook for the SIMD's ceiling (its 6.1e11 "peak" was
# exceeded by a leg that printed frac 1.002); a packed v_pk_*_f32 counts as two. Round 2's scale was the right one.
VALU_PEAK_WAVE_INSTR_PER_S = 9.8e11
VALU_PEAK_SOURCE = "profiles/r04_valu_issue.txt: plain f32, 8 waves per SIMD, every CU busy, whole-launch rate; clock under load 2.0-2.25 GHz"


def triangle_pmc(config, n, ordering="rows"):
    """Counters per launch of the triangle kernel from the committed PMC pass, if it was taken from this source of
    triangle_kernels.hpp (and this mesh layout)."""
    import glob

    sha = hashlib.sha256(open(os.path.join(ROOT, "renderer_amd", "csrc", "triangle_kernels.hpp"), "rb").read()).hexdigest()[:16]
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*triangle*pmc_summary.json")), reverse=True):
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            continue
        if (doc.get("config") == config and doc.get("instances") == n and doc.get("triangle_source_sha") == sha
                and doc.get("ordering", "rows") == ordering):
            return doc, os.path.relpath(path, ROOT)
    return None, None


def triangle_leg(torch, renderer_amd, scene, make_frame, s, device, local_rank, with_cpu, config=2, ordering="rows"):
    """generate_work.comp:68-200 for every emitted command of scene `s` (synthetic torus geometry with the mesh
    table's triangle counts, listed row by row or as first-use-ordered strips): frame = instance kernel + triangle
    kernel + re-compaction."""
    n = s["n"]
    vertices, indices = scene.make_geometry(s["meshes"], ordering=ordering)
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
        "mesh_layout": ordering,
        "note": "bound by VALU issue time, not HBM: two mat4*vec4 per vertex without FMA per triangle corner (the reference's "
                "pv * (model * vec4(v, 1)), generate_work.comp:132-136); geometry is L2-resident, HBM traffic is the 12 B per "
                "surviving triangle",
    }
    doc, src = triangle_pmc(config, n, ordering)
    if doc and doc["counters_per_launch"].get("SQ_INSTS_VALU"):
        valu = doc["counters_per_launch"]["SQ_INSTS_VALU"]
        kern_ns = list(doc["kernel_ns_under_pmc"].values())[0]
        row["roofline"] = {
            "bound": "valu", "kernel": list(doc["kernel_ns_under_pmc"].keys())[0], "achieved": valu / (kern_ns * 1e-9),
            "peak": VALU_PEAK_WAVE_INSTR_PER_S, "peak_source": VALU_PEAK_SOURCE, "unit": "wave-instructions/s",
            "frac": valu / (kern_ns * 1e-9) / VALU_PEAK_WAVE_INSTR_PER_S,
            "valu_wave_instructions_per_launch": valu, "kernel_ms": kern_ns * 1e-6, "source": src,
            "note": "SQ_INSTS_VALU per launch / the kernel's duration in the same rocprofv3 pass; peak = the MEASURED issue ceiling of plain "
                    "f32 wave64 instructions, one per 2 cycles per SIMD at the clock the chip holds under load (peak_source); the kernel has no "
                    "packed instructions (its translation unit is built without the SLP vectoriser). Round 3 printed 0.98 here against a ceiling "
                    "of one instruction per 4 cycles, which is what ONE wave sustains, not what the SIMD does",
        }
        c = doc["counters_per_launch"]
        if c.get("SQ_INSTS_VMEM_RD") and c.get("GRBM_GUI_ACTIVE"):
            # second reading from the same PMC pass: how busy the vector-load path is. Ceilings for the SAME instruction
            # (global_load_dwordx3, 12 B per lane) measured by tools/micro/l1_line_rate.hip (profiles/r04_l1_line_rate.txt), in
            # wave-loads per cycle per CU at 8 waves per SIMD: 0.114 when every load hits the L1 (any stride up to 48 B);
            # L2-served: 0.134 consecutive vec3, 0.039 every 4th vertex, 0.016-0.019 a line per lane.
            cycles = c["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
            rate = c["SQ_INSTS_VMEM_RD"] / 256.0 / cycles
            row["roofline"]["vector_load_path"] = {
                "wave_loads_per_cycle_per_cu": rate, "shader_cycles": cycles,
                "l1_accesses_per_wave_load": (c["TCP_TOTAL_CACHE_ACCESSES_sum"] / c["SQ_INSTS_VMEM_RD"]) if c.get("TCP_TOTAL_CACHE_ACCESSES_sum") else None,
                "l1_hit_rate": (1.0 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]) if c.get("TCP_TCC_READ_REQ_sum") and c.get("TCP_TOTAL_CACHE_ACCESSES_sum") else None,
                "measured_ceilings_wave_loads_per_cycle_per_cu": {"l1_hits": 0.114, "l2_consecutive_vec3": 0.134, "l2_every_4th_vertex": 0.039, "l2_a_line_per_lane": 0.017},
                "note": "the gathers use 0.22-0.72 of what the load path sustains for comparable access patterns: neither VALU issue (frac above) nor the "
                        "load path is saturated, and deeper pipelining of the gathers changes nothing (profiles/r04_triangle_bound_experiments.txt)",
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


def spawn_ranks(args):
    """`python bench.py --gpus N` (N > 1) started WITHOUT a launcher: become the launcher. Runs
    `python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>` as a CHILD process — this
    process has made no GPU call yet and never replaces itself (no exec) — lets rank 0's JSON line through on
    stdout and returns the launcher's exit code, so one failed rank fails the whole run."""
    import socket
    import subprocess

    import torch  # device_count() does not initialise the GPU

    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s); no line printed\n")
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: no WORLD_SIZE in the environment, launching the ranks: " + " ".join(cmd) + "\n")
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    args.steps = max(1, args.steps)
    args.warmup = max(0, args.warmup)
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and os.environ.get("MIP_BENCH_FORCE_DIST") != "1":
        sys.exit(spawn_ranks(args))
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
    if os.environ.get("MIP_BENCH_DEVICE") is not None:  # rehearsal only: several ranks on one GPU
        local_rank = int(os.environ["MIP_BENCH_DEVICE"])
    distributed = world > 1
    if os.environ.get("MIP_BENCH_FORCE_DIST") == "1":  # rehearsal of the N>1 code path with one rank
        distributed = True
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if args.gpus != world and os.environ.get("MIP_BENCH_FORCE_DIST") != "1":
        # n_gpus in the line is the world size: a line whose n_gpus is not what was asked for is never printed
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE {world}; no line printed")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the instance pipeline has no CPU path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        # "nccl" IS RCCL on ROCm. MIP_BENCH_BACKEND=gloo exists only to rehearse the N>1 control flow with several
        # ranks on ONE GPU (RCCL refuses two ranks on one device); its numbers mean nothing.
        backend = os.environ.get("MIP_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    if not os.path.exists(renderer_amd.library_path()):  # bare checkout: build the HIP library first
        import __graft_entry__

        __graft_entry__.build()
    renderer_amd.load_library()

    # The context launches on torch's current stream (a real, non-null one), so that the HIP events recorded
    # by torch bracket exactly the launches, and the sharded path's kernels stay ordered with torch's NCCL ops.
    torch_stream = torch.cuda.Stream(device=device)
    torch.cuda.set_stream(torch_stream)
    stream = torch_stream.cuda_stream

    result = {
        "metric": "instances/sec through transform+cull+compact",
        "unit": "instances/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "higher_is_better": True,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "parity": PARITY,
    }
    if distributed:
        result["backend"] = "nccl (RCCL)" if os.environ.get("MIP_BENCH_BACKEND", "nccl") == "nccl" else os.environ["MIP_BENCH_BACKEND"] + " (REHEARSAL: not a measurement)"
    extra = {}

    if not distributed:
        run_single(args, torch, renderer_amd, scene, make_frame, device, local_rank, stream, result, extra)
    else:
        run_sharded(args, torch, dist, renderer_amd, scene, make_frame, device, local_rank, rank, world, stream, result, extra)

    # Everything below is reported beside the headline. If a secondary leg stalls the headline line must still
    # come out: a watchdog prints it with what is there and ends the process with a NON-ZERO code (a process that
    # has given up on a GPU stall must not look like a clean run).
    watchdog = None
    if not args.no_extra:
        import threading

        def bail():
            try:
                if rank == 0:
                    try:
                        line = json.dumps(dict(result, extra=extra, watchdog="secondary legs did not finish within 300 s; abandoned, exit code 3"))
                    except Exception:  # noqa: BLE001  (the main thread was writing into `extra`)
                        line = json.dumps(dict(result, watchdog="secondary legs did not finish within 300 s; abandoned, exit code 3"))
                    os.write(json_fd, (line + "\n").encode())
            finally:
                os._exit(3)

        watchdog = threading.Timer(300.0, bail)
        watchdog.daemon = True
        watchdog.start()

    if not args.no_extra and not distributed and args.config is None and args.instances is None and not args.all_visible:
        secondary_single(args, torch, renderer_amd, scene, make_frame, device, local_rank, stream, extra)
    if not args.no_extra and distributed and args.config is None and args.instances is None and (args.sharded_extras or args.native_rccl_leg or world == 1):
        secondary_sharded(args, torch, dist, renderer_amd, scene, make_frame, device, local_rank, rank, world, stream, extra)

    if extra:
        result["extra"] = extra
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(result) + "\n").encode())


def run_single(args, torch, renderer_amd, scene, make_frame, device, local_rank, stream, result, extra):
    config = 3 if args.config is None else args.config
    cfg = scene.CONFIGS[config]
    n = args.instances if args.instances is not None else cfg["n"]
    s = scene.make_scene(config, n=n, all_visible=args.all_visible)
    samples = max(args.samples, MIN_SAMPLES)
    r = serialized_leg(torch, renderer_amd, make_frame, s, device, local_rank, stream, args.steps, args.warmup, samples)
    ms = float(np.median(r["event_ms"]))
    v_emit = r["count"] / max(n, 1)
    result.update({
        "value": n / (ms * 1e-3),
        "ms_per_step": ms,
        "scaling": "weak",
        "config": {
            "workload": cfg["workload"],
            "baseline_config": f"BASELINE.json configs[{config - 1}]",
            "instances_per_gpu": n,
            "instances_total": n,
            "meshes": int(len(s["meshes"])),
            "visible_fraction": r["visible"] / max(n, 1),
            "emitted_fraction": v_emit,
            "draw_list_exchange": "none (1 GPU)",
            "frames_in_flight": 1,
            "host_loop": "python: one mip_run per step (direct launches on one stream, each ordered behind the previous one)",
            "outputs": "model[N] mat4 + visibility bitmap + compacted VkDrawIndexedIndirectCommand stream, HBM-resident",
        },
        "timing": dict(stats(r["event_ms"]), method=f"HIP events on the launch stream around {args.steps} back-to-back steps per sample; "
                                                      f"value = N / median sample", wall_clock_ms_per_step=stats(r["wall_ms"])),
    })
    bytes_per_launch = n * algorithmic_bytes_per_instance(v_emit)
    achieved = bytes_per_launch / (ms * 1e-3) / 1e9
    pmc, stale = pmc_traffic(config, n) if not args.all_visible else (None, None)
    result["roofline"] = {
        "bound": "hbm",
        "kernel": "mip_instance_pipeline_kernel",
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "traffic": pmc["hbm_bytes_per_launch"] if pmc else None,
        "traffic_source": pmc["source"] if pmc else (f"stale: {stale} was collected from an older kernel source" if stale else None),
        "algorithmic_bytes_per_launch": bytes_per_launch,
        "bytes_per_instance": algorithmic_bytes_per_instance(v_emit),
        "kernel_ms": ms,
        "kernel_ms_min": float(r["event_ms"].min()),
        "kernel_source_sha": kernel_source_sha(),
        "read_only_frac": 36.0 * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "frac_of_measured_copy_ceiling_6290": achieved / 6290.0,
        "note": "one launch = one step: kernel_ms is the median per-step time of the timed samples themselves; "
                "profiles/ holds the rocprofv3 --kernel-trace --stats summary of the same command",
    }
    if not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(s, args.cpu_seconds)


def secondary_single(args, torch, renderer_amd, scene, make_frame, device, local_rank, stream, extra):
    def guarded(label, fn):
        try:  # an extra leg must never cost the headline line
            extra[label] = fn()
        except Exception as e:  # noqa: BLE001
            extra[label] = {"error": f"{type(e).__name__}: {e}"}

    s3 = scene.make_scene(3)
    s2 = scene.make_scene(2)
    guarded("frames_in_flight_2_1m", lambda: frames_in_flight_leg(torch, renderer_amd, make_frame, s3, device, local_rank, 2, 256))
    guarded("damaged_helmet_100k_serialized", lambda: leg_summary(serialized_leg(
        torch, renderer_amd, make_frame, s2, device, local_rank, stream, args.steps, args.warmup, MIN_SAMPLES)))
    guarded("damaged_helmet_100k_frames_in_flight_2", lambda: frames_in_flight_leg(torch, renderer_amd, make_frame, s2, device, local_rank, 2, 1024))
    guarded("mixed_1m_all_visible_serialized", lambda: leg_summary(serialized_leg(
        torch, renderer_amd, make_frame, scene.make_scene(3, all_visible=True), device, local_rank, stream, args.steps, args.warmup, MIN_SAMPLES)))
    def ten_million():
        # the honest HBM figure (the 1 M frame's 105 MB are helped by the memory-side cache): with the PMC traffic of the same kernel source at 10 M
        row = leg_summary(serialized_leg(torch, renderer_amd, make_frame, scene.make_scene(4), device, local_rank, stream, 5, 5, 20))
        pmc4, stale4 = pmc_traffic(4, scene.CONFIGS[4]["n"])
        row["algorithmic_bytes_per_launch"] = row["instances"] * algorithmic_bytes_per_instance(row["emitted_fraction"])
        row["traffic"] = pmc4["hbm_bytes_per_launch"] if pmc4 else None
        row["traffic_source"] = pmc4["source"] if pmc4 else (f"stale: {stale4} was collected from an older kernel source" if stale4 else None)
        return row

    guarded("mixed_10m_one_gpu_serialized", ten_million)
    # Since ABI 4 EVERY launch is independent of the order workgroups start in (a tile whose predecessor has not published computes
    # that aggregate itself): there is no separate "ordered tiles" mode any more. The two keys round 3 reported for it are kept, and
    # measure the one kernel there is, on a context created with the (now ignored) flag — so they can be read against round 3's
    # 25.45 / 7.55 us and against the default legs above.
    guarded("mixed_1m_ordered_tiles_serialized", lambda: dict(leg_summary(serialized_leg(
        torch, renderer_amd, make_frame, s3, device, local_rank, stream, args.steps, args.warmup, MIN_SAMPLES, ordered_tiles=True)),
        note="MIP_CFG_ORDERED_TILES is accepted and ignored: this is the default kernel, which no longer depends on dispatch order "
             "(decoupled look-back with a fallback; round 3: three wait-free launches, 25.45 us)"))
    guarded("damaged_helmet_100k_ordered_tiles_serialized", lambda: dict(leg_summary(serialized_leg(
        torch, renderer_amd, make_frame, s2, device, local_rank, stream, args.steps, args.warmup, MIN_SAMPLES, ordered_tiles=True)),
        note="as above (round 3: two launches, 7.55 us)"))
    guarded("mixed_1m_scrambled_dispatch", lambda: scrambled_dispatch_leg())
    guarded("zero_copy_semaphore_frame", lambda: semaphore_leg())
    guarded("shard_merge_8x336k", lambda: merge_leg(torch, renderer_amd, scene, make_frame, device, local_rank, stream))
    # next-tier rows (not the headline)
    guarded("triangle_cull_100k", lambda: triangle_leg(torch, renderer_amd, scene, make_frame, s2, device, local_rank, not args.no_cpu_baseline))
    guarded("triangle_cull_100k_strips", lambda: triangle_leg(torch, renderer_amd, scene, make_frame, s2, device, local_rank, False, ordering="strips"))

    def mixed_triangles():
        # the mixed 64-mesh scene at 100 k instances: commands from 12 to 23 k triangles; the stage launches both of its large-frame
        # grids and the range kernel takes the frame (plan_tri_choice_is_ranges; the size-sorted wave-per-command kernel: 0.45-0.47 ms)
        row = triangle_leg(torch, renderer_amd, scene, make_frame, scene.make_scene(3, n=100_000), device, local_rank, False, config=3)
        row["note"] = ("mixed scene: the largest commands' walk is long against a wave's share of the frame and far above the mean command, so the range "
                       "kernel (equal ranges of the triangle stream, commands cut between waves) is chosen on the device (profiles/r05_triangle_stage_modes.txt)")
        return row

    guarded("triangle_cull_mixed_100k", mixed_triangles)
    guarded("light_draw_lists", lambda: light_leg(torch, renderer_amd, scene, make_frame, s3, device, local_rank))
    guarded("culled_views_x4", lambda: views_leg(torch, renderer_amd, scene, make_frame, s3, device, local_rank, stream))
    guarded("skinned_256k", lambda: skinned_leg(torch, renderer_amd, scene, make_frame, None, device, local_rank))


def scrambled_dispatch_leg():
    """The degraded mode, tracked: the frame kernel when workgroups do NOT start in tile order (another tenant of the GPU, RCCL's own
    kernels beside a shard kernel). The diagnostic build of the library numbers its tiles by a scrambling permutation of the workgroup
    index (MIP_DEBUG_TILE_ORDER=scramble; never the product), so the tiles that start first miss predecessors that are not even resident
    and compute their aggregates themselves; serialized launches, HIP events, tools/kbench.py as a child process (the library is chosen at
    import). Rows: 1 M and 2.5 M instances of the mixed scene, scrambled (as the product behaves, and with the first-mover rule switched
    off), and 1 M in order from the same build for comparison."""
    import re
    import subprocess

    dbg = os.path.join(ROOT, "renderer_amd", "lib", "libmi_instance_pipeline_dbg.so")
    if not os.path.exists(dbg):
        return {"error": "renderer_amd/lib/libmi_instance_pipeline_dbg.so has not been built (__graft_entry__.build())"}
    rows = {}
    for label, order, rule in (("scrambled", "scramble", None), ("scrambled_rule_never", "scramble", "never"), ("in_order_same_build", None, None)):
        env = dict(os.environ)
        env.pop("MIP_DEBUG_TILE_ORDER", None)
        env.pop("MIP_TUNE_FIRST_MOVER", None)
        if order:
            env["MIP_DEBUG_TILE_ORDER"] = order
        if rule:
            env["MIP_TUNE_FIRST_MOVER"] = rule
        sizes = "1000000,2500000" if order else "1000000"
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kbench.py"), "--configs", ",".join("3" for _ in sizes.split(",")), "--n", sizes,
                              "--libs", dbg], capture_output=True, text=True, timeout=300, env=env)
        for line in out.stdout.split("\n"):
            m = re.search(r"cfg3 n=(\d+)\s+full\s+median\s+([\d.]+) us\s+min\s+([\d.]+)\s+p90\s+([\d.]+)", line)
            if m:
                rows[f"{label}_{m.group(1)}"] = {"ms_per_step": float(m.group(2)) * 1e-3, "min_ms": float(m.group(3)) * 1e-3, "p90_ms": float(m.group(4)) * 1e-3}
        if out.returncode != 0:
            rows[f"{label}_error"] = (out.stderr or out.stdout)[-400:]
    rows["note"] = ("diagnostic build, tiles numbered by a scrambled permutation of the workgroup index: what a frame costs when the hardware starts "
                    "workgroups in an order the launch did not ask for; round 4: 0.30-0.40 ms at 1 M, 1.0-1.1 ms at 2.5 M (host wall clock, "
                    "profiles/r04_selfhelp_any_order.txt); round 5 (claims): profiles/r05_selfhelp_claims.txt. `scrambled`: the product's "
                    "behaviour — after a launch that had to help, the next launches follow the first-mover rule (tiles mark themselves STARTED, "
                    "helping waves complete the group accumulators: profiles/r05_first_mover.txt); `scrambled_rule_never`: the rule of rounds 3-4 "
                    "for every launch (MIP_TUNE_FIRST_MOVER=never)")
    return rows


def semaphore_leg():
    """Row f-2: what ordering a frame against a Vulkan queue by exported timeline semaphores costs per frame. Runs
    renderer_amd/lib/mip_semaphore_bench (tools/micro/semaphore_frames.cpp: frames issued from compiled code against two kernel
    DRM timeline sync objects, what vkGetSemaphoreFdKHR exports on amdgpu) as a child process, for 1 M and 100 k instances."""
    import subprocess

    exe = os.path.join(ROOT, "renderer_amd", "lib", "mip_semaphore_bench")
    if not os.path.exists(exe):
        return {"error": "renderer_amd/lib/mip_semaphore_bench has not been built (make -C renderer_amd/host)"}
    rows = {}
    for n in (1_000_000, 100_000):
        out = subprocess.run([exe, str(n), "2000"], capture_output=True, text=True, timeout=120)
        line = [l for l in out.stdout.split("\n") if l.startswith("{")]
        rows[f"{n}"] = json.loads(line[-1]) if out.returncode == 0 and line else {"error": (out.stderr or out.stdout)[-400:]}
    rows["note"] = ("microseconds per frame, wall clock over 2 000 frames from compiled code: bare = mip_run(ASYNC) back to back; free_running = "
                    "mip_wait_external -> mip_run -> mip_signal_external per frame with the consumer timeline already ahead (frames in flight: "
                    "the hand-over's own cost); ping_pong = every frame waits for a consumer THREAD that has seen the previous frame's signal "
                    "(two kernel wake-ups per frame). profiles/r04_external_semaphore_handover.txt has round 3's host-function path beside it")
    return rows


def merge_leg(torch, renderer_amd, scene, make_frame, device, local_rank, stream):
    """Row e, the shard merge alone, on the 8-rank shape of BASELINE configs[3] (8 chunks of a 1.25 M-instance shard's list): back to
    back on one stream, HIP events, both wire forms; bytes = wire bytes read + 20 B written per command."""
    from renderer_amd.pipeline import SHARD_HEADER_BYTES
    from renderer_amd.sharded import chunk_stride_bytes

    n, ranks, cap = 1_250_000, 8, 360_000
    s = scene.make_scene(4, n=n)
    p = make_pipe(renderer_amd, s, local_rank, stream=stream)
    rows = {}
    for form, name, per_cmd in (("packed", "packed_wire", 4.25 + 20), (True, "wire_8_byte", 8.0625 + 20)):
        stride = chunk_stride_bytes(cap, wire=form)
        recv = torch.zeros(ranks * stride // 4, dtype=torch.int32, device=device)
        merged = torch.zeros((ranks * cap, 5), dtype=torch.int32, device=device)
        oc = torch.zeros(2, dtype=torch.int32, device=device)
        torch.cuda.synchronize()
        for k in range(ranks):
            base = recv.data_ptr() + k * stride
            p.run_device(make_frame(s["planes"], s["cam_pos"], first_instance_base=k * n), draw_cmds=base + SHARD_HEADER_BYTES,
                         draw_count=base, draw_index_total=base + 4, wire=form)
        count = int(recv[0].item())

        def step():
            p.merge_wire_lists(recv.data_ptr(), ranks, stride, merged.data_ptr(), oc.data_ptr(), chunk_capacity=cap, packed=(form == "packed"), async_=True)

        ev, _ = event_samples(torch, step, 20, 20, 20)
        p.wait()
        ms = float(np.median(ev))
        nbytes = ranks * count * per_cmd
        rows[name] = {"chunks": ranks, "commands_per_chunk": count, "ms_per_merge": ms, "algorithmic_GBps": nbytes / (ms * 1e-3) / 1e9,
                      "frac_of_8000": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "merged_commands": int(oc[0].item()), "timing": stats(ev)}
    p.close()
    rows["note"] = ("mip_merge_wire_lists[_packed] alone, 20 back-to-back merges per HIP-event sample (as the step issues it: behind the "
                    "all-gather, the GPU busy); round 3's kernel: 16.8 / 15.9 us in the kernel trace, this round's: profiles/r04_wire_merge.txt")
    return rows


def run_sharded(args, torch, dist, renderer_amd, scene, make_frame, device, local_rank, rank, world, stream, result, extra):
    from renderer_amd.sharded import DrawListExchange, shard_range

    config = 4 if args.config is None else args.config
    cfg = scene.CONFIGS[config]
    n_total = args.instances if args.instances is not None else cfg["n"]
    lo, hi = shard_range(n_total, world, rank)
    n_local = hi - lo
    s = scene.make_scene(config, n=n_local, first=lo, all_visible=args.all_visible)
    pipe = make_pipe(renderer_amd, s, local_rank, stream=stream)
    outs = DeviceOutputs(torch, n_local, device)
    frame = make_frame(s["planes"], s["cam_pos"], first_instance_base=lo)
    torch.cuda.synchronize()
    # MIP_BENCH_WIRE: A/B of the exchanged form — unset / 2: packed 4-byte records when the largest shard fits (else 8-byte),
    # 1: 8-byte records, 0: 20-byte commands
    wire_env = os.environ.get("MIP_BENCH_WIRE", "2")
    ex = DrawListExchange(pipe, n_local, world, rank, device, wire={"0": False, "1": 1}.get(wire_env, True))
    def timed():
        out = barrier_samples(torch, dist, lambda: ex.step(frame, outs), args.steps, args.warmup, samples, True)
        ex.complete()
        return out

    # (round 3 carried a collective "did anybody time out" protocol here: a frame kernel's bounded wait could expire when several
    #  processes shared one GPU. No kernel waits for another workgroup any more, so a frame cannot time out; what is reported
    #  instead is how many tile aggregates waiting tiles computed themselves — 0 on GPUs the ranks have to themselves.)
    ex.step(frame, outs)
    ex.complete()
    capacity_full = ex.capacity
    ex.tighten()   # the exchanged chunk = the largest shard list of the first frame + 6 %; an overflow is repaired, not lost
    ex.step(frame, outs)
    ex.complete()
    counts, _ = ex.counts()
    samples = max(10, min(args.samples, 30))
    ms = timed()
    helps = torch.tensor([pipe.timings()["prefix_helps"]], dtype=torch.int64, device="cuda")
    dist.all_reduce(helps, op=dist.ReduceOp.MAX)
    med = float(np.median(ms))
    result.update({
        "value": n_total / (med * 1e-3),
        "ms_per_step": med,
        "scaling": "strong",
        "config": {
            "workload": cfg["workload"] + f" — {n_total} instances in {world} contiguous shards, one RCCL all-gather of the draw lists + merge per step",
            "baseline_config": f"BASELINE.json configs[{config - 1}]",
            "instances_per_gpu": n_local if world == 1 else (n_total + world - 1) // world,
            "instances_total": n_total,
            "meshes": int(len(s["meshes"])),
            "commands_total": int(counts.sum()),
            "emitted_fraction": float(counts.sum()) / max(n_total, 1),
            "draw_list_exchange": "rccl all-gather (torch.distributed nccl backend) + merge kernel, inside the timed region",
            "chunk_bytes_per_rank": int(ex.stride),
            "chunk_format": {0: "20-byte commands",
                             1: "wire: 8-byte records {firstInstance, mesh | lod} in blocks of 256 (MIP_OUT_WIRE), expanded by the merge",
                             2: "packed wire: 4-byte records {instance index | mesh | lod} in blocks of 64 (MIP_OUT_WIRE_PACKED, 4.25 B per command), expanded by the merge"}[ex.form],
            "chunk_bytes_per_rank_as_20_byte_commands": int((32 + ex.capacity * 20 + 255) // 256 * 256),
            "chunk_capacity_commands": int(ex.capacity),
            "chunk_capacity_untightened": int(capacity_full),
            "n_ranks_seen": int(dist.get_world_size()),
            "prefix_helps_max_over_ranks": int(helps.item()),
            "frames_in_flight": 1,
            "host_loop": "python: kernel -> all_gather_into_tensor -> merge per step on one stream",
            "outputs": "per rank: its shard's model[] + bitmap; every rank: the merged global draw list",
        },
        "timing": dict(stats(ms), method=f"barrier + synchronize around EXACTLY {args.steps} steps per sample, MAX over ranks per sample, "
                                         f"value = N_total / median sample"),
    })

    if True:  # collective legs: every rank takes part, rank 0 reports
        kw = outs.kwargs()
        k_ms = barrier_samples(torch, dist, lambda: pipe.run_device(frame, async_=True, **kw), args.steps, args.warmup, 10, True)
        pipe.wait()
        g_ms = barrier_samples(torch, dist, lambda: dist.all_gather_into_tensor(ex.recv, ex.send), args.steps, 3, 10, True)
        merge_fn = pipe.merge_wire_lists if ex.form else pipe.merge_draw_lists
        merge_kw = {"packed": True} if ex.form == 2 else {}
        m_ms = barrier_samples(torch, dist, lambda: merge_fn(ex.recv.data_ptr(), world, ex.stride, ex.merged.data_ptr(),
                                                             ex.merged_count.data_ptr(), async_=True, chunk_capacity=ex.capacity, **merge_kw),
                               args.steps, 3, 10, True)
        pipe.wait()
        result["breakdown_ms_per_step"] = {
            "shard_kernel_only": float(np.median(k_ms)), "all_gather_only": float(np.median(g_ms)), "merge_only": float(np.median(m_ms)),
            "note": "each leg alone, same barrier-bracketed timing, MAX over ranks; the step runs them back to back on one stream",
        }
        local_v = float(counts[rank]) / max(n_local, 1)
        kb = n_local * algorithmic_bytes_per_instance(local_v)
        pmc3, stale3 = pmc_traffic(3, 1_000_000)
        kms = float(np.median(k_ms))
        result["roofline"] = {
            "bound": "hbm", "kernel": "mip_instance_pipeline_kernel", "achieved": kb / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": kb / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            # (the committed PMC pass is BASELINE configs[2]: 1 M instances, emitted fraction 0.268553 = 105.50 MB algorithmic)
            "traffic": (pmc3["hbm_bytes_per_launch"] / (1_000_000 * algorithmic_bytes_per_instance(0.268553)) * kb) if pmc3 else None,
            "traffic_source": (pmc3["source"] + ": PMC bytes per algorithmic byte of the same kernel source at 1 M instances, scaled to this shard's algorithmic bytes"
                               if pmc3 else (f"stale: {stale3} was collected from an older kernel source" if stale3 else None)),
            "algorithmic_bytes_per_launch": kb, "kernel_ms": kms,
            "note": "the shard kernel of rank 0 alone (wall clock incl. host launch); the step as a whole is bound by the all-gather over xGMI",
        }
    # the same 10 M scene on ONE GPU (rank 0), so that a reader can compute the strong-scaling speed-up
    if not args.no_extra and args.instances is None:
        dist.barrier()
        if rank == 0:
            try:
                one = leg_summary(serialized_leg(torch, renderer_amd, make_frame, scene.make_scene(config, n=n_total), device, local_rank,
                                                 stream, 5, 5, 20))
                extra["single_gpu_same_workload"] = one
            except Exception as e:  # noqa: BLE001
                extra["single_gpu_same_workload"] = {"error": f"{type(e).__name__}: {e}"}
        dist.barrier()
    # the CPU restatement beside the GPU number on EVERY line (round 4 carried it at N = 1 only): rank 0 times a bounded sample of its
    # own shard's scene outside the timed region while the other ranks wait at the barrier
    if rank == 0 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(s, args.cpu_seconds)
        if world > 1:
            result["cpu_baseline"]["sample"] = f"rank 0's shard of the {n_total}-instance scene; " + str(result["cpu_baseline"].get("sample", ""))
    elif rank == 0:
        result["cpu_baseline"] = None  # --no-cpu-baseline
    if world > 1:
        dist.barrier()
    pipe.close()


def secondary_sharded(args, torch, dist, renderer_amd, scene, make_frame, device, local_rank, rank, world, stream, extra):
    """Beside the headline: the same exchange with two frames in flight, and (optionally) the native RCCL path."""
    from renderer_amd.sharded import PipelinedExchange, shard_range

    try:
        n_total = scene.CONFIGS[4]["n"]
        lo, hi = shard_range(n_total, world, rank)
        s4 = scene.make_scene(4, n=hi - lo, first=lo)
        n4 = s4["n"]
        f4 = make_frame(s4["planes"], s4["cam_pos"], first_instance_base=lo)

        def mk(stream_handle):
            return make_pipe(renderer_amd, s4, local_rank, stream=stream_handle)

        px = PipelinedExchange(mk, n4, world, rank, device, frames=2)
        o4s = [DeviceOutputs(torch, n4, device), DeviceOutputs(torch, n4, device)]
        torch.cuda.synchronize()
        for _ in range(2):
            px.step(f4, o4s)
        px.wait()
        px.tighten()
        ms = barrier_samples(torch, dist, lambda: px.step(f4, o4s), max(args.steps, 4), 4, 10, True)
        px.wait()
        row = {"frames_in_flight_2": {"instances_per_s": n_total / (float(np.median(ms)) * 1e-3), "ms_per_step": float(np.median(ms)),
                                      "note": "two contexts on two streams: frame k+1's kernel runs under frame k's all-gather"}}
        px.close()
        if args.native_rccl_leg:
            ids = [renderer_amd.InstancePipeline.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            pn = make_pipe(renderer_amd, s4, local_rank)
            pn.comm_init(ids[0], rank, world)
            merged = torch.empty((world * n4, 5), dtype=torch.int32, device=device)
            mcount = torch.zeros(2, dtype=torch.int32, device=device)
            torch.cuda.synchronize()
            cap = int(n4 * 0.3)

            def native_step():
                pn.run_sharded(f4, merged.data_ptr(), mcount.data_ptr(), model=o4s[0].model.data_ptr(),
                               visible_bitmap=o4s[0].bitmap.data_ptr(), chunk_capacity=cap, async_=True)

            ms = barrier_samples(torch, dist, native_step, max(args.steps, 4), 4, 10, True)
            pn.wait()
            row["native_rccl"] = {"instances_per_s": n_total / (float(np.median(ms)) * 1e-3), "ms_per_step": float(np.median(ms)),
                                  "commands_total": int(mcount[0].item())}
            pn.comm_destroy()
            pn.close()
        if rank == 0:
            extra["sharded_exchange_variants"] = row
    except Exception as exc:  # the headline must survive a failure of the secondary leg
        if rank == 0:
            extra["sharded_exchange_variants"] = {"error": repr(exc)}


if __name__ == "__main__":
    main()
