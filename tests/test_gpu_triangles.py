"""Row f-1 on the GPU: per-triangle cull + index-stream append (generate_work.comp:68-200) vs the
oracle. Bit-exact final commands, count and culled index stream."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import renderer_amd

    renderer_amd.load_library()
    return renderer_amd


def _run_gpu(ra, s, vertices, indices, pv, capacity, frames=1, first_instance_base=0):
    import torch

    from renderer_amd.pipeline import make_frame

    n = s["n"]
    dev = torch.device("cuda", 0)
    with ra.InstancePipeline(max_instances=max(n, 1), max_meshes=len(s["meshes"])) as p:
        p.set_mesh_table(s["meshes"])
        p.set_geometry(vertices, indices)
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        model = torch.zeros((max(n, 1), 16), dtype=torch.float32, device=dev)
        cmds = torch.zeros((max(n, 1), 5), dtype=torch.int32, device=dev)
        scal = torch.zeros(8, dtype=torch.int32, device=dev)
        out = torch.full((capacity,), -1, dtype=torch.int32, device=dev)
        frame = make_frame(s["planes"], s["cam_pos"], first_instance_base=first_instance_base, pv=pv)
        for _ in range(frames):
            out.fill_(-1)
            torch.cuda.synchronize()
            p.run_device(frame, model=model.data_ptr(), draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(),
                         draw_index_total=scal.data_ptr() + 4, culled_index_buffer=out.data_ptr(),
                         culled_index_capacity=capacity)
        count, total = (int(x) & 0xFFFFFFFF for x in scal[:2].cpu().tolist())
        got_cmds = cmds[:count].cpu().numpy().view(np.uint32).reshape(-1).view(ra.DRAW_CMD_DTYPE)
        _run_gpu.last_timings = p.timings()
        return got_cmds, count, total, out.cpu().numpy().view(np.uint32)


def _oracle(oracle_mod, s, vertices, indices, pv, capacity, first_instance_base=0):
    r = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"],
                       first_instance_base=first_instance_base, threads=8)
    cmds, out, _ = oracle_mod.cull_all_triangles(r, s["pos"], s["mesh_id"], s["meshes"], s["cam_pos"], pv, vertices, indices,
                                                 first_instance_base=first_instance_base, out_capacity=capacity)
    return r, cmds, out


# (3, 40 000): above 32 768 instances the workgroup-per-command kernel pulls its commands from the ticket counter
@pytest.mark.parametrize("config,n,allvis", [(1, 1024, False), (2, 4000, False), (3, 20_000, False), (3, 3000, True), (3, 40_000, False)])
def test_triangle_cull_matches_oracle(ra, oracle_mod, config, n, allvis):
    s = ra.scene.make_scene(config, n=n, all_visible=allvis)
    vertices, indices = ra.scene.make_geometry(s["meshes"])
    pv = ra.scene.default_pv()
    r0 = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"], want=("draw_cmds",))
    capacity = r0["draw_index_total"] + 3
    r, want_cmds, want_out = _oracle(oracle_mod, s, vertices, indices, pv, capacity)
    got_cmds, count, total, got_out = _run_gpu(ra, s, vertices, indices, pv, capacity, frames=2)
    assert count == len(want_cmds) and total == r["draw_index_total"]
    assert got_cmds.tobytes() == want_cmds.tobytes()
    assert np.array_equal(got_out, want_out)
    survivors = int(want_cmds["indexCount"].astype(np.int64).sum())
    assert 0 < survivors < int(r["draw_cmds"]["indexCount"].astype(np.int64).sum())  # something was culled, something survived


@pytest.mark.parametrize("mode", ["ranges", "ranges_ticketed", "ranges_or_sorted_waves", "sorted_waves", "ranges_of_a_large_frame", "recompact_three_launches",
                                  "wave", "wave_only", "tickets", "tickets_x4", "block256", "block512", "block1024", "parts"])
def test_every_triangle_kernel_variant(ra, oracle_mod, monkeypatch, mode):
    """Round 5: the stage is the range kernel (equal ranges of the triangle stream, one per wave, or — long streams — ranges
    of MIP_TUNE_TRI_RANGE_SLOTS pulled from a counter) and, above 65 536 instances, the range kernel or the wave-per-command
    kernel over commands sorted by size class, chosen on the device. The round-4 kernels stay selectable
    (MIP_TUNE_TRI_CHUNKS_FROM=4294967295): sixteen parts per command, a 256/512/1024-thread workgroup per command (dealt by
    a stride, or pulling tickets — one or four commands each), one wave per command in list order. Each variant is forced
    (tuning variables, read by mip_create and at launch) onto the same mixed scene."""
    if mode == "ranges":
        pass                                                     # the default at this size: one range per wave
    elif mode == "ranges_ticketed":
        monkeypatch.setenv("MIP_TUNE_TRI_RANGE_SLOTS", "256")    # a long stream by this measure: 256-slot ranges, most of them pulled from the counter
    elif mode == "recompact_three_launches":
        monkeypatch.setenv("MIP_TUNE_TRI_BLOCK_MAX", "0")
        monkeypatch.setenv("MIP_TUNE_TRI_RECOMPACT_LAUNCHES", "3")  # round 4's count / scan / scatter instead of the one-launch re-compaction
    elif mode in ("ranges_or_sorted_waves", "sorted_waves", "ranges_of_a_large_frame"):
        monkeypatch.setenv("MIP_TUNE_TRI_BLOCK_MAX", "0")        # the large-frame path: prepare + scatter kernels, one grid, the decomposition chosen on the device ...
        if mode == "sorted_waves":
            monkeypatch.setenv("MIP_TUNE_TRI_CHOICE", "waves")   # ... forced to the wave-per-command grid
        elif mode == "ranges_of_a_large_frame":
            monkeypatch.setenv("MIP_TUNE_TRI_CHOICE", "block")   # ... forced to the range kernel's
            monkeypatch.setenv("MIP_TUNE_TRI_RANGE_SLOTS", "512")
    else:
        monkeypatch.setenv("MIP_TUNE_TRI_CHUNKS_FROM", "4294967295")  # the round-4 kernels
    if mode in ("wave", "wave_only", "tickets", "tickets_x4"):
        monkeypatch.setenv("MIP_TUNE_TRI_BLOCK_MAX", "0")   # the large-frame path: both grids, one chosen on the device ...
        monkeypatch.setenv("MIP_TUNE_TRI_PARTS_MAX", "0")
        if mode == "wave":
            monkeypatch.setenv("MIP_TUNE_TRI_CHOICE", "waves")  # ... forced to the wave-per-command grid
        elif mode == "wave_only":
            monkeypatch.setenv("MIP_TUNE_TRI_NO_CHOICE", "1")   # the second grid is not launched at all
        else:
            monkeypatch.setenv("MIP_TUNE_TRI_CHOICE", "block")  # ... to the ticket-pulling workgroup grid
            if mode == "tickets_x4":
                monkeypatch.setenv("MIP_TUNE_TRI_BATCH_FROM", "100")  # four consecutive commands per ticket (default: from 65 536 commands)
    elif mode == "parts":  # 16 work items per command (small frames), forced onto this larger frame
        monkeypatch.setenv("MIP_TUNE_TRI_PARTS_MAX", "100000000")
    else:
        monkeypatch.setenv("MIP_TUNE_TRI_BLOCK_MAX", "100000000")
        monkeypatch.setenv("MIP_TUNE_TRI_BLOCK_THREADS", mode[5:])
    s = ra.scene.make_scene(3, n=7000)
    s["pos"][11, 1] = np.nan  # one command takes the literal (non-affine) path
    vertices, indices = ra.scene.make_geometry(s["meshes"])
    pv = ra.scene.default_pv()
    r0 = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"], want=("draw_cmds",))
    capacity = r0["draw_index_total"] + 3
    r, want_cmds, want_out = _oracle(oracle_mod, s, vertices, indices, pv, capacity, first_instance_base=17)
    got_cmds, count, total, got_out = _run_gpu(ra, s, vertices, indices, pv, capacity, frames=2, first_instance_base=17)
    assert count == len(want_cmds) and total == r["draw_index_total"], mode
    assert got_cmds.tobytes() == want_cmds.tobytes(), mode
    assert np.array_equal(got_out, want_out), mode


@pytest.mark.parametrize("ordering", ["rows", "strips", "shuffled"])
@pytest.mark.parametrize("config,n", [(3, 7000), (2, 1500)])
def test_wave_kernel_vertex_ring_on_every_mesh_layout(ra, oracle_mod, monkeypatch, ordering, config, n):
    """The wave-per-command kernel transforms a step's vertex range once into its LDS ring when that is cheaper than
    the per-corner path. The bytes must not depend on which path a step took: the same surfaces listed row by row
    (ranges of a few hundred vertices: ring, window carried from step to step), as first-use-ordered strips (the layout
    of an optimised mesh), and with shuffled triangles (every step spans the mesh: per-corner path), with non-finite
    positions (literal chain), a NaN matrix, tiny meshes whose window runs past their vertices, and the last mesh of
    the consolidated buffer (the ring loads ahead of the indices it has seen: clamped at the buffer's end)."""
    monkeypatch.setenv("MIP_TUNE_TRI_BLOCK_MAX", "0")
    monkeypatch.setenv("MIP_TUNE_TRI_PARTS_MAX", "0")
    monkeypatch.setenv("MIP_TUNE_TRI_CHOICE", "waves")  # (round 5: the large-frame pairing is range kernel | sorted wave-per-command kernel)
    s = ra.scene.make_scene(config, n=n, all_visible=(config == 2))
    s["pos"][11, 1] = np.nan
    vertices, indices = ra.scene.make_geometry(s["meshes"], ordering=ordering)
    pv = ra.scene.default_pv()
    r0 = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"], want=("draw_cmds",))
    capacity = r0["draw_index_total"] + 3
    for poison in (False, True):
        if poison:  # non-finite positions switch the affine shortcut off for the whole geometry
            vertices = vertices.copy()
            vertices[min(10, len(vertices) - 1), 0] = np.inf
            vertices[len(vertices) // 2, 2] = np.nan
        r, want_cmds, want_out = _oracle(oracle_mod, s, vertices, indices, pv, capacity, first_instance_base=5)
        got_cmds, count, total, got_out = _run_gpu(ra, s, vertices, indices, pv, capacity, frames=2, first_instance_base=5)
        assert count == len(want_cmds) and total == r["draw_index_total"], (ordering, poison)
        assert got_cmds.tobytes() == want_cmds.tobytes(), (ordering, poison)
        assert np.array_equal(got_out, want_out), (ordering, poison)


def test_triangle_cull_special_instances_and_bases(ra, oracle_mod):
    s = ra.scene.make_scene(3, n=2000, all_visible=True)
    s["pos"][5, 0] = np.nan          # NaN model matrix: every comparison is false, so every triangle survives
    s["scale"][7] = 0.0              # degenerate: all vertices coincide
    s["scale"][9] = -1.0             # mirrored: winding flips
    s["rot"][11] = (0, 0, 0, 3.0)    # non-unit quaternion
    vertices, indices = ra.scene.make_geometry(s["meshes"])
    pv = ra.scene.default_pv()
    r0 = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"], want=("draw_cmds",))
    capacity = r0["draw_index_total"] + 3
    r, want_cmds, want_out = _oracle(oracle_mod, s, vertices, indices, pv, capacity, first_instance_base=1000)
    got_cmds, count, total, got_out = _run_gpu(ra, s, vertices, indices, pv, capacity, first_instance_base=1000)
    assert count == len(want_cmds) and got_cmds.tobytes() == want_cmds.tobytes()
    assert np.array_equal(got_out, want_out)


@pytest.mark.parametrize("range_slots", [None, "256"])
def test_triangle_cull_index_base_and_index_counts_that_are_no_multiple_of_three(ra, oracle_mod, monkeypatch, range_slots):
    """The range kernel numbers the frame's triangles by (firstIndex - first_index_base) / 3: a first_index_base that is no
    multiple of 3, and meshes whose index counts are not (the last one or two indices of a command are no triangle; such
    commands leave unused slots in the stream, and a command of one or two indices has no triangle at all), with one range
    per wave and with 256-slot ranges: bytes equal to the oracle's."""
    import torch

    from renderer_amd.pipeline import make_frame

    if range_slots:
        monkeypatch.setenv("MIP_TUNE_TRI_RANGE_SLOTS", range_slots)
    s = ra.scene.make_scene(3, n=5000)
    meshes = s["meshes"].copy()
    for k in range(len(meshes)):
        for lod in range(int(meshes["n_lods"][k])):
            if (k + lod) % 3 == 1 and meshes["index_len"][k][lod] > 4:
                meshes["index_len"][k][lod] -= 1 + (k % 2)
    meshes["index_len"][5][:] = 2   # commands without a triangle
    meshes["index_len"][9][:] = 1
    s["meshes"] = meshes
    vertices, indices = ra.scene.make_geometry(ra.scene.make_scene(3, n=5000)["meshes"])  # the geometry of the unshortened table
    pv = ra.scene.default_pv()
    base = 5
    r = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"], first_index_base=base, threads=8)
    capacity = base + r["draw_index_total"] + 3
    want_cmds, want_out, _ = oracle_mod.cull_all_triangles(r, s["pos"], s["mesh_id"], s["meshes"], s["cam_pos"], pv, vertices, indices, out_capacity=capacity)
    dev = torch.device("cuda", 0)
    n = s["n"]
    with ra.InstancePipeline(max_instances=n, max_meshes=len(s["meshes"])) as p:
        p.set_mesh_table(s["meshes"])
        p.set_geometry(vertices, indices)
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        model = torch.zeros((n, 16), dtype=torch.float32, device=dev)
        cmds = torch.zeros((n, 5), dtype=torch.int32, device=dev)
        scal = torch.zeros(8, dtype=torch.int32, device=dev)
        out = torch.full((capacity,), -1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for _ in range(2):
            out.fill_(-1)
            torch.cuda.synchronize()
            p.run_device(make_frame(s["planes"], s["cam_pos"], first_index_base=base, pv=pv), model=model.data_ptr(), draw_cmds=cmds.data_ptr(),
                         draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4, culled_index_buffer=out.data_ptr(), culled_index_capacity=capacity)
        count = int(scal[0].item())
        got_cmds = cmds[:count].cpu().numpy().view(np.uint32).reshape(-1).view(ra.DRAW_CMD_DTYPE)
        assert count == len(want_cmds) and got_cmds.tobytes() == want_cmds.tobytes()
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want_out)


@pytest.mark.parametrize("large_frame_path", [False, True])
def test_triangle_stage_with_frames_in_flight(ra, oracle_mod, monkeypatch, large_frame_path):
    """Two frame slots, two cameras, asynchronous frames alternating between them: every slot owns its scratch (command list, final
    counts, range map and granules, size-class table and ticket counter), so the culled streams of frames that overlap on the device
    are each the oracle's — with the range kernel alone and with the large-frame pairing (sort kernels + both grids)."""
    import torch

    from renderer_amd.pipeline import make_frame

    if large_frame_path:
        monkeypatch.setenv("MIP_TUNE_TRI_BLOCK_MAX", "0")
    s = ra.scene.make_scene(3, n=6000)
    vertices, indices = ra.scene.make_geometry(s["meshes"])
    pv = ra.scene.default_pv()
    cams = [np.array(c, np.float32) for c in ((0, 1, 2), (5, 1, -20))]
    planes = [s["planes"], oracle_mod.project_camera(cams[1], (0.0, 0.0, 0.0, 1.0))]
    wants = []
    for k in range(2):
        r = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], planes[k], cams[k], threads=8)
        cap = r["draw_index_total"] + 3
        wc, wo, _ = oracle_mod.cull_all_triangles(r, s["pos"], s["mesh_id"], s["meshes"], cams[k], pv, vertices, indices, out_capacity=cap)
        wants.append((cap, wc, wo))
    dev = torch.device("cuda", 0)
    n = s["n"]
    with ra.InstancePipeline(max_instances=n, max_meshes=len(s["meshes"]), frames_in_flight=2) as p:
        p.set_mesh_table(s["meshes"])
        p.set_geometry(vertices, indices)
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        bufs = []
        for k in range(2):
            bufs.append(dict(model=torch.zeros((n, 16), dtype=torch.float32, device=dev), cmds=torch.zeros((n, 5), dtype=torch.int32, device=dev),
                             scal=torch.zeros(8, dtype=torch.int32, device=dev), out=torch.full((wants[k][0],), -1, dtype=torch.int32, device=dev)))
        torch.cuda.synchronize()
        for rep in range(6):  # 12 asynchronous frames rotating over the two slots
            for k in range(2):
                b = bufs[k]
                p.run_device(make_frame(planes[k], cams[k], pv=pv), model=b["model"].data_ptr(), draw_cmds=b["cmds"].data_ptr(), draw_count=b["scal"].data_ptr(),
                             draw_index_total=b["scal"].data_ptr() + 4, culled_index_buffer=b["out"].data_ptr(), culled_index_capacity=wants[k][0], async_=True)
        p.wait()
        for k in range(2):
            b = bufs[k]
            count = int(b["scal"][0].item())
            got = b["cmds"][:count].cpu().numpy().view(np.uint32).reshape(-1).view(ra.DRAW_CMD_DTYPE)
            assert count == len(wants[k][1]) and got.tobytes() == wants[k][1].tobytes(), k
            assert np.array_equal(b["out"].cpu().numpy().view(np.uint32), wants[k][2]), k


def test_ranges_of_a_stream_whose_every_command_has_two_indices_too_many(ra, oracle_mod, monkeypatch):
    """One mesh whose index counts are 3 k + 2, every instance visible, an index buffer far larger than the frame needs: the stream is
    then LONGER in slots than instances x triangles of the largest command (the slots are numbered by the running sum of indexCount / 3,
    and every command pushes the later ones up by 2/3 of a slot) — the bound the host sizes the range map by has to allow for that, or
    the last ranges of the frame are never walked."""
    import torch

    from renderer_amd.pipeline import make_frame

    monkeypatch.setenv("MIP_TUNE_TRI_RANGE_SLOTS", "256")
    s = ra.scene.make_scene(2, n=3000, all_visible=True)
    vertices, indices = ra.scene.make_geometry(s["meshes"])
    meshes = s["meshes"].copy()
    for lod in range(int(meshes["n_lods"][0])):
        meshes["index_len"][0][lod] = meshes["index_len"][0][lod] // 3 * 3 - 1   # 3 k + 2
    s["meshes"] = meshes
    pv = ra.scene.default_pv()
    r = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"], threads=8)
    capacity = 2 * r["draw_index_total"] + 64
    want_cmds, want_out, _ = oracle_mod.cull_all_triangles(r, s["pos"], s["mesh_id"], s["meshes"], s["cam_pos"], pv, vertices, indices, out_capacity=capacity)
    got_cmds, count, total, got_out = _run_gpu(ra, s, vertices, indices, pv, capacity)
    assert count == len(want_cmds) and total == r["draw_index_total"]
    assert got_cmds.tobytes() == want_cmds.tobytes()
    assert np.array_equal(got_out, want_out)


def test_triangle_cull_with_non_finite_positions(ra, oracle_mod):
    """Positions that are not finite switch off the affine shortcut (0 * inf is not 0): the literal
    mat4 * vec4 chain must then reproduce the oracle, including for instances with ordinary matrices."""
    for n in (300, 30_000):  # workgroup-per-command and wave-per-command kernels
        s = ra.scene.make_scene(2, n=n, all_visible=(n == 300))
        vertices, indices = ra.scene.make_geometry(s["meshes"])
        vertices = vertices.copy()
        vertices[10, 0] = np.inf
        vertices[200, 1] = -np.inf
        vertices[3000, 2] = np.nan
        pv = ra.scene.default_pv()
        r0 = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"], want=("draw_cmds",))
        capacity = r0["draw_index_total"] + 3
        r, want_cmds, want_out = _oracle(oracle_mod, s, vertices, indices, pv, capacity)
        got_cmds, count, total, got_out = _run_gpu(ra, s, vertices, indices, pv, capacity)
        assert count == len(want_cmds) and got_cmds.tobytes() == want_cmds.tobytes(), n
        assert np.array_equal(got_out, want_out), n


def test_triangle_cull_reports_a_short_index_buffer(ra, oracle_mod):
    s = ra.scene.make_scene(2, n=500, all_visible=True)
    vertices, indices = ra.scene.make_geometry(s["meshes"])
    with pytest.raises(ra.MipError) as e:
        _run_gpu(ra, s, vertices, indices, ra.scene.default_pv(), capacity=46356 * 10)
    assert e.value.code == -4
    # and the stage refuses to run without geometry / on host outputs
    import torch

    from renderer_amd.pipeline import make_frame

    with ra.InstancePipeline(max_instances=500, max_meshes=1) as p:
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        buf = torch.zeros(1024, dtype=torch.int32, device="cuda")
        with pytest.raises(ra.MipError) as e:
            p.run_device(make_frame(s["planes"], s["cam_pos"]), model=buf.data_ptr(), draw_cmds=buf.data_ptr(),
                         draw_count=buf.data_ptr(), culled_index_buffer=buf.data_ptr(), culled_index_capacity=1024)
        assert e.value.code == -6


@pytest.mark.parametrize("config,n", [(2, 100_000), (3, 200_000)])
def test_triangle_cull_at_baseline_sizes(ra, oracle_mod, config, n):
    """Row f-1 at the sizes the bench runs: BASELINE configs[1] in full (100 k instances, 27 k commands, 213 M
    triangles in) and the mixed scene at 200 k (114 M triangles). Above 65 536 instances the library selects
    the one-wave-per-command kernel and the wide re-compaction by itself — nothing is forced here. Every
    final command, the count and the WHOLE culled index stream (gaps included) against the oracle."""
    s = ra.scene.make_scene(config, n=n)
    vertices, indices = ra.scene.make_geometry(s["meshes"])
    pv = ra.scene.default_pv()
    r = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"], threads=16)
    capacity = r["draw_index_total"] + 3
    assert capacity > 300_000_000  # hundreds of millions of indices: the regime of the bench leg
    want_cmds, want_out, _ = oracle_mod.cull_all_triangles(r, s["pos"], s["mesh_id"], s["meshes"], s["cam_pos"], pv, vertices, indices,
                                                           out_capacity=capacity, threads=16)
    got_cmds, count, total, got_out = _run_gpu(ra, s, vertices, indices, pv, capacity, frames=1)
    assert count == len(want_cmds) and total == r["draw_index_total"]
    assert got_cmds.tobytes() == want_cmds.tobytes()
    assert np.array_equal(got_out, want_out)
    survivors = int(want_cmds["indexCount"].astype(np.int64).sum())
    assert 0.2 < survivors / int(r["draw_cmds"]["indexCount"].astype(np.int64).sum()) < 0.8


@pytest.mark.parametrize("fault", ["reverse", "skip"])
def test_parts_kernel_does_not_depend_on_who_runs_when(oracle_mod, fault):
    """The kernels that cut a command between waves — round 5's range kernel (a range's continuing segment goes behind the
    survivors its command has in the earlier ranges), with one range per wave and with 256-slot ranges pulled from the
    counter, and round 4's parts kernel (16 work items per command) — in the diagnostic build, with the work dealt from the
    LAST item down ("reverse": later ranges start first), and with every 16th range / one part of every command that NEVER
    publishes ("skip": its successors have to count its survivors themselves — MipTimings.prefix_helps says they did): the
    stream still equals the oracle's. No wait depends on another workgroup ever running (rounds 2-3: a bounded wait,
    MIP_ERR_TIMEOUT and the kernel switched off for the context)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
os.environ["MIP_LIBRARY"] = os.path.join(sys.argv[1], "renderer_amd", "lib", "libmi_instance_pipeline_dbg.so")
os.environ["MIP_TUNE_TRI_PARTS_MAX"] = "100000000"
if sys.argv[2] == "reverse":
    os.environ["MIP_DEBUG_TILE_ORDER"] = "reverse"
else:
    os.environ["MIP_DEBUG_SKIP_PART"] = "3"
import numpy as np
import oracle, renderer_amd
from test_gpu_triangles import _oracle, _run_gpu
import torch
for config, n, knobs in ((3, 3000, {}), (2, 700, {}), (3, 3000, {"MIP_TUNE_TRI_RANGE_SLOTS": "256"}), (2, 700, {"MIP_TUNE_TRI_RANGE_SLOTS": "256"}),
                         (3, 9000, {"MIP_TUNE_TRI_BLOCK_MAX": "0"}),   # the large-frame path: the one-launch re-compaction over 9 workgroups, every fourth silent under "skip"
                         (3, 9000, {"MIP_TUNE_TRI_BLOCK_MAX": "0", "MIP_TUNE_TRI_CHOICE": "waves"}),
                         (3, 3000, {"MIP_TUNE_TRI_CHUNKS_FROM": "4294967295"}), (2, 700, {"MIP_TUNE_TRI_CHUNKS_FROM": "4294967295"})):
    for k in ("MIP_TUNE_TRI_RANGE_SLOTS", "MIP_TUNE_TRI_CHUNKS_FROM", "MIP_TUNE_TRI_BLOCK_MAX", "MIP_TUNE_TRI_CHOICE"):
        os.environ.pop(k, None)
    os.environ.update(knobs)   # (read when the context is created)
    s = renderer_amd.scene.make_scene(config, n=n)
    vertices, indices = renderer_amd.scene.make_geometry(s["meshes"])
    pv = renderer_amd.scene.default_pv()
    r0 = oracle.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"], want=("draw_cmds",))
    capacity = r0["draw_index_total"] + 3
    r, want_cmds, want_out = _oracle(oracle, s, vertices, indices, pv, capacity)
    got_cmds, count, total, got_out = _run_gpu(renderer_amd, s, vertices, indices, pv, capacity, frames=2)
    assert count == len(want_cmds), (config, n, knobs, count, len(want_cmds))
    assert got_cmds.tobytes() == want_cmds.tobytes(), (config, n, knobs, "commands", int((got_cmds != want_cmds).sum()))
    assert np.array_equal(got_out, want_out), (config, n, knobs, "index stream", int((got_out != want_out).sum()), int(np.nonzero(got_out != want_out)[0][0]))
    if sys.argv[2] == "skip":   # (in reverse order the predecessors are usually published in time by their own workgroups)
        assert _run_gpu.last_timings["prefix_helps"] > 0, _run_gpu.last_timings
print("PARTS OK")
'''
    out = subprocess.run([sys.executable, "-c", code, root, fault], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "PARTS OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
