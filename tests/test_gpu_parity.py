"""GPU parity: the HIP path through the C ABI vs the CPU oracle on identical inputs.
Bit-exact visibility bitmap, draw count, command bytes; numerically identical matrices."""
import numpy as np
import pytest

from helpers import assert_parity, popcount_bitmap, report_timing_property, run_gpu, run_oracle

pytestmark = pytest.mark.gpu


def _free_port():
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


@pytest.fixture(scope="module")
def ra():
    import renderer_amd

    renderer_amd.load_library()  # fails loudly if the HIP library is missing
    return renderer_amd


@pytest.mark.parametrize("config,n", [(1, None), (2, None), (3, 200_000)])
def test_configs_match_oracle(ra, oracle_mod, config, n):
    s = ra.scene.make_scene(config, n=n)
    got = run_gpu(ra, s)
    want = run_oracle(oracle_mod, s, threads=8)
    assert_parity(got, want, f"config {config}")
    assert popcount_bitmap(got["visible_bitmap"]) >= got["draw_count"]


@pytest.mark.parametrize("n", [0, 1, 2, 31, 32, 33, 63, 64, 65, 255, 256, 257, 511, 513, 1023, 1025, 4097, 65537])
def test_ragged_sizes(ra, oracle_mod, n):
    s = ra.scene.make_scene(3, n=n)
    got = run_gpu(ra, s)
    want = run_oracle(oracle_mod, s)
    assert_parity(got, want, f"n={n}")


def test_all_visible_and_none_visible(ra, oracle_mod):
    s = ra.scene.make_scene(3, n=50_000, all_visible=True)
    got = run_gpu(ra, s)
    want = run_oracle(oracle_mod, s)
    assert_parity(got, want, "all visible")
    assert got["draw_count"] == s["n"]
    s2 = ra.scene.make_scene(3, n=50_000)
    s2["pos"][:, 2] = -50.0 - np.abs(s2["pos"][:, 2])  # everything behind the camera
    got = run_gpu(ra, s2)
    want = run_oracle(oracle_mod, s2)
    assert_parity(got, want, "none visible")
    assert got["draw_count"] == 0


def _special_values():
    f = np.float32
    return np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e-38, 3.4e38, -3.4e38,
                     1e19, -1e19, 1e-20, 0.5, 2.0, 10.0, 100.0], dtype=f)


def test_non_finite_and_extreme_inputs(ra, oracle_mod):
    """Every wave gets a few poisoned lanes, so both the general path (taken by the whole
    wave) and its agreement with the fast path on the finite lanes are exercised."""
    s = ra.scene.make_scene(3, n=20_000, all_visible=True)
    rng = np.random.default_rng(7)
    sv = _special_values()
    n = s["n"]
    for col, width in (("pos", 3), ("rot", 4)):
        rows = rng.choice(n, 600, replace=False)
        comps = rng.integers(0, width, 600)
        s[col][rows, comps] = rng.choice(sv, 600)
    rows = rng.choice(n, 300, replace=False)
    s["scale"][rows] = rng.choice(sv, 300)
    got = run_gpu(ra, s)
    want = run_oracle(oracle_mod, s)
    assert_parity(got, want, "special values")


def test_denormal_and_huge_scales(ra, oracle_mod):
    s = ra.scene.make_scene(3, n=8192, all_visible=True)
    s["scale"][::3] = np.float32(1e-42)
    s["scale"][1::3] = np.float32(1e30)
    s["rot"][::5] *= np.float32(1e-20)  # products underflow into denormals
    got = run_gpu(ra, s)
    want = run_oracle(oracle_mod, s)
    assert_parity(got, want, "denormal/huge")


def test_boxes_tangent_to_planes(ra, oracle_mod):
    """Instances whose box touches a plane exactly (s - e == 0 => visible) and one ulp either side."""
    s = ra.scene.make_scene(1, n=4096)
    n = s["n"]
    s["rot"][:] = (0, 0, 0, 1)
    s["scale"][:] = 1.0
    # far plane: z*0.001001 - 0.1021021 ; near: -2.001 z + 4.102 ; sweep z finely around both
    z_far = np.float32(102.0) + (np.arange(n // 2, dtype=np.float32) - n // 4) * np.float32(2 ** -17)
    z_near = np.float32(2.05) + (np.arange(n - n // 2, dtype=np.float32) - n // 4) * np.float32(2 ** -22)
    s["pos"][:, 0] = 0
    s["pos"][:, 1] = 1
    s["pos"][: n // 2, 2] = z_far + np.float32(0.5)
    s["pos"][n // 2 :, 2] = z_near - np.float32(0.5)
    got = run_gpu(ra, s)
    want = run_oracle(oracle_mod, s)
    assert_parity(got, want, "tangent")
    vis = popcount_bitmap(got["visible_bitmap"])
    assert 0 < vis < n  # the sweep really crosses the planes


def test_lod_switch_and_zero_length_meshes(ra, oracle_mod):
    s = ra.scene.make_scene(3, n=30_000, all_visible=True)
    m = s["meshes"]
    m["index_len"][::7, 0] = 0  # LOD0 empty: dropped by compaction when near
    m["index_len"][3::7, 1] = 0  # LOD1 empty: dropped when far
    # ring of instances at distance ~10 from the camera (LOD threshold)
    cam = s["cam_pos"]
    k = 10_000
    ang = np.linspace(0, 0.5, k).astype(np.float32)
    r = np.float32(10.0) + (np.arange(k, dtype=np.float32) - k // 2) * np.float32(2 ** -20)
    s["pos"][:k, 0] = cam[0] + r * np.sin(ang)
    s["pos"][:k, 1] = cam[1]
    s["pos"][:k, 2] = cam[2] + r * np.cos(ang)
    got = run_gpu(ra, s)
    want = run_oracle(oracle_mod, s)
    assert_parity(got, want, "lod")
    assert got["draw_count"] < popcount_bitmap(got["visible_bitmap"])


def test_bases_and_partial_outputs(ra, oracle_mod):
    s = ra.scene.make_scene(3, n=10_000)
    with ra.InstancePipeline(max_instances=20_000, max_meshes=64) as p:
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        got = p.run_host(s["planes"], s["cam_pos"], first_instance_base=123456, first_index_base=0xFFFFFF00)
        want = run_oracle(oracle_mod, s, first_instance_base=123456, first_index_base=0xFFFFFF00)
        assert_parity(got, want, "bases")
        only = p.run_host(s["planes"], s["cam_pos"], want=("visible_bitmap",))
        assert np.array_equal(only["visible_bitmap"], want0 := run_oracle(oracle_mod, s)["visible_bitmap"])
        cmds_only = p.run_host(s["planes"], s["cam_pos"], want=("draw_cmds",))
        assert cmds_only["draw_cmds"].tobytes() == run_oracle(oracle_mod, s)["draw_cmds"].tobytes()
        # repeated runs (fresh look-back epoch every launch) stay identical
        for _ in range(20):
            again = p.run_host(s["planes"], s["cam_pos"], want=("draw_cmds", "visible_bitmap"))
            assert again["draw_cmds"].tobytes() == cmds_only["draw_cmds"].tobytes()
            assert np.array_equal(again["visible_bitmap"], want0)
        # shrink the resident set and run again
        p.set_instances(s["pos"][:777], s["rot"][:777], s["scale"][:777], s["mesh_id"][:777])
        got = p.run_host(s["planes"], s["cam_pos"])
        s777 = dict(s, n=777, pos=s["pos"][:777], rot=s["rot"][:777], scale=s["scale"][:777], mesh_id=s["mesh_id"][:777])
        assert_parity(got, run_oracle(oracle_mod, s777), "shrunk")


def test_error_paths(ra):
    from renderer_amd import MipError

    s = ra.scene.make_scene(1, n=16)
    with ra.InstancePipeline(max_instances=8, max_meshes=1) as p:
        with pytest.raises(MipError) as e:
            p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        assert e.value.code == -6  # mesh table first
        p.set_mesh_table(s["meshes"])
        with pytest.raises(MipError) as e:
            p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        assert e.value.code == -4  # capacity
        bad = s["mesh_id"][:8].copy()
        bad[3] = 5
        with pytest.raises(MipError) as e:
            p.set_instances(s["pos"][:8], s["rot"][:8], s["scale"][:8], bad)
        assert e.value.code == -1
        two = np.concatenate([s["meshes"], s["meshes"]])
        with pytest.raises(MipError) as e:
            p.set_mesh_table(two)
        assert e.value.code == -4
        nf = s["meshes"].copy()
        nf["aabb_max"][0, 1] = np.inf
        with pytest.raises(MipError) as e:
            p.set_mesh_table(nf)
        assert e.value.code == -1


def test_full_size_properties(ra, oracle_mod):
    """BASELINE sizes (1 M): size-independent properties plus a full oracle comparison (the
    threaded oracle finishes 1 M in about a second)."""
    s = ra.scene.make_scene(3)
    got = run_gpu(ra, s)
    cmds = got["draw_cmds"]
    n = s["n"]
    # stable order, one command per kept instance, firstIndex is the running sum of indexCount
    fi = cmds["firstInstance"].astype(np.int64)
    assert np.all(np.diff(fi) > 0)
    assert np.all(cmds["instanceCount"] == 1) and np.all(cmds["indexCount"] > 0)
    csum = np.cumsum(cmds["indexCount"].astype(np.uint64))
    run = np.concatenate([np.zeros(1, np.uint64), csum[:-1]]) & np.uint64(0xFFFFFFFF)
    vis = np.unpackbits(got["visible_bitmap"].view(np.uint8), bitorder="little")[:n].astype(bool)
    assert np.all(vis[fi])
    lens_all_positive = np.all(s["meshes"]["index_len"][:, :2][s["meshes"]["n_lods"] > 1] > 0)
    if lens_all_positive:
        assert np.array_equal(cmds["firstIndex"].astype(np.uint64), run)
        assert len(cmds) == int(vis.sum())
    assert got["draw_index_total"] == int(csum[-1]) & 0xFFFFFFFF
    # matrices: bottom row is (0,0,0,1), translation column is the position
    m = got["model"].reshape(n, 4, 4)  # [col][row]
    assert np.all(m[:, :3, 3] == 0) and np.all(m[:, 3, 3] == 1)
    assert np.array_equal(m[:, 3, :3], s["pos"])
    want = run_oracle(oracle_mod, s, threads=8)
    assert_parity(got, want, "1M")


def _golden_files():
    import glob
    import os

    here = os.path.dirname(os.path.abspath(__file__))
    return sorted(glob.glob(os.path.join(here, "golden", "*.npz")))


@pytest.mark.parametrize("path", _golden_files(), ids=lambda p: p.split("/")[-1])
def test_golden_fixtures(ra, path):
    """The committed vectors (inputs + oracle outputs) reproduce on the GPU without the oracle."""
    g = np.load(path)
    s = dict(pos=g["pos"], rot=g["rot"], scale=g["scale"], mesh_id=g["mesh_id"], meshes=g["meshes"],
             planes=g["planes"], cam_pos=g["cam_pos"], n=len(g["scale"]))
    got = run_gpu(ra, s, first_instance_base=int(g["first_instance_base"]), first_index_base=int(g["first_index_base"]))
    want = dict(visible_bitmap=g["visible_bitmap"], draw_count=int(g["draw_count"]), draw_cmds=g["draw_cmds"],
                draw_index_total=int(g["draw_index_total"]), model=g["model"], world_aabb=g["world_aabb"])
    assert_parity(got, want, path)


def test_merge_draw_lists_on_device(ra, oracle_mod):
    """Three shards (one empty) computed on this GPU, laid out as an all-gather would, merged by
    the device kernel == the unsharded oracle run; an undersized chunk is reported, not cut."""
    import torch

    from renderer_amd.pipeline import SHARD_HEADER_BYTES, make_frame
    from renderer_amd.sharded import chunk_stride_bytes

    s = ra.scene.make_scene(3, n=50_000)
    spans = [(0, 17_000), (17_000, 17_000), (17_000, 50_000)]
    cap = max(hi - lo for lo, hi in spans)
    stride = chunk_stride_bytes(cap)
    dev = torch.device("cuda", 0)
    recv = torch.zeros(len(spans) * stride // 4, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    with ra.InstancePipeline(max_instances=cap, max_meshes=64) as p:
        p.set_mesh_table(s["meshes"])
        for k, (lo, hi) in enumerate(spans):
            p.set_instances(s["pos"][lo:hi], s["rot"][lo:hi], s["scale"][lo:hi], s["mesh_id"][lo:hi])
            base = recv.data_ptr() + k * stride
            p.run_device(make_frame(s["planes"], s["cam_pos"], first_instance_base=lo),
                         draw_cmds=base + SHARD_HEADER_BYTES, draw_count=base, draw_index_total=base + 4)
        merged = torch.zeros((s["n"], 5), dtype=torch.int32, device=dev)
        count = torch.zeros(2, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        p.merge_draw_lists(recv.data_ptr(), len(spans), stride, merged.data_ptr(), count.data_ptr())
        want = run_oracle(oracle_mod, s, threads=8)
        total, index_total = (int(x) & 0xFFFFFFFF for x in count.cpu().tolist())
        assert total == want["draw_count"] and index_total == want["draw_index_total"]
        got = merged[:total].cpu().numpy().view(np.uint32).reshape(-1).view(ra.DRAW_CMD_DTYPE)
        assert got.tobytes() == want["draw_cmds"].tobytes()
        # a stride too small for shard 2's list must raise MIP_ERR_CAPACITY
        small = chunk_stride_bytes(64)
        with pytest.raises(ra.MipError) as e:
            p.merge_draw_lists(recv.data_ptr(), 1, small, merged.data_ptr(), count.data_ptr())
        assert e.value.code == -4
        # the stride is rounded up to 256 B and physically holds a few commands more than the capacity the
        # caller sized its output for: a count of capacity + 1 is cut at the CAPACITY and reported, and
        # nothing lands past n_chunks x capacity commands
        n0 = int(recv[0].item())                       # shard 0's count
        cap0 = n0 - 1
        stride0 = chunk_stride_bytes(cap0)
        assert (stride0 - 32) // 20 >= n0               # ... although it would fit the padded stride
        out = torch.full((cap0 + 8, 5), -1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        with pytest.raises(ra.MipError) as e:
            p.merge_draw_lists(recv.data_ptr(), 1, stride0, out.data_ptr(), count.data_ptr(), chunk_capacity=cap0)
        assert e.value.code == -4
        assert int(count[0].item()) == cap0 and bool((out[cap0:] == -1).all())
        assert out[:cap0].cpu().numpy().tobytes() == want["draw_cmds"][:cap0].tobytes()
        with pytest.raises(ra.MipError):                # a capacity the stride cannot hold is an argument error
            p.merge_draw_lists(recv.data_ptr(), 1, small, out.data_ptr(), count.data_ptr(), chunk_capacity=1000)


def test_config4_ten_million_in_eight_shards(ra, oracle_mod):
    """BASELINE configs[3] at full size on one GPU: the 10 M scene as one launch, and as 8 contiguous
    shards of 1.25 M (each with its own draw_index base, as 8 ranks would run them) laid out like the
    all-gather's receive buffer and merged on the device. Both must be the oracle's list, byte for byte."""
    import torch

    from renderer_amd.pipeline import SHARD_HEADER_BYTES, make_frame
    from renderer_amd.sharded import chunk_stride_bytes, shard_range

    s = ra.scene.make_scene(4)
    n, world = s["n"], 8
    assert n == 10_000_000
    want = run_oracle(oracle_mod, s, threads=8, want=("visible_bitmap", "draw_cmds"))
    dev = torch.device("cuda", 0)
    with ra.InstancePipeline(max_instances=n, max_meshes=64) as p:
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        bitmap = torch.zeros((n + 31) // 32, dtype=torch.int32, device=dev)
        cmds = torch.zeros((want["draw_count"] + 16, 5), dtype=torch.int32, device=dev)  # exactly enough: nothing may land past the count
        scal = torch.zeros(8, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()  # torch fills on its own stream; the library's streams do not wait for it
        p.run_device(make_frame(s["planes"], s["cam_pos"]), visible_bitmap=bitmap.data_ptr(), draw_cmds=cmds.data_ptr(),
                     draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4)
        count, index_total = (int(x) & 0xFFFFFFFF for x in scal[:2].cpu().tolist())
        assert count == want["draw_count"] and index_total == want["draw_index_total"]
        assert np.array_equal(bitmap.cpu().numpy().view(np.uint32), want["visible_bitmap"])
        assert cmds[:count].cpu().numpy().tobytes() == want["draw_cmds"].tobytes()
        assert not cmds[count:].any()
        del bitmap

        spans = [shard_range(n, world, r) for r in range(world)]
        assert spans[0] == (0, 1_250_000) and spans[-1][1] == n
        cap = 400_000  # v = 0.27: ~337 k commands per shard
        stride = chunk_stride_bytes(cap)
        recv = torch.zeros(world * stride // 4, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for r, (lo, hi) in enumerate(spans):
            p.set_instances(s["pos"][lo:hi], s["rot"][lo:hi], s["scale"][lo:hi], s["mesh_id"][lo:hi])
            base = recv.data_ptr() + r * stride
            p.run_device(make_frame(s["planes"], s["cam_pos"], first_instance_base=lo),
                         draw_cmds=base + SHARD_HEADER_BYTES, draw_count=base, draw_index_total=base + 4)
        cmds.zero_()
        torch.cuda.synchronize()
        p.merge_draw_lists(recv.data_ptr(), world, stride, cmds.data_ptr(), scal.data_ptr())
        count, index_total = (int(x) & 0xFFFFFFFF for x in scal[:2].cpu().tolist())
        assert count == want["draw_count"] and index_total == want["draw_index_total"]
        assert cmds[:count].cpu().numpy().tobytes() == want["draw_cmds"].tobytes()


def test_exchange_step_world_size_one(ra, oracle_mod):
    """The real frame driver (kernel -> all_gather_into_tensor over RCCL -> merge) with one rank."""
    import os

    import torch
    import torch.distributed as dist

    from renderer_amd.sharded import DrawListExchange, make_shard_frame

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        s = ra.scene.make_scene(3, n=300_000)
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            with ra.InstancePipeline(max_instances=s["n"], max_meshes=64, stream=st.cuda_stream) as p:
                p.set_mesh_table(s["meshes"])
                p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
                ex = DrawListExchange(p, s["n"], 1, 0, dev)
                frame = make_shard_frame(s["planes"], s["cam_pos"], s["n"], 1, 0)
                ex.step(frame)
                p.wait()
                ex.tighten()
                for _ in range(3):
                    ex.step(frame)
                p.wait()
                cmds, total, index_total = ex.merged_draw_list()
        want = run_oracle(oracle_mod, s, threads=8, want=("draw_cmds",))
        assert total == want["draw_count"] and index_total == want["draw_index_total"]
        assert cmds.tobytes() == want["draw_cmds"].tobytes()
    finally:
        dist.destroy_process_group()


def test_three_million_instances_multi_window_prefix(ra, oracle_mod):
    """> 2 M instances needs more than one round of level-1 accumulator reads per tile."""
    s = ra.scene.make_scene(4, n=3_000_000)
    got = run_gpu(ra, s, want=("visible_bitmap", "draw_cmds"))
    want = run_oracle(oracle_mod, s, threads=8, want=("visible_bitmap", "draw_cmds"))
    assert_parity(got, want, "3M")


def test_plain_c_host(ra):
    """integration/c/mip_smoke.c: a C99 program that links the library and runs a frame with host pointers —
    the boundary as the reference-side shim would use it."""
    import os
    import subprocess
    import tempfile

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "renderer_amd", "lib")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "mip_smoke")
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                               os.path.join(root, "integration", "c", "mip_smoke.c"), "-L", lib_dir, "-lmi_instance_pipeline",
                               "-Wl,-rpath," + lib_dir, "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "C_SMOKE OK" in out.stdout, out.stdout + out.stderr
    fields = dict(x.split("=") for x in out.stdout.split()[2:])
    assert 0 < int(fields["commands"]) == int(fields["visible"]) < 1000


def test_ordered_tiles_flag_is_accepted_and_changes_nothing(ra, oracle_mod):
    """MIP_CFG_ORDERED_TILES (ABI <= 3: tile numbers from a counter / three wait-free launches) is accepted and ignored
    since ABI 4 — every launch is independent of dispatch order. Same results, for the plain frame, frames in flight,
    recorded launch graphs and a skinned frame."""
    import torch

    from renderer_amd.pipeline import make_frame

    dev = torch.device("cuda", 0)
    for n in (1, 255, 70_000, 1_000_000):
        s = ra.scene.make_scene(3, n=n)
        want = run_oracle(oracle_mod, s, threads=8)
        with ra.InstancePipeline(max_instances=n, max_meshes=64, ordered_tiles=True) as p:
            p.set_mesh_table(s["meshes"])
            p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
            for rep in range(3):
                assert_parity(p.run_host(s["planes"], s["cam_pos"]), want, f"ordered n={n} rep={rep}")
    s = ra.scene.make_scene(3, n=40_000)
    want = run_oracle(oracle_mod, s, threads=8, want=("draw_cmds",))
    with ra.InstancePipeline(max_instances=s["n"], max_meshes=64, frames_in_flight=2, ordered_tiles=True) as p:
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        sets = []
        for _ in range(2):
            cmds = torch.zeros((s["n"], 5), dtype=torch.int32, device=dev)
            scal = torch.zeros(8, dtype=torch.int32, device=dev)
            sets.append((cmds, scal, p.prepare_outputs(draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(),
                                                        draw_index_total=scal.data_ptr() + 4)))
        torch.cuda.synchronize()
        p.run_many(make_frame(s["planes"], s["cam_pos"]), [x[2] for x in sets], 150)  # graphs + single launches
        p.wait()
        assert p.timings()["graph_frames"] == 128
        for cmds, scal, _ in sets:
            count = int(scal[0].item())
            assert count == want["draw_count"] and cmds[:count].cpu().numpy().tobytes() == want["draw_cmds"].tobytes()
    sk = ra.scene.make_skinned_scene(5000)
    ws = oracle_mod.run_skinned(sk["pos"], sk["rot"], sk["scale"], sk["mesh_id"], sk["meshes"], sk["skeleton"], sk["poses"],
                                sk["planes"], sk["cam_pos"])
    with ra.InstancePipeline(max_instances=5000, max_meshes=1, ordered_tiles=True) as p:
        p.set_mesh_table(sk["meshes"])
        p.set_instances(sk["pos"], sk["rot"], sk["scale"], sk["mesh_id"])
        p.set_skeleton(sk["skeleton"]["parent"], sk["skeleton"]["inverse_bind"], sk["skeleton"]["joint_box"])
        p.set_poses(sk["poses"])
        cmds = torch.zeros((5000, 5), dtype=torch.int32, device=dev)
        scal = torch.zeros(8, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for _ in range(2):
            p.run_skinned(make_frame(sk["planes"], sk["cam_pos"]), draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(),
                          draw_index_total=scal.data_ptr() + 4)
        count = int(scal[0].item())
        assert count == ws["draw_count"] and cmds[:count].cpu().numpy().tobytes() == ws["draw_cmds"].tobytes()


_ORDER_CHILD = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
os.environ["MIP_LIBRARY"] = os.path.join(sys.argv[1], "renderer_amd", "lib", "libmi_instance_pipeline_dbg.so")
what = sys.argv[2]
import numpy as np, torch
import oracle, renderer_amd
from renderer_amd import scene
from renderer_amd.pipeline import make_frame
from cpu_pipeline import decode_wire, unpack_wire
dev = torch.device("cuda", 0)

def frame_case(n, order, wire=False, nonfinite=False):
    os.environ["MIP_TUNE_ORDER"] = order
    s = scene.make_scene(3, n=n)
    if nonfinite:
        s["pos"][7] = np.nan; s["scale"][300] = np.inf; s["rot"][n - 3] = [3e38, 3e38, 0, 0]
    want = oracle.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"], threads=8,
                      want=("draw_cmds", "visible_bitmap"))
    with renderer_amd.InstancePipeline(max_instances=n, max_meshes=64) as p:
        p.set_mesh_table(s["meshes"]); p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        cmds = torch.zeros((n + 1024, 5), dtype=torch.int32, device=dev)
        bitmap = torch.zeros(((n + 31) // 32,), dtype=torch.int32, device=dev)
        scal = torch.zeros(8, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for rep in range(4):   # (launches that had to help make the ones after the next follow the first-mover rule: both kinds run)
            p.run_device(make_frame(s["planes"], s["cam_pos"]), visible_bitmap=bitmap.data_ptr(), draw_cmds=cmds.data_ptr(),
                         draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4, wire=wire)
            count = int(scal[0].item())
            assert count == want["draw_count"] and int(scal[1].item()) == want["draw_index_total"], (n, order, wire, rep, count)
            words = cmds.cpu().numpy().view(np.uint32).reshape(-1)
            if wire == "packed":
                got = decode_wire(unpack_wire(words, count), count, s["meshes"])
            elif wire:
                got = decode_wire(words, count, s["meshes"])
            else:
                got = words[:count * 5]
            assert got.tobytes() == want["draw_cmds"].tobytes(), (n, order, wire, rep)
            assert np.array_equal(bitmap.cpu().numpy().view(np.uint32), want["visible_bitmap"]), (n, order, wire, rep)
        return p.timings()["prefix_helps"]

helps = []
if what == "frames":
    # 32 tiles: resident as a whole; 3 907 tiles: not (a stores-first workgroup shrinks to one wave while it looks up its
    # prefix, so ~2 800 of those fit the chip at once: 600 k instances would still all become resident)
    for n in (8192, 1_000_000):
        for order in ("1", "3"):
            helps.append(frame_case(n, order))
    helps.append(frame_case(1_000_000, "1", wire=True))
    helps.append(frame_case(1_000_000, "3", wire="packed"))
    helps.append(frame_case(1_000_000, "1", nonfinite=True))   # the kernel with the fall-back arithmetic tiers helps with them
elif what == "graphs":                 # recorded launches: the cold path reads its frame from the ring, like the hot path
    s = scene.make_scene(3, n=1_000_000)
    cams = [np.array(c, np.float32) for c in ((0, 1, 2), (5, 1, 2))]
    wants = [oracle.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], c, threads=8, want=("draw_cmds",)) for c in cams]
    os.environ["MIP_TUNE_GRAPH_ROUND"] = "4"   # (read when the context is created) a round = 4 frames: 2 per frame slot
    with renderer_amd.InstancePipeline(max_instances=s["n"], max_meshes=64, frames_in_flight=2) as p:
        p.set_mesh_table(s["meshes"]); p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        sets = []
        for _ in cams:
            cmds = torch.zeros((s["n"], 5), dtype=torch.int32, device=dev); scal = torch.zeros(8, dtype=torch.int32, device=dev)
            sets.append((cmds, scal, p.prepare_outputs(draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4)))
        torch.cuda.synchronize()
        p.run_many([make_frame(s["planes"], c) for c in cams], [x[2] for x in sets], 4)
        p.wait()
        assert p.timings()["graph_frames"] == 4, p.timings()
        for k, (cmds, scal, _) in enumerate(sets):
            count = int(scal[0].item())
            assert count == wants[k]["draw_count"] and cmds[:count].cpu().numpy().tobytes() == wants[k]["draw_cmds"].tobytes(), k
        helps.append(p.timings()["prefix_helps"])
elif what == "skinned":
    sk = scene.make_skinned_scene(1_000_000)
    ws = oracle.run_skinned(sk["pos"], sk["rot"], sk["scale"], sk["mesh_id"], sk["meshes"], sk["skeleton"], sk["poses"], sk["planes"], sk["cam_pos"])
    with renderer_amd.InstancePipeline(max_instances=sk["n"], max_meshes=1) as p:
        p.set_mesh_table(sk["meshes"]); p.set_instances(sk["pos"], sk["rot"], sk["scale"], sk["mesh_id"])
        p.set_skeleton(sk["skeleton"]["parent"], sk["skeleton"]["inverse_bind"], sk["skeleton"]["joint_box"]); p.set_poses(sk["poses"])
        cmds = torch.zeros((sk["n"], 5), dtype=torch.int32, device=dev); scal = torch.zeros(8, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        p.run_skinned(make_frame(sk["planes"], sk["cam_pos"]), draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4)
        count = int(scal[0].item())
        assert count == ws["draw_count"] and cmds[:count].cpu().numpy().tobytes() == ws["draw_cmds"].tobytes()
        helps.append(p.timings()["prefix_helps"])
elif what == "views":
    s = scene.make_scene(3, n=1_000_000)
    cams = [np.array(c, np.float32) for c in ((0, 1, 2), (5, 1, 2), (-3, 2, 8))]
    with renderer_amd.InstancePipeline(max_instances=s["n"], max_meshes=64) as p:
        p.set_mesh_table(s["meshes"]); p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        sets = []
        for _ in cams:
            cmds = torch.zeros((s["n"], 5), dtype=torch.int32, device=dev); scal = torch.zeros(8, dtype=torch.int32, device=dev)
            sets.append((cmds, scal, p.prepare_outputs(draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4)))
        torch.cuda.synchronize()
        p.run_views([make_frame(s["planes"], c) for c in cams], [x[2] for x in sets])
        p.wait()   # (prepared outputs are asynchronous by default)
        for k, (cmds, scal, _) in enumerate(sets):
            want = oracle.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], cams[k], threads=8, want=("draw_cmds",))
            count = int(scal[0].item())
            assert count == want["draw_count"] and cmds[:count].cpu().numpy().tobytes() == want["draw_cmds"].tobytes(), k
        helps.append(p.timings()["prefix_helps"])
print("HELPS", " ".join(str(h) for h in helps))
"""


def _run_order_child(what, **env_add):
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_add)     # (the diagnostic library is built by __graft_entry__.build(), not from inside a test)
    env.pop("MIP_TUNE_ORDER", None)
    out = subprocess.run([sys.executable, "-c", _ORDER_CHILD, root, what], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.split("\n") if l.startswith("HELPS")]
    assert line, out.stdout
    return [int(x) for x in line[0].split()[1:]]


@pytest.mark.parametrize("tile_order,first_mover", [("reverse", None), ("scramble", None), ("scramble", "always"), ("scramble", "never")])
def test_any_dispatch_order_gives_the_same_bytes(tile_order, first_mover):
    """No kernel of the library depends on the order the hardware starts workgroups in. The diagnostic build numbers
    the tiles by a PERMUTATION of the workgroup index — reversed (the first workgroups to run are the LAST tiles, every
    one of which needs every earlier tile), or scrambled: a tile whose predecessors have not published after the
    patient polls computes their aggregates itself (instance_kernel.hpp, resolve_prefix). Command list, count, index
    total and bitmap equal the oracle's, in both kernel orders, both wire forms, with non-finite instances, for a
    launch that is resident as a whole (nothing to help) and for one that is not (thousands of helps).
    Three frames each: the first finds the state of an ordinary launch (only a tile's owner adds it to its group's accumulator),
    the later ones follow the first-mover rule because their predecessor had to help (mark_tile_started: tiles mark themselves
    STARTED, the first to touch a granule adds the tile); MIP_TUNE_FIRST_MOVER pins either rule for every frame."""
    env = {"MIP_TUNE_FIRST_MOVER": first_mover} if first_mover else {}
    helps = _run_order_child("frames", MIP_DEBUG_TILE_ORDER=tile_order, **env)
    # 32 tiles, all resident: the predecessors normally publish within the patient polls (0 helps); a cold first launch makes some
    # late (round 4: 34 on the builder's box, 91 on the driver's). 3 907 tiles: the early workgroups normally help themselves.
    # How many is a timing property — reported; the bytes the child compared are the test.
    report_timing_property(f"{tile_order}: helps of the two resident 32-tile launches", helps[:2], "small", max(helps[:2]) <= 64)
    report_timing_property(f"{tile_order}: helps of the non-resident launches", helps[2:], "> 0 each", all(h > 0 for h in helps[2:]))


def test_any_dispatch_order_recorded_graphs_skinned_and_views():
    """The same for frames replayed from recorded launch graphs (the helper reads its frame from the ring), a skinned
    frame (per-instance boxes, the general kernel) and the multi-view kernel, tiles in reverse order."""
    for what in ("graphs", "skinned", "views"):
        helps = _run_order_child(what, MIP_DEBUG_TILE_ORDER="reverse")
        report_timing_property(f"{what}, tiles reversed: helps", helps, "> 0", helps[-1] > 0)


@pytest.mark.parametrize("first_mover", [None, "always"])
def test_a_tile_that_never_publishes_is_helped(first_mover):
    """Fault injection (diagnostic build): tile 5 never publishes its aggregate (rounds 1-3: every later tile's bounded
    wait expired after 0.5 s and the frame ended in MIP_ERR_TIMEOUT). Now the tiles that need it compute it: right
    outputs, no error, and the count of helps says it happened."""
    env = {"MIP_TUNE_FIRST_MOVER": first_mover} if first_mover else {}
    helps = _run_order_child("frames", MIP_DEBUG_SKIP_PUBLISH_TILE="5", **env)
    assert all(h > 0 for h in helps), helps


def test_an_idle_gpu_needs_no_help(ra, oracle_mod):
    """The product build on a GPU it has to itself: workgroups start in index order, nobody waits long enough to help
    (MipTimings.prefix_helps stays 0) — self-help is the safety net, not the steady state."""
    s = ra.scene.make_scene(3, n=1_000_000)
    want = run_oracle(oracle_mod, s, threads=8)
    with ra.InstancePipeline(max_instances=s["n"], max_meshes=64) as p:
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        for rep in range(20):
            got = p.run_host(s["planes"], s["cam_pos"])
        assert_parity(got, want, "1 M")
        helps = p.timings()["prefix_helps"]
        report_timing_property("product build, 20 frames of 1 M: prefix_helps", helps, "0 on an idle GPU", helps == 0)


def test_frames_in_flight_rotate_independent_state(ra, oracle_mod):
    """Three frames in flight with different cameras and their own output buffers: all correct."""
    import torch

    from renderer_amd.pipeline import make_frame

    s = ra.scene.make_scene(3, n=300_000)
    dev = torch.device("cuda", 0)
    cams = [np.array(c, np.float32) for c in ((0, 1, 2), (5, 1, 2), (0, 1, 30))]
    with ra.InstancePipeline(max_instances=s["n"], max_meshes=64, frames_in_flight=3) as p:
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        bufs = []
        for _ in range(3):
            cmds = torch.zeros((s["n"], 5), dtype=torch.int32, device=dev)
            scal = torch.zeros(8, dtype=torch.int32, device=dev)
            bitmap = torch.zeros((s["n"] + 31) // 32, dtype=torch.int32, device=dev)
            bufs.append((cmds, scal, bitmap))
        torch.cuda.synchronize()
        for rep in range(4):  # 12 async launches rotating over the 3 slots
            for k in range(3):
                cmds, scal, bitmap = bufs[k]
                p.run_device(make_frame(s["planes"], cams[k]), draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(),
                             draw_index_total=scal.data_ptr() + 4, visible_bitmap=bitmap.data_ptr(), async_=True)
        p.wait()
        for k in range(3):
            cmds, scal, bitmap = bufs[k]
            want = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], cams[k],
                                  threads=8, want=("draw_cmds", "visible_bitmap"))
            count, total = (int(x) & 0xFFFFFFFF for x in scal[:2].cpu().tolist())
            assert count == want["draw_count"] and total == want["draw_index_total"]
            got = cmds[:count].cpu().numpy().view(np.uint32).reshape(-1).view(ra.DRAW_CMD_DTYPE)
            assert got.tobytes() == want["draw_cmds"].tobytes()
            assert np.array_equal(bitmap.cpu().numpy().view(np.uint32), want["visible_bitmap"])
    with pytest.raises(ra.MipError):
        ra.InstancePipeline(max_instances=16, max_meshes=1, frames_in_flight=99)


def test_per_light_culled_lists_as_concurrent_views(ra, oracle_mod):
    """SURVEY section 8 f-4 "per-light cull lists": one view per light — the light's own six planes and its
    position as the LOD reference — queued back to back on a context with as many frame slots as lights,
    so the views overlap on the device. Every view's bitmap and list must be the oracle's for that view."""
    import torch

    from renderer_amd.pipeline import make_frame

    s = ra.scene.make_scene(3, n=200_000)
    rng = np.random.default_rng(21)
    lights = [np.array(p, np.float32) for p in ((30, 20, -40.1), (0.1, 17, 0.1), (-20, 5, 10), (0, 40, 0))]  # main.rs:368-382 + two
    views = []
    for lp in lights:
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        views.append((lp, oracle_mod.project_camera(lp, q.astype(np.float32), aspect=1.0, fovy_degrees=90.0, near=0.5, far=200.0)))
    dev = torch.device("cuda", 0)
    n = s["n"]
    with ra.InstancePipeline(max_instances=n, max_meshes=64, frames_in_flight=len(views)) as p:
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        bufs = [(torch.zeros((n, 5), dtype=torch.int32, device=dev), torch.zeros(8, dtype=torch.int32, device=dev),
                 torch.zeros((n + 31) // 32, dtype=torch.int32, device=dev)) for _ in views]
        torch.cuda.synchronize()
        for rep in range(2):
            for (lp, planes), (cmds, scal, bitmap) in zip(views, bufs):
                p.run_device(make_frame(planes, lp), draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(),
                             draw_index_total=scal.data_ptr() + 4, visible_bitmap=bitmap.data_ptr(), async_=True)
        p.wait()
        counts = []
        for (lp, planes), (cmds, scal, bitmap) in zip(views, bufs):
            want = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], planes, lp, threads=8,
                                  want=("draw_cmds", "visible_bitmap"))
            count = int(scal[0].item())
            counts.append(count)
            assert count == want["draw_count"] and cmds[:count].cpu().numpy().tobytes() == want["draw_cmds"].tobytes()
            assert np.array_equal(bitmap.cpu().numpy().view(np.uint32), want["visible_bitmap"])
        assert len(set(counts)) > 1 and all(0 < c < n for c in counts)  # the views do differ


def test_run_views_matches_one_run_per_view(ra, oracle_mod):
    """mip_run_views: 1..4 frusta over the same instances in one launch (per-light culled lists). Every view
    must be byte-identical to the oracle's frame for that view — including bases, non-finite instances,
    partial tiles, an empty scene, repeated launches — and the argument checks must hold."""
    import torch

    from renderer_amd.pipeline import make_frame

    rng = np.random.default_rng(33)
    dev = torch.device("cuda", 0)
    for n, k in ((0, 2), (1, 1), (255, 3), (70_001, 4), (40_000, 7), (3000, 16), (1_000_000, 4)):
        s = ra.scene.make_scene(3, n=max(n, 1))
        if n == 0:
            s = {key: (val[:0] if isinstance(val, np.ndarray) and key in ("pos", "rot", "scale", "mesh_id") else val) for key, val in s.items()}
        if n > 100:
            s["pos"][17] = np.nan
            s["scale"][33] = np.inf
        views = []
        for v in range(k):
            lp = rng.normal(0, 15, 3).astype(np.float32)
            q = rng.normal(size=4)
            q /= np.linalg.norm(q)
            planes = oracle_mod.project_camera(lp, q.astype(np.float32), aspect=float(rng.uniform(0.8, 2)), fovy_degrees=float(rng.uniform(50, 110)),
                                               near=0.2, far=300.0)
            views.append((lp, planes, int(rng.integers(0, 1000)), int(rng.integers(0, 2 ** 32))))
        with ra.InstancePipeline(max_instances=max(n, 1), max_meshes=64) as p:
            p.set_mesh_table(s["meshes"])
            p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
            bufs = [(torch.full((max(n, 1), 5), -1, dtype=torch.int32, device=dev), torch.full((8,), -1, dtype=torch.int32, device=dev),
                     torch.zeros((n + 31) // 32 + 1, dtype=torch.int32, device=dev)) for _ in views]
            torch.cuda.synchronize()
            frames = [make_frame(pl, lp, first_instance_base=fib, first_index_base=fxb) for lp, pl, fib, fxb in views]
            outs = [p.prepare_outputs(draw_cmds=c.data_ptr(), draw_count=sc.data_ptr(), draw_index_total=sc.data_ptr() + 4,
                                      visible_bitmap=b.data_ptr(), async_=False) for c, sc, b in bufs]
            for rep in range(3):
                p.run_views(frames, outs)
            for (lp, pl, fib, fxb), (cmds, scal, bitmap) in zip(views, bufs):
                want = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], pl, lp, first_instance_base=fib,
                                      first_index_base=fxb, threads=8, want=("draw_cmds", "visible_bitmap"))
                count, total = (int(x) & 0xFFFFFFFF for x in scal[:2].cpu().tolist())
                assert count == want["draw_count"] and total == want["draw_index_total"], (n, k)
                assert cmds[:count].cpu().numpy().tobytes() == want["draw_cmds"].tobytes(), (n, k)
                assert np.array_equal(bitmap[:(n + 31) // 32].cpu().numpy().view(np.uint32), want["visible_bitmap"]), (n, k)
            if n == 255:
                with pytest.raises(ra.MipError):
                    p.run_views(frames * 6, outs * 6)                      # more than 16 views
                model = torch.zeros((n, 16), dtype=torch.float32, device=dev)
                with pytest.raises(ra.MipError):                           # matrices are not per view
                    p.run_views(frames[:1], [p.prepare_outputs(model=model.data_ptr(), draw_cmds=bufs[0][0].data_ptr(),
                                                               draw_count=bufs[0][1].data_ptr(), async_=False)])
                with pytest.raises(ra.MipError):                           # a view needs its command list
                    p.run_views(frames[:1], [p.prepare_outputs(visible_bitmap=bufs[0][2].data_ptr(), async_=False)])


def test_tlas_instance_rows(ra, oracle_mod):
    """Row f-4: VkAccelerationStructureInstanceKHR rows for every instance, with and without the
    matrix output, with BLAS addresses and a draw_index base."""
    import torch

    from helpers import same_floats
    from renderer_amd.pipeline import make_frame

    s = ra.scene.make_scene(3, n=10_001)
    s["pos"][17, 1] = np.nan
    blas = (np.arange(len(s["meshes"]), dtype=np.uint64) << np.uint64(20)) + np.uint64(0xABC0000000)
    dev = torch.device("cuda", 0)
    want_run = run_oracle(oracle_mod, s)
    want = oracle_mod.tlas_instances(want_run["model"], s["mesh_id"], blas, first_instance_base=5)
    with ra.InstancePipeline(max_instances=s["n"], max_meshes=64) as p:
        p.set_mesh_table(s["meshes"])
        p.set_blas_addresses(blas)
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        frame = make_frame(s["planes"], s["cam_pos"], first_instance_base=5)
        for with_model in (True, False):
            tlas = torch.zeros((s["n"], 16), dtype=torch.int32, device=dev)
            model = torch.zeros((s["n"], 16), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            p.run_device(frame, model=model.data_ptr() if with_model else 0, tlas_instances=tlas.data_ptr())
            got = tlas.cpu().numpy().view(np.uint32)
            assert np.array_equal(got[:, 12:], want[:, 12:])                       # index|mask, sbt|flags, BLAS address
            assert same_floats(got[:, :12].view(np.float32), want[:, :12].view(np.float32))
            if with_model:
                assert same_floats(model.cpu().numpy(), want_run["model"])
        with pytest.raises(ra.MipError):
            p.set_blas_addresses(blas[:3])


def test_run_many_and_pipelined_exchange(ra, oracle_mod):
    """mip_run_many (frames issued from compiled code) and two sharded frames in flight (world 1)."""
    import os

    import torch
    import torch.distributed as dist

    from renderer_amd.pipeline import make_frame
    from renderer_amd.sharded import PipelinedExchange, make_shard_frame

    s = ra.scene.make_scene(3, n=120_000)
    want = run_oracle(oracle_mod, s, threads=8, want=("draw_cmds",))
    dev = torch.device("cuda", 0)
    with ra.InstancePipeline(max_instances=s["n"], max_meshes=64, frames_in_flight=2) as p:
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        sets = []
        for _ in range(2):
            cmds = torch.zeros((s["n"], 5), dtype=torch.int32, device=dev)
            scal = torch.zeros(8, dtype=torch.int32, device=dev)
            sets.append((cmds, scal, p.prepare_outputs(draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(),
                                                        draw_index_total=scal.data_ptr() + 4)))
        torch.cuda.synchronize()
        p.run_many(make_frame(s["planes"], s["cam_pos"]), [x[2] for x in sets], 25)
        p.wait()
        for cmds, scal, _ in sets:
            count = int(scal[0].item())
            assert count == want["draw_count"]
            got = cmds[:count].cpu().numpy().view(np.uint32).reshape(-1).view(ra.DRAW_CMD_DTYPE)
            assert got.tobytes() == want["draw_cmds"].tobytes()
        with pytest.raises(ra.MipError):  # host-pointer outputs are not accepted by run_many
            p.run_many(make_frame(s["planes"], s["cam_pos"]), [p.prepare_outputs(async_=False)], 1)

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        def make_pipe(stream_handle):
            q = ra.InstancePipeline(max_instances=s["n"], max_meshes=64, stream=stream_handle)
            q.set_mesh_table(s["meshes"])
            q.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
            return q

        px = PipelinedExchange(make_pipe, s["n"], 1, 0, dev, frames=2)
        frame = make_shard_frame(s["planes"], s["cam_pos"], s["n"], 1, 0)
        torch.cuda.synchronize()
        for _ in range(6):
            px.step(frame, [None, None])
        px.wait()
        px.tighten()
        for _ in range(4):
            px.step(frame, [None, None])
        px.wait()
        for ex in px.exchanges:
            cmds, total, index_total = ex.merged_draw_list()
            assert total == want["draw_count"] and cmds.tobytes() == want["draw_cmds"].tobytes()
        px.close()
    finally:
        dist.destroy_process_group()


def test_light_draw_lists(ra, oracle_mod):
    """Row f-4, shadow pass (shadow_mapping.rs:405-478): n_lights x n commands, bit-exact, for instance
    counts that do and do not keep the rows 16-byte aligned, partial tiles, 1 and 16 lights."""
    import torch

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    for n, n_lights in ((1, 1), (255, 3), (1024, 16), (100_003, 4), (120_000, 5)):
        s = ra.scene.make_scene(3, n=n)
        pos = s["pos"].copy()
        if n > 100:
            pos[17] = np.nan
            pos[33] = np.inf
        lights = rng.uniform(-40, 40, size=(n_lights, 3)).astype(np.float32)
        lights[0] = pos[0]
        want = oracle_mod.light_draw_lists(pos, s["mesh_id"], s["meshes"], lights, first_instance_base=3)
        with ra.InstancePipeline(max_instances=n, max_meshes=64) as p:
            p.set_mesh_table(s["meshes"])
            p.set_instances(pos, s["rot"], s["scale"], s["mesh_id"])
            out = torch.full((n_lights * n + 1, 5), -1, dtype=torch.int32, device=dev)
            out2 = torch.full((n_lights * n + 2, 5), -1, dtype=torch.int32, device=dev)  # 4-byte-aligned destination only
            torch.cuda.synchronize()  # torch fills on its own stream; the library's stream does not wait for it
            p.light_draw_lists(lights, out.data_ptr(), first_instance_base=3)
            got = out.cpu().numpy()
            assert got[:-1].tobytes() == want.tobytes(), (n, n_lights)
            assert (got[-1] == -1).all()  # nothing past the last list
            p.light_draw_lists(lights, out2.data_ptr() + 20, first_instance_base=3, async_=True)
            p.wait()
            assert out2.cpu().numpy()[1:-1].tobytes() == want.tobytes(), (n, n_lights, "unaligned")
            with pytest.raises(ra.MipError):
                p.light_draw_lists(np.zeros((17, 3), np.float32), out.data_ptr())
            with pytest.raises(ra.MipError):
                p.light_draw_lists(np.zeros((0, 3), np.float32), out.data_ptr())


def test_run_many_replays_recorded_launch_graphs(ra, oracle_mod):
    """mip_run_many sends whole rounds as one hipGraph per frame slot. Replays, replays after
    single launches changed the accumulator parity, other output rotations and a changed
    instance count all have to give the oracle's list in every output set."""
    import torch

    from renderer_amd.pipeline import make_frame

    big = ra.scene.make_scene(3, n=150_000)
    small = ra.scene.make_scene(3, n=33_000)
    want = {id(x): run_oracle(oracle_mod, x, threads=8, want=("draw_cmds",)) for x in (big, small)}
    dev = torch.device("cuda", 0)

    def check(sets, s, what):
        for k, (cmds, scal, _) in enumerate(sets):
            w = want[id(s)]
            count = int(scal[0].item())
            assert count == w["draw_count"] and int(scal[1].item()) == w["draw_index_total"], (what, k)
            got = cmds[:count].cpu().numpy().view(np.uint32).reshape(-1).view(ra.DRAW_CMD_DTYPE)
            assert got.tobytes() == w["draw_cmds"].tobytes(), (what, k)
            cmds.zero_()
            scal.zero_()
        torch.cuda.synchronize()

    for frames, n_sets in ((1, 1), (2, 2), (3, 3), (2, 3)):
        with ra.InstancePipeline(max_instances=big["n"], max_meshes=64, frames_in_flight=frames) as p:
            p.set_mesh_table(big["meshes"])
            p.set_instances(big["pos"], big["rot"], big["scale"], big["mesh_id"])
            sets = []
            for _ in range(n_sets):
                cmds = torch.zeros((big["n"], 5), dtype=torch.int32, device=dev)
                scal = torch.zeros(8, dtype=torch.int32, device=dev)
                sets.append((cmds, scal, p.prepare_outputs(draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(),
                                                            draw_index_total=scal.data_ptr() + 4)))
            torch.cuda.synchronize()
            frame = make_frame(big["planes"], big["cam_pos"])
            outs = [x[2] for x in sets]
            tag = (frames, n_sets)
            p.run_many(frame, outs, 200)  # rounds + a remainder of single launches
            p.wait()
            t = p.timings()
            assert t["graph_records"] == 1 and 128 <= t["graph_frames"] <= 200, t
            check(sets, big, (tag, "first"))
            p.run_many(frame, outs, 130)
            p.wait()
            check(sets, big, (tag, "again"))
            # odd numbers of single launches flip the accumulator parity under the recorded chains
            for k in range(3):
                p.run_prepared(p.frame_ref(frame), outs[k % n_sets])
            p.wait()
            p.run_many(frame, outs, 64 * 3)
            p.wait()
            check(sets, big, (tag, "after singles"))
            p.run_prepared(p.frame_ref(frame), outs[0])
            p.wait()
            check(sets[:1], big, (tag, "single after replay"))
            # a new instance count invalidates everything recorded
            p.set_instances(small["pos"], small["rot"], small["scale"], small["mesh_id"])
            p.run_many(make_frame(small["planes"], small["cam_pos"]), outs, 150)
            p.wait()
            check(sets, small, (tag, "resized"))
            assert p.timings()["graph_records"] >= 2


def test_epoch_tag_wrap_clears_the_prefix_state(ra, oracle_mod):
    """The 23-bit launch tag wraps after 8 388 606 launches per frame slot; the host then clears the
    prefix state and restarts at 1. Start two launches before the wrap and cross it."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
os.environ["MIP_TEST_EPOCH_START"] = str((1 << 23) - 1 - 3)
import numpy as np, renderer_amd, oracle
from renderer_amd import scene
s = scene.make_scene(3, n=70_000)
want = oracle.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"], threads=4, want=("draw_cmds",))
for frames in (1, 2):
    with renderer_amd.InstancePipeline(s["n"], 64, frames_in_flight=frames) as p:
        p.set_mesh_table(s["meshes"]); p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        for k in range(12):
            got = p.run_host(s["planes"], s["cam_pos"], want=("draw_cmds",))
            assert got["draw_count"] == want["draw_count"] and got["draw_cmds"].tobytes() == want["draw_cmds"].tobytes(), (frames, k)
# recorded chains need G+2 tags: a context that starts next to the wrap clears its state first
import torch
from renderer_amd.pipeline import make_frame
dev = torch.device("cuda", 0)
with renderer_amd.InstancePipeline(s["n"], 64, frames_in_flight=2) as p:
    p.set_mesh_table(s["meshes"]); p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
    sets = []
    for _ in range(2):
        cmds = torch.zeros((s["n"], 5), dtype=torch.int32, device=dev); scal = torch.zeros(8, dtype=torch.int32, device=dev)
        sets.append((cmds, scal, p.prepare_outputs(draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4)))
    torch.cuda.synchronize()
    for rep in range(3):
        p.run_many(make_frame(s["planes"], s["cam_pos"]), [x[2] for x in sets], 100); p.wait()
        for cmds, scal, _ in sets:
            c = int(scal[0].item())
            assert c == want["draw_count"] and cmds[:c].cpu().numpy().tobytes() == want["draw_cmds"].tobytes(), rep
    assert p.timings()["graph_frames"] == 3 * 64
print("WRAP_OK")
'''
    out = subprocess.run([sys.executable, "-c", code, root], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "WRAP_OK" in out.stdout, out.stderr[-2000:]


def test_device_resident_upload_and_timings(ra, oracle_mod):
    """mip_set_instances_device (columns already in HBM) and the hipEvent timing counters."""
    import torch

    from renderer_amd.pipeline import make_frame

    s = ra.scene.make_scene(3, n=50_000)
    dev = torch.device("cuda", 0)
    cols = [torch.from_numpy(np.ascontiguousarray(s[k])).to(dev) for k in ("pos", "rot", "scale")]
    mesh = torch.from_numpy(s["mesh_id"].astype(np.int32)).to(dev)
    torch.cuda.synchronize()
    with ra.InstancePipeline(max_instances=s["n"], max_meshes=64, timing=True) as p:
        p.set_mesh_table(s["meshes"])
        p.set_instances_device(cols[0].data_ptr(), cols[1].data_ptr(), cols[2].data_ptr(), mesh.data_ptr(), s["n"])
        got = p.run_host(s["planes"], s["cam_pos"])
        assert_parity(got, run_oracle(oracle_mod, s, threads=8), "device upload")
        model = torch.empty((s["n"], 16), dtype=torch.float32, device=dev)
        for _ in range(5):
            p.run_device(make_frame(s["planes"], s["cam_pos"]), model=model.data_ptr())
        t = p.timings()
        assert t["runs"] == 6 and 0 < t["last_kernel_ms"] < 5 and t["total_kernel_ms"] >= t["last_kernel_ms"]
        p.reset_timings()
        assert p.timings()["runs"] == 0


def test_gpu_against_float64_formulas(ra):
    """The north star's own criterion, without the oracle: matrices within 1e-5 relative of a float64
    evaluation; visibility identical wherever the float64 margin is not within rounding of zero."""
    import float64_reference as f64

    s = ra.scene.make_scene(3, n=200_000)
    got = run_gpu(ra, s, want=("model", "visible_bitmap"))
    b = f64.run(s)
    denom = np.maximum(np.abs(b["model"]), 1e-3 * np.abs(b["model"]).max(axis=1, keepdims=True))
    assert np.max(np.abs(got["model"].astype(np.float64) - b["model"]) / denom) < 1e-5  # relative tolerance 1e-5
    vis = np.unpackbits(got["visible_bitmap"].view(np.uint8), bitorder="little")[: s["n"]].astype(bool)
    d = b["decided"]
    assert np.array_equal(~vis[d], b["culled"][d]) and d.mean() > 0.99


def test_partial_instance_updates(ra, oracle_mod):
    """Moving entities: overwrite ranges of single columns between frames."""
    s = ra.scene.make_scene(3, n=10_000)
    with ra.InstancePipeline(max_instances=s["n"], max_meshes=64) as p:
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        rng = np.random.default_rng(5)
        for step in range(4):
            first = int(rng.integers(0, 9000))
            count = int(rng.integers(0, 1000))
            new_pos = rng.normal(0, 20, (count, 3)).astype(np.float32)
            s["pos"][first : first + count] = new_pos
            p.update_instances(first, pos_xyz=new_pos)
            if step % 2:
                new_scale = rng.uniform(0.1, 3, count).astype(np.float32)
                new_mesh = rng.integers(0, 64, count).astype(np.uint32)
                s["scale"][first : first + count] = new_scale
                s["mesh_id"][first : first + count] = new_mesh
                p.update_instances(first, scale=new_scale, mesh_id=new_mesh)
            assert_parity(p.run_host(s["planes"], s["cam_pos"]), run_oracle(oracle_mod, s), f"update {step}")
        with pytest.raises(ra.MipError):
            p.update_instances(9_990, pos_xyz=np.zeros((20, 3), np.float32))
        with pytest.raises(ra.MipError):
            p.update_instances(0, mesh_id=np.array([64], np.uint32))


def test_native_rccl_exchange_world_size_one(ra, oracle_mod):
    """mip_comm_* / mip_run_sharded: the library opens RCCL itself; one rank gathers with itself."""
    import torch

    from renderer_amd.pipeline import make_frame

    s = ra.scene.make_scene(3, n=150_000)
    want = run_oracle(oracle_mod, s, threads=8, want=("draw_cmds", "visible_bitmap"))
    dev = torch.device("cuda", 0)
    uid = ra.InstancePipeline.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    with ra.InstancePipeline(max_instances=s["n"], max_meshes=64) as p:
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        frame = make_frame(s["planes"], s["cam_pos"])
        merged = torch.zeros((s["n"], 5), dtype=torch.int32, device=dev)
        count = torch.zeros(2, dtype=torch.int32, device=dev)
        bitmap = torch.zeros((s["n"] + 31) // 32, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        with pytest.raises(ra.MipError):  # not initialised yet
            p.run_sharded(frame, merged.data_ptr(), count.data_ptr())
        p.comm_init(uid, 0, 1)
        for cap in (0, 60_000):   # full capacity, then a tightened chunk
            p.run_sharded(frame, merged.data_ptr(), count.data_ptr(), visible_bitmap=bitmap.data_ptr(), chunk_capacity=cap)
            total, index_total = (int(x) & 0xFFFFFFFF for x in count.cpu().tolist())
            assert total == want["draw_count"] and index_total == want["draw_index_total"]
            got = merged[:total].cpu().numpy().view(np.uint32).reshape(-1).view(ra.DRAW_CMD_DTYPE)
            assert got.tobytes() == want["draw_cmds"].tobytes()
            assert np.array_equal(bitmap.cpu().numpy().view(np.uint32), want["visible_bitmap"])
        # a chunk smaller than the rank's list: the frame is NOT lost — the library re-gathers it at full
        # capacity (synchronous call: at once; asynchronous call: inside mip_wait)
        for async_ in (False, True):
            merged.zero_(); count.zero_(); torch.cuda.synchronize()
            before = p.timings()["sharded_retries"]
            p.run_sharded(frame, merged.data_ptr(), count.data_ptr(), chunk_capacity=1000, async_=async_)
            if async_:
                p.wait()
            assert p.timings()["sharded_retries"] == before + 1
            total, index_total = (int(x) & 0xFFFFFFFF for x in count.cpu().tolist())
            assert total == want["draw_count"] and index_total == want["draw_index_total"]
            assert merged[:total].cpu().numpy().tobytes() == want["draw_cmds"].tobytes()
        # two overflowing frames queued before a wait: the first one's list is gone — reported, not repaired
        p.run_sharded(frame, merged.data_ptr(), count.data_ptr(), chunk_capacity=1000, async_=True)
        p.run_sharded(frame, merged.data_ptr(), count.data_ptr(), chunk_capacity=1000, async_=True)
        with pytest.raises(ra.MipError) as e:
            p.wait()
        assert e.value.code == -4
        p.run_sharded(frame, merged.data_ptr(), count.data_ptr())  # and the context is fine afterwards
        assert int(count[0].item()) == want["draw_count"]
        p.comm_destroy()
    with ra.InstancePipeline(max_instances=16, max_meshes=1, frames_in_flight=2) as p2:
        with pytest.raises(ra.MipError):
            p2.comm_init(uid, 0, 1)


def test_exchange_refuses_a_pipeline_on_another_stream(ra):
    """kernel -> all-gather -> merge are ordered by ONE stream; a context created on its own stream would race with
    the collective that torch issues on the current stream, so DrawListExchange.step says so instead of racing."""
    import torch

    from renderer_amd.sharded import DrawListExchange, make_shard_frame

    dev = torch.device("cuda", 0)
    s = ra.scene.make_scene(3, n=4096)
    with ra.InstancePipeline(max_instances=s["n"], max_meshes=64) as p:  # its own stream
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        ex = DrawListExchange(p, s["n"], 1, 0, dev)
        with pytest.raises(ValueError, match="current stream"):
            ex.step(make_shard_frame(s["planes"], s["cam_pos"], s["n"], 1, 0))
