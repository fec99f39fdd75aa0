"""Round-2 additions, all through the C ABI on the GPU: launch graphs that survive a moving camera,
the upload-time finite census that picks the kernel, both store/command orders of the kernel, and
zero-copy import of another process's allocation (row f-2)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers import assert_parity, run_oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ra():
    import renderer_amd

    renderer_amd.load_library()  # fails loudly if the HIP library is missing
    return renderer_amd


def _camera_path(s, k_frames):
    """k frames of a camera that translates and yaws: planes from the oracle's project_camera."""
    import oracle

    out = []
    for k in range(k_frames):
        t = k / max(k_frames - 1, 1)
        pos = np.array([-20.0 + 40.0 * t, 1.0 + 3.0 * np.sin(7 * t), 2.0 - 30.0 * t], np.float32)
        half = 0.5 * (0.9 * np.sin(5.0 * t))
        rot = np.array([0.0, np.sin(half), 0.0, np.cos(half)], np.float32)  # [i,j,k,w]
        out.append((oracle.project_camera(pos, rot), pos))
    return out


def test_moving_camera_replays_one_recorded_graph(ra, oracle_mod, monkeypatch):
    """mip_run_many with a different MipFrame every step (the renderer's frame loop: project_camera
    runs every frame, src/ecs.rs:66-91): the launches are recorded ONCE, every replay carries new
    planes through the device-side frame ring, and every frame's bitmap and command list is the
    oracle's for that frame's camera."""
    import torch

    from renderer_amd.pipeline import make_frame

    s = ra.scene.make_scene(3, n=8_000)
    n, K = s["n"], 256
    dev = torch.device("cuda", 0)
    monkeypatch.setenv("MIP_TUNE_GRAPH_ROUND", str(2 * K))  # one round = 2 K frames = two passes over K output sets
    with ra.InstancePipeline(max_instances=n, max_meshes=64) as p:
        monkeypatch.delenv("MIP_TUNE_GRAPH_ROUND")
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        cmds = torch.zeros((K, n, 5), dtype=torch.int32, device=dev)
        scal = torch.zeros((K, 8), dtype=torch.int32, device=dev)
        bits = torch.zeros((K, (n + 31) // 32), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        outs = [p.prepare_outputs(draw_cmds=cmds[k].data_ptr(), draw_count=scal[k].data_ptr(),
                                  draw_index_total=scal[k].data_ptr() + 4, visible_bitmap=bits[k].data_ptr())
                for k in range(K)]
        for rep in range(2):  # the second call moves the camera along a different path: same graphs
            path = _camera_path(s, 2 * K)
            if rep:
                path = path[::-1]
            frames = [make_frame(pl, pos) for pl, pos in path]
            p.run_many(frames, outs, 2 * K)
            p.wait()
            t = p.timings()
            assert t["graph_records"] == 1 and t["graph_frames"] == (rep + 1) * 2 * K, t
            counts = scal[:, 0].cpu().numpy()
            assert len(set(counts.tolist())) > K // 4  # the camera really moved
            for k in range(K):  # output set k holds step K + k of this call
                pl, pos = path[K + k]
                want = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], pl, pos,
                                      want=("visible_bitmap", "draw_cmds"))
                c = int(counts[k])
                assert c == want["draw_count"] and int(scal[k, 1].item()) == want["draw_index_total"], (rep, k)
                assert np.array_equal(bits[k].cpu().numpy().view(np.uint32), want["visible_bitmap"]), (rep, k)
                assert cmds[k, :c].cpu().numpy().tobytes() == want["draw_cmds"].tobytes(), (rep, k)
        # two frames in flight, four output sets, default round: still one recording per context
    with ra.InstancePipeline(max_instances=n, max_meshes=64, frames_in_flight=2) as p:
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        outs4 = outs[:4]
        path = _camera_path(s, 192)
        p.run_many([make_frame(pl, pos) for pl, pos in path], outs4, 192)
        p.wait()
        t = p.timings()
        assert t["graph_records"] == 1 and t["graph_frames"] == 192, t
        for k in range(4):  # the last four steps
            pl, pos = path[188 + k]
            want = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], pl, pos, want=("draw_cmds",))
            c = int(scal[k, 0].item())
            assert c == want["draw_count"] and cmds[k, :c].cpu().numpy().tobytes() == want["draw_cmds"].tobytes(), k


def test_finite_census_picks_the_kernel(ra, oracle_mod):
    """Uploads are scanned once with the kernel's own finite test. All finite: the frame runs the kernel
    without the literal cold path (general_launches stays 0). One NaN / inf / overflowing quaternion
    anywhere: the kernel with that path. Partial updates keep the count exact in both directions."""
    s = ra.scene.make_scene(3, n=30_000)
    with ra.InstancePipeline(max_instances=s["n"], max_meshes=64) as p:
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        assert_parity(p.run_host(s["planes"], s["cam_pos"]), run_oracle(oracle_mod, s), "finite")
        assert p.timings()["general_launches"] == 0
        for what, col, row, val in (("nan position", "pos", 12_345, [np.nan, 1.0, 2.0]),
                                    ("inf scale", "scale", 77, np.inf),
                                    ("huge quaternion", "rot", 29_999, [3e19, 3e19, 0.0, 1.0]),  # w*w, i*i overflow
                                    # finite inputs whose corner arithmetic overflows: the corner-enumerating tier
                                    ("huge scale", "scale", 123, 3e37), ("far position", "pos", 4_000, [2e37, -3e37, 1e37])):
            old = s[col][row].copy()
            s[col][row] = val
            kw = {"pos": "pos_xyz", "rot": "rot_ijkw", "scale": "scale"}[col]
            p.update_instances(row, **{kw: s[col][row:row + 1]})
            before = p.timings()["general_launches"]
            assert_parity(p.run_host(s["planes"], s["cam_pos"]), run_oracle(oracle_mod, s), what)
            assert p.timings()["general_launches"] == before + 1, what
            s[col][row] = old
            p.update_instances(row, **{kw: s[col][row:row + 1]})
            before = p.timings()["general_launches"]
            assert_parity(p.run_host(s["planes"], s["cam_pos"]), run_oracle(oracle_mod, s), what + " restored")
            assert p.timings()["general_launches"] == before, what
        # two bad instances, one repaired: still the general kernel
        s["pos"][5] = np.nan
        s["pos"][6] = np.inf
        p.update_instances(5, pos_xyz=s["pos"][5:7])
        s["pos"][5] = 0.0
        p.update_instances(5, pos_xyz=s["pos"][5:6])
        before = p.timings()["general_launches"]
        assert_parity(p.run_host(s["planes"], s["cam_pos"]), run_oracle(oracle_mod, s), "one of two repaired")
        assert p.timings()["general_launches"] == before + 1


@pytest.mark.parametrize("order", [1, 3])
@pytest.mark.parametrize("general", [0, 1])
def test_both_kernel_orders_and_both_arithmetic_paths(ra, oracle_mod, order, general, monkeypatch):
    """The host picks the store/command order by launch size and the arithmetic variant by the census;
    every combination has to give the same bytes (forced here on sizes either side of the switch)."""
    monkeypatch.setenv("MIP_TUNE_ORDER", str(order))
    monkeypatch.setenv("MIP_TUNE_FORCE_GENERAL", str(general))
    for n in (1, 255, 257, 70_001):
        s = ra.scene.make_scene(3, n=n)
        with ra.InstancePipeline(max_instances=n, max_meshes=64) as p:
            p.set_mesh_table(s["meshes"])
            p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
            assert_parity(p.run_host(s["planes"], s["cam_pos"]), run_oracle(oracle_mod, s), (order, general, n))


_EXPORTER = r'''
import ctypes as C, os, socket, sys
import numpy as np, torch
sock_path, nbytes = sys.argv[1], int(sys.argv[2])
buf = torch.zeros(nbytes // 4, dtype=torch.int32, device="cuda:0")
torch.cuda.synchronize()
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipMemGetHandleForAddressRange.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_ulonglong]
hip.hipMemGetHandleForAddressRange.restype = C.c_int
fd = C.c_int(-1)
rc = hip.hipMemGetHandleForAddressRange(C.byref(fd), C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel() * 4), 1, 0)  # hipMemRangeHandleTypeDmaBufFd
s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
s.connect(sock_path)
if rc != 0 or fd.value < 0:
    s.sendall(b"EXPORT_FAILED %d" % rc); sys.exit(0)
socket.send_fds(s, [b"FD"], [fd.value])
assert s.recv(16) == b"DONE"           # the importer has run its frame and waited for it
torch.cuda.synchronize()
host = buf.cpu().numpy()
s.sendall(len(host.tobytes()).to_bytes(8, "little") + host.tobytes())
s.close()
'''


def test_external_memory_fd_import_zero_copy(ra, oracle_mod, tmp_path):
    """Row f-2, the HIP half: a buffer that ANOTHER process allocated (standing in for the renderer's VMA
    allocation exported with vkGetMemoryFdKHR; on amdgpu such an fd is a dma-buf) is imported with
    mip_import_external_fd, a frame writes its model matrices, bitmap and command list straight into it,
    and the OWNER of the allocation reads the oracle's bytes out of its own pointer."""
    s = ra.scene.make_scene(3, n=20_000)
    n = s["n"]
    want = run_oracle(oracle_mod, s, want=("model", "visible_bitmap", "draw_cmds"))
    off_model, off_cmds, off_bits, off_scal = 0, n * 64, n * 64 + n * 20, n * 64 + n * 20 + ((n + 31) // 32) * 4
    off_scal = (off_scal + 255) // 256 * 256
    nbytes = (off_scal + 256 + (1 << 21) - 1) // (1 << 21) * (1 << 21)  # whole 2 MiB pages
    sock_path = str(tmp_path / "fd.sock")
    srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    srv.bind(sock_path)
    srv.listen(1)
    srv.settimeout(120)
    child = subprocess.Popen([sys.executable, "-c", _EXPORTER, sock_path, str(nbytes)], stderr=subprocess.PIPE)
    try:
        conn, _ = srv.accept()
        conn.settimeout(120)
        msg, fds, _, _ = socket.recv_fds(conn, 64, 1)
        if not fds:
            pytest.fail(f"the exporter could not export a dma-buf fd: {msg!r}")
        with ra.InstancePipeline(max_instances=n, max_meshes=64) as p:
            p.set_mesh_table(s["meshes"])
            p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
            base = p.import_external_fd(fds[0], nbytes)
            assert base
            from renderer_amd.pipeline import make_frame

            p.run_device(make_frame(s["planes"], s["cam_pos"]), model=base + off_model, draw_cmds=base + off_cmds,
                         visible_bitmap=base + off_bits, draw_count=base + off_scal, draw_index_total=base + off_scal + 4)
            p.wait()
            with pytest.raises(ra.MipError):
                p.release_external(base + 4)
            p.release_external(base)
        conn.sendall(b"DONE")
        size = int.from_bytes(_recv_exact(conn, 8), "little")
        raw = np.frombuffer(_recv_exact(conn, size), dtype=np.uint8)
    finally:
        child.wait(timeout=120)
        srv.close()
    assert child.returncode == 0, child.stderr.read().decode()[-2000:]
    count, index_total = (int(x) for x in raw[off_scal:off_scal + 8].view(np.uint32))
    assert count == want["draw_count"] and index_total == want["draw_index_total"]
    assert raw[off_model:off_model + n * 64].tobytes() == want["model"].tobytes()
    assert raw[off_cmds:off_cmds + count * 20].tobytes() == want["draw_cmds"].tobytes()
    assert raw[off_bits:off_bits + ((n + 31) // 32) * 4].tobytes() == want["visible_bitmap"].tobytes()


def _recv_exact(conn, k):
    out = bytearray()
    while len(out) < k:
        chunk = conn.recv(min(1 << 20, k - len(out)))
        if not chunk:
            raise RuntimeError("peer closed")
        out += chunk
    return bytes(out)


def test_views_with_the_ignored_ordered_tiles_flag(ra, oracle_mod):
    """(ABI <= 3: a context in ordered-tiles mode ran one ticketed frame per view; the flag is ignored now)
    five views (two launches of the multi-view kernel): each equal to the oracle's frame for that view."""
    import torch

    from renderer_amd.pipeline import make_frame

    s = ra.scene.make_scene(3, n=30_011)
    n = s["n"]
    rng = np.random.default_rng(5)
    dev = torch.device("cuda", 0)
    views = []
    for v in range(5):
        lp = rng.normal(0, 12, 3).astype(np.float32)
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        views.append((lp, oracle_mod.project_camera(lp, q.astype(np.float32), aspect=1.3, fovy_degrees=80.0, near=0.2, far=250.0), 7 * v, 1000 * v))
    with ra.InstancePipeline(max_instances=n, max_meshes=64, ordered_tiles=True) as p:
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        bufs = [(torch.full((n, 5), -1, dtype=torch.int32, device=dev), torch.full((8,), -1, dtype=torch.int32, device=dev),
                 torch.zeros((n + 31) // 32 + 1, dtype=torch.int32, device=dev)) for _ in views]
        torch.cuda.synchronize()
        frames = [make_frame(pl, lp, first_instance_base=fib, first_index_base=fxb) for lp, pl, fib, fxb in views]
        outs = [p.prepare_outputs(draw_cmds=c.data_ptr(), draw_count=sc.data_ptr(), draw_index_total=sc.data_ptr() + 4,
                                  visible_bitmap=b.data_ptr(), async_=False) for c, sc, b in bufs]
        for _ in range(2):
            p.run_views(frames, outs)
        for (lp, pl, fib, fxb), (cmds, scal, bitmap) in zip(views, bufs):
            want = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], pl, lp, first_instance_base=fib,
                                  first_index_base=fxb, threads=8, want=("draw_cmds", "visible_bitmap"))
            count, total = (int(x) & 0xFFFFFFFF for x in scal[:2].cpu().tolist())
            assert count == want["draw_count"] and total == want["draw_index_total"]
            assert cmds[:count].cpu().numpy().tobytes() == want["draw_cmds"].tobytes()
            assert np.array_equal(bitmap[:(n + 31) // 32].cpu().numpy().view(np.uint32), want["visible_bitmap"])


def test_mesh_table_and_geometry_must_agree_before_the_triangle_stage(ra):
    """The per-triangle kernels gather vertices[vertex_offset + index] and indices[index_offset ..] unchecked, so
    the host refuses a frame whose mesh table points outside the uploaded geometry (checked once per upload)."""
    import torch

    from renderer_amd.pipeline import make_frame

    s = ra.scene.make_scene(2, n=500)
    vertices, indices = ra.scene.make_geometry(s["meshes"])
    dev = torch.device("cuda", 0)
    model = torch.zeros((s["n"], 16), dtype=torch.float32, device=dev)
    cmds = torch.zeros((s["n"], 5), dtype=torch.int32, device=dev)
    scal = torch.zeros(8, dtype=torch.int32, device=dev)
    out = torch.zeros(1 << 22, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    frame = make_frame(s["planes"], s["cam_pos"], pv=ra.scene.default_pv())

    def run(p):
        p.run_device(frame, model=model.data_ptr(), draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(),
                     culled_index_buffer=out.data_ptr(), culled_index_capacity=out.numel())

    with ra.InstancePipeline(max_instances=s["n"], max_meshes=len(s["meshes"])) as p:
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        p.set_geometry(vertices, indices)
        run(p)                                                # consistent: fine
        p.set_geometry(vertices, indices[: len(indices) // 2])  # LOD ranges now end outside the index buffer
        with pytest.raises(ra.MipError) as e:
            run(p)
        assert e.value.code == -1 and "outside" in str(e.value)
        bad = indices.copy()
        bad[5] = len(vertices)                                # one index one past the vertices
        p.set_geometry(vertices, bad)
        with pytest.raises(ra.MipError) as e:
            run(p)
        assert e.value.code == -1 and "largest index" in str(e.value)
        m2 = s["meshes"].copy()
        m2["vertex_offset"][0] = -1
        p.set_geometry(vertices, indices)
        p.set_mesh_table(m2)
        with pytest.raises(ra.MipError):
            run(p)
        p.set_mesh_table(s["meshes"])
        run(p)                                                # and consistent again
        # frames that do not ask for the per-triangle stage are not affected by any of this
        p.set_geometry(vertices, indices[:10])
        p.run_device(frame, model=model.data_ptr(), draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr())


_SHARDED_RANK = r'''
import os, sys
root, rank, world, n_global, id_path, out_path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import time
import numpy as np, torch
import renderer_amd, oracle
from renderer_amd import scene
from renderer_amd.pipeline import make_frame
from renderer_amd.sharded import shard_range

full = scene.make_scene(3, n=n_global)
lo, hi = shard_range(n_global, world, rank)
s = scene.make_scene(3, n=hi - lo, first=lo)
if rank == 0:
    uid = renderer_amd.InstancePipeline.comm_unique_id()
    open(id_path + ".tmp", "wb").write(uid); os.rename(id_path + ".tmp", id_path)
else:
    t0 = time.time()
    while not os.path.exists(id_path):
        assert time.time() - t0 < 900, 'rank 0 never wrote the communicator id (a rendezvous guard, far above any start-up time)'
        time.sleep(0.01)
    uid = open(id_path, "rb").read()
dev = torch.device("cuda", 0)
cam_b = np.array([0.0, 1.0, -40.0], np.float32)            # sees more than the default camera
planes_b = oracle.project_camera(cam_b, (0.0, 0.0, 0.0, 1.0))
want_a = oracle.run(full["pos"], full["rot"], full["scale"], full["mesh_id"], full["meshes"], full["planes"], full["cam_pos"], want=("draw_cmds", "visible_bitmap"))
want_b = oracle.run(full["pos"], full["rot"], full["scale"], full["mesh_id"], full["meshes"], planes_b, cam_b, want=("draw_cmds",))
with renderer_amd.InstancePipeline(max_instances=hi - lo, max_meshes=64) as p:
    p.set_mesh_table(s["meshes"]); p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
    p.comm_init(uid, rank, world)
    merged = torch.zeros((n_global, 5), dtype=torch.int32, device=dev)
    count = torch.zeros(2, dtype=torch.int32, device=dev)
    bitmap = torch.zeros((hi - lo + 31) // 32, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    frame_a = make_frame(full["planes"], full["cam_pos"], first_instance_base=lo)
    frame_b = make_frame(planes_b, cam_b, first_instance_base=lo)
    def check(want, what):
        total, index_total = (int(x) & 0xFFFFFFFF for x in count.cpu().tolist())
        assert total == want["draw_count"] and index_total == want["draw_index_total"], (what, total, want["draw_count"])
        assert merged[:total].cpu().numpy().tobytes() == want["draw_cmds"].tobytes(), what
    p.run_sharded(frame_a, merged.data_ptr(), count.data_ptr(), visible_bitmap=bitmap.data_ptr())           # full capacity
    check(want_a, "A full")
    vis_local = np.unpackbits(bitmap.cpu().numpy().view(np.uint8), bitorder="little")[: hi - lo]
    assert np.array_equal(vis_local, np.unpackbits(want_a["visible_bitmap"].view(np.uint8), bitorder="little")[lo:hi])
    cap = int(max(1, want_a["draw_count"] // world * 1.1))    # fits camera A's shards, not camera B's
    p.run_sharded(frame_a, merged.data_ptr(), count.data_ptr(), chunk_capacity=cap)
    check(want_a, "A tightened")
    assert p.timings()["sharded_retries"] == 0
    for async_ in (False, True):                              # the camera moves: every rank sees the overflow and repairs it together
        merged.zero_(); torch.cuda.synchronize()
        before = p.timings()["sharded_retries"]
        p.run_sharded(frame_b, merged.data_ptr(), count.data_ptr(), chunk_capacity=cap, async_=async_)
        if async_:
            p.wait()
        check(want_b, "B repaired")
        assert p.timings()["sharded_retries"] == before + 1
    p.run_sharded(frame_a, merged.data_ptr(), count.data_ptr(), chunk_capacity=cap)
    check(want_a, "A again")
    p.comm_destroy()
open(out_path, "w").write("ok")
'''


@pytest.mark.parametrize("world,wire", [(2, 2), (3, 2), (2, 1), (2, 0)])
def test_native_sharded_frame_with_several_ranks_on_one_gpu(ra, tmp_path, world, wire):
    """mip_comm_init / mip_run_sharded with world size 2 and 3: one process per rank, all on this box's one GPU,
    the collective library replaced through the MIP_COMM_LIBRARY seam by tests/fake_ccl (a shared-memory double of
    the five nccl* entry points). The library's own sequence — shard kernel -> all-gather -> merge — and its
    collective overflow repair run unchanged; every rank must hold the unsharded oracle's list."""
    fake = str(tmp_path / "libfake_rccl.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "fake_ccl", "fake_rccl.cpp"), "-o", fake, "-L/opt/rocm/lib", "-lamdhip64", "-lrt"])
    # wire = 2 (default): the lists travel as packed 4-byte records (the shards fit), 1: as 8-byte records, and the merge expands
    # them; 0: as 20-byte commands
    env = dict(os.environ, MIP_COMM_LIBRARY=fake, MIP_TUNE_SHARD_WIRE=str(wire))
    id_path = str(tmp_path / "uid")
    procs = [subprocess.Popen([sys.executable, "-c", _SHARDED_RANK, ROOT, str(r), str(world), "90001", id_path, str(tmp_path / f"ok{r}")],
                              env=env, stderr=subprocess.PIPE, text=True) for r in range(world)]
    errs = [p.communicate(timeout=300)[1] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0 and os.path.exists(tmp_path / f"ok{r}"), f"rank {r}:\n{errs[r][-3000:]}"


def test_recorded_graphs_follow_the_census(ra, oracle_mod):
    """Recorded launch graphs name a kernel instantiation. When a partial update makes the first instance non-finite
    (or the last one finite again) the choice changes, and mip_run_many has to re-record instead of replaying the
    kernel without the fall-back arithmetic."""
    import torch

    from renderer_amd.pipeline import make_frame

    s = ra.scene.make_scene(3, n=40_000)
    n = s["n"]
    dev = torch.device("cuda", 0)
    with ra.InstancePipeline(max_instances=n, max_meshes=64, frames_in_flight=2) as p:
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        sets = []
        for _ in range(2):
            cmds = torch.zeros((n, 5), dtype=torch.int32, device=dev)
            scal = torch.zeros(8, dtype=torch.int32, device=dev)
            model = torch.zeros((n, 16), dtype=torch.float32, device=dev)
            sets.append((cmds, scal, model, p.prepare_outputs(draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(),
                                                                draw_index_total=scal.data_ptr() + 4, model=model.data_ptr())))
        torch.cuda.synchronize()
        frame = make_frame(s["planes"], s["cam_pos"])
        outs = [x[3] for x in sets]

        def check(what):
            want = run_oracle(oracle_mod, s, want=("draw_cmds", "model"))
            for cmds, scal, model, _ in sets:
                c = int(scal[0].item())
                assert c == want["draw_count"] and cmds[:c].cpu().numpy().tobytes() == want["draw_cmds"].tobytes(), what
                got = model.cpu().numpy()
                assert bool(np.all((got == want["model"]) | (np.isnan(got) & np.isnan(want["model"])))), what

        p.run_many(frame, outs, 128); p.wait()
        check("finite")
        assert p.timings()["graph_records"] == 1 and p.timings()["general_launches"] == 0
        s["pos"][123] = [np.nan, 0.0, 1.0]
        p.update_instances(123, pos_xyz=s["pos"][123:124])
        p.run_many(frame, outs, 128); p.wait()
        check("one NaN")
        assert p.timings()["graph_records"] == 2
        s["pos"][123] = [1.0, 2.0, 3.0]
        p.update_instances(123, pos_xyz=s["pos"][123:124])
        p.run_many(frame, outs, 128); p.wait()
        check("finite again")
        assert p.timings()["graph_records"] == 3


@pytest.mark.parametrize("order", [1, 3])
def test_one_mesh_scene_through_the_gather_path(ra, oracle_mod, order, monkeypatch):
    """The one-mesh scene with MIP_TUNE_NO_ONE_MESH (read when the context is created): mesh ids loaded and the entry gathered
    like for any other table — the same bytes as the scalar path the other tests of this scene take."""
    monkeypatch.setenv("MIP_TUNE_NO_ONE_MESH", "1")
    monkeypatch.setenv("MIP_TUNE_ORDER", str(order))
    for n in (1, 300, 100_000):
        s = ra.scene.make_scene(2, n=n)
        with ra.InstancePipeline(max_instances=n, max_meshes=1) as p:
            p.set_mesh_table(s["meshes"])
            p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
            assert_parity(p.run_host(s["planes"], s["cam_pos"]), run_oracle(oracle_mod, s), (order, n))


def test_a_mesh_table_that_grows_from_one_entry_rerecords_and_gathers(ra, oracle_mod):
    """A table of ONE mesh is read as a scalar by the frame kernel (no id load, no gather: KernelArgs.one_mesh), and a recorded
    launch carries that choice in its arguments. The table growing to two entries and instances moving to the new one must give
    the oracle's bytes — from direct launches (the gather path, chosen per launch) and from run_many, which has to record again."""
    import torch

    from renderer_amd.pipeline import make_frame

    s = ra.scene.make_scene(2, n=30_000)   # the one-mesh scene
    n = s["n"]
    dev = torch.device("cuda", 0)
    two = np.concatenate([s["meshes"], ra.scene.mixed_mesh_table()[7:8]])
    with ra.InstancePipeline(max_instances=n, max_meshes=2, frames_in_flight=2) as p:
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        sets = []
        for _ in range(2):
            cmds = torch.zeros((n, 5), dtype=torch.int32, device=dev)
            scal = torch.zeros(8, dtype=torch.int32, device=dev)
            sets.append((cmds, scal, p.prepare_outputs(draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4)))
        torch.cuda.synchronize()
        frame = make_frame(s["planes"], s["cam_pos"])
        outs = [x[2] for x in sets]

        def check(what):
            want = run_oracle(oracle_mod, s, want=("draw_cmds",))
            for cmds, scal, _ in sets:
                c = int(scal[0].item())
                assert c == want["draw_count"] and cmds[:c].cpu().numpy().tobytes() == want["draw_cmds"].tobytes(), what
            got = p.run_host(s["planes"], s["cam_pos"])
            assert_parity(got, run_oracle(oracle_mod, s), what + ", direct launch")

        p.run_many(frame, outs, 64); p.wait()
        check("one mesh")
        records = p.timings()["graph_records"]
        s["meshes"] = two
        p.set_mesh_table(two)
        s["mesh_id"] = s["mesh_id"].copy()
        s["mesh_id"][1000:20_000:3] = 1
        p.update_instances(0, mesh_id=s["mesh_id"])
        p.run_many(frame, outs, 64); p.wait()
        check("two meshes")
        assert p.timings()["graph_records"] == records + 1


@pytest.mark.parametrize("order", [1, 3])
def test_streamed_outputs_stop_at_the_last_instance(ra, oracle_mod, order, monkeypatch):
    """The matrix and TLAS streams go out through buffer descriptors bounded at the tile's last instance
    (the hardware drops lanes past it, there is no range test in the kernel): ragged sizes must fill exactly
    n x 64 bytes — the words before and after stay untouched — with the oracle's values."""
    import torch

    from renderer_amd.pipeline import make_frame

    monkeypatch.setenv("MIP_TUNE_ORDER", str(order))
    dev = torch.device("cuda", 0)
    guard = 1024  # floats either side
    for n in (1, 3, 255, 256, 257, 1000, 65_537):
        s = ra.scene.make_scene(3, n=n)
        want = run_oracle(oracle_mod, s)
        with ra.InstancePipeline(max_instances=n, max_meshes=64) as p:
            p.set_mesh_table(s["meshes"])
            p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
            model = torch.full((guard + n * 16 + guard,), -7.25, dtype=torch.float32, device=dev)
            tlas = torch.full((guard + n * 16 + guard,), -7.25, dtype=torch.float32, device=dev)
            cmds = torch.zeros((n, 5), dtype=torch.int32, device=dev)
            scal = torch.zeros(8, dtype=torch.int32, device=dev)
            p.run_device(make_frame(s["planes"], s["cam_pos"]), model=model.data_ptr() + guard * 4,
                         tlas_instances=tlas.data_ptr() + guard * 4, draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(),
                         draw_index_total=scal.data_ptr() + 4)
            torch.cuda.synchronize()
            m, t = model.cpu().numpy(), tlas.cpu().numpy()
        for name, arr in (("model", m), ("tlas", t)):
            assert (arr[:guard] == -7.25).all() and (arr[guard + n * 16:] == -7.25).all(), (name, order, n)
            assert not (arr[guard:guard + n * 16].view(np.uint32) == np.float32(-7.25).view(np.uint32)).all(), (name, order, n)
        got = m[guard:guard + n * 16].reshape(n, 16)
        np.testing.assert_array_equal(got.view(np.uint32), np.asarray(want["model"], np.float32).reshape(n, 16).view(np.uint32), err_msg=str((order, n)))
