"""Round-3 additions, all through the C ABI on the GPU: the wire form of a shard's draw list (8-byte records
through the all-gather, expanded by the merge), the sharded frame beside a collective kernel that spin-waits on the
device, the overflow repair of a pipelined exchange under its own stream, and the hardened device-pointer upload."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers import report_timing_property, run_oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ra():
    import renderer_amd

    renderer_amd.load_library()  # fails loudly if the HIP library is missing
    return renderer_amd


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _far_bits(oracle_mod, s, cmds, base, cam):
    inst = (cmds["firstInstance"] - np.uint32(base)).astype(np.int64)
    d = np.asarray(cam, np.float32)[None, :] - s["pos"][inst]
    # pick_lod's distance test (helpers.rs:4) vectorised: sqrt_rn(q) > 10 <=> q > nextafter(100)
    q = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + (d[:, 2] * d[:, 2]).astype(np.float32)
    far = (q.astype(np.float32) > np.float32(100.00000762939453125)).astype(np.uint32)
    for k in range(0, len(inst), max(1, len(inst) // 64)):  # spot-check against the oracle's scalar pick_lod
        assert far[k] == oracle_mod.pick_lod(2, cam, s["pos"][inst[k]])
    return inst, far


@pytest.mark.parametrize("n", [1, 255, 300, 1024, 4_097, 100_000, 1_000_003])
def test_wire_list_is_the_command_list(ra, oracle_mod, n):
    """MIP_OUT_WIRE: the kernel's wire bytes are the numpy restatement's (tests/cpu_pipeline.py encode_wire) of the
    oracle's command list, and expanding them against the mesh table gives the oracle's 20-byte commands back."""
    import torch

    from cpu_pipeline import decode_wire, encode_wire, encode_wire_packed, unpack_wire, wire_live_mask
    from renderer_amd.pipeline import make_frame, wire_body_bytes

    s = ra.scene.make_scene(3, n=n)
    if n == 300:
        s = ra.scene.make_scene(3, n=n, all_visible=True)  # 300 commands: crosses a block boundary at a non-tile position
    dev = torch.device("cuda", 0)
    base, index_base = 1000, 77
    want = run_oracle(oracle_mod, s, threads=8, want=("draw_cmds",), first_instance_base=base, first_index_base=index_base)
    cmds = want["draw_cmds"]
    with ra.InstancePipeline(max_instances=n, max_meshes=64) as p:
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        body = torch.full((wire_body_bytes(n) // 4 + 4,), 0x5A5A5A5A, dtype=torch.int32, device=dev)
        scal = torch.zeros(8, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        frame = make_frame(s["planes"], s["cam_pos"], first_instance_base=base, first_index_base=index_base)
        for order in (None, "1", "3"):
            if order:
                os.environ["MIP_TUNE_ORDER"] = order
            try:
                with ra.InstancePipeline(max_instances=n, max_meshes=64) as q:
                    q.set_mesh_table(s["meshes"])
                    q.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
                    body.fill_(0x5A5A5A5A)
                    torch.cuda.synchronize()
                    q.run_device(frame, draw_cmds=body.data_ptr(), draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4, wire=True)
            finally:
                os.environ.pop("MIP_TUNE_ORDER", None)
            count, total = (int(x) & 0xFFFFFFFF for x in scal[:2].cpu().tolist())
            assert count == want["draw_count"] and total == want["draw_index_total"]
            got = body.cpu().numpy().view(np.uint32)
            inst, far = _far_bits(oracle_mod, s, cmds, base, s["cam_pos"])
            ref = encode_wire(cmds, s["mesh_id"][inst], far)
            live = wire_live_mask(count)   # anchors of the sub-blocks that exist + the records of the live commands; the rest is never written
            assert np.array_equal(got[: ref.size][live], ref[live]), f"order {order}: anchors and records"
            assert np.all(got[: ref.size][~live] == 0x5A5A5A5A) and np.all(got[ref.size:] == 0x5A5A5A5A), "nothing else is written"
            assert decode_wire(got, count, s["meshes"]).tobytes() == cmds.tobytes()
            # the packed form (MIP_OUT_WIRE_PACKED) of the same frame, same kernel order
            if order:
                os.environ["MIP_TUNE_ORDER"] = order
            try:
                with ra.InstancePipeline(max_instances=n, max_meshes=64) as q:
                    q.set_mesh_table(s["meshes"])
                    q.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
                    body.fill_(0x5A5A5A5A)
                    scal.zero_()
                    torch.cuda.synchronize()
                    q.run_device(frame, draw_cmds=body.data_ptr(), draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4, wire="packed")
            finally:
                os.environ.pop("MIP_TUNE_ORDER", None)
            assert [int(x) & 0xFFFFFFFF for x in scal[:2].cpu().tolist()] == [count, total]
            got = body.cpu().numpy().view(np.uint32)
            refp = encode_wire_packed(cmds, s["mesh_id"][inst], far, base, len(s["meshes"]))
            livep = wire_live_mask(count, packed=True)   # every block header {firstIndex, base, index bits, 0} + the live records
            assert np.array_equal(got[: refp.size][livep], refp[livep]), f"order {order}: packed headers and records"
            assert np.all(got[: refp.size][~livep] == 0x5A5A5A5A) and np.all(got[refp.size:] == 0x5A5A5A5A), "nothing else is written"
            assert decode_wire(unpack_wire(got, count), count, s["meshes"]).tobytes() == cmds.tobytes()
        with pytest.raises(ra.MipError):  # the wire form cannot carry the per-triangle stage's counts
            p.run_device(frame, model=body.data_ptr(), draw_cmds=body.data_ptr(), draw_count=scal.data_ptr(), wire=True,
                         culled_index_buffer=body.data_ptr(), culled_index_capacity=4)


def test_wire_form_with_non_finite_instances_and_the_ignored_ordered_tiles_flag(ra, oracle_mod, monkeypatch):
    """The wire form is a template parameter of the frame kernel: the general (literal arithmetic) instantiation carries it
    too. (MIP_CFG_ORDERED_TILES, which up to ABI 3 selected a ticketed instantiation, is accepted and changes nothing.)"""
    import torch

    from cpu_pipeline import decode_wire, unpack_wire
    from renderer_amd.pipeline import make_frame, wire_body_bytes

    dev = torch.device("cuda", 0)
    s = ra.scene.make_scene(3, n=90_001)
    s["pos"][17] = np.nan
    s["scale"][4000] = np.inf
    want = run_oracle(oracle_mod, s, threads=8, want=("draw_cmds",))
    for ordered in (False, True):
        with ra.InstancePipeline(max_instances=s["n"], max_meshes=64, ordered_tiles=ordered) as p:
            p.set_mesh_table(s["meshes"])
            p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
            body = torch.zeros(wire_body_bytes(s["n"]) // 4, dtype=torch.int32, device=dev)
            scal = torch.zeros(8, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            for _ in range(2):
                p.run_device(make_frame(s["planes"], s["cam_pos"]), draw_cmds=body.data_ptr(), draw_count=scal.data_ptr(), wire=True)
            count = int(scal[0].item())
            assert count == want["draw_count"] and p.timings()["general_launches"] == 2  # the NaN / inf instances select the literal tier
            assert decode_wire(body.cpu().numpy().view(np.uint32), count, s["meshes"]).tobytes() == want["draw_cmds"].tobytes(), ordered
            scal.zero_()
            torch.cuda.synchronize()
            p.run_device(make_frame(s["planes"], s["cam_pos"]), draw_cmds=body.data_ptr(), draw_count=scal.data_ptr(), wire="packed")
            assert int(scal[0].item()) == count and p.timings()["general_launches"] == 3
            assert decode_wire(unpack_wire(body.cpu().numpy().view(np.uint32), count), count, s["meshes"]).tobytes() == want["draw_cmds"].tobytes(), ordered


@pytest.mark.parametrize("n_global,world", [(10, 3), (70_001, 3), (1_000_000, 8)])
def test_wire_merge_equals_the_command_merge(ra, oracle_mod, n_global, world):
    """Shards of one scene run into wire chunks and into 20-byte chunks laid out as an all-gather would; both merges
    give the unsharded oracle's list, byte for byte — also with a tightened capacity, whose overflow is reported."""
    import torch

    from renderer_amd.pipeline import SHARD_HEADER_BYTES, make_frame
    from renderer_amd.sharded import chunk_stride_bytes, shard_range

    dev = torch.device("cuda", 0)
    full = ra.scene.make_scene(3, n=n_global)
    want = run_oracle(oracle_mod, full, threads=8, want=("draw_cmds",))
    per = (n_global + world - 1) // world
    forms = (True, "packed", False)
    strides = {w: chunk_stride_bytes(per, wire=w) for w in forms}
    assert strides["packed"] < 0.54 * strides[True] or per < 512
    recv = {w: torch.zeros(world * strides[w] // 4, dtype=torch.int32, device=dev) for w in forms}
    counts = []
    for r in range(world):
        lo, hi = shard_range(n_global, world, r)
        sh = ra.scene.make_scene(3, n=hi - lo, first=lo)
        with ra.InstancePipeline(max_instances=max(hi - lo, 1), max_meshes=64) as p:
            p.set_mesh_table(sh["meshes"])
            p.set_instances(sh["pos"], sh["rot"], sh["scale"], sh["mesh_id"])
            frame = make_frame(full["planes"], full["cam_pos"], first_instance_base=lo)
            for w in forms:
                base = recv[w].data_ptr() + r * strides[w]
                p.run_device(frame, draw_cmds=base + SHARD_HEADER_BYTES, draw_count=base, draw_index_total=base + 4, wire=w)
            counts.append(int(recv[True][r * strides[True] // 4].item()))
    assert sum(counts) == want["draw_count"]
    merged = torch.zeros((world * per + 1, 5), dtype=torch.int32, device=dev)
    scal = torch.zeros(2, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    with ra.InstancePipeline(max_instances=1, max_meshes=64) as p:
        p.set_mesh_table(full["meshes"])
        for w in forms:
            merged.fill_(-1)
            torch.cuda.synchronize()
            if w:
                p.merge_wire_lists(recv[w].data_ptr(), world, strides[w], merged.data_ptr(), scal.data_ptr(), chunk_capacity=per, packed=w == "packed")
            else:
                p.merge_draw_lists(recv[w].data_ptr(), world, strides[w], merged.data_ptr(), scal.data_ptr(), chunk_capacity=per)
            count, index_total = (int(x) & 0xFFFFFFFF for x in scal.cpu().tolist())
            assert count == want["draw_count"] and index_total == want["draw_index_total"], (w, count)
            assert merged[:count].cpu().numpy().tobytes() == want["draw_cmds"].tobytes(), f"wire={w}"
            assert bool((merged[count:] == -1).all()), "nothing is written past the merged list"
        if max(counts) > 256:
            # a tightened slice: whole blocks only; the largest shard no longer fits -> cut there and reported
            cap = (max(counts) - 1) // 256 * 256
            tight = chunk_stride_bytes(cap, wire=True)
            small = torch.zeros(world * tight // 4, dtype=torch.int32, device=dev)
            for r in range(world):
                small[r * tight // 4:(r + 1) * tight // 4] = recv[True][r * strides[True] // 4: r * strides[True] // 4 + tight // 4]
            torch.cuda.synchronize()
            with pytest.raises(ra.MipError) as e:
                p.merge_wire_lists(small.data_ptr(), world, tight, merged.data_ptr(), scal.data_ptr(), chunk_capacity=cap)
            assert e.value.code == -4
            assert int(scal[0].item()) == sum(min(c, cap) for c in counts)
            tightp = chunk_stride_bytes(cap, wire="packed")
            smallp = torch.zeros(world * tightp // 4, dtype=torch.int32, device=dev)
            for r in range(world):
                smallp[r * tightp // 4:(r + 1) * tightp // 4] = recv["packed"][r * strides["packed"] // 4: r * strides["packed"] // 4 + tightp // 4]
            torch.cuda.synchronize()
            with pytest.raises(ra.MipError) as e:
                p.merge_wire_lists(smallp.data_ptr(), world, tightp, merged.data_ptr(), scal.data_ptr(), chunk_capacity=cap, packed=True)
            assert e.value.code == -4
            assert int(scal[0].item()) == sum(min(c, cap) for c in counts)
        # a record that names a mesh outside the table is never followed: reported, expanded as mesh 0
        if counts[0] > 0:
            bad = recv[True].clone()
            bad[SHARD_HEADER_BYTES // 4 + 4 + 1] = 64  # first record's second word: mesh 64 of a 64-entry table
            torch.cuda.synchronize()
            with pytest.raises(ra.MipError) as e:
                p.merge_wire_lists(bad.data_ptr(), world, strides[True], merged.data_ptr(), scal.data_ptr(), chunk_capacity=per)
            assert e.value.code == -5 and "mesh" in str(e.value)
            # the packed form: an index-bit count that cannot be (block header word 2) is reported, never used as a shift
            badp = recv["packed"].clone()
            badp[SHARD_HEADER_BYTES // 4 + 2] = 40
            torch.cuda.synchronize()
            with pytest.raises(ra.MipError) as e:
                p.merge_wire_lists(badp.data_ptr(), world, strides["packed"], merged.data_ptr(), scal.data_ptr(), chunk_capacity=per, packed=True)
            assert e.value.code == -5
        with pytest.raises(ra.MipError):  # a stride that cannot hold the capacity
            p.merge_wire_lists(recv[True].data_ptr(), world, 256, merged.data_ptr(), scal.data_ptr(), chunk_capacity=per)


_SPIN_RANK = r'''
import os, sys
root, rank, world, n_global, id_path, out_path, frames = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6], int(sys.argv[7])
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
want = oracle.run(full["pos"], full["rot"], full["scale"], full["mesh_id"], full["meshes"], full["planes"], full["cam_pos"], threads=4, want=("draw_cmds",))
per = (n_global + world - 1) // world
timeouts = 0
with renderer_amd.InstancePipeline(max_instances=hi - lo, max_meshes=64) as p:
    p.set_mesh_table(s["meshes"]); p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
    p.comm_init(uid, rank, world)
    merged = torch.zeros((world * per, 5), dtype=torch.int32, device=dev)
    count = torch.zeros(2, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    frame = make_frame(full["planes"], full["cam_pos"], first_instance_base=lo)
    def check(what):
        total, index_total = (int(x) & 0xFFFFFFFF for x in count.cpu().tolist())
        assert total == want["draw_count"] and index_total == want["draw_index_total"], (what, total, want["draw_count"])
        assert merged[:total].cpu().numpy().tobytes() == want["draw_cmds"].tobytes(), what
    # `frames` asynchronous frames back to back: while this rank's collective kernel of frame k spin-waits on the
    # device for its peers, the peers' shard kernels of frame k (and this rank's of k + 1, queued behind) need the
    # same GPU. No frame may fail: a shard kernel whose predecessors' tiles cannot start (the peers' spinning
    # workgroups hold the compute units) computes what it needs itself (MipTimings.prefix_helps says whether it had to).
    for k in range(frames):
        p.run_sharded(frame, merged.data_ptr(), count.data_ptr(), async_=True)
        if k % 8 == 7:
            p.wait()
            check(f"frame {k}")
    p.wait()
    for k in range(2):
        p.run_sharded(frame, merged.data_ptr(), count.data_ptr())
        check(f"synchronous frame {k}")
    timeouts = p.timings()["prefix_helps"]
    sent = p.timings()["sharded_bytes_sent"]
    p.comm_destroy()
open(out_path, "w").write(f"ok helps={timeouts} sent={sent}")
'''


def _build_double(tmp_path, name):
    so = str(tmp_path / f"lib{name}.so")
    src = os.path.join(ROOT, "tests", "fake_ccl", name + (".hip" if name == "spin_rccl" else ".cpp"))
    if name == "spin_rccl":
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", so, "-lrt"])
    else:
        subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-o", so,
                               "-L/opt/rocm/lib", "-lamdhip64", "-lrt"])
    return so


@pytest.mark.parametrize("world,workgroups", [(2, 64), (3, 512)])
def test_sharded_frames_beside_a_collective_that_spin_waits_on_the_device(ra, tmp_path, world, workgroups):
    """The co-tenant shape of DESIGN.md section 4: the collective is a KERNEL of persistent workgroups that spin-wait on
    the device for the peers (tests/fake_ccl/spin_rccl.hip), enqueued asynchronously like RCCL's, and the ranks —
    one process each — share this box's one GPU, so shard kernels of one rank run beside spinning collective
    workgroups of another. Every frame of every rank must equal the unsharded oracle's list — no error, no hang: a
    tile whose predecessor cannot start computes that predecessor's aggregate itself (the note printed below carries
    each rank's MipTimings.prefix_helps)."""
    fake = _build_double(tmp_path, "spin_rccl")
    env = dict(os.environ, MIP_COMM_LIBRARY=fake, SPIN_CCL_WORKGROUPS=str(workgroups))
    id_path = str(tmp_path / "uid")
    n_global = 600_001
    procs = [subprocess.Popen([sys.executable, "-c", _SPIN_RANK, ROOT, str(r), str(world), str(n_global), id_path, str(tmp_path / f"ok{r}"), "48"],
                              env=env, stderr=subprocess.PIPE, text=True) for r in range(world)]
    errs = [p.communicate(timeout=420)[1] for p in procs]
    notes = []
    for r, p in enumerate(procs):
        assert p.returncode == 0 and os.path.exists(tmp_path / f"ok{r}"), f"rank {r}:\n{errs[r][-3000:]}"
        notes.append(open(tmp_path / f"ok{r}").read())
    print("spin ccl:", notes)
    # the chunk a rank sent is the wire form: 8.06 B per command (+ header), 40 % of the 20-byte form
    per = (n_global + world - 1) // world
    sent = int(notes[0].split("sent=")[1])
    # the default form: packed 4-byte records (these shards fit beside the 64 mesh ids) — a fifth of the 20-byte commands
    assert sent == (32 + (per + 63) // 64 * 272 + 255) // 256 * 256 < 0.22 * (32 + per * 20)


def test_pipelined_exchange_repairs_an_overflow_under_its_own_stream(ra, oracle_mod):
    """the repair of an overflowed tightened chunk (all-gather + merge again at full capacity) must run on
    the slot's stream — with two frames in flight the current stream is not it. Camera A, tighten, then camera B
    (sees more): every slot's merged list must be B's, complete."""
    import torch
    import torch.distributed as dist

    from renderer_amd.sharded import PipelinedExchange, make_shard_frame

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        s = ra.scene.make_scene(3, n=700_000)
        cam_b = np.array([0.0, 1.0, -40.0], np.float32)
        planes_b = oracle_mod.project_camera(cam_b, (0.0, 0.0, 0.0, 1.0))
        want_a = run_oracle(oracle_mod, s, threads=8, want=("draw_cmds",))
        want_b = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], planes_b, cam_b, threads=8, want=("draw_cmds",))
        assert want_b["draw_count"] > 1.2 * want_a["draw_count"]

        def make_pipe(stream_handle):
            q = ra.InstancePipeline(max_instances=s["n"], max_meshes=64, stream=stream_handle)
            q.set_mesh_table(s["meshes"])
            q.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
            return q

        for wire in (True, 1, False):  # packed records (they fit), 8-byte records, 20-byte commands
            px = PipelinedExchange(make_pipe, s["n"], 1, 0, dev, frames=2, wire=wire)
            assert px.exchanges[0].form == (2 if wire is True else int(bool(wire)))
            frame_a = make_shard_frame(s["planes"], s["cam_pos"], s["n"], 1, 0)
            frame_b = make_shard_frame(planes_b, cam_b, s["n"], 1, 0)
            torch.cuda.synchronize()
            for _ in range(4):
                px.step(frame_a, [None, None])
            assert px.wait() == [False, False]
            caps = px.tighten()
            assert max(caps) < s["n"]
            for f in (frame_a, frame_b, frame_b, frame_a, frame_b, frame_b):  # overflows are repaired inside step() and wait()
                px.step(f, [None, None])
            px.wait()
            assert sum(ex.retries for ex in px.exchanges) >= 1
            for k in range(2):
                cmds, total, index_total = px.merged_draw_list(k)
                assert total == want_b["draw_count"] and index_total == want_b["draw_index_total"], (wire, total)
                assert cmds.tobytes() == want_b["draw_cmds"].tobytes(), f"wire={wire}"
            # with a frame in flight, complete() refuses to run (a possible repair) on a foreign stream instead of racing
            px.step(frame_a, [None, None])
            with pytest.raises(ValueError):
                px.exchanges[px.last].complete()
            px.wait()
            px.close()
    finally:
        dist.destroy_process_group()


def test_device_upload_rejects_mesh_ids_outside_the_table(ra):
    """mip_set_instances_device used to trust mesh ids (an id >= m became an out-of-bounds gather in the frame kernel);
    the upload-time census now counts them and the call fails like mip_set_instances does."""
    import torch

    dev = torch.device("cuda", 0)
    s = ra.scene.make_scene(3, n=10_000)
    cols = [torch.from_numpy(np.ascontiguousarray(s[k])).to(dev) for k in ("pos", "rot", "scale")]
    good = torch.from_numpy(s["mesh_id"].astype(np.int32)).to(dev)
    bad = good.clone()
    bad[9_999] = 64
    torch.cuda.synchronize()
    with ra.InstancePipeline(max_instances=s["n"], max_meshes=64) as p:
        p.set_mesh_table(s["meshes"])
        with pytest.raises(ra.MipError) as e:
            p.set_instances_device(cols[0].data_ptr(), cols[1].data_ptr(), cols[2].data_ptr(), bad.data_ptr(), s["n"])
        assert e.value.code == -1 and "mesh id" in str(e.value)
        with pytest.raises(ra.MipError) as e:  # nothing became resident
            p.run_host(s["planes"], s["cam_pos"])
        assert e.value.code == -6
        p.set_instances_device(cols[0].data_ptr(), cols[1].data_ptr(), cols[2].data_ptr(), good.data_ptr(), s["n"])
        got = p.run_host(s["planes"], s["cam_pos"], want=("draw_cmds",))
        assert got["draw_count"] > 0
        # a smaller table afterwards (a new scene: table first, instances next): ids that fall outside it make the
        # old instances non-resident instead of letting a frame gather outside the table
        p.set_mesh_table(s["meshes"][:10])
        with pytest.raises(ra.MipError) as e:
            p.run_host(s["planes"], s["cam_pos"])
        assert e.value.code == -6
        p.set_mesh_table(s["meshes"])
        p.set_instances_device(cols[0].data_ptr(), cols[1].data_ptr(), cols[2].data_ptr(), good.data_ptr(), s["n"])
        assert p.run_host(s["planes"], s["cam_pos"], want=("draw_cmds",))["draw_count"] == got["draw_count"]


# ---- row f-2, the semaphore half -------------------------------------------------------------------------------
# A Vulkan timeline semaphore exported with vkGetSemaphoreFdKHR(OPAQUE_FD) is, on amdgpu, the fd of a DRM sync object.
# There is no Vulkan device on the build boxes, but the kernel object itself can be made through the render node's
# DRM_IOCTL_SYNCOBJ_* ioctls — standing in for the renderer's export, as the dma-buf of another process does for memory.

def _drm_iowr(nr, size):
    return (3 << 30) | (size << 16) | (ord("d") << 8) | nr


class _SyncObj:
    """A DRM timeline sync object on the first render node that allows it, exportable as an fd."""

    def __init__(self):
        import fcntl
        import glob
        import struct

        self.fcntl, self.struct = fcntl, struct
        self.fd, self.handle, self.why = -1, 0, "no /dev/dri/renderD* node"
        for node in sorted(glob.glob("/dev/dri/renderD*")):
            try:
                fd = os.open(node, os.O_RDWR | os.O_CLOEXEC)
            except OSError as e:
                self.why = f"{node}: {e}"
                continue
            try:
                buf = bytearray(struct.pack("II", 0, 0))
                fcntl.ioctl(fd, _drm_iowr(0xBF, 8), buf)  # DRM_IOCTL_SYNCOBJ_CREATE
                self.fd, self.handle = fd, struct.unpack("II", buf)[0]
                return
            except OSError as e:
                self.why = f"{node}: SYNCOBJ_CREATE: {e}"
                os.close(fd)

    def export_fd(self):
        buf = bytearray(self.struct.pack("IIiI", self.handle, 0, -1, 0))
        self.fcntl.ioctl(self.fd, _drm_iowr(0xC1, 16), buf)  # DRM_IOCTL_SYNCOBJ_HANDLE_TO_FD
        return self.struct.unpack("IIiI", buf)[2]

    def query(self):
        import ctypes as C

        h, p = (C.c_uint32 * 1)(self.handle), (C.c_uint64 * 1)(0)
        buf = bytearray(self.struct.pack("QQII", C.addressof(h), C.addressof(p), 1, 0))
        self.fcntl.ioctl(self.fd, _drm_iowr(0xCB, 24), buf)  # DRM_IOCTL_SYNCOBJ_QUERY
        return int(p[0])

    def signal(self, value):
        import ctypes as C

        h, p = (C.c_uint32 * 1)(self.handle), (C.c_uint64 * 1)(value)
        buf = bytearray(self.struct.pack("QQII", C.addressof(h), C.addressof(p), 1, 0))
        self.fcntl.ioctl(self.fd, _drm_iowr(0xCD, 24), buf)  # DRM_IOCTL_SYNCOBJ_TIMELINE_SIGNAL

    def wait_binary(self, seconds):
        """True once the (binary) sync object has a signalled fence."""
        import ctypes as C
        import time

        h = (C.c_uint32 * 1)(self.handle)
        deadline = int((time.monotonic() + seconds) * 1e9)
        buf = bytearray(self.struct.pack("QqIIII", C.addressof(h), deadline, 1, 2, 0, 0))  # flags 2 = WAIT_FOR_SUBMIT
        try:
            self.fcntl.ioctl(self.fd, _drm_iowr(0xC3, 32), buf)  # DRM_IOCTL_SYNCOBJ_WAIT
            return True
        except OSError:
            return False

    def close(self):
        if self.fd >= 0:
            os.close(self.fd)


def test_external_semaphore_entry_points(ra, oracle_mod):
    """mip_import_external_semaphore_fd / mip_wait_external / mip_signal_external / mip_release_external_semaphore: the
    error paths always; and, when the runtime accepts the handle, the whole hand-over against a DRM timeline sync
    object (what vkGetSemaphoreFdKHR exports on amdgpu): a frame that waits for value 1, runs, and signals value 2."""
    import time

    import torch

    from renderer_amd.pipeline import make_frame

    dev = torch.device("cuda", 0)
    s = ra.scene.make_scene(3, n=50_000)
    want = run_oracle(oracle_mod, s, threads=8, want=("draw_cmds",))
    notes = []
    with ra.InstancePipeline(max_instances=s["n"], max_meshes=64) as p, ra.InstancePipeline(max_instances=1, max_meshes=1) as other:
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        # -- error paths --
        with pytest.raises(ra.MipError) as e:
            p.import_external_semaphore_fd(-1)
        assert e.value.code == -1
        with pytest.raises(ra.MipError) as e:
            p.import_external_semaphore_fd(0, timeline=7)  # not a kind
        assert e.value.code == -1
        r, w = os.pipe()
        try:
            with pytest.raises(ra.MipError) as e:  # an fd that is no semaphore: refused by the runtime, reported, nothing leaks
                p.import_external_semaphore_fd(r)
            assert e.value.code == -5 and "hipImportExternalSemaphore" in str(e.value)
            notes.append(f"pipe fd as TimelineSemaphoreFd -> {e.value}")
        finally:
            os.close(r)
            os.close(w)
        bogus = 0x1234
        for call in (lambda: p.wait_external(bogus, 1), lambda: p.signal_external(bogus, 1), lambda: p.release_external_semaphore(bogus)):
            with pytest.raises(ra.MipError) as e:
                call()
            assert e.value.code == -1
        # -- the hand-over against a real kernel sync object --
        for timeline in (True, False):
            so = _SyncObj()  # a fresh kernel object per kind
            if so.fd < 0:
                notes.append(f"no DRM sync object available ({so.why}): hand-over not exercised")
                continue
            try:
                if True:
                    fd = so.export_fd()
                    try:
                        sem = p.import_external_semaphore_fd(fd, timeline=timeline)
                    except ra.MipError as e:
                        assert e.code == -5
                        notes.append(f"DRM syncobj fd as {'Timeline' if timeline else 'Opaque'}Fd -> {e}")
                        os.close(fd)
                        continue
                    notes.append(f"DRM syncobj fd as {'Timeline' if timeline else 'Opaque'}Fd -> imported, "
                                 f"{'device-side (HIP runtime)' if p.external_semaphore_on_device(sem) else 'host functions on the stream (DRM sync object)'}")
                    with pytest.raises(ra.MipError):  # a handle belongs to the context that imported it
                        other.release_external_semaphore(sem)
                    if timeline:
                        cmds = torch.zeros((s["n"], 5), dtype=torch.int32, device=dev)
                        scal = torch.zeros(8, dtype=torch.int32, device=dev)
                        torch.cuda.synchronize()
                        frame = make_frame(s["planes"], s["cam_pos"])
                        p.wait_external(sem, 1)                      # the frame may not start before the "renderer" reaches 1
                        p.run_device(frame, draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(), async_=True)
                        p.signal_external(sem, 2)                    # reached when the frame's kernel has finished
                        time.sleep(0.2)
                        assert so.query() < 2, "the frame ran before the semaphore it waits for was signalled"
                        so.signal(1)                                 # the renderer's submit completes
                        t0 = time.time()
                        while so.query() < 2 and time.time() - t0 < 60:
                            time.sleep(0.001)
                        assert so.query() == 2, "the signal behind the frame never arrived"
                        # value 2 means the frame's outputs are complete — read without any mip_wait
                        count = int(scal[0].item())
                        assert count == want["draw_count"] and cmds[:count].cpu().numpy().tobytes() == want["draw_cmds"].tobytes()
                        p.wait()
                        notes.append("timeline hand-over: wait(1) -> frame -> signal(2) observed through the DRM sync object")
                    else:
                        cmds = torch.zeros((s["n"], 5), dtype=torch.int32, device=dev)
                        scal = torch.zeros(8, dtype=torch.int32, device=dev)
                        torch.cuda.synchronize()
                        assert not so.wait_binary(0.05), "a fresh binary semaphore is unsignalled"
                        p.run_device(make_frame(s["planes"], s["cam_pos"]), draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(), async_=True)
                        p.signal_external(sem)
                        assert so.wait_binary(10.0), "the binary signal behind the frame never arrived"
                        assert int(scal[0].item()) == want["draw_count"]
                        p.wait()
                        notes.append("binary hand-over: frame -> signal observed through the DRM sync object")
                    p.release_external_semaphore(sem)
                    with pytest.raises(ra.MipError):
                        p.release_external_semaphore(sem)            # released once
            finally:
                so.close()
    print("external semaphores:", notes)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        open(os.path.join(out, "external_semaphore_notes.txt"), "w").write("\n".join(notes) + "\n")


def test_external_signals_of_frames_in_flight_keep_their_order(ra, oracle_mod, monkeypatch):
    """Two frames in flight on their own streams, each followed by a timeline signal (1 behind the first, 2 behind the second);
    the FIRST frame is held back by a wait on a gate semaphore, so the second finishes long before it. The timeline must not
    move until the first frame has run: every signal request polls its own word (round 4 shared one word per semaphore — the
    second frame's sequence number released the first frame's signal early)."""
    import time

    import torch

    from renderer_amd.pipeline import make_frame

    monkeypatch.setenv("MIP_TUNE_SEMAPHORE_VIA_DRM", "1")
    dev = torch.device("cuda", 0)
    big = ra.scene.make_scene(3, n=400_000)
    want_big = run_oracle(oracle_mod, big, threads=8, want=("draw_cmds",))
    gate, out = _SyncObj(), _SyncObj()
    if gate.fd < 0 or out.fd < 0:
        pytest.skip(f"no DRM sync object available ({gate.why})")
    try:
        with ra.InstancePipeline(max_instances=big["n"], max_meshes=64, frames_in_flight=2) as p:
            p.set_mesh_table(big["meshes"])
            p.set_instances(big["pos"], big["rot"], big["scale"], big["mesh_id"])
            sem_gate = p.import_external_semaphore_fd(gate.export_fd(), timeline=True)
            sem_out = p.import_external_semaphore_fd(out.export_fd(), timeline=True)
            cmds = [torch.zeros((big["n"], 5), dtype=torch.int32, device=dev) for _ in range(2)]
            scal = [torch.zeros(8, dtype=torch.int32, device=dev) for _ in range(2)]
            torch.cuda.synchronize()
            p.wait_external(sem_gate, 1)                                   # slot 0: held back
            p.run_device(make_frame(big["planes"], big["cam_pos"]), draw_cmds=cmds[0].data_ptr(), draw_count=scal[0].data_ptr(), async_=True)
            p.signal_external(sem_out, 1)
            p.run_device(make_frame(big["planes"], big["cam_pos"]), draw_cmds=cmds[1].data_ptr(), draw_count=scal[1].data_ptr(), async_=True)
            p.signal_external(sem_out, 2)                                  # slot 1: free to run, finishes at once
            time.sleep(0.5)
            assert out.query() == 0, f"the timeline reached {out.query()} although the first frame has not started"
            assert int(scal[0][0].item()) == 0
            gate.signal(1)
            t0 = time.time()
            while out.query() < 2 and time.time() - t0 < 60:
                time.sleep(0.001)
            assert out.query() == 2, "the signals behind the two frames never arrived"
            for k in range(2):
                count = int(scal[k][0].item())
                assert count == want_big["draw_count"] and cmds[k][:count].cpu().numpy().tobytes() == want_big["draw_cmds"].tobytes(), k
            p.wait()
            # more signals than the ring has words, back to back: every one is performed, in order
            for v in range(3, 3 + 80):
                p.run_device(make_frame(big["planes"], big["cam_pos"]), draw_cmds=cmds[v & 1].data_ptr(), draw_count=scal[v & 1].data_ptr(), async_=True)
                p.signal_external(sem_out, v)
            p.wait()
            assert out.query() == 82
            p.release_external_semaphore(sem_gate)
            p.release_external_semaphore(sem_out)
    finally:
        gate.close()
        out.close()


def test_bases_non_finite_and_wire_forms_across_launch_sizes(ra, oracle_mod):
    """(Round 3 ran this sweep against the three wait-free launches of the ordered-tiles mode, which is gone: every launch is
    order-independent now.) One frame at sizes either side of every switch the host takes — one tile, a partial tile, the
    commands-first / stores-first crossover, more tiles than are resident — with non-finite instances (the general kernel),
    instance and index bases, host outputs, both wire forms and plain device outputs: all equal to the oracle's bytes."""
    import torch

    from cpu_pipeline import decode_wire, unpack_wire
    from helpers import assert_parity
    from renderer_amd.pipeline import make_frame, wire_body_bytes

    dev = torch.device("cuda", 0)
    for n in (1, 255, 4_097, 40_961, 131_073, 1_200_003):
        s = ra.scene.make_scene(3, n=n)
        if n > 300:
            s["pos"][17] = np.nan
            s["scale"][200] = np.inf
        want = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"], threads=8,
                              first_instance_base=11, first_index_base=5)
        with ra.InstancePipeline(max_instances=n, max_meshes=64) as p:
            p.set_mesh_table(s["meshes"])
            p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
            for rep in range(2):
                got = p.run_host(s["planes"], s["cam_pos"], first_instance_base=11, first_index_base=5)
                assert_parity(got, want, f"n={n} rep={rep}")
            body = torch.zeros(wire_body_bytes(n) // 4, dtype=torch.int32, device=dev)
            scal = torch.zeros(8, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            p.run_device(make_frame(s["planes"], s["cam_pos"], first_instance_base=11, first_index_base=5), draw_cmds=body.data_ptr(),
                         draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4, wire=True)
            count, total = (int(x) & 0xFFFFFFFF for x in scal[:2].cpu().tolist())
            assert count == want["draw_count"] and total == want["draw_index_total"]
            assert decode_wire(body.cpu().numpy().view(np.uint32), count, s["meshes"]).tobytes() == want["draw_cmds"].tobytes(), n
            body.zero_()
            scal.zero_()
            torch.cuda.synchronize()
            p.run_device(make_frame(s["planes"], s["cam_pos"], first_instance_base=11, first_index_base=5), draw_cmds=body.data_ptr(),
                         draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4, wire="packed")
            assert [int(x) & 0xFFFFFFFF for x in scal[:2].cpu().tolist()] == [count, total]
            assert decode_wire(unpack_wire(body.cpu().numpy().view(np.uint32), count), count, s["meshes"]).tobytes() == want["draw_cmds"].tobytes(), n
            cmds = torch.zeros((n, 5), dtype=torch.int32, device=dev)
            scal.zero_()
            torch.cuda.synchronize()
            p.run_device(make_frame(s["planes"], s["cam_pos"], first_instance_base=11, first_index_base=5), draw_cmds=cmds.data_ptr(),
                         draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4)
            assert [int(x) & 0xFFFFFFFF for x in scal[:2].cpu().tolist()] == [count, total]
            assert cmds[:count].cpu().numpy().tobytes() == want["draw_cmds"].tobytes(), n
            report_timing_property(f"wire forms at {n}: prefix_helps", p.timings()["prefix_helps"], "0 on an idle GPU", p.timings()["prefix_helps"] == 0)
            # a draw list that is only 4-byte aligned is fine for 20-byte commands and refused for the wire forms
            with pytest.raises(ra.MipError):
                p.run_device(make_frame(s["planes"], s["cam_pos"]), draw_cmds=body.data_ptr() + 4, draw_count=scal.data_ptr(), wire=True)


def test_kernel_wire_bytes_equal_the_committed_fixture(ra):
    """The frame kernel's MIP_OUT_WIRE output for the committed golden inputs is the committed wire fixture, word for word
    (live record slots and block headers), and the merge kernel expands it to the golden command bytes."""
    import torch

    from renderer_amd.pipeline import SHARD_HEADER_BYTES, make_frame
    from renderer_amd.sharded import chunk_stride_bytes

    here = os.path.dirname(os.path.abspath(__file__))
    w = np.load(os.path.join(here, "golden", "ext", "wire_4097_bases.npz"))
    g = np.load(os.path.join(here, "golden", str(w["source"]) + ".npz"))
    n = len(g["pos"])
    dev = torch.device("cuda", 0)
    stride = chunk_stride_bytes(n, wire=True)
    chunk = torch.zeros(stride // 4, dtype=torch.int32, device=dev)
    merged = torch.zeros((n, 5), dtype=torch.int32, device=dev)
    scal = torch.zeros(2, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    with ra.InstancePipeline(max_instances=n, max_meshes=len(g["meshes"])) as p:
        p.set_mesh_table(g["meshes"])
        p.set_instances(g["pos"], g["rot"], g["scale"], g["mesh_id"])
        frame = make_frame(g["planes"], g["cam_pos"], first_instance_base=int(g["first_instance_base"]), first_index_base=int(g["first_index_base"]))
        base = chunk.data_ptr()
        p.run_device(frame, draw_cmds=base + SHARD_HEADER_BYTES, draw_count=base, draw_index_total=base + 4, wire=True)
        host = chunk.cpu().numpy().view(np.uint32)
        count = int(host[0])
        assert count == int(w["draw_count"]) and int(host[1]) == int(w["draw_index_total"])
        from cpu_pipeline import wire_live_mask
        got = host[SHARD_HEADER_BYTES // 4: SHARD_HEADER_BYTES // 4 + w["body"].size]
        live = wire_live_mask(count)
        assert np.array_equal(got[live], w["body"][live])
        p.merge_wire_lists(base, 1, stride, merged.data_ptr(), scal.data_ptr(), chunk_capacity=n)
        assert merged[:count].cpu().numpy().tobytes() == g["draw_cmds"].tobytes()
        # the packed form (MIP_OUT_WIRE_PACKED) against its committed bytes
        stride_p = chunk_stride_bytes(n, wire="packed")
        chunk.zero_()
        merged.zero_()
        torch.cuda.synchronize()
        p.run_device(frame, draw_cmds=base + SHARD_HEADER_BYTES, draw_count=base, draw_index_total=base + 4, wire="packed")
        host = chunk.cpu().numpy().view(np.uint32)
        assert int(host[0]) == count
        got = host[SHARD_HEADER_BYTES // 4: SHARD_HEADER_BYTES // 4 + w["body_packed"].size]
        livep = wire_live_mask(count, packed=True)
        assert np.array_equal(got[livep], w["body_packed"][livep])
        p.merge_wire_lists(base, 1, stride_p, merged.data_ptr(), scal.data_ptr(), chunk_capacity=n, packed=True)
        assert merged[:count].cpu().numpy().tobytes() == g["draw_cmds"].tobytes()
