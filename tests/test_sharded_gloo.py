"""World-size-2 (and 3) gloo test of the N>1 path on CPU: contiguous shards, one all-gather of
the fixed-size draw-list chunks, merge with firstIndex rebasing == the unsharded oracle run."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, init_file, n_global, tighten, out_dir, wire=True):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import oracle
    from cpu_pipeline import OraclePipeline
    from renderer_amd import scene
    from renderer_amd.sharded import DrawListExchange, make_shard_frame, shard_range

    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        full = scene.make_scene(3, n=n_global)
        lo, hi = shard_range(n_global, world, rank)
        # the shard is regenerated from the stream position, exactly as a rank would do it
        shard = scene.make_scene(3, n=hi - lo, first=lo)
        assert np.array_equal(shard["pos"], full["pos"][lo:hi])
        pipe = OraclePipeline(shard)
        n_local = hi - lo
        ex = DrawListExchange(pipe, n_local, world, rank, torch.device("cpu"), dist=dist, torch=torch, wire=wire)
        assert ex.wire == bool(wire)
        # True: packed (these shards fit beside 64 mesh ids); the integer 1 asks for 8-byte records; False: 20-byte commands
        assert ex.form == (2 if wire is True else (1 if wire else 0))
        frame = make_shard_frame(full["planes"], full["cam_pos"], n_global, world, rank)
        bitmap = torch.zeros((n_local + 31) // 32 + 1, dtype=torch.int32)
        model = torch.zeros((max(n_local, 1), 16), dtype=torch.float32)
        ex.step(frame, model=model.data_ptr(), visible_bitmap=bitmap.data_ptr())
        if tighten:
            cap = ex.tighten()
            assert cap <= max(ex.n_max, 256)
            ex.step(frame, model=model.data_ptr(), visible_bitmap=bitmap.data_ptr())
        cmds, total, index_total = ex.merged_draw_list()
        want = oracle.run(full["pos"], full["rot"], full["scale"], full["mesh_id"], full["meshes"], full["planes"],
                          full["cam_pos"])
        assert total == want["draw_count"], (total, want["draw_count"])
        assert cmds.tobytes() == want["draw_cmds"].tobytes()
        assert index_total == want["draw_index_total"]
        # the sharded outputs stay sharded: this rank's matrices/bitmap are its slice of the global ones
        assert np.array_equal(model.numpy()[:n_local], want["model"][lo:hi])
        vis_local = np.unpackbits(bitmap.numpy()[: (n_local + 31) // 32].view(np.uint8), bitorder="little")[:n_local]
        vis_global = np.unpackbits(want["visible_bitmap"].view(np.uint8), bitorder="little")[lo:hi]
        assert np.array_equal(vis_local, vis_global)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_global,tighten,wire", [(2, 20_000, False, True), (2, 4_097, True, True), (3, 1_000, True, True),
                                                         (2, 1, False, True), (2, 4_097, True, 1), (3, 1_000, True, 1),
                                                         (2, 4_097, True, False), (3, 1_000, False, False)])
def test_sharded_exchange_matches_unsharded_oracle(world, n_global, tighten, wire):
    """wire=True: the lists travel as packed 4-byte records (MIP_OUT_WIRE_PACKED; the shards fit), wire=1: as 8-byte records
    (MIP_OUT_WIRE), and the merge expands them; wire=False: as 20-byte commands (round 2). The merged list is the
    unsharded oracle's every time."""
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, "init")
        mp.spawn(_worker, args=(world, init_file, n_global, tighten, d, wire), nprocs=world, join=True)
        for r in range(world):
            assert os.path.exists(os.path.join(d, f"ok{r}"))


def test_shard_ranges_cover_everything():
    from renderer_amd.sharded import chunk_stride_bytes, shard_range

    for n in (0, 1, 7, 8, 9, 1000, 10_000_000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, e) in zip(spans, spans[1:]):
                assert b == c and a <= b and c <= e
    assert chunk_stride_bytes(0) == 256 and chunk_stride_bytes(12) % 256 == 0
    # the wire form: whole blocks of 256 records (2064 B) behind the 32-byte header; ~40 % of the 20-byte form
    assert chunk_stride_bytes(0, wire=True) == 256 and chunk_stride_bytes(1, wire=True) == 2304
    assert chunk_stride_bytes(256, wire=True) == 2304 and chunk_stride_bytes(257, wire=True) == 4352
    assert chunk_stride_bytes(336_000, wire=True) / chunk_stride_bytes(336_000) < 0.41
    # the packed form: blocks of 64 records = 272 B; ~21 % of the 20-byte form
    assert chunk_stride_bytes(0, wire="packed") == 256 and chunk_stride_bytes(1, wire="packed") == 512
    assert chunk_stride_bytes(64, wire=2) == 512 and chunk_stride_bytes(257, wire=2) == 1536
    assert chunk_stride_bytes(336_000, wire="packed") / chunk_stride_bytes(336_000) < 0.22


def test_wire_form_round_trips_through_the_numpy_restatement():
    """encode_wire / decode_wire (tests/cpu_pipeline.py) are the CPU statement of the wire form the kernels speak."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import oracle
    from cpu_pipeline import WIRE_BLOCK_WORDS, WIRE_PACKED_BLOCK_WORDS, decode_wire, encode_wire, encode_wire_packed, unpack_wire, wire_live_mask
    from renderer_amd import scene
    from renderer_amd.pipeline import wire_index_bits

    assert [wire_index_bits(m) for m in (0, 1, 2, 3, 64, 65, 1 << 20, (1 << 31) - 1)] == [31, 31, 30, 29, 25, 24, 11, 0]

    for n in (0, 1, 255, 256, 257, 3000):
        s = scene.make_scene(3, n=max(n, 1))
        r = oracle.run(s["pos"][:n], s["rot"][:n], s["scale"][:n], s["mesh_id"][:n], s["meshes"], s["planes"], s["cam_pos"],
                       first_instance_base=77, first_index_base=12345)
        cmds = r["draw_cmds"]
        inst = (cmds["firstInstance"] - np.uint32(77)).astype(np.int64)
        far = np.array([oracle.pick_lod(2, s["cam_pos"], s["pos"][i]) for i in inst], np.uint32)
        body = encode_wire(cmds, s["mesh_id"][inst], far)
        assert body.size == (len(cmds) + 255) // 256 * WIRE_BLOCK_WORDS
        back = decode_wire(body, len(cmds), s["meshes"])
        assert back.tobytes() == cmds.tobytes(), n
        packed = encode_wire_packed(cmds, s["mesh_id"][inst], far, 77, len(s["meshes"]))
        assert packed.size == (len(cmds) + 63) // 64 * WIRE_PACKED_BLOCK_WORDS
        live = wire_live_mask(len(cmds))  # the unused slots of the last block are zero in one form, {base, 0} in the other
        assert np.array_equal(unpack_wire(packed, len(cmds))[live], body[live]), n
        assert decode_wire(unpack_wire(packed, len(cmds)), len(cmds), s["meshes"]).tobytes() == cmds.tobytes(), n


def _camera_b():
    """A second camera that sees more of the scene than the default one (so a chunk tightened on the
    default camera's counts overflows)."""
    import oracle

    pos = np.array([0.0, 1.0, -40.0], np.float32)
    return oracle.project_camera(pos, (0.0, 0.0, 0.0, 1.0)), pos


def _worker_overflow(rank, world, init_file, n_global, out_dir):
    """tighten() on camera A, then the camera moves: the overflowing frame must come out complete
    (one collective re-gather at full capacity), on every rank, with nothing lost."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import oracle
    from cpu_pipeline import OraclePipeline
    from renderer_amd import scene
    from renderer_amd.sharded import DrawListExchange, PipelinedExchange, make_shard_frame, shard_range

    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    try:
        full = scene.make_scene(3, n=n_global)
        lo, hi = shard_range(n_global, world, rank)
        shard = scene.make_scene(3, n=hi - lo, first=lo)
        n_local = hi - lo
        planes_b, cam_b = _camera_b()
        frame_a = make_shard_frame(full["planes"], full["cam_pos"], n_global, world, rank)
        frame_b = make_shard_frame(planes_b, cam_b, n_global, world, rank)
        want_a = oracle.run(full["pos"], full["rot"], full["scale"], full["mesh_id"], full["meshes"], full["planes"], full["cam_pos"])
        want_b = oracle.run(full["pos"], full["rot"], full["scale"], full["mesh_id"], full["meshes"], planes_b, cam_b)
        assert want_b["draw_count"] > 1.2 * want_a["draw_count"]

        def check(ex, want, what):
            cmds, total, index_total = ex.merged_draw_list()
            assert total == want["draw_count"] and index_total == want["draw_index_total"], (what, total, want["draw_count"])
            assert cmds.tobytes() == want["draw_cmds"].tobytes(), what

        ex = DrawListExchange(OraclePipeline(shard), n_local, world, rank, torch.device("cpu"), dist=dist, torch=torch)
        ex.step(frame_a)
        check(ex, want_a, "A, full capacity")
        cap = ex.tighten()
        assert cap < n_local
        ex.step(frame_a)
        check(ex, want_a, "A, tightened")
        assert ex.retries == 0
        ex.step(frame_b)                      # the camera moved between tighten() and this frame
        check(ex, want_b, "B, overflow repaired")
        assert ex.retries == 1 and ex.capacity == ex.n_max
        ex.step(frame_a)
        check(ex, want_a, "A again")
        assert ex.retries == 1
        # One rank's mip_wait reports ANOTHER error for the overflowing frame (a local error outranks the overflow there):
        # that rank must still take part in the collective repair — decided from the gathered headers, which every rank
        # holds — and report its own error afterwards; nobody hangs, everybody ends with the complete list.
        from renderer_amd._lib import MipError
        ex.tighten()
        ex.step(frame_a)
        check(ex, want_a, "A, tightened again")
        if rank == world - 1:
            ex.pipe.other_error_once = -5
        ex.step(frame_b)
        if rank == world - 1:
            with pytest.raises(MipError) as e:
                ex.complete()
            assert e.value.code == -5
        else:
            assert ex.complete() is True
        assert ex.retries == 2 and ex.capacity == ex.n_max
        check(ex, want_b, "B, repaired by every rank although one had a different error")

        # the same with two frames in flight (two exchanges issued round-robin)
        px = PipelinedExchange(lambda stream: OraclePipeline(shard), n_local, world, rank, torch.device("cpu"),
                               frames=2, dist=dist, torch=torch)
        for f in (frame_a, frame_a):
            px.step(f, [None, None])
        assert px.wait() == [False, False]
        px.tighten()
        for k, f in enumerate((frame_a, frame_b, frame_b, frame_a)):
            slot = px.step(f, [None, None])
            assert slot == k % 2
        repaired = px.wait()                  # slot 0 last ran B, slot 1 last ran A (after B was completed inside step)
        check(px.exchanges[0], want_b, "pipelined slot 0")
        check(px.exchanges[1], want_a, "pipelined slot 1")
        assert sum(e.retries for e in px.exchanges) == 2, [e.retries for e in px.exchanges]
        assert repaired == [True, False]
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_global", [(2, 6_000), (3, 5_001)])
def test_overflow_of_a_tightened_chunk_is_repaired_not_lost(world, n_global):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_overflow, args=(world, os.path.join(d, "init"), n_global, d), nprocs=world, join=True)
        for r in range(world):
            assert os.path.exists(os.path.join(d, f"ok{r}"))


def test_wire_fixture_pins_the_byte_layout():
    """tests/golden/ext/wire_4097_bases.npz: the wire form of the committed golden frame mixed_4097_bases (bases 1000 /
    0xFFFFF000, so the block headers wrap). The numpy statement of the format must still produce and expand these bytes."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import oracle
    from cpu_pipeline import decode_wire, encode_wire, encode_wire_packed, unpack_wire

    w = np.load(os.path.join(HERE, "golden", "ext", "wire_4097_bases.npz"))
    g = np.load(os.path.join(HERE, "golden", str(w["source"]) + ".npz"))
    cmds = g["draw_cmds"]
    assert int(w["draw_count"]) == len(cmds) == 1116 and w["body"].size == 5 * 516
    assert decode_wire(w["body"], len(cmds), g["meshes"]).tobytes() == cmds.tobytes()
    inst = (cmds["firstInstance"] - np.uint32(g["first_instance_base"])).astype(np.int64)
    far = np.array([oracle.pick_lod(2, g["cam_pos"], g["pos"][i]) for i in inst], np.uint32)
    assert np.array_equal(encode_wire(cmds, g["mesh_id"][inst], far), w["body"])
    # the layout itself: block b starts at word 516 b; header word q = firstIndex of the block's record 64 q; records from word 4
    assert int(w["body"][0]) == int(cmds["firstIndex"][0]) and int(w["body"][516]) == int(cmds["firstIndex"][256])
    assert [int(x) for x in w["body"][516:520]] == [int(cmds["firstIndex"][256 + 64 * q]) for q in range(4)]
    assert int(w["body"][4]) == int(cmds["firstInstance"][0]) and int(w["body"][516 + 4 + 2]) == int(cmds["firstInstance"][257])
    # the packed form of the same list: blocks of 64 records = 68 words, header {firstIndex, first_instance_base, index bits (64
    # meshes: 25), 0}, one word per record: instance index | mesh << 25 | far << 31
    pk = w["body_packed"]
    assert pk.size == 18 * 68 and np.array_equal(encode_wire_packed(cmds, g["mesh_id"][inst], far, g["first_instance_base"], len(g["meshes"])), pk)
    assert [int(x) for x in pk[4 * 68:4 * 68 + 4]] == [int(cmds["firstIndex"][256]), int(g["first_instance_base"]), 25, 0]
    assert int(pk[4 * 68 + 4 + 1]) == int(inst[257]) | (int(g["mesh_id"][inst[257]]) << 25) | (int(far[257]) << 31)
    assert decode_wire(unpack_wire(pk, len(cmds)), len(cmds), g["meshes"]).tobytes() == cmds.tobytes()
