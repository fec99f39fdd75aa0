"""Randomised parity: oracle vs the independent numpy restatement on CPU, HIP path vs oracle on the GPU."""
import numpy as np
import pytest

import numpy_restatement as npr
from fuzz_scenes import random_scene
from helpers import assert_parity, same_floats


@pytest.mark.parametrize("seed", range(12))
def test_oracle_vs_numpy_restatement_random_scenes(oracle_mod, seed):
    rng = np.random.default_rng(1000 + seed)
    s = random_scene(rng, oracle_mod, n_max=3000)
    a = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"],
                       first_instance_base=s["first_instance_base"], first_index_base=s["first_index_base"])
    b = npr.run(s, first_instance_base=s["first_instance_base"], first_index_base=s["first_index_base"])
    assert same_floats(a["model"], b["model"]) and same_floats(a["world_aabb"], b["world_aabb"])
    assert np.array_equal(a["coarse_culled"].astype(bool), b["coarse_culled"])
    c = a["draw_cmds"]
    assert np.array_equal(c["indexCount"], b["cmds"]["indexCount"]) and np.array_equal(c["firstIndex"], b["cmds"]["firstIndex"])
    assert np.array_equal(c["firstInstance"], b["cmds"]["firstInstance"]) and np.array_equal(c["vertexOffset"], b["cmds"]["vertexOffset"])


def _check_wire_form(ra, p, s, want, what):
    """The same frame emitted in the wire form (MIP_OUT_WIRE) and expanded against the mesh table (the numpy statement of what
    mip_merge_wire_lists does, tests/cpu_pipeline.py) must give the oracle's command bytes — special values, empty LODs, random
    bases and all — and the merge kernel itself must agree (one chunk)."""
    import torch

    from cpu_pipeline import decode_wire, unpack_wire
    from renderer_amd.pipeline import SHARD_HEADER_BYTES, make_frame
    from renderer_amd.sharded import chunk_stride_bytes

    n = s["n"]
    dev = torch.device("cuda", 0)
    frame = make_frame(s["planes"], s["cam_pos"], first_instance_base=s["first_instance_base"], first_index_base=s["first_index_base"])
    for form in (True, "packed"):  # 8-byte records, packed 4-byte records
        stride = chunk_stride_bytes(max(n, 1), wire=form)
        chunk = torch.zeros(stride // 4, dtype=torch.int32, device=dev)
        merged = torch.full((max(n, 1), 5), -1, dtype=torch.int32, device=dev)
        scal = torch.zeros(2, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        base = chunk.data_ptr()
        p.run_device(frame, draw_cmds=base + SHARD_HEADER_BYTES, draw_count=base, draw_index_total=base + 4, wire=form)
        host = chunk.cpu().numpy().view(np.uint32)
        count = int(host[0])
        assert count == want["draw_count"] and int(host[1]) == want["draw_index_total"], (what, form)
        body = host[SHARD_HEADER_BYTES // 4:]
        if form == "packed":
            body = unpack_wire(body, count)
        assert decode_wire(body, count, s["meshes"]).tobytes() == want["draw_cmds"].tobytes(), (what, form)
        p.merge_wire_lists(base, 1, stride, merged.data_ptr(), scal.data_ptr(), chunk_capacity=max(n, 1), packed=form == "packed")
        assert int(scal[0].item()) == count and merged[:count].cpu().numpy().tobytes() == want["draw_cmds"].tobytes(), (what, form)


@pytest.mark.gpu
def test_gpu_vs_oracle_random_scenes(oracle_mod):
    import renderer_amd as ra

    rng = np.random.default_rng(77)
    with ra.InstancePipeline(max_instances=20_000, max_meshes=64) as p:
        for trial in range(60):
            s = random_scene(rng, oracle_mod, n_max=20_000 if trial % 10 == 0 else 3000)
            p.set_mesh_table(s["meshes"])
            p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
            got = p.run_host(s["planes"], s["cam_pos"], first_instance_base=s["first_instance_base"],
                             first_index_base=s["first_index_base"])
            want = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"],
                                  first_instance_base=s["first_instance_base"], first_index_base=s["first_index_base"])
            assert_parity(got, want, f"trial {trial} n={s['n']}")
            _check_wire_form(ra, p, s, want, f"trial {trial} n={s['n']}")
