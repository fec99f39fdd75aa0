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
