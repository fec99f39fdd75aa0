"""Row f-3: glTF -> instance columns / mesh table / consolidated geometry (renderer_amd/host/gltf_scene.cpp),
following the reference loader's rules (scene_loader.rs:642-789)."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import gltf_fixture
from renderer_amd.pipeline import MESH_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXTRACT = os.path.join(ROOT, "renderer_amd", "lib", "mip_gltf_extract")


def _build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "renderer_amd", "csrc"), "-s"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "renderer_amd", "host"), "-s"])


def read_scene(path):
    raw = open(path, "rb").read()
    n, m = np.frombuffer(raw, np.uint32, 2)
    off = 8
    meshes = np.frombuffer(raw, MESH_DTYPE, m, off); off += m * 80
    pos = np.frombuffer(raw, np.float32, n * 3, off).reshape(n, 3); off += n * 12
    rot = np.frombuffer(raw, np.float32, n * 4, off).reshape(n, 4); off += n * 16
    scale = np.frombuffer(raw, np.float32, n, off); off += n * 4
    mesh_id = np.frombuffer(raw, np.uint32, n, off); off += n * 4
    nv, ni = np.frombuffer(raw, np.uint32, 2, off); off += 8
    vertices = np.frombuffer(raw, np.float32, nv * 3, off).reshape(nv, 3); off += nv * 12
    indices = np.frombuffer(raw, np.uint32, ni, off)
    return dict(n=int(n), meshes=meshes, pos=pos, rot=rot, scale=scale, mesh_id=mesh_id, vertices=vertices, indices=indices)


def _extract(src, copies=None):
    if not os.path.exists(EXTRACT):
        _build()
    out = src + ".bin"
    cmd = [EXTRACT, src, out] + ([str(copies)] if copies else [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    return r, out


@pytest.mark.parametrize("container", ["gltf", "glb"])
def test_extractor_follows_the_reference_loader(container):
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "scene." + container)
        exp = gltf_fixture.write_gltf(src) if container == "gltf" else gltf_fixture.write_glb(src)
        r, out = _extract(src)
        assert r.returncode == 0, r.stderr
        assert "entities=3 meshes=3 primitives=5 skipped_no_base_color=1 skipped_small=1" in r.stdout
        s = read_scene(out)
    big_pos, big_idx = exp["big"]
    med_pos, med_idx = exp["med"]
    # traversal order: scene 0 root -> child "trs" (-> grandchild, both primitives skipped) -> child "matrix"; scene 1
    assert np.array_equal(s["pos"][0], [1, 2, 3])            # local TRS only: the root's (100,100,100) is not accumulated
    assert np.allclose(s["rot"][0], [0, 0.6, 0, 0.8]) and s["scale"][0] == np.float32(1.5)   # scale[0] of a non-uniform scale
    t, q, sc = exp["matrix_trs"]
    assert np.allclose(s["pos"][1], t) and np.allclose(s["rot"][1], q, atol=1e-6) and np.isclose(s["scale"][1], sc)
    assert np.array_equal(s["pos"][2], [-5, 0, 20]) and np.array_equal(s["rot"][2], [0, 0, 0, 1]) and s["scale"][2] == 1
    assert s["mesh_id"].tolist() == [0, 1, 2]                # no de-duplication: one mesh per entity
    m = s["meshes"]
    assert np.array_equal(m["aabb_min"][0], big_pos.min(0)) and np.array_equal(m["aabb_max"][0], big_pos.max(0))
    assert np.array_equal(m["aabb_min"][1], med_pos.min(0)) and np.array_equal(m["aabb_max"][1], med_pos.max(0))
    # LOD chain: len * 0.5^x rounded down to whole triangles, kept while it shrinks
    def chain(n):
        out = [n]
        for x in range(1, 6):
            t = int(np.float32(n) * np.float32(0.5) ** np.float32(x))
            t -= t % 3
            if 0 < t < n:
                out.append(t)
        return out
    for k, idx in ((0, big_idx), (1, med_idx), (2, big_idx)):
        want = chain(len(idx))
        assert m["n_lods"][k] == len(want) and m["index_len"][k, : len(want)].tolist() == want
        off0 = int(m["index_offset"][k, 0])
        assert np.array_equal(s["indices"][off0 : off0 + len(idx)], idx)      # u16 and u32 sources
    assert m["vertex_offset"].tolist() == [0, len(big_pos), len(big_pos) + len(med_pos)]
    assert np.array_equal(s["vertices"][: len(big_pos)], big_pos)
    assert np.array_equal(s["vertices"][len(big_pos) : len(big_pos) + len(med_pos)], med_pos)  # strided view
    # offsets run on without gaps across meshes and LODs
    flat = [(int(m["index_offset"][k, l]), int(m["index_len"][k, l])) for k in range(3) for l in range(int(m["n_lods"][k]))]
    for (o0, l0), (o1, _) in zip(flat, flat[1:]):
        assert o0 + l0 == o1
    assert flat[-1][0] + flat[-1][1] == len(s["indices"])


def test_extractor_rejects_malformed_input():
    with tempfile.TemporaryDirectory() as d:
        bad = os.path.join(d, "bad.gltf")
        open(bad, "w").write('{"scenes": [{"nodes": [0]}], "nodes": [{"mesh": 3}]}')
        r, _ = _extract(bad)
        assert r.returncode == 3 and "mesh index out of range" in r.stderr
        open(bad, "w").write('{"scenes": [')
        r, _ = _extract(bad)
        assert r.returncode == 3 and "gltf json" in r.stderr
        r, _ = _extract(os.path.join(d, "missing.gltf"))
        assert r.returncode == 3


@pytest.mark.gpu
def test_gltf_scene_through_the_frame_driver_matches_oracle(oracle_mod):
    """An N-instance scene made from one glTF asset, run through the C++ schedule on the GPU."""
    from test_host_mirror import DRIVER, _read_out

    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "scene.gltf")
        gltf_fixture.write_gltf(src)
        r, scene_bin = _extract(src, copies=700)   # 2100 entities on a grid
        assert r.returncode == 0, r.stderr
        s = read_scene(scene_bin)
        out = subprocess.run([DRIVER, scene_bin, os.path.join(d, "o.bin"), "2"], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        got = _read_out(os.path.join(d, "o.bin"))
    cam = np.array([0, 1, 2], np.float32)
    want = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], got["planes"], cam)
    assert got["n"] == s["n"] == 2100 and 0 < got["count"] == want["draw_count"] < s["n"]
    assert np.array_equal(got["culled"], want["coarse_culled"]) and got["cmds"].tobytes() == want["draw_cmds"].tobytes()
