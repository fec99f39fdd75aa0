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


def _target(n, x):
    return int(np.float32(n) * np.float32(0.5) ** np.float32(x))  # `(indices.len() as f32 * factor) as usize`


def _lod_chain(positions, indices):
    """What the loader keeps, from the independent numpy restatement of the simplifier (tests/simplify_sloppy_np.py)."""
    from simplify_sloppy_np import simplify_sloppy

    out = [np.asarray(indices, np.uint32)]
    for x in range(1, 6):
        res = simplify_sloppy(indices, positions, _target(len(indices), x))
        if 0 < len(res) < len(indices):
            out.append(res)
    return out


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
    # LOD chain (scene_loader.rs:739-753): simplify_sloppy(indices, positions, len * 0.5^x) for x = 1..5, a level kept
    # only if it is shorter than LOD 0 and non-empty; its length is whatever the simplifier returns, not the target
    for k, (p_, idx) in ((0, (big_pos, big_idx)), (1, (med_pos, med_idx)), (2, (big_pos, big_idx))):
        want = _lod_chain(p_, idx)
        assert m["n_lods"][k] == len(want), (k, m["n_lods"][k], [len(w) for w in want])
        for l, w in enumerate(want):
            off = int(m["index_offset"][k, l])
            assert m["index_len"][k, l] == len(w) and np.array_equal(s["indices"][off : off + len(w)], w), (k, l)   # u16 and u32 sources
    assert m["n_lods"][0] >= 3 and any(m["index_len"][0, l] < _target(len(big_idx), l) for l in range(1, int(m["n_lods"][0])))
    assert m["vertex_offset"].tolist() == [0, len(big_pos), len(big_pos) + len(med_pos)]
    assert np.array_equal(s["vertices"][: len(big_pos)], big_pos)
    assert np.array_equal(s["vertices"][len(big_pos) : len(big_pos) + len(med_pos)], med_pos)  # strided view
    # offsets run on without gaps across meshes and LODs
    flat = [(int(m["index_offset"][k, l]), int(m["index_len"][k, l])) for k in range(3) for l in range(int(m["n_lods"][k]))]
    for (o0, l0), (o1, _) in zip(flat, flat[1:]):
        assert o0 + l0 == o1
    assert flat[-1][0] + flat[-1][1] == len(s["indices"])


def test_lod_levels_are_what_the_simplifier_returns_not_the_target():
    """VERDICT r02 'missing 1': index_len[lod >= 1] feeds indexCount and the running firstIndex, so the LODs must be the
    simplifier's output (data-dependent length <= target; dropped when empty), not a fixed-size stand-in. Shapes that
    exercise the search: a dense torus (levels shorter than their targets), a sparse primitive — 120 positions, 4
    triangles — whose every level comes out empty and is DROPPED, a flat plate (one grid layer: many duplicate
    triangles to filter), two tori in one primitive (cells shared between surfaces), and a needle-thin stretched mesh
    (the grid is isotropic: most cells empty). The C++ extractor and the numpy restatement must agree index for index."""
    rng = np.random.default_rng(7)
    torus_pos, torus_idx = gltf_fixture.torus(40, 25)                                 # 1000 positions, 2000 triangles
    sparse_pos = torus_pos[:120]
    sparse_idx = np.array([0, 1, 30, 1, 31, 30, 60, 61, 90, 61, 91, 90], np.uint32)
    gx, gy = np.meshgrid(np.arange(20), np.arange(15), indexing="ij")
    plate_pos = np.stack([gx.ravel() * 0.1, np.zeros(300), gy.ravel() * 0.1], 1).astype(np.float32)
    q = (gx[:-1, :-1] * 15 + gy[:-1, :-1]).ravel()
    plate_idx = np.concatenate([np.stack([q, q + 15, q + 16], 1), np.stack([q, q + 16, q + 1], 1)]).astype(np.uint32).ravel()
    t2_pos, t2_idx = gltf_fixture.torus(16, 12, (0.5, 0.5, 0.5))
    two_pos = np.concatenate([t2_pos, t2_pos + np.float32([0.2, 0.1, 0.0])]).astype(np.float32)
    two_idx = np.concatenate([t2_idx, t2_idx + len(t2_pos)]).astype(np.uint32)
    thin_pos, thin_idx = gltf_fixture.torus(30, 8, (5.0, 0.02, 0.02))
    noisy_pos = (torus_pos + rng.normal(0, 0.003, torus_pos.shape)).astype(np.float32)
    prims = [(torus_pos, torus_idx), (sparse_pos, sparse_idx), (plate_pos, plate_idx), (two_pos, two_idx), (thin_pos, thin_idx),
             (noisy_pos, torus_idx)]
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "lods.gltf")
        gltf_fixture.write_simple(src, prims)
        r, out = _extract(src)
        assert r.returncode == 0, r.stderr
        s = read_scene(out)
    m = s["meshes"]
    assert s["n"] == len(prims)
    shorter = dropped = 0
    for k, (p_, idx) in enumerate(prims):
        want = _lod_chain(p_, idx)
        assert m["n_lods"][k] == len(want), (k, int(m["n_lods"][k]), [len(w) for w in want])
        for l, w in enumerate(want):
            off = int(m["index_offset"][k, l])
            assert m["index_len"][k, l] == len(w), (k, l, int(m["index_len"][k, l]), len(w))
            assert np.array_equal(s["indices"][off : off + len(w)], w), (k, l)
            assert len(w) % 3 == 0 and (l == 0 or len(w) <= _target(len(idx), 1))
        shorter += sum(1 for l in range(1, len(want)) if len(want[l]) < _target(len(idx), l) - 2)
        dropped += 5 - (len(want) - 1)
    assert m["n_lods"][1] <= 2                      # the sparse primitive (12 indices): targets 6, 3, 1, 0, 0 -> at most one level survives
    assert shorter >= 5 and dropped >= 5            # lengths are data-dependent, and levels do get dropped
    # the levels reference vertices of the primitive only, and every triangle of a level is non-degenerate and unique
    for k, (p_, idx) in enumerate(prims):
        for l in range(1, int(m["n_lods"][k])):
            off, ln = int(m["index_offset"][k, l]), int(m["index_len"][k, l])
            t = s["indices"][off : off + ln].reshape(-1, 3)
            assert t.max() < len(p_) and np.all((t[:, 0] != t[:, 1]) & (t[:, 0] != t[:, 2]) & (t[:, 1] != t[:, 2]))
            assert len({tuple(x) for x in t.tolist()}) == len(t)


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


def test_buffer_uris_cannot_leave_the_assets_directory():
    """an untrusted .gltf must not make the extractor read arbitrary local files into its output blob."""
    import json

    with tempfile.TemporaryDirectory() as d:
        g, blob, _ = gltf_fixture.build(embed=False)
        os.mkdir(os.path.join(d, "a"))
        open(os.path.join(d, "a", "buf.bin"), "wb").write(blob)
        open(os.path.join(d, "secret.bin"), "wb").write(blob)
        os.symlink(os.path.join(d, "secret.bin"), os.path.join(d, "a", "link.bin"))
        src = os.path.join(d, "a", "s.gltf")
        for uri, ok in (("buf.bin", True), ("./buf.bin", True), ("../secret.bin", False), (os.path.join(d, "secret.bin"), False),
                        ("file:///etc/passwd", False), ("link.bin", False), ("sub/../buf.bin", False), ("..", False)):
            g["buffers"][0]["uri"] = uri
            json.dump(g, open(src, "w"))
            r, _ = _extract(src)
            assert (r.returncode == 0) == ok, (uri, r.returncode, r.stderr)
            assert ok or r.returncode == 3


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
