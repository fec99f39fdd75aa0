"""CPU tests of the oracle: committed golden vectors, hand-computable known answers, edge
cases, and an independent numpy restatement. No GPU involved."""
import glob
import os

import numpy as np
import pytest

import numpy_restatement as npr
from helpers import same_floats

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def _scene_from(g):
    return dict(pos=g["pos"], rot=g["rot"], scale=g["scale"], mesh_id=g["mesh_id"], meshes=g["meshes"],
                planes=g["planes"], cam_pos=g["cam_pos"], n=len(g["scale"]))


def _inside_planes():
    """Six planes nothing is outside of: (0,0,0,-1) gives s = -1, e = 0."""
    return np.tile(np.array([0, 0, 0, -1], np.float32), 6)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_reproduces_golden(oracle_mod, path):
    g = np.load(path)
    s = _scene_from(g)
    for threads in (None, 3):
        r = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"],
                           first_instance_base=int(g["first_instance_base"]), first_index_base=int(g["first_index_base"]),
                           threads=threads)
        assert np.array_equal(r["visible_bitmap"], g["visible_bitmap"])
        assert np.array_equal(r["coarse_culled"], g["coarse_culled"])
        assert r["draw_count"] == int(g["draw_count"]) and r["draw_index_total"] == int(g["draw_index_total"])
        assert r["draw_cmds"].tobytes() == g["draw_cmds"].tobytes()
        assert same_floats(r["model"], g["model"]) and same_floats(r["world_aabb"], g["world_aabb"])


def test_extension_fixtures_reproduce(oracle_mod):
    """tests/golden/ext/: the skinned frame (BASELINE config 5 extension) and the per-light draw lists."""
    g = np.load(os.path.join(HERE, "golden", "ext", "skinned_301.npz"))
    sk = dict(parent=g["parent"], inverse_bind=g["inverse_bind"], joint_box=g["joint_box"])
    r = oracle_mod.run_skinned(g["pos"], g["rot"], g["scale"], g["mesh_id"], g["meshes"], sk, g["poses"], g["planes"], g["cam_pos"])
    for key in ("palette", "local_box", "model", "world_aabb"):
        assert same_floats(r[key].reshape(g[key].shape), g[key]), key
    assert np.array_equal(r["visible_bitmap"], g["visible_bitmap"]) and r["draw_count"] == int(g["draw_count"])
    assert r["draw_cmds"].tobytes() == g["draw_cmds"].tobytes() and r["draw_index_total"] == int(g["draw_index_total"])
    g = np.load(os.path.join(HERE, "golden", "ext", "lights_1001.npz"))
    lists = oracle_mod.light_draw_lists(g["pos"], g["mesh_id"], g["meshes"], g["lights"], first_instance_base=int(g["first_instance_base"]))
    assert lists.tobytes() == g["lists"].tobytes()


def test_golden_set_is_present():
    names = {os.path.basename(p) for p in GOLDEN}
    assert {"box_1024.npz", "special_513.npz", "mixed_4097_bases.npz"} <= names


def test_model_matrix_known_answers(oracle_mod):
    # identity rotation, unit scale: M = T
    m = oracle_mod.model_matrix([1, 2, 3], [0, 0, 0, 1], 1.0).reshape(4, 4).T
    assert np.array_equal(m, np.array([[1, 0, 0, 1], [0, 1, 0, 2], [0, 0, 1, 3], [0, 0, 0, 1]], np.float32))
    # 90 degrees about +Y (q = (0, sqrt(.5), 0, sqrt(.5))), scale 2: z -> x
    r = np.float32(np.sqrt(0.5))
    m = oracle_mod.model_matrix([0, 0, 0], [0, r, 0, r], 2.0).reshape(4, 4).T
    two_wj = np.float32(np.float32(r * r) * np.float32(2.0)) * np.float32(2.0)  # (w*j*2)*s
    assert m[0, 2] == two_wj and m[2, 0] == -two_wj and m[1, 1] == np.float32(np.float32(r * r) + np.float32(r * r)) * 2
    assert m[0, 0] == 0 and m[2, 2] == 0 and np.array_equal(m[3], [0, 0, 0, 1])
    # the quaternion is not renormalised: |q| = 2 scales the rotation block by 4
    m = oracle_mod.model_matrix([0, 0, 0], [0, 0, 0, 2], 1.0).reshape(4, 4).T
    assert np.array_equal(np.diag(m), [4, 4, 4, 1])
    # zero scale collapses the linear part, translation survives
    m = oracle_mod.model_matrix([5, 6, 7], [0.1, 0.2, 0.3, 0.9], 0.0).reshape(4, 4).T
    assert np.all(m[:3, :3] == 0) and np.array_equal(m[:3, 3], [5, 6, 7])


def test_world_aabb_roundtrip_and_nan_fold(oracle_mod):
    m = oracle_mod.model_matrix([10, 0, 0], [0, 0, 0, 1], 2.0)
    mins, maxs = oracle_mod.world_aabb(m, [-0.5, -1, -2], [0.5, 1, 2])
    assert np.array_equal(mins, [9, -2, -4]) and np.array_equal(maxs, [11, 2, 4])
    # all corners NaN: f32::min/max ignore NaN, so the fold keeps (MAX, MIN); centre 0, half -inf
    m = oracle_mod.model_matrix([np.nan, np.nan, np.nan], [0, 0, 0, 1], 1.0)
    mins, maxs = oracle_mod.world_aabb(m, [-1, -1, -1], [1, 1, 1])
    assert np.all(np.isposinf(mins)) and np.all(np.isneginf(maxs))


def test_plane_tangency_is_visible(oracle_mod):
    planes = _inside_planes()
    planes[0:4] = [1, 0, 0, -5]  # outside where x - 5 > 0
    one_up = np.nextafter(np.float32(6), np.float32(7))
    for cx, want_culled in ((4.0, False), (6.0, False), (one_up, True), (7.0, True)):
        mins = np.array([cx - 1, -1, -1], np.float32)
        maxs = np.array([cx + 1, 1, 1], np.float32)
        # at cx = 6 the box spans [5, 7]: s - e = (6 - 5) - 1 = 0, not > 0
        assert oracle_mod.coarse_culled(mins, maxs, planes) == want_culled, cx
    # NaN box => every comparison false => visible
    assert oracle_mod.coarse_culled([np.nan] * 3, [np.nan] * 3, planes) is False


def test_default_camera_planes(oracle_mod):
    from renderer_amd import scene

    planes = scene.default_planes()
    committed = np.load(os.path.join(HERE, "golden", "default_planes.npy"))
    assert np.array_equal(planes, committed)
    # the oracle's float32 project_camera agrees with the float64-rounded planes to a few ulp
    assert np.allclose(oracle_mod.project_camera(), planes, rtol=2e-6, atol=1e-6)
    p = planes.reshape(6, 4)

    def outside(pt):
        return [float(np.dot(pl[:3], pt) + pl[3]) > 0 for pl in p]

    assert not any(outside([0, 1, 12]))           # straight ahead
    assert outside([0, 1, 1.0])[4]                # nearer than the (GL-style, conservative) near plane
    assert outside([0, 1, 103.0])[5]              # beyond far = 100 from z = 2
    assert outside([-100, 1, 12])[0] and outside([100, 1, 12])[1]
    assert outside([0, -100, 12])[2] and outside([0, 100, 12])[3]


def test_pick_lod_threshold(oracle_mod):
    cam = np.array([0, 1, 2], np.float32)
    assert oracle_mod.pick_lod(3, cam, [0, 1, 12]) == 0          # distance exactly 10: `>` is false
    assert oracle_mod.pick_lod(3, cam, [0, 1, np.nextafter(np.float32(12), np.float32(13))]) == 1
    assert oracle_mod.pick_lod(1, cam, [0, 1, 500]) == 0         # a single LOD never switches
    assert oracle_mod.pick_lod(3, cam, [np.nan, 1, 12]) == 0     # NaN distance: comparison false


def test_lod_squared_distance_threshold_equivalence():
    """The kernel tests dist^2 > 100 + 2^-17 instead of sqrt(dist^2) > 10 (kLodDistSqThreshold):
    equivalent for every float because sqrt is correctly rounded and monotonic."""
    thr = np.float32(100.00000762939453125)
    assert thr == np.nextafter(np.float32(100), np.float32(101))
    lo, hi = np.float32(99.5).view(np.uint32), np.float32(100.5).view(np.uint32)
    q = np.arange(lo, hi + 1, dtype=np.uint32).view(np.float32)  # every float in [99.5, 100.5]
    assert len(q) > 100_000
    assert np.array_equal(np.sqrt(q) > np.float32(10.0), q > thr)
    for special in (np.float32(np.inf), np.float32(np.nan), np.float32(0), np.float32(3.4e38), np.float32(1e-45)):
        with np.errstate(invalid="ignore"):
            assert bool(np.sqrt(special) > np.float32(10.0)) == bool(special > thr)


def test_plane_test_without_the_subtraction():
    """The multi-view kernel evaluates the frustum test `sd - e > 0` (cull_pipeline.rs:108-119) as `sd > e`. With subnormals
    kept (the build's mode, and numpy's) the two agree for every pair of floats: a difference that underflows is exact, so its
    sign is the comparison's; inf - inf and NaN are false on both sides. Checked on neighbours over every exponent, subnormals,
    zeros, infinities, NaNs and random bit patterns."""
    rng = np.random.default_rng(11)
    base = []
    for e in list(range(-149, 128, 2)) + [-126, -127, 127]:
        v = np.float32(np.ldexp(1.0, e))
        for frac in (1.0, 1.0000001, 1.5, 1.9999999):
            base.append(np.float32(v * np.float32(frac)))
    base = np.array(base + [0.0, np.inf, np.nan, 1e-45, 1.1754942e-38, 1.1754944e-38, 3.4028235e38], np.float32)
    base = np.concatenate([base, -base])
    sd, e = [], []
    with np.errstate(all="ignore"):
        for v in base:
            near = [v]
            up, dn = v, v
            for _ in range(4):
                up = np.nextafter(up, np.float32(np.inf)); dn = np.nextafter(dn, np.float32(-np.inf))
                near += [up, dn]
            for x in near:
                sd.append(np.float32(x)); e.append(v)
                sd.append(v); e.append(np.float32(x))
        bits = rng.integers(0, 2**32, 400_000, dtype=np.uint64).astype(np.uint32).view(np.float32)
        sd = np.concatenate([np.array(sd, np.float32), bits[:200_000], rng.standard_normal(100_000).astype(np.float32)])
        e = np.concatenate([np.array(e, np.float32), bits[200_000:], np.abs(rng.standard_normal(100_000)).astype(np.float32)])
        assert np.array_equal((sd - e) > np.float32(0), sd > e)


def test_ndc_comparison_without_division():
    """The triangle kernel tests RN(x/w) > 1 as x*sgn(w) > |w| (and < -1 as x*sgn(w) < -|w|) instead of
    dividing. Check the equivalence against float32 divisions on the cases where it could break: x within a
    few ulps of +-w over every exponent, subnormals, zeros of both signs, infinities, NaNs, random pairs."""
    rng = np.random.default_rng(7)
    ws = []
    for e in list(range(-149, 128, 3)) + [-126, -127, -125, 127]:
        base = np.float32(2.0) ** np.float32(e) if e > -127 else np.float32(np.ldexp(1.0, e))
        for frac in (1.0, 1.0000001, 1.25, 1.5, 1.9999999):
            ws.append(np.float32(base * np.float32(frac)))
    ws = np.array(ws + [0.0, np.inf, np.nan, 1e-45, 3e-45, 1.1754942e-38, 1.1754944e-38, 3.4028235e38], np.float32)
    ws = np.concatenate([ws, -ws])
    xs_all, ws_all = [], []
    np.seterr(all="ignore")
    for w in ws:
        near = np.full(17, abs(w), np.float32)
        for k in range(8):
            near[1 + k:] = np.nextafter(near[1 + k:], np.float32(np.inf))
        down = np.full(9, abs(w), np.float32)
        for k in range(8):
            down[1 + k:] = np.nextafter(down[1 + k:], np.float32(0))
        cand = np.concatenate([near[:9], down, [0.0, np.inf, np.nan, abs(w) * np.float32(2), abs(w) * np.float32(0.5)]]).astype(np.float32)
        cand = np.concatenate([cand, -cand])
        xs_all.append(cand)
        ws_all.append(np.full(len(cand), w, np.float32))
    xs = np.concatenate(xs_all + [rng.standard_normal(200_000).astype(np.float32) * np.float32(3),
                                   (rng.integers(0, 2**32, 200_000, dtype=np.uint64).astype(np.uint32)).view(np.float32)])
    wv = np.concatenate(ws_all + [rng.standard_normal(200_000).astype(np.float32),
                                   (rng.integers(0, 2**32, 200_000, dtype=np.uint64).astype(np.uint32)).view(np.float32)])
    with np.errstate(all="ignore"):
        q = xs / wv                                              # IEEE binary32 division, round to nearest even
        sign = wv.view(np.uint32) & np.uint32(0x80000000)
        xf = (xs.view(np.uint32) ^ sign).view(np.float32)
        wa = np.abs(wv)
        assert np.array_equal(q > np.float32(1), xf > wa)
        assert np.array_equal(q < np.float32(-1), xf < -wa)
    np.seterr(all="warn")
    assert len(xs) > 400_000


def test_emit_and_compact_semantics(oracle_mod):
    from renderer_amd import scene

    s = scene.make_scene(3, n=3000, all_visible=True)
    s["meshes"]["index_len"][::5, 0] = 0
    s["meshes"]["index_len"][::5, 1] = 0
    r = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"],
                       first_instance_base=7, first_index_base=100)
    cmds = r["draw_cmds"]
    vis = r["coarse_culled"] == 0
    assert vis.all()
    assert np.all(cmds["indexCount"] > 0) and np.all(cmds["instanceCount"] == 1)
    assert np.all(np.diff(cmds["firstInstance"].astype(np.int64)) > 0)  # stable order
    dropped = s["n"] - len(cmds)
    assert dropped == int((s["meshes"]["index_len"][s["mesh_id"], 0] == 0).sum()) > 0
    run = 100 + np.concatenate([[0], np.cumsum(cmds["indexCount"].astype(np.int64))[:-1]])
    assert np.array_equal(cmds["firstIndex"].astype(np.int64), run)
    assert np.array_equal(cmds["vertexOffset"], s["meshes"]["vertex_offset"][s["mesh_id"][cmds["firstInstance"] - 7]])
    # compaction on its own: zero entries vanish, order kept, in-place capable
    sparse = np.zeros(10, oracle_mod.DRAW_CMD_DTYPE)
    sparse["indexCount"][[1, 4, 9]] = (3, 6, 9)
    sparse["firstInstance"] = np.arange(10)
    packed = oracle_mod.compact_draw_stream(sparse)
    assert packed["firstInstance"].tolist() == [1, 4, 9]


def test_first_index_wraps_like_u32(oracle_mod):
    from renderer_amd import scene

    s = scene.make_scene(2, n=200_000, all_visible=True)  # 200k x 46 356 indices overflows 2^32
    r = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"], threads=8,
                       want=("draw_cmds",))
    total = int(r["draw_cmds"]["indexCount"].astype(np.uint64).sum())
    assert total > 2 ** 32 and r["draw_index_total"] == total % 2 ** 32


def test_merge_draw_lists(oracle_mod):
    from renderer_amd import scene

    s = scene.make_scene(3, n=5000)
    whole = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"])
    lists, totals = [], []
    for lo, hi in ((0, 1700), (1700, 1700), (1700, 5000)):  # includes an empty shard
        part = oracle_mod.run(s["pos"][lo:hi], s["rot"][lo:hi], s["scale"][lo:hi], s["mesh_id"][lo:hi], s["meshes"],
                              s["planes"], s["cam_pos"], first_instance_base=lo)
        lists.append(part["draw_cmds"])
        totals.append(part["draw_index_total"])
    merged, index_total = oracle_mod.merge_draw_lists(lists, totals)
    assert merged.tobytes() == whole["draw_cmds"].tobytes() and index_total == whole["draw_index_total"]


@pytest.mark.parametrize("special", [False, True])
def test_numpy_restatement_agrees(oracle_mod, special):
    from renderer_amd import scene

    s = scene.make_scene(3, n=20_000, all_visible=special)
    if special:
        rng = np.random.default_rng(3)
        sv = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, 1e-38, 3.4e38, -3.4e38, 1e19, 1e-20], np.float32)
        for col, width in (("pos", 3), ("rot", 4)):
            rows = rng.choice(s["n"], 500, replace=False)
            s[col][rows, rng.integers(0, width, 500)] = rng.choice(sv, 500)
        s["scale"][rng.choice(s["n"], 300, replace=False)] = rng.choice(sv, 300)
    a = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"])
    b = npr.run(s)
    assert same_floats(a["model"], b["model"])
    assert same_floats(a["world_aabb"], b["world_aabb"])
    assert np.array_equal(a["coarse_culled"].astype(bool), b["coarse_culled"])
    c = a["draw_cmds"]
    assert np.array_equal(c["indexCount"], b["cmds"]["indexCount"]) and np.array_equal(c["firstIndex"], b["cmds"]["firstIndex"])
    assert np.array_equal(c["firstInstance"], b["cmds"]["firstInstance"]) and np.array_equal(c["vertexOffset"], b["cmds"]["vertexOffset"])
    assert a["draw_index_total"] == b["cmds"]["total"]


def test_light_draw_lists_against_numpy(oracle_mod):
    """Row f-4, shadow pass: every light lists every instance; only the LOD depends on the light."""
    import numpy_restatement as npr
    from renderer_amd import scene

    s = scene.make_scene(3, n=5000)
    pos = s["pos"].copy()
    pos[7] = np.nan  # NaN distance compares false: LOD 0 (helpers.rs:6)
    lights = np.array([[30, 20, -40.1], [0.1, 17.0, -0.1], [0.0, 0.0, 0.0], pos[11]], np.float32)  # main.rs:368-382 + on top of an instance
    got = oracle_mod.light_draw_lists(pos, s["mesh_id"], s["meshes"], lights, first_instance_base=9)
    want = npr.light_draw_lists(pos, s["mesh_id"], s["meshes"], lights, first_instance_base=9)
    assert got.shape == (4, 5000)
    assert np.array_equal(got.view(np.uint32).reshape(4, 5000, 5), want)
    m = s["meshes"][s["mesh_id"][7]]
    assert all(got[l, 7]["indexCount"] == m["index_len"][0] for l in range(4))
    assert got[3, 11]["indexCount"] == s["meshes"][s["mesh_id"][11]]["index_len"][0]  # distance 0: LOD 0
    assert len({int(got[l]["indexCount"].astype(np.uint64).sum()) for l in range(4)}) > 1  # the lights do differ


def test_skinned_extension_against_float64(oracle_mod):
    """Extension (BASELINE config 5; no reference semantics): palette and skinned boxes of the oracle
    against an independent float64 evaluation of glTF's J_k = G_k * IBM_k, and the bind pose."""
    from renderer_amd import scene

    s = scene.make_skinned_scene(300)
    sk, poses = s["skeleton"], s["poses"]
    got = oracle_mod.run_skinned(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], sk, poses, s["planes"], s["cam_pos"])
    j = len(sk["parent"])

    def trs(t, q, sc):
        i, jj, k, w = (float(x) for x in q)
        r = np.array([[w * w + i * i - jj * jj - k * k, 2 * (i * jj - w * k), 2 * (w * jj + i * k)],
                      [2 * (w * k + i * jj), w * w - i * i + jj * jj - k * k, 2 * (jj * k - w * i)],
                      [2 * (i * k - w * jj), 2 * (w * i + jj * k), w * w - i * i - jj * jj + k * k]])
        m = np.eye(4)
        m[:3, :3] = r * np.asarray(sc, np.float64)[None, :]
        m[:3, 3] = t
        return m

    for inst in (0, 17, 299):
        g = [None] * j
        lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
        model = trs(s["pos"][inst], s["rot"][inst], [s["scale"][inst]] * 3)
        for k in range(j):
            l = trs(poses[inst, k, 0:3], poses[inst, k, 3:7], poses[inst, k, 7:10])
            g[k] = l if sk["parent"][k] < 0 else g[sk["parent"][k]] @ l
            jm = g[k] @ sk["inverse_bind"][k].reshape(4, 4).T.astype(np.float64)
            assert np.allclose(got["palette"][inst, k].reshape(4, 4).T, jm, rtol=1e-5, atol=1e-5)
            b = sk["joint_box"][k].astype(np.float64)
            for c in range(8):
                v = jm @ np.array([b[3 if c & 1 else 0], b[4 if c & 4 else 1], b[5 if c & 2 else 2], 1.0])
                lo, hi = np.minimum(lo, v[:3]), np.maximum(hi, v[:3])
        assert np.allclose(got["local_box"][inst], np.concatenate([lo, hi]), rtol=1e-5, atol=1e-5)
        # the posed box then goes through aabb_calculation like any mesh box
        wl, wh = np.full(3, np.inf), np.full(3, -np.inf)
        for c in range(8):
            v = model @ np.array([hi[0] if c & 1 else lo[0], hi[1] if c & 4 else lo[1], hi[2] if c & 2 else lo[2], 1.0])
            wl, wh = np.minimum(wl, v[:3]), np.maximum(wh, v[:3])
        assert np.allclose(got["world_aabb"][inst], np.concatenate([wl, wh]), rtol=1e-4, atol=1e-4)
    bind = poses[:4].copy()
    bind[:, :, 3:6], bind[:, :, 6], bind[:, :, 7:] = 0.0, 1.0, 1.0
    r0 = oracle_mod.run_skinned(s["pos"][:4], s["rot"][:4], s["scale"][:4], s["mesh_id"][:4], s["meshes"], sk, bind,
                                s["planes"], s["cam_pos"])
    assert np.abs(r0["palette"] - np.eye(4, dtype=np.float32).reshape(16)).max() < 1e-6  # bind pose: identity palette
    # one identity joint whose box is the mesh box: the skinned frame IS the rigid frame
    rigid = scene.make_scene(2, n=3000)
    m = rigid["meshes"][0]
    one = dict(parent=np.array([-1], np.int32), inverse_bind=np.eye(4, dtype=np.float32).reshape(1, 16),
               joint_box=np.concatenate([m["aabb_min"], m["aabb_max"]]).reshape(1, 6))
    ident = np.zeros((3000, 1, 10), np.float32)
    ident[:, :, 6], ident[:, :, 7:] = 1.0, 1.0
    a = oracle_mod.run_skinned(rigid["pos"], rigid["rot"], rigid["scale"], rigid["mesh_id"], rigid["meshes"], one, ident,
                               rigid["planes"], rigid["cam_pos"])
    b = oracle_mod.run(rigid["pos"], rigid["rot"], rigid["scale"], rigid["mesh_id"], rigid["meshes"], rigid["planes"], rigid["cam_pos"])
    assert a["draw_cmds"].tobytes() == b["draw_cmds"].tobytes() and np.array_equal(a["world_aabb"], b["world_aabb"])
    assert np.array_equal(a["visible_bitmap"], b["visible_bitmap"])
    with pytest.raises(ValueError):
        oracle_mod.run_skinned(s["pos"][:4], s["rot"][:4], s["scale"][:4], s["mesh_id"][:4], s["meshes"],
                               dict(sk, parent=np.array([0] + list(sk["parent"][1:]), np.int32)), bind, s["planes"], s["cam_pos"])


def test_mesh_id_out_of_range_is_rejected(oracle_mod):
    from renderer_amd import scene

    s = scene.make_scene(1, n=8)
    s["mesh_id"][3] = 9
    with pytest.raises(ValueError):
        oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"])


def test_oracle_against_float64_formulas(oracle_mod):
    """Independent of operation order: the oracle's matrices are within 1e-5 relative of the float64
    formulas (the north star's tolerance) and its visibility equals the float64 decision for every
    instance that is not within rounding distance of a plane."""
    import float64_reference as f64
    from renderer_amd import scene

    for config, n in ((2, 30_000), (3, 30_000)):
        s = scene.make_scene(config, n=n)
        a = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"])
        b = f64.run(s)
        denom = np.maximum(np.abs(b["model"]), 1e-3 * np.abs(b["model"]).max(axis=1, keepdims=True))
        assert np.max(np.abs(a["model"].astype(np.float64) - b["model"]) / denom) < 1e-5
        assert np.allclose(a["world_aabb"][:, :3], b["mins"], rtol=1e-5, atol=1e-4)
        assert np.allclose(a["world_aabb"][:, 3:], b["maxs"], rtol=1e-5, atol=1e-4)
        d = b["decided"]
        assert d.mean() > 0.99
        assert np.array_equal(a["coarse_culled"].astype(bool)[d], b["culled"][d])
