"""Extension, BASELINE config 5 (skinned instances; the reference has no skinning): the HIP path
against this repository's oracle — palette, skinned world boxes, bitmap, commands, all bit-exact
(same operation order on both sides, no FMA)."""
import numpy as np
import pytest

from helpers import float_mismatches

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import renderer_amd

    renderer_amd.load_library()
    return renderer_amd


def _random_skeleton(rng, j):
    parent = np.array([-1] + [int(rng.integers(-1 if k % 5 == 4 else 0, k)) for k in range(1, j)], np.int32)
    ibm = np.tile(np.eye(4, dtype=np.float32).reshape(16), (j, 1))
    ibm[:, 12:15] = rng.uniform(-1, 1, (j, 3))
    ibm[:, [0, 5, 10]] = rng.uniform(0.8, 1.2, (j, 3))
    ibm[:, [1, 4, 6, 9]] = rng.uniform(-0.2, 0.2, (j, 4))
    lo = rng.uniform(-1, 0, (j, 3)).astype(np.float32)
    box = np.concatenate([lo, lo + rng.uniform(0.1, 1.0, (j, 3)).astype(np.float32)], axis=1)
    if j > 2:
        box[j // 2, 0] = box[j // 2, 3] + 1.0  # a joint that binds no vertex
    return dict(parent=parent, inverse_bind=ibm, joint_box=box)


def _random_poses(rng, n, j):
    poses = np.empty((n, j, 10), np.float32)
    poses[:, :, 0:3] = rng.uniform(-0.5, 0.5, (n, j, 3))
    q = rng.normal(size=(n, j, 4))
    poses[:, :, 3:7] = q / np.linalg.norm(q, axis=2, keepdims=True)
    poses[:, :, 7:10] = rng.uniform(0.7, 1.3, (n, j, 3))
    return poses


def _run_gpu(ra, s, sk, poses, first_instance_base=0, first_index_base=0, device_poses=False, frames_in_flight=1, repeat=1):
    import torch

    from renderer_amd.pipeline import make_frame

    n, j = s["n"], len(sk["parent"])
    dev = torch.device("cuda", 0)
    with ra.InstancePipeline(max_instances=max(n, 1), max_meshes=len(s["meshes"]), frames_in_flight=frames_in_flight) as p:
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        p.set_skeleton(sk["parent"], sk["inverse_bind"], sk["joint_box"])
        if device_poses:
            dposes = torch.from_numpy(poses).to(dev)
            torch.cuda.synchronize()
            p.set_poses_device(dposes.data_ptr(), n)
        else:
            p.set_poses(poses)
        model = torch.zeros((max(n, 1), 16), dtype=torch.float32, device=dev)
        palette = torch.zeros((max(n, 1), j, 16), dtype=torch.float32, device=dev)
        aabb = torch.zeros((max(n, 1), 6), dtype=torch.float32, device=dev)
        bitmap = torch.zeros((n + 31) // 32 + 1, dtype=torch.int32, device=dev)
        cmds = torch.zeros((max(n, 1), 5), dtype=torch.int32, device=dev)
        scal = torch.zeros(8, dtype=torch.int32, device=dev)
        frame = make_frame(s["planes"], s["cam_pos"], first_instance_base=first_instance_base, first_index_base=first_index_base)
        torch.cuda.synchronize()  # torch fills on its own stream; the library's streams do not wait for it
        for _ in range(repeat):
            p.run_skinned(frame, palette=palette.data_ptr(), model=model.data_ptr(), visible_bitmap=bitmap.data_ptr(),
                          draw_cmds=cmds.data_ptr(), draw_count=scal.data_ptr(), draw_index_total=scal.data_ptr() + 4,
                          world_aabb=aabb.data_ptr(), async_=repeat > 1)
        p.wait()
        count, total = (int(x) & 0xFFFFFFFF for x in scal[:2].cpu().tolist())
        return dict(model=model[:n].cpu().numpy(), palette=palette[:n].cpu().numpy(), world_aabb=aabb[:n].cpu().numpy(),
                    visible_bitmap=bitmap[:(n + 31) // 32].cpu().numpy().view(np.uint32), draw_count=count, draw_index_total=total,
                    draw_cmds=cmds[:count].cpu().numpy().view(np.uint32).reshape(-1).view(ra.DRAW_CMD_DTYPE))


def _check(got, want, what):
    for key in ("palette", "world_aabb", "model"):
        mm = float_mismatches(got[key].reshape(want[key].shape), want[key])
        assert len(mm) == 0, f"{what}: {len(mm)} {key} entries differ, first {mm[:4].tolist()}"
    assert np.array_equal(got["visible_bitmap"], want["visible_bitmap"]), f"{what}: bitmap"
    assert got["draw_count"] == want["draw_count"] and got["draw_index_total"] == want["draw_index_total"], what
    assert got["draw_cmds"].tobytes() == want["draw_cmds"].tobytes(), f"{what}: commands"


@pytest.mark.parametrize("n", [1, 2, 3, 11, 12, 13, 1000, 40_003, 256_000])
def test_rigged_figure_scene(ra, oracle_mod, n):
    s = ra.scene.make_skinned_scene(n)
    want = oracle_mod.run_skinned(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["skeleton"], s["poses"],
                                  s["planes"], s["cam_pos"], first_instance_base=7, first_index_base=11)
    got = _run_gpu(ra, s, s["skeleton"], s["poses"], first_instance_base=7, first_index_base=11)
    _check(got, want, f"rigged n={n}")
    if n >= 1000:
        assert 0 < want["draw_count"] < n  # the frustum does cut the scene


def test_committed_extension_fixtures(ra):
    """tests/golden/ext/ reproduces on the GPU without the oracle: the skinned frame and the per-light lists."""
    import os

    import torch

    here = os.path.dirname(os.path.abspath(__file__))
    g = np.load(os.path.join(here, "golden", "ext", "skinned_301.npz"))
    s = dict(n=len(g["scale"]), pos=g["pos"], rot=g["rot"], scale=g["scale"], mesh_id=g["mesh_id"], meshes=g["meshes"],
             planes=g["planes"], cam_pos=g["cam_pos"])
    sk = dict(parent=g["parent"], inverse_bind=g["inverse_bind"], joint_box=g["joint_box"])
    got = _run_gpu(ra, s, sk, g["poses"])
    want = {k: g[k] for k in ("palette", "world_aabb", "model", "visible_bitmap", "draw_cmds")}
    want["draw_count"], want["draw_index_total"] = int(g["draw_count"]), int(g["draw_index_total"])
    _check(got, want, "fixture skinned_301")
    g = np.load(os.path.join(here, "golden", "ext", "lights_1001.npz"))
    n, lights = len(g["scale"]), g["lights"]
    dev = torch.device("cuda", 0)
    with ra.InstancePipeline(max_instances=n, max_meshes=64) as p:
        p.set_mesh_table(g["meshes"])
        p.set_instances(g["pos"], g["rot"], g["scale"], g["mesh_id"])
        out = torch.zeros((len(lights) * n, 5), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        p.light_draw_lists(lights, out.data_ptr(), first_instance_base=int(g["first_instance_base"]))
        assert out.cpu().numpy().tobytes() == g["lists"].tobytes()


@pytest.mark.parametrize("j", [1, 2, 7, 16, 21, 32])
def test_other_joint_counts_and_hierarchies(ra, oracle_mod, j):
    rng = np.random.default_rng(100 + j)
    n = 3001
    s = ra.scene.make_scene(3, n=n)
    sk = _random_skeleton(rng, j)
    poses = _random_poses(rng, n, j)
    want = oracle_mod.run_skinned(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], sk, poses, s["planes"], s["cam_pos"])
    got = _run_gpu(ra, s, sk, poses, device_poses=(j % 2 == 0))
    _check(got, want, f"j={j}")


def test_identity_skin_reproduces_the_rigid_path(ra, oracle_mod):
    """One joint, identity pose and bind matrix, joint box = the mesh box: the skinned frame must be the
    rigid frame of rows a-1..a-7, bit for bit (oracle.run, not run_skinned, is the expectation)."""
    s = ra.scene.make_scene(2, n=20_000)
    m = s["meshes"][0]
    sk = dict(parent=np.array([-1], np.int32), inverse_bind=np.eye(4, dtype=np.float32).reshape(1, 16),
              joint_box=np.concatenate([m["aabb_min"], m["aabb_max"]]).reshape(1, 6))
    poses = np.zeros((s["n"], 1, 10), np.float32)
    poses[:, :, 6] = 1.0
    poses[:, :, 7:] = 1.0
    want = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["planes"], s["cam_pos"])
    got = _run_gpu(ra, s, sk, poses)
    want["palette"] = np.tile(np.eye(4, dtype=np.float32).reshape(16), (s["n"], 1, 1))
    _check(got, want, "identity skin")


def test_non_finite_and_degenerate_inputs(ra, oracle_mod):
    s = ra.scene.make_skinned_scene(600)
    poses = s["poses"].copy()
    pos, rot, scale = s["pos"].copy(), s["rot"].copy(), s["scale"].copy()
    poses[3, 4, 0] = np.nan          # a NaN joint translation poisons that joint and its children
    poses[5, 0, 6] = np.inf          # infinite root rotation component
    poses[9, :, 7:10] = 0.0          # zero joint scales
    poses[11, 7, 3:7] = 0.0          # zero quaternion
    pos[20] = np.nan                 # NaN instance position: the rigid general path, every joint box NaN
    scale[21] = 0.0
    rot[22] = 0.0
    pos[23, 1] = np.inf
    s2 = dict(s, pos=pos, rot=rot, scale=scale)
    want = oracle_mod.run_skinned(pos, rot, scale, s["mesh_id"], s["meshes"], s["skeleton"], poses, s["planes"], s["cam_pos"])
    got = _run_gpu(ra, s2, s["skeleton"], poses)
    _check(got, want, "specials")
    sk = dict(s["skeleton"])
    box = sk["joint_box"].copy()
    box[:, 0] = box[:, 3] + 1.0      # no joint binds a vertex: the fold keeps its seeds
    sk["joint_box"] = box
    want = oracle_mod.run_skinned(pos, rot, scale, s["mesh_id"], s["meshes"], sk, poses, s["planes"], s["cam_pos"])
    got = _run_gpu(ra, s2, sk, poses)
    _check(got, want, "all boxes empty")


def test_two_pose_buffers_alternate_with_frames_in_flight(ra, oracle_mod):
    """An animation system double-buffers its poses: frame k reads buffer A while frame k+1, queued behind it on
    the other frame slot, reads buffer B. Each output set must carry its own pose's result."""
    import torch

    from renderer_amd.pipeline import make_frame

    s = ra.scene.make_skinned_scene(60_000)
    sk = s["skeleton"]
    poses_a = s["poses"]
    poses_b = ra.scene.make_poses(s["n"], sk, 0xB0B, max_angle_deg=90.0)
    wants = [oracle_mod.run_skinned(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], sk, p, s["planes"], s["cam_pos"],
                                    want=("draw_cmds", "palette")) for p in (poses_a, poses_b)]
    assert wants[0]["draw_cmds"].tobytes() != wants[1]["draw_cmds"].tobytes()
    dev = torch.device("cuda", 0)
    n, j = s["n"], len(sk["parent"])
    with ra.InstancePipeline(max_instances=n, max_meshes=1, frames_in_flight=2) as p:
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"], s["rot"], s["scale"], s["mesh_id"])
        p.set_skeleton(sk["parent"], sk["inverse_bind"], sk["joint_box"])
        bufs = [torch.from_numpy(x).to(dev) for x in (poses_a, poses_b)]
        outs = []
        for _ in range(2):
            outs.append(dict(cmds=torch.zeros((n, 5), dtype=torch.int32, device=dev), scal=torch.zeros(8, dtype=torch.int32, device=dev),
                             palette=torch.zeros((n, j, 16), dtype=torch.float32, device=dev)))
        torch.cuda.synchronize()
        frame = make_frame(s["planes"], s["cam_pos"])
        for k in range(8):  # A, B, A, B ... queued without waiting
            o = outs[k % 2]
            p.set_poses_device(bufs[k % 2].data_ptr(), n)
            p.run_skinned(frame, palette=o["palette"].data_ptr(), draw_cmds=o["cmds"].data_ptr(), draw_count=o["scal"].data_ptr(),
                          draw_index_total=o["scal"].data_ptr() + 4, async_=True)
        p.wait()
        for o, w in zip(outs, wants):
            count = int(o["scal"][0].item())
            assert count == w["draw_count"] and o["cmds"][:count].cpu().numpy().tobytes() == w["draw_cmds"].tobytes()
            assert len(float_mismatches(o["palette"].cpu().numpy(), w["palette"])) == 0


def test_frames_in_flight_and_errors(ra, oracle_mod):
    import torch

    from renderer_amd.pipeline import make_frame

    s = ra.scene.make_skinned_scene(30_000)
    want = oracle_mod.run_skinned(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], s["skeleton"], s["poses"],
                                  s["planes"], s["cam_pos"])
    got = _run_gpu(ra, s, s["skeleton"], s["poses"], frames_in_flight=2, repeat=7)
    _check(got, want, "7 frames, 2 in flight")
    with ra.InstancePipeline(max_instances=64, max_meshes=1) as p:
        p.set_mesh_table(s["meshes"])
        p.set_instances(s["pos"][:64], s["rot"][:64], s["scale"][:64], s["mesh_id"][:64])
        frame = make_frame(s["planes"], s["cam_pos"])
        with pytest.raises(ra.MipError):   # no skeleton yet
            p.run_skinned(frame)
        with pytest.raises(ra.MipError):
            p.set_poses(s["poses"][:64])
        sk = s["skeleton"]
        with pytest.raises(ra.MipError):   # a parent must precede its child
            p.set_skeleton(np.array([1, -1], np.int32), sk["inverse_bind"][:2], sk["joint_box"][:2])
        p.set_skeleton(sk["parent"], sk["inverse_bind"], sk["joint_box"])
        with pytest.raises(ra.MipError):   # poses for another instance count
            p.set_poses(s["poses"][:63])
        with pytest.raises(ra.MipError):   # skeleton but no poses
            p.run_skinned(frame)
        p.set_poses(s["poses"][:64])
        p.run_skinned(frame)               # no outputs at all is allowed
        p.set_instances(s["pos"][:32], s["rot"][:32], s["scale"][:32], s["mesh_id"][:32])
        with pytest.raises(ra.MipError):   # the instance count changed under the poses
            p.run_skinned(frame)
        dev = torch.device("cuda", 0)
        stream_out = torch.zeros(16, dtype=torch.int32, device=dev)
        p.set_poses(s["poses"][:32])
        with pytest.raises(ra.MipError):   # the per-triangle stage does not skin
            p.run_skinned(frame, culled_index_buffer=stream_out.data_ptr(), culled_index_capacity=16)
