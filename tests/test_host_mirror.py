"""The C++ host mirror of the reference's ECS systems (renderer_amd/host) driven through the
reference's per-frame schedule; its outputs are compared with the oracle fed the same planes."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from helpers import same_floats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "renderer_amd", "lib", "mip_frame_driver")


def _build_driver():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "renderer_amd", "csrc"), "-s"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "renderer_amd", "host"), "-s"])


def _write_scene(path, s):
    with open(path, "wb") as f:
        f.write(np.uint32(s["n"]).tobytes())
        f.write(np.uint32(len(s["meshes"])).tobytes())
        f.write(s["meshes"].tobytes())
        for k in ("pos", "rot", "scale", "mesh_id"):
            f.write(np.ascontiguousarray(s[k]).tobytes())


def _read_out(path):
    from renderer_amd.pipeline import DRAW_CMD_DTYPE

    raw = open(path, "rb").read()
    n, count, total = np.frombuffer(raw, np.uint32, 3)
    off = 12
    planes = np.frombuffer(raw, np.float32, 24, off); off += 96
    model = np.frombuffer(raw, np.float32, n * 16, off).reshape(n, 16); off += n * 64
    aabb = np.frombuffer(raw, np.float32, n * 6, off).reshape(n, 6); off += n * 24
    culled = np.frombuffer(raw, np.uint8, n, off); off += n
    cmds = np.frombuffer(raw, DRAW_CMD_DTYPE, count, off)
    return dict(n=int(n), count=int(count), total=int(total), planes=planes, model=model, aabb=aabb, culled=culled, cmds=cmds)


def test_driver_fails_loudly_without_gpu(gpu_available):
    if gpu_available:
        pytest.skip("a GPU is present")
    from renderer_amd import scene

    _build_driver()
    with tempfile.TemporaryDirectory() as d:
        _write_scene(os.path.join(d, "s.bin"), scene.make_scene(1, n=16))
        r = subprocess.run([DRIVER, os.path.join(d, "s.bin"), os.path.join(d, "o.bin")], capture_output=True, text=True)
    assert r.returncode == 12 and "no CPU fallback" in r.stderr  # 10 + MIP_ERR_NO_DEVICE


@pytest.mark.gpu
@pytest.mark.parametrize("config,n,frames", [(1, None, 1), (3, 30_000, 3)])
def test_schedule_through_cpp_mirror_matches_oracle(oracle_mod, config, n, frames):
    from renderer_amd import scene

    if not os.path.exists(DRIVER):
        _build_driver()
    s = scene.make_scene(config, n=n)
    with tempfile.TemporaryDirectory() as d:
        _write_scene(os.path.join(d, "s.bin"), s)
        r = subprocess.run([DRIVER, os.path.join(d, "s.bin"), os.path.join(d, "o.bin"), str(frames)],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        got = _read_out(os.path.join(d, "o.bin"))
    # the mirror's project_camera (float32, basis-vector look_at) lands within a few ulp of the planes
    assert np.allclose(got["planes"], s["planes"], rtol=2e-6, atol=1e-6)
    want = oracle_mod.run(s["pos"], s["rot"], s["scale"], s["mesh_id"], s["meshes"], got["planes"], s["cam_pos"], threads=4)
    assert got["n"] == s["n"] and got["count"] == want["draw_count"] and got["total"] == want["draw_index_total"]
    assert np.array_equal(got["culled"], want["coarse_culled"])
    assert got["cmds"].tobytes() == want["draw_cmds"].tobytes()
    assert same_floats(got["model"], want["model"]) and same_floats(got["aabb"], want["world_aabb"])
