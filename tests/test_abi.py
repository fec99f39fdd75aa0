"""The C-ABI library loads, exports every symbol include/mi_instance_pipeline.h declares, keeps
its struct layouts, and fails loudly (never falls back to a CPU path) without a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mi_instance_pipeline.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mip_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import renderer_amd
    from renderer_amd import _lib

    lib = renderer_amd.load_library()
    declared = _declared_functions()
    assert len(declared) >= 12
    assert set(declared) == set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mip_abi_version() == 4
    # a pure function of the ABI (no device): the bits a packed wire record leaves for the instance index — the library's
    # answer, the Python mirror's and the header's wording (31 - ceil(log2(n_meshes))) agree
    from renderer_amd.pipeline import wire_index_bits
    for m in (0, 1, 2, 3, 4, 5, 63, 64, 65, 1023, 1024, 1025, 1 << 20, (1 << 31) - 1, 1 << 31, (1 << 32) - 1):
        want = 31 - max(0, min(31, (m - 1).bit_length() if m > 1 else 0))
        assert lib.mip_wire_index_bits(m) == wire_index_bits(m) == want, m


def test_struct_layouts_match_header():
    from renderer_amd import _lib
    from renderer_amd.pipeline import DRAW_CMD_DTYPE, MESH_DTYPE

    assert C.sizeof(_lib.MipConfig) == 32
    assert C.sizeof(_lib.MipFrame) == 24 * 4 + 3 * 4 + 8 + 64
    assert C.sizeof(_lib.MipOutputs) == 6 * 8 + 8 + 16 + 8
    assert C.sizeof(_lib.MipTimings) == 104
    assert MESH_DTYPE.itemsize == 80 and DRAW_CMD_DTYPE.itemsize == 20
    # compile the header as C and compare sizeof/offsetof with the Python mirrors
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "mi_instance_pipeline.h"
    int main(void) {
      printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(MipConfig), sizeof(MipMesh), sizeof(MipFrame),
             sizeof(MipOutputs), sizeof(MipTimings), sizeof(MipDrawIndexedIndirectCommand), sizeof(MipShardHeader),
             offsetof(MipMesh, vertex_offset), offsetof(MipOutputs, flags));
      return 0;
    }'''
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes == [32, 80, 180, 80, 104, 20, 32, 76, 48]
    assert MESH_DTYPE.fields["vertex_offset"][1] == 76


def test_create_fails_loudly_without_a_gpu(gpu_available):
    if gpu_available:
        pytest.skip("a GPU is present")
    import renderer_amd

    with pytest.raises(renderer_amd.MipError) as e:
        renderer_amd.InstancePipeline(max_instances=16, max_meshes=1)
    assert e.value.code == -2  # MIP_ERR_NO_DEVICE: there is no CPU backend to fall back to


def test_null_and_bad_arguments_are_status_codes_not_crashes():
    import renderer_amd
    from renderer_amd import _lib

    lib = renderer_amd.load_library()
    ctx = C.c_void_p()
    assert lib.mip_create(None, C.byref(ctx)) == -1
    cfg = _lib.MipConfig()
    cfg.struct_size = 8  # wrong ABI size
    assert lib.mip_create(C.byref(cfg), C.byref(ctx)) == -1 and not ctx.value
    lib.mip_destroy(None)  # no-op
    assert lib.mip_run(None, None, None) == -1
    assert lib.mip_wait(None) == -1
    assert lib.mip_set_mesh_table(None, None, 0) == -1
    assert lib.mip_instance_count(None) == 0
    assert lib.mip_last_error(None) == b"null context"


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under renderer_amd/ or include/ may import,
    include or link it."""
    offenders = []
    for base in ("renderer_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            if os.sep + "lib" in dirpath:
                continue
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"import oracle|from oracle|mip_oracle|liboracle|libmip_oracle", text):
                        offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders


def test_scene_generator_is_deterministic_and_shardable():
    from renderer_amd import scene

    a = scene.make_scene(3, n=5000)
    b = scene.make_scene(3, n=2000, first=3000)
    assert np.array_equal(a["pos"][3000:], b["pos"]) and np.array_equal(a["rot"][3000:], b["rot"])
    assert np.array_equal(a["mesh_id"][3000:], b["mesh_id"]) and a["mesh_id"].max() < len(a["meshes"])
    # splitmix64 known answers (seed 0): first outputs of the reference implementation
    z = scene.splitmix64(0, 0, 3)
    assert [int(v) for v in z] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]
    norms = np.linalg.norm(a["rot"].astype(np.float64), axis=1)
    assert np.allclose(norms, 1.0, atol=1e-6)


def test_rust_binding_sizes_and_symbols_follow_the_header():
    """integration/rust/mip-sys/src/lib.rs cannot be compiled here; keep its layout guards and its
    extern block in step with the C header mechanically."""
    from renderer_amd import _lib

    text = open(os.path.join(ROOT, "integration", "rust", "mip-sys", "src", "lib.rs")).read()
    guards = dict(re.findall(r"size_of::<(\w+)>\(\) == (\d+)", text))
    want = {"MipConfig": C.sizeof(_lib.MipConfig), "MipMesh": 80, "MipFrame": C.sizeof(_lib.MipFrame),
            "MipOutputs": C.sizeof(_lib.MipOutputs), "MipDrawIndexedIndirectCommand": 20}
    assert {k: int(v) for k, v in guards.items()} == want
    rust_fns = set(re.findall(r"pub fn (mip_[a-z_]+)\(", text))
    assert rust_fns <= set(_declared_functions())
    assert {"mip_create", "mip_run", "mip_run_many", "mip_set_geometry", "mip_merge_draw_lists",
            "mip_import_external_fd", "mip_release_external", "mip_merge_wire_lists", "mip_merge_wire_lists_packed", "mip_wire_index_bits", "mip_import_external_semaphore_fd",
            "mip_wait_external", "mip_signal_external", "mip_release_external_semaphore"} <= rust_fns
    # argument counts of the calls whose signature changed with ABI 2 (header vs Rust extern block)
    header = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for fn in ("mip_run_many", "mip_merge_draw_lists", "mip_import_external_fd", "mip_merge_wire_lists", "mip_import_external_semaphore_fd",
               "mip_wait_external", "mip_signal_external"):
        c_args = re.search(fn + r"\s*\(([^;]*?)\)\s*;", header, flags=re.S).group(1).count(",") + 1
        r_args = re.search(r"pub fn " + fn + r"\((.*?)\)\s*->", text, flags=re.S).group(1).count(",") + 1
        assert c_args == r_args, (fn, c_args, r_args)
    # every call the shim text makes is declared in the binding
    shim = open(os.path.join(ROOT, "integration", "rust", "instance_pipeline.rs")).read()
    assert set(re.findall(r"mip_sys::(mip_[a-z_]+)\(", shim)) <= rust_fns


def test_no_instruction_touches_the_registers_of_the_pending_started_swap(tmp_path):
    """The frame kernel issues its STARTED swap from inline assembly and asks for the answer hundreds of instructions later
    (instance_kernel.hpp, mark_tile_started_issue / _answer): the compiler does not know the destination registers are pending.
    The gfx950 ISA of every kernel that carries the swap, generated with the product's flags (make asm), is checked: nothing
    reads, writes or spills them in between."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-s", "-C", os.path.join(root, "renderer_amd", "csrc"), "asm-frame", f"ASM_DIR={tmp_path}"], check=True,
                   capture_output=True, text=True, timeout=900)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_pending_swap.py"), str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 violations" in r.stdout and not r.stdout.startswith("0 pending"), r.stdout
