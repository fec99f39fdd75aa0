"""Mutation fuzz of the glTF / GLB extractor (renderer_amd/host/gltf_scene.cpp: a hand-written JSON, base64 and GLB
reader of untrusted files) under AddressSanitizer + UndefinedBehaviorSanitizer — `make -C renderer_amd/host asan`.
Every mutated file must end in a clean exit: 0 (still a valid scene) or 3 (reported as malformed); never a sanitizer
report, a crash, or a hang. Host code only: no GPU sanitizer exists on this pool."""
import copy
import json
import os
import random
import struct
import subprocess
import tempfile

import pytest

import gltf_fixture

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN = os.path.join(ROOT, "renderer_amd", "lib", "mip_gltf_extract_asan")
HOSTILE_NUMBERS = [-1, -0.5, 0.5, 1e308, -1e308, 4294967295, 4294967296, 1 << 40, (1 << 40) - 1, 1 << 62, 1e18, 18446744073709551615,
                   2147483648, 65536, 3, 0]
BACKSLASH = chr(92)


@pytest.fixture(scope="module")
def asan_binary():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "renderer_amd", "host"), "-s", "asan"])
    return ASAN


def _run(binary, path):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1:max_allocation_size_mb=2048",
               UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([binary, path, path + ".bin"], capture_output=True, text=True, errors="replace", timeout=60, env=env)
    report = "AddressSanitizer" in r.stderr or "runtime error" in r.stderr or "LeakSanitizer" in r.stderr
    assert r.returncode in (0, 3) and not report, f"{path}: exit {r.returncode}\n{r.stderr[-3000:]}"
    return r.returncode


def _paths(node, prefix=()):
    """Every (path, value) of numbers and strings in a JSON tree."""
    if isinstance(node, dict):
        for k, v in node.items():
            yield from _paths(v, prefix + (k,))
    elif isinstance(node, list):
        for k, v in enumerate(node):
            yield from _paths(v, prefix + (k,))
    elif isinstance(node, (int, float, str)) and not isinstance(node, bool):
        yield prefix, node


def _set(tree, path, value):
    for k in path[:-1]:
        tree = tree[k]
    tree[path[-1]] = value


def test_structured_mutations_of_every_number_and_key(asan_binary):
    """Each numeric field in turn replaced by hostile values (negative, fractional, 2^32, 2^40, 1e308 ...),
    each string by junk, each value by another JSON kind; plus hand-made graph attacks."""
    gltf, _, _ = gltf_fixture.build(embed=True)
    rng = random.Random(1234)
    cases = []
    leaves = [(p, v) for p, v in _paths(gltf) if not (len(p) >= 3 and p[0] == "buffers" and p[-1] == "uri")]
    for path, value in leaves:
        if isinstance(value, str):
            cases.append((path, rng.choice(["", "VEC9", " ", "A" * 300, 7])))
        else:
            for h in rng.sample(HOSTILE_NUMBERS, 3):
                cases.append((path, h))
            cases.append((path, rng.choice(["x", [], {}, None, True])))
    with tempfile.TemporaryDirectory() as d:
        ok = bad = 0
        for k, (path, value) in enumerate(cases):
            g = copy.deepcopy(gltf)
            _set(g, path, value)
            f = os.path.join(d, f"m{k}.gltf")
            json.dump(g, open(f, "w"))
            rc = _run(asan_binary, f)
            ok += rc == 0
            bad += rc == 3
            os.remove(f)
            if os.path.exists(f + ".bin"):
                os.remove(f + ".bin")
        assert ok > 20 and bad > 100, (ok, bad)  # the mutations bite, and harmless ones still load

        def write(name, g):
            f = os.path.join(d, name)
            json.dump(g, open(f, "w"))
            return f

        # graph attacks
        g = copy.deepcopy(gltf)
        g["nodes"][3]["children"] = [0]                                                 # cycle
        assert _run(asan_binary, write("cycle.gltf", g)) == 3
        g = copy.deepcopy(gltf)
        g["nodes"][0]["children"] = [0]                                                 # self reference
        assert _run(asan_binary, write("self.gltf", g)) == 3
        g = copy.deepcopy(gltf)                                                          # DAG: 2^40 paths under the depth limit
        g["nodes"] = [{"children": [k + 1, k + 1]} for k in range(40)] + [{"name": "leaf"}]
        g["scenes"] = [{"nodes": [0]}]
        assert _run(asan_binary, write("fanout.gltf", g)) == 3
        g = copy.deepcopy(gltf)
        g["scenes"] = [{"nodes": [99]}]
        assert _run(asan_binary, write("root_oob.gltf", g)) == 3
        # an index one past the primitive's positions: rejected (the device gathers vertices unchecked)
        g = copy.deepcopy(gltf)
        g["accessors"][0]["count"] = 119                                                 # 120 positions -> 119: index 119 is now out of range
        assert _run(asan_binary, write("index_oob.gltf", g)) == 3
        # deep nesting must not overflow the parser's stack
        for text in ("[" * 200000, '{"a":' * 100000, '{"nodes": ' + "[" * 5000 + "]" * 5000 + "}"):
            f = os.path.join(d, "deep.gltf")
            open(f, "w").write(text)
            assert _run(asan_binary, f) == 3
        # strings, escapes and numbers cut short
        for text in ('{"a": "' + BACKSLASH, '{"a": "' + BACKSLASH + "u12", '{"a": "abc', '{"a": 1e', '{"a": -', '{"a"', "{", ""):
            f = os.path.join(d, "cut.gltf")
            open(f, "w").write(text)
            assert _run(asan_binary, f) == 3


def test_byte_level_mutations_of_gltf_and_glb(asan_binary):
    """Truncations along the whole file, random byte flips and random splices of the text and the binary container."""
    rng = random.Random(99)
    with tempfile.TemporaryDirectory() as d:
        src_json = os.path.join(d, "a.gltf")
        src_glb = os.path.join(d, "a.glb")
        gltf_fixture.write_gltf(src_json)
        gltf_fixture.write_glb(src_glb)
        assert _run(asan_binary, src_json) == 0 and _run(asan_binary, src_glb) == 0
        for src, ext in ((src_json, ".gltf"), (src_glb, ".glb")):
            raw = open(src, "rb").read()
            outcomes = {0: 0, 3: 0}
            variants = [raw[:k] for k in range(0, len(raw), max(97, len(raw) // 60))]
            for _ in range(120):
                b = bytearray(raw)
                for _ in range(rng.choice((1, 1, 2, 8))):
                    b[rng.randrange(len(b))] = rng.randrange(256)
                variants.append(bytes(b))
            for _ in range(30):  # splice: copy a random span over another place
                b = bytearray(raw)
                a0, ln, dst = rng.randrange(len(b)), rng.randrange(1, 64), rng.randrange(len(b))
                b[dst:dst + ln] = b[a0:a0 + ln]
                variants.append(bytes(b))
            if ext == ".glb":  # header fields: lengths that lie
                for off in (8, 12, 12 + 8 + struct.unpack_from("<I", raw, 12)[0]):
                    for val in (0, 1, 0x7FFFFFFF, 0xFFFFFFFF, len(raw), len(raw) + 1):
                        b = bytearray(raw)
                        struct.pack_into("<I", b, off, val)
                        variants.append(bytes(b))
            for k, v in enumerate(variants):
                f = os.path.join(d, f"v{k}{ext}")
                open(f, "wb").write(v)
                outcomes[_run(asan_binary, f)] += 1
                os.remove(f)
                if os.path.exists(f + ".bin"):
                    os.remove(f + ".bin")
            assert outcomes[3] > 40, outcomes
