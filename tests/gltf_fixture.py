"""Builds small glTF 2.0 / GLB files for the extractor tests (authored here from the glTF 2.0
specification; no asset of the Khronos sample repository is available offline)."""
import base64
import json
import struct

import numpy as np


def torus(w, h, scale=(1.0, 0.4, 1.0)):
    u = np.arange(w) / w * 2 * np.pi
    v = np.arange(h) / h * 2 * np.pi
    uu, vv = np.meshgrid(u, v, indexing="ij")
    pos = np.stack([(0.7 + 0.3 * np.cos(vv)) * np.cos(uu) * scale[0], 0.3 * np.sin(vv) * scale[1] / 0.3,
                    (0.7 + 0.3 * np.cos(vv)) * np.sin(uu) * scale[2]], -1).reshape(-1, 3).astype(np.float32)
    i, j = np.meshgrid(np.arange(w), np.arange(h), indexing="ij")
    a, b = (i * h + j).ravel(), (((i + 1) % w) * h + j).ravel()
    c, d = (((i + 1) % w) * h + (j + 1) % h).ravel(), (i * h + (j + 1) % h).ravel()
    tris = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)]).astype(np.uint32)
    return pos, tris.reshape(-1)


def build(embed=True):
    """Returns (gltf_dict, binary_blob). Two scenes; nodes exercise TRS, matrix, children,
    a primitive without base-colour texture, a primitive with < 100 positions, u16/u32 indices and
    a strided POSITION view."""
    big_pos, big_idx = torus(12, 10)           # 120 positions, 240 triangles (u16 indices)
    med_pos, med_idx = torus(16, 8, (2, 1, 1))  # 128 positions, 256 triangles (u32 indices, strided positions)
    small_pos, small_idx = torus(4, 4)          # 16 positions: skipped (< 100)
    blob = bytearray()
    views, accessors = [], []

    def add_view(data, stride=None):
        while len(blob) % 4:
            blob.append(0)
        off = len(blob)
        blob.extend(data)
        v = {"buffer": 0, "byteOffset": off, "byteLength": len(data)}
        if stride:
            v["byteStride"] = stride
        views.append(v)
        return len(views) - 1

    def add_positions(pos, strided=False):
        if strided:  # 12 bytes of position + 4 bytes of padding per vertex
            raw = b"".join(struct.pack("<3fI", *p, 0xDEADBEEF) for p in pos.tolist())
            view = add_view(raw, stride=16)
        else:
            view = add_view(pos.tobytes())
        accessors.append({"bufferView": view, "componentType": 5126, "count": len(pos), "type": "VEC3",
                          "min": pos.min(0).tolist(), "max": pos.max(0).tolist()})
        return len(accessors) - 1

    def add_indices(idx, ctype):
        data = idx.astype({5123: np.uint16, 5125: np.uint32, 5121: np.uint8}[ctype]).tobytes()
        view = add_view(data)
        accessors.append({"bufferView": view, "componentType": ctype, "count": len(idx), "type": "SCALAR"})
        return len(accessors) - 1

    p_big, i_big = add_positions(big_pos), add_indices(big_idx, 5123)
    p_med, i_med = add_positions(med_pos, strided=True), add_indices(med_idx, 5125)
    p_small, i_small = add_positions(small_pos), add_indices(small_idx, 5121)
    c, s = np.cos(0.5), np.sin(0.5)
    matrix = [2 * c, 0, -2 * s, 0, 0, 2, 0, 0, 2 * s, 0, 2 * c, 0, 7, 8, 9, 1]  # column-major: scale 2, rotation about +Y, translation
    gltf = {
        "asset": {"version": "2.0"},
        "scene": 0,
        "scenes": [{"nodes": [0]}, {"nodes": [4]}],
        "nodes": [
            {"name": "root", "children": [1, 2], "translation": [100, 100, 100]},                       # no mesh; its translation must NOT propagate
            {"name": "trs", "mesh": 0, "translation": [1, 2, 3], "rotation": [0, 0.6, 0, 0.8], "scale": [1.5, 9, 9], "children": [3]},
            {"name": "matrix", "mesh": 1, "matrix": matrix},
            {"name": "grandchild", "mesh": 2},                                                          # two primitives: untextured + tiny -> both skipped
            {"name": "second scene", "mesh": 0, "translation": [-5, 0, 20]},
        ],
        "meshes": [
            {"primitives": [{"attributes": {"POSITION": p_big}, "indices": i_big, "material": 0}]},
            {"primitives": [{"attributes": {"POSITION": p_med}, "indices": i_med, "material": 0}]},
            {"primitives": [{"attributes": {"POSITION": p_big}, "indices": i_big, "material": 1},
                            {"attributes": {"POSITION": p_small}, "indices": i_small, "material": 0}]},
        ],
        "materials": [
            {"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}, "normalTexture": {"index": 0}},
            {"pbrMetallicRoughness": {"baseColorFactor": [1, 0, 0, 1]}},
        ],
        "textures": [{"source": 0}],
        "images": [{"uri": "albedo.png"}],
        "accessors": accessors,
        "bufferViews": views,
        "buffers": [{"byteLength": len(blob)}],
    }
    if embed:
        gltf["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(bytes(blob)).decode()
    expected = dict(big=(big_pos, big_idx), med=(med_pos, med_idx), matrix_trs=((7, 8, 9), (0, np.sin(0.25), 0, np.cos(0.25)), 2.0))
    return gltf, bytes(blob), expected


def write_gltf(path):
    gltf, _, expected = build(embed=True)
    with open(path, "w") as f:
        json.dump(gltf, f)
    return expected


def write_glb(path):
    gltf, blob, expected = build(embed=False)
    js = json.dumps(gltf).encode()
    js += b" " * (-len(js) % 4)
    blob += b"\0" * (-len(blob) % 4)
    with open(path, "wb") as f:
        f.write(struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(blob)))
        f.write(struct.pack("<I4s", len(js), b"JSON") + js)
        f.write(struct.pack("<I4s", len(blob), b"BIN\0") + blob)
    return expected


def write_simple(path, primitives):
    """One scene, one textured node per (positions, indices) pair — for tests that only care about the geometry."""
    blob = bytearray()
    views, accessors, nodes, meshes = [], [], [], []
    for k, (pos, idx) in enumerate(primitives):
        pos = np.ascontiguousarray(pos, np.float32)
        idx = np.ascontiguousarray(idx, np.uint32)
        for data, acc in ((pos.tobytes(), {"componentType": 5126, "count": len(pos), "type": "VEC3", "min": pos.min(0).tolist(), "max": pos.max(0).tolist()}),
                          (idx.tobytes(), {"componentType": 5125, "count": len(idx), "type": "SCALAR"})):
            while len(blob) % 4:
                blob.append(0)
            views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": len(data)})
            blob.extend(data)
            accessors.append(dict(acc, bufferView=len(views) - 1))
        meshes.append({"primitives": [{"attributes": {"POSITION": 2 * k}, "indices": 2 * k + 1, "material": 0}]})
        nodes.append({"name": f"n{k}", "mesh": k, "translation": [float(3 * k), 0.0, 10.0]})
    gltf = {
        "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": list(range(len(nodes)))}], "nodes": nodes, "meshes": meshes,
        "materials": [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}, "normalTexture": {"index": 0}}],
        "textures": [{"source": 0}], "images": [{"uri": "albedo.png"}], "accessors": accessors, "bufferViews": views,
        "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(bytes(blob)).decode()}],
    }
    with open(path, "w") as f:
        json.dump(gltf, f)
