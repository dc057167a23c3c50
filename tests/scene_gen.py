"""Generated glTF scenes with real geometry (vertices + triangles), for the tests that rasterise depth from the
visible meshlets instead of taking a synthetic depth image.  Test infrastructure only."""
import json

import numpy as np


def _grid(n: int, wave: float):
    xs, ys = np.meshgrid(np.linspace(-1, 1, n), np.linspace(-1, 1, n))
    pos = np.stack([xs, ys, wave * np.sin(3 * xs) * np.cos(2 * ys)], -1).reshape(-1, 3).astype(np.float32)
    idx = []
    for y in range(n - 1):
        for x in range(n - 1):
            a = y * n + x
            idx += [a, a + 1, a + n, a + 1, a + n + 1, a + n]          # counter-clockwise seen from +z
    return pos, np.array(idx, np.uint16)


def _sphere(seg: int, rings: int):
    pos, idx = [], []
    for r in range(rings + 1):
        t = np.pi * r / rings
        for s in range(seg):
            p = 2 * np.pi * s / seg
            pos.append([np.sin(t) * np.cos(p), np.cos(t), np.sin(t) * np.sin(p)])
    for r in range(rings):
        for s in range(seg):
            a, b = r * seg + s, r * seg + (s + 1) % seg
            c, d = a + seg, b + seg
            idx += [a, b, c, b, d, c]                                  # outward facing
    return np.array(pos, np.float32), np.array(idx, np.uint16)


def write_city_gltf(tmp_path, num_spheres: int = 120, num_cutouts: int = 16, seed: int = 7) -> str:
    """A wall (24x24-vertex wavy grid, ~20 meshlets) close to the camera, a field of spheres behind and beside it, and
    a few alpha-masked quads.  Camera node at the origin looking down -z: the wall hides part of the field, so the
    two-phase loop has work to do once the depth is the frame's own."""
    rng = np.random.default_rng(seed)
    meshes = [_grid(24, 0.05), _sphere(16, 10), _grid(2, 0.0)]
    blob, views, accessors = b"", [], []
    for pos, idx in meshes:
        for arr, ctype, typ in ((pos, 5126, "VEC3"), (idx, 5123, "SCALAR")):
            raw = arr.tobytes()
            views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": len(raw)})
            accessors.append({"bufferView": len(views) - 1, "componentType": ctype, "count": int(len(arr)), "type": typ})
            blob += raw + b"\0" * (-len(raw) % 4)
    nodes = [{"camera": 0}, {"mesh": 0, "translation": [0.3, 0.0, -8.0], "scale": [3.0, 2.2, 1.0]}]
    for i in range(num_spheres):
        s = float(rng.uniform(0.4, 1.4))
        nodes.append({"mesh": 1, "translation": [float(rng.uniform(-12, 12)), float(rng.uniform(-6, 6)), float(rng.uniform(-40, -11))],
                      "scale": [s, s * float(rng.uniform(0.7, 1.3)), s]})
    for i in range(num_cutouts):
        nodes.append({"mesh": 2, "translation": [float(rng.uniform(-8, 8)), float(rng.uniform(-4, 4)), float(rng.uniform(-30, -5))],
                      "rotation": [0.0, float(np.sin(0.3)), 0.0, float(np.cos(0.3))]})
    g = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": list(range(len(nodes)))}], "nodes": nodes,
         "cameras": [{"type": "perspective", "perspective": {"yfov": 0.7, "znear": 0.1, "aspectRatio": 16 / 9}}],
         "materials": [{"name": "opaque"}, {"name": "cutout", "alphaMode": "MASK"}],
         "meshes": [{"primitives": [{"attributes": {"POSITION": 0}, "indices": 1, "material": 0}]},
                    {"primitives": [{"attributes": {"POSITION": 2}, "indices": 3, "material": 0}]},
                    {"primitives": [{"attributes": {"POSITION": 4}, "indices": 5, "material": 1}]}],
         "accessors": accessors, "bufferViews": views, "buffers": [{"byteLength": len(blob), "uri": "city.bin"}]}
    (tmp_path / "city.bin").write_bytes(blob)
    (tmp_path / "city.gltf").write_text(json.dumps(g))
    return str(tmp_path / "city.gltf")


def all_meshlets_visible(scene):
    """Records + visible list that draw every LOD-0 meshlet of every instance (what the cull emits with all tests off)."""
    from toyrenderer_amd import interop as I
    records, visible = [], []
    for i, inst in enumerate(scene.instances):
        n = int(scene.meshData[int(inst["m_MeshDataIdx"])]["m_MeshLODDatas"]["m_NumMeshlets"][0])
        for off in range(0, n, 32):
            g = len(records)
            records.append((i, 0, off))
            visible += [(g << 5) | lane for lane in range(min(32, n - off))]
    rec = np.zeros(len(records), I.MeshletAmplificationData)
    for j, (i, lod, off) in enumerate(records):
        rec[j] = (i, lod, off)
    return rec, np.array(visible, np.uint32)
