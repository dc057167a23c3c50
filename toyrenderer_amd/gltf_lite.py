"""glTF-lite scene ingestion for the visibility path: what SceneLoader + Mesh::Initialize produce for the cull
(reference: source/SceneLoading.cpp:203-224,812-866, source/Visual.cpp:302-507, source/Scene.cpp:282-362), without
the third-party libraries the reference uses for it (cgltf, meshoptimizer, DirectXMath: all absent from the
reference snapshot -- SURVEY.md 8(c)).  The partition of a primitive into meshlets and the bounding volumes are
therefore THIS build's own (restating the published algorithms), not bit-comparable with the reference's; what is
kept verbatim is everything the cull path depends on: the limits (64 vertices / 96 triangles per meshlet,
ShaderInterop.h:19-20), the wire formats, the cone packing (Visual.cpp:421-431: axis (a+1)/2*255 truncated, cutoff =
2 * cone_cutoff_s8), one `Primitive` per glTF primitive of every mesh node (SceneLoading.cpp:853-862), world matrices
through the node-transform pass, opaque / alpha-mask id lists, and the LOD chain CONTRACT of Mesh::Initialize
(Visual.cpp:326-491: up to 8 LODs, each simplified from the previous one towards 65 % of its indices under a relative
error bound of 0.1, the stop rules, the error accumulation `max(1.5 * previous, result)` scaled by the mesh extent) --
with this build's own edge-collapse simplifier (`simplify`) in the place of meshopt_simplifyWithAttributes.

Supported: .gltf + external .bin (or a dict + bytes), float32 POSITION, u8/u16/u32 indices or none, node TRS or
matrix-free hierarchies, perspective cameras, alphaMode MASK.
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass

import numpy as np

from . import interop as I

MAX_MESHLET_VERTICES = 64      # ShaderInterop.h:19
MAX_MESHLET_TRIANGLES = 96     # ShaderInterop.h:20

_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NUM = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


@dataclass
class Camera:
    name: str
    position: tuple
    orientation: tuple        # quaternion xyzw (world)
    yfov: float
    znear: float
    aspect: float


@dataclass
class LoadedScene:
    instances: np.ndarray      # BasePassInstanceConstants (world matrices filled by update_instance_consts)
    meshData: np.ndarray
    meshlets: np.ndarray
    opaqueIds: np.ndarray
    alphaMaskIds: np.ndarray
    nodes: np.ndarray          # NodeLocalTransform
    primToNode: np.ndarray     # u32 per primitive
    cameras: list
    vertices: np.ndarray           # RawVertexFormat: the global vertex buffer (positions; packed normals when present)
    meshletVertexIds: np.ndarray   # per meshlet vertex: index into the global vertex buffer
    meshletTriangles: np.ndarray   # packed a | b << 8 | c << 16 (Visual.cpp:396-403)

    def as_oracle(self) -> dict:
        return dict(instances=self.instances, meshData=self.meshData, meshlets=self.meshlets,
                    opaqueIds=self.opaqueIds, alphaMaskIds=self.alphaMaskIds)


# ----------------------------------------------------------------------------------------------- glTF access
def _accessor(g: dict, blobs: list, idx: int) -> np.ndarray:
    a = g["accessors"][idx]
    bv = g["bufferViews"][a["bufferView"]]
    dt = np.dtype(_COMPONENT[a["componentType"]])
    n = _NUM[a["type"]]
    off = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
    stride = bv.get("byteStride", 0) or dt.itemsize * n
    buf = blobs[bv["buffer"]]
    if stride == dt.itemsize * n:
        arr = np.frombuffer(buf, dt, a["count"] * n, off).reshape(a["count"], n)
    else:
        arr = np.stack([np.frombuffer(buf, dt, n, off + i * stride) for i in range(a["count"])])
    return arr.copy()


def _quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return (aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz)


def _quat_rotate(q, v):
    x, y, z, w = q
    vx, vy, vz = v
    tx, ty, tz = 2 * (y * vz - z * vy), 2 * (z * vx - x * vz), 2 * (x * vy - y * vx)
    return (vx + w * tx + (y * tz - z * ty), vy + w * ty + (z * tx - x * tz), vz + w * tz + (x * ty - y * tx))


# ----------------------------------------------------------------------------------------------- bounding volumes
def bounding_sphere(points: np.ndarray) -> np.ndarray:
    """Ritter's sphere the way meshoptimizer / DirectXMath seed it: the most separated pair of the six axis-extreme
    points gives the first sphere, then every point outside grows it.  float32 result (centre xyz, radius)."""
    p = np.asarray(points, np.float64).reshape(-1, 3)
    if len(p) == 0:
        return np.zeros(4, np.float32)
    best, pair = -1.0, (0, 0)
    for ax in range(3):
        lo, hi = int(np.argmin(p[:, ax])), int(np.argmax(p[:, ax]))
        d = float(((p[hi] - p[lo]) ** 2).sum())
        if d > best:
            best, pair = d, (lo, hi)
    c = (p[pair[0]] + p[pair[1]]) * 0.5
    r = math.sqrt(best) * 0.5
    for q in p:
        d = math.sqrt(float(((q - c) ** 2).sum()))
        if d > r:
            k = (d - r) / (2 * d) if d > 0 else 0.0
            c = c + (q - c) * k
            r = (r + d) * 0.5
    return np.array([c[0], c[1], c[2], r], np.float32)


def meshlet_cone(points: np.ndarray, tris: np.ndarray):
    """Normal cone of a meshlet (the published meshopt_computeMeshletBounds construction): axis = centre of the
    bounding sphere of the triangle normals, cutoff = sqrt(1 - min dot^2), degenerate (no culling) when the
    normals span more than a hemisphere.  Returns (axis float32[3], cone_cutoff_s8 in [0, 127])."""
    p = np.asarray(points, np.float64)
    a, b, c = p[tris[:, 0]], p[tris[:, 1]], p[tris[:, 2]]
    n = np.cross(b - a, c - a)
    ln = np.linalg.norm(n, axis=1)
    n = n[ln > 0] / ln[ln > 0, None]
    if len(n) == 0:
        return np.zeros(3, np.float32), 127
    s = bounding_sphere(n).astype(np.float64)
    axis = s[:3]
    la = float(np.linalg.norm(axis))
    axis = axis / la if la > 0 else np.array([1.0, 0.0, 0.0])
    mindp = float((n @ axis).min())
    if mindp <= 0.1:
        return np.zeros(3, np.float32), 127                   # cutoff 1: the cone test never culls
    cutoff = math.sqrt(max(0.0, 1.0 - mindp * mindp))
    s8 = min(127, int(cutoff * 127.0) + 1)                     # rounded up: conservative
    return axis.astype(np.float32), s8


def pack_cone(axis: np.ndarray, cutoff_s8: int) -> int:
    """Visual.cpp:421-431: u8 axis = (a + 1) * 0.5 * 255 truncated (float32 arithmetic), cutoff byte = 2 * cutoff_s8."""
    f = np.float32
    px, py, pz = (int((f(v) + f(1.0)) * f(0.5) * f(255.0)) for v in axis)
    assert 0 <= px <= 255 and 0 <= py <= 255 and 0 <= pz <= 255 and 0 <= cutoff_s8 <= 127
    return px | (py << 8) | (pz << 16) | ((cutoff_s8 * 2) << 24)


# ----------------------------------------------------------------------------------------------- meshlets
def build_meshlets(indices: np.ndarray):
    """Greedy partition of a triangle list, in index order, into meshlets of at most 64 unique vertices and 96
    triangles.  Returns a list of (vertex ids [<=64], local triangles uint8 [n,3])."""
    tris = np.asarray(indices, np.int64).reshape(-1, 3)
    out, verts, local, cur = [], [], {}, []
    for t in tris:
        new = [v for v in dict.fromkeys(int(x) for x in t) if v not in local]
        if cur and (len(verts) + len(new) > MAX_MESHLET_VERTICES or len(cur) + 1 > MAX_MESHLET_TRIANGLES):
            out.append((np.array(verts, np.uint32), np.array(cur, np.uint8).reshape(-1, 3)))
            verts, local, cur = [], {}, []
            new = list(dict.fromkeys(int(x) for x in t))
        for v in new:
            local[v] = len(verts)
            verts.append(v)
        cur.append([local[int(x)] for x in t])
    if cur:
        out.append((np.array(verts, np.uint32), np.array(cur, np.uint8).reshape(-1, 3)))
    return out


# ----------------------------------------------------------------------------------------------- LOD chain
def simplify_scale(positions: np.ndarray) -> float:
    """meshopt_simplifyScale (Visual.cpp:324): the factor between relative and absolute error = the largest extent of the
    bounding box."""
    p = np.asarray(positions, np.float64).reshape(-1, 3)
    return float((p.max(0) - p.min(0)).max()) if len(p) else 0.0


def simplify(positions: np.ndarray, indices: np.ndarray, target_index_count: int, target_error: float):
    """Edge-collapse simplification in the role of meshopt_simplifyWithAttributes(options = 0) at Visual.cpp:456-471: the
    result uses a SUBSET of the input vertices (a collapse moves one end of an edge onto the other), stops at
    `target_index_count` indices or when the cheapest remaining collapse would exceed `target_error` (relative to
    simplify_scale), never flips a triangle and never moves a boundary vertex off its boundary.  Cost of a collapse = the
    quadric error (area-weighted squared distance to the planes around the moved vertex, Garland-Heckbert) as an RMS
    distance.  Returns (indices uint32, result_error relative).  Own algorithm: same contract, other triangles than
    meshoptimizer would pick."""
    import heapq
    P = np.asarray(positions, np.float64).reshape(-1, 3)
    if len(indices) <= target_index_count:
        return np.asarray(indices, np.uint32).copy(), 0.0
    tris = [tuple(int(x) for x in t) for t in np.asarray(indices, np.int64).reshape(-1, 3)]
    tris = [t for t in tris if len(set(t)) == 3]
    scale = simplify_scale(P[np.unique(np.array(tris).ravel())]) if tris else 0.0
    if not tris or scale == 0.0:
        return np.asarray(indices, np.uint32).copy(), 0.0
    limit = (target_error * scale) ** 2
    # weld by position: collapses work on position classes so that attribute seams (duplicated positions) stay closed
    _, remap = np.unique(np.round(P / (scale * 1e-7)).astype(np.int64), axis=0, return_inverse=True)
    remap = remap.ravel()
    Q = {}                                             # class -> [4x4 quadric, weight]
    vt = {}                                            # class -> set of triangle ids
    alive = {}
    tri_cls = []

    def plane_quadric(a, b, c):
        n = np.cross(b - a, c - a)
        area = float(np.linalg.norm(n))
        if area == 0.0:
            return None, 0.0
        n = n / area
        pl = np.append(n, -float(n @ a))
        return np.outer(pl, pl) * area, area

    for ti, t in enumerate(tris):
        c = tuple(int(remap[v]) for v in t)
        tri_cls.append(c)
        if len(set(c)) < 3:
            continue
        alive[ti] = True
        q, w = plane_quadric(P[t[0]], P[t[1]], P[t[2]])
        for v in c:
            vt.setdefault(v, set()).add(ti)
            if q is not None:
                e = Q.setdefault(v, [np.zeros((4, 4)), 0.0])
                e[0] += q; e[1] += w
    rep = {}                                           # class -> a representative input vertex (position)
    for v, c in enumerate(remap):
        rep.setdefault(int(c), v)
    # boundary edges (one adjacent triangle) get a perpendicular plane so that borders keep their shape
    edge_count = {}
    for ti in alive:
        c = tri_cls[ti]
        for i in range(3):
            e = (min(c[i], c[(i + 1) % 3]), max(c[i], c[(i + 1) % 3]))
            edge_count.setdefault(e, []).append(ti)
    border = set()
    for (a, b), ts in edge_count.items():
        if len(ts) == 1:
            border.update((a, b))
            c = tri_cls[ts[0]]
            pa, pb = P[rep[a]], P[rep[b]]
            other = P[rep[[x for x in c if x not in (a, b)][0]]]
            n = np.cross(pb - pa, other - pa)
            d = np.cross(n, pb - pa)
            ld = float(np.linalg.norm(d))
            if ld > 0:
                d /= ld
                w = float(np.linalg.norm(pb - pa)) ** 2 * 10.0
                pl = np.append(d, -float(d @ pa))
                for v in (a, b):
                    e = Q.setdefault(v, [np.zeros((4, 4)), 0.0])
                    e[0] += np.outer(pl, pl) * w; e[1] += w
    version = {v: 0 for v in vt}

    def cost(u, v):                                    # move class u onto class v
        q, w = Q.get(u, (None, 0.0))
        if q is None or w <= 0.0:
            return 0.0
        x = np.append(P[rep[v]], 1.0)
        return max(float(x @ q @ x) / w, 0.0)

    heap = []

    def push(u):
        nbrs = set()
        for ti in vt.get(u, ()):
            nbrs.update(tri_cls[ti])
        nbrs.discard(u)
        for v in nbrs:
            if u in border and v not in border:
                continue                               # a boundary vertex only slides along the boundary
            heapq.heappush(heap, (cost(u, v), u, v, version[u], version[v]))
    for u in list(vt):
        push(u)
    n_idx = 3 * len(alive)
    worst = 0.0
    while heap and n_idx > target_index_count:
        c_, u, v, vu, vv = heapq.heappop(heap)
        if u not in version or v not in version or version[u] != vu or version[v] != vv:
            continue
        if c_ > limit:
            break
        # reject collapses that flip or degenerate a surviving triangle
        ok = True
        for ti in vt[u]:
            c = tri_cls[ti]
            if v in c:
                continue                               # disappears
            a, b, d = (P[rep[x]] for x in c)
            n0 = np.cross(b - a, d - a)
            a2, b2, d2 = (P[rep[v if x == u else x]] for x in c)
            n1 = np.cross(b2 - a2, d2 - a2)
            if float(n0 @ n1) <= 1e-12 * float(n0 @ n0):
                ok = False
                break
        if not ok:
            continue
        worst = max(worst, c_)
        touched = set()
        for ti in list(vt[u]):
            c = tri_cls[ti]
            if v in c:                                  # collapses to a line
                for x in c:
                    vt[x].discard(ti)
                    touched.add(x)
                del alive[ti]
                n_idx -= 3
            else:
                nc = tuple(v if x == u else x for x in c)
                tri_cls[ti] = nc
                vt[v].add(ti)
                touched.update(nc)
        Q.setdefault(v, [np.zeros((4, 4)), 0.0])
        if u in Q:
            Q[v][0] += Q[u][0]; Q[v][1] += Q[u][1]
        if u in border:
            border.add(v)
        del vt[u]; del version[u]
        touched.discard(u)
        for x in touched:
            if x in version:
                version[x] += 1
        for x in touched:
            if x in version:
                push(x)
    # a class is drawn with its representative vertex; keep the input triangle order
    out = []
    for ti in sorted(alive):
        out.extend(rep[x] for x in tri_cls[ti])
    return np.array(out, np.uint32), math.sqrt(worst) / scale


def build_lod_chain(positions: np.ndarray, indices: np.ndarray, simplifier=simplify):
    """The LOD loop of Mesh::Initialize (Visual.cpp:326-491): returns [(indices, error)] for LOD 0..n-1, n <= 8, with
    error = accumulated relative error * simplify_scale (the value MeshLODData::m_Error carries, :345)."""
    # :336-338: the reference's constants are FLOATs (0.65f, 0.85f) multiplied into double(size): 60 indices -> 38.99999
    # -> 38 -> target 36 (with the double 0.65 it would be 39 -> 39)
    kTargetError = float(np.float32(0.1))                                                         # :334 0.1f
    kTargetIndexCountPercentage, kMinIndexReductionPercentage = float(np.float32(0.65)), float(np.float32(0.85))
    scale = np.float32(simplify_scale(positions))                                                 # :324
    lod_indices = np.asarray(indices, np.uint32).copy()
    lod_error = np.float32(0.0)
    out = []
    for _ in range(I.kMaxNumMeshLODs):                                                            # :329
        out.append((lod_indices, float(np.float32(lod_error * scale))))                           # :343-345
        target = (int(float(len(lod_indices)) * kTargetIndexCountPercentage) // 3) * 3            # :454
        simplified, result_error = simplifier(positions, lod_indices, target, kTargetError)      # :455-471
        assert len(simplified) <= len(lod_indices)
        if len(simplified) == len(lod_indices) or len(simplified) == 0:                           # :475-479 error bound reached
            break
        if len(simplified) >= int(float(len(lod_indices)) * kMinIndexReductionPercentage):        # :481-485 too close to the last one
            break
        lod_indices = simplified                                                                  # :487
        lod_error = max(np.float32(lod_error * np.float32(1.5)), np.float32(result_error))        # :488 errors accumulate
    return out


# ----------------------------------------------------------------------------------------------- loader
def load(path_or_gltf, blobs=None, lods: bool = True) -> LoadedScene:
    """lods=False: LOD 0 only (no simplification)."""
    if isinstance(path_or_gltf, dict):
        g = path_or_gltf
        blobs = list(blobs or [])
    else:
        with open(path_or_gltf) as f:
            g = json.load(f)
        base = os.path.dirname(os.path.abspath(path_or_gltf))
        blobs = []
        for b in g.get("buffers", []):
            with open(os.path.join(base, b["uri"]), "rb") as f:
                blobs.append(f.read())

    materials = g.get("materials", [])
    mesh_prims = []                                   # per glTF mesh: [(meshIdx, alphaMask)]
    md_rows, ml_rows, mvid, mtri, vtx = [], [], [], [], []
    vertex_base = 0
    for mesh in g.get("meshes", []):
        prims = []
        for prim in mesh["primitives"]:
            if prim.get("mode", 4) != 4:
                continue                               # triangles only
            pos = _accessor(g, blobs, prim["attributes"]["POSITION"]).astype(np.float32)
            idx = (_accessor(g, blobs, prim["indices"]).reshape(-1).astype(np.uint32) if "indices" in prim
                   else np.arange(len(pos), dtype=np.uint32))
            idx = idx[:len(idx) // 3 * 3]
            row = np.zeros((), I.MeshData)
            row["m_BoundingSphere"] = bounding_sphere(pos)              # Visual.cpp:321
            row["m_GlobalVertexBufferIdx"] = vertex_base
            chain = build_lod_chain(pos, idx) if lods else [(idx, 0.0)]
            row["m_NumLODs"] = len(chain)
            meshlets = []
            for lod, (lod_idx, lod_err) in enumerate(chain):                             # Visual.cpp:341-431 per LOD
                lod_meshlets = build_meshlets(lod_idx)
                row["m_MeshLODDatas"]["m_MeshletDataBufferIdx"][lod] = len(ml_rows) + len(meshlets)
                row["m_MeshLODDatas"]["m_NumMeshlets"][lod] = len(lod_meshlets)
                row["m_MeshLODDatas"]["m_Error"][lod] = lod_err
                meshlets.extend(lod_meshlets)
            for vids, tri in meshlets:
                m = np.zeros((), I.MeshletData)
                m["m_BoundingSphere"] = bounding_sphere(pos[vids])
                axis, s8 = meshlet_cone(pos[vids], tri.astype(np.int64))
                m["m_ConeAxisAndCutoff"] = pack_cone(axis, s8)
                m["m_MeshletVertexIDsBufferIdx"] = len(mvid)
                m["m_MeshletIndexIDsBufferIdx"] = len(mtri)
                m["m_VertexAndTriangleCount"] = len(vids) | (len(tri) << 8)             # Visual.cpp:418
                mvid.extend((vids + vertex_base).tolist())
                mtri.extend((tri[:, 0].astype(np.uint32) | (tri[:, 1].astype(np.uint32) << 8) | (tri[:, 2].astype(np.uint32) << 16)).tolist())
                ml_rows.append(m)
            mat = materials[prim["material"]] if "material" in prim and prim["material"] < len(materials) else {}
            vrows = np.zeros(len(pos), I.RawVertexFormat)
            vrows["m_Position"] = pos
            if "NORMAL" in prim["attributes"]:                                       # R10G10B10A2: x bits 20-29, y 10-19, z 0-9 (Visual.cpp:440-449)
                nrm = np.clip(_accessor(g, blobs, prim["attributes"]["NORMAL"]).astype(np.float32), -1, 1)
                q = np.round((nrm * 0.5 + 0.5) * 1023.0).astype(np.uint32)
                vrows["m_PackedNormal"] = (q[:, 0] << 20) | (q[:, 1] << 10) | q[:, 2]
            vtx.append(vrows)
            prims.append((len(md_rows), mat.get("alphaMode", "OPAQUE") == "MASK"))
            md_rows.append(row)
            vertex_base += len(pos)
        mesh_prims.append(prims)

    # nodes: local TRS + parent (SceneLoading.cpp:812-866); world transform of cameras by walking the parents
    nodes_in = g.get("nodes", [])
    parent = [0xFFFFFFFF] * len(nodes_in)
    for i, n in enumerate(nodes_in):
        for c in n.get("children", []):
            parent[c] = i
    nodes = np.zeros(len(nodes_in), I.NodeLocalTransform)
    for i, n in enumerate(nodes_in):
        assert "matrix" not in n, "glTF-lite: node matrices are not supported (TRS only)"
        nodes[i]["m_ParentNodeIdx"] = parent[i]
        nodes[i]["m_Position"] = n.get("translation", (0, 0, 0))
        nodes[i]["m_Rotation"] = n.get("rotation", (0, 0, 0, 1))
        nodes[i]["m_Scale"] = n.get("scale", (1, 1, 1))

    def world_of(i):
        pos, rot = (0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0)
        chain = []
        while i != 0xFFFFFFFF:
            chain.append(i)
            i = parent[i]
        for j in reversed(chain):                    # root first
            n = nodes_in[j]
            t, r, s = n.get("translation", (0, 0, 0)), n.get("rotation", (0, 0, 0, 1)), n.get("scale", (1, 1, 1))
            assert all(abs(x - 1) < 1e-6 for x in s) or "camera" not in nodes_in[chain[0]], "scaled camera chain"
            lp = _quat_rotate(rot, t)
            pos = (pos[0] + lp[0], pos[1] + lp[1], pos[2] + lp[2])
            rot = _quat_mul(rot, r)
        return pos, rot

    inst_rows, prim_to_node, opaque, alpha, cameras = [], [], [], [], []
    for i, n in enumerate(nodes_in):
        if "mesh" in n:
            for mesh_idx, is_mask in mesh_prims[n["mesh"]]:
                pid = len(inst_rows)
                r = np.zeros((), I.BasePassInstanceConstants)
                r["m_WorldMatrix"] = np.eye(4, dtype=np.float32)
                r["m_PrevWorldMatrix"] = np.eye(4, dtype=np.float32)
                r["m_MeshDataIdx"] = mesh_idx
                inst_rows.append(r)
                prim_to_node.append(i)
                (alpha if is_mask else opaque).append(pid)             # Scene.cpp:282-362 buckets
        if "camera" in n:
            c = g["cameras"][n["camera"]]
            assert c["type"] == "perspective"
            p, q = world_of(i)
            per = c["perspective"]
            cameras.append(Camera(n.get("name", "Un-named Camera"), p, q, float(per["yfov"]), float(per["znear"]), float(per.get("aspectRatio", 16 / 9))))

    def stack(rows, dt):
        return np.array(rows, dt) if rows else np.zeros(0, dt)
    return LoadedScene(stack(inst_rows, I.BasePassInstanceConstants), stack(md_rows, I.MeshData), stack(ml_rows, I.MeshletData),
                       np.array(opaque, np.uint32), np.array(alpha, np.uint32), nodes, np.array(prim_to_node, np.uint32), cameras,
                       np.concatenate(vtx) if vtx else np.zeros(0, I.RawVertexFormat), np.array(mvid, np.uint32), np.array(mtri, np.uint32))


def view_of(camera: Camera, render=(1920, 1080)):
    """synth.View of a loaded camera (RH, reverse-Z, infinite far plane: Scene.cpp:121-133, MathUtilities.cpp:3-38)."""
    from . import synth
    P = synth.perspective_rh_reverse_z_infinite(camera.yfov, render[0] / render[1], camera.znear)
    V = synth.world_to_view(camera.position, camera.orientation)
    return synth.View(V, V.copy(), P, float(np.float32(camera.znear)), render[0], render[1])
