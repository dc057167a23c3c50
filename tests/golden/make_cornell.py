#!/usr/bin/env python3
"""Fixture generator: the reference's only in-tree asset (resources/cornell.gltf + cornell.bin; BASELINE.json
configs[0] "cornell.gltf ... scalar C++ frustum+cone cull on CPU") ingested with toyrenderer_amd/gltf_lite.py and
culled by the CPU oracle.  Run in the build container (the reference tree does not travel to the GPU box):

    python tests/golden/make_cornell.py [/root/reference/resources/cornell.gltf]

Writes tests/golden/cornell_scene.npz = the DERIVED scene in the path's wire formats (instances, mesh table,
meshlets, id lists, node transforms, camera) + the oracle's outputs for culling flags 5 (frustum + cone, the
config's CPU case) and 7.  No reference file is copied: the meshlet partition and bounds are this build's own
(meshoptimizer is absent from the reference snapshot), so the fixture pins THIS build's behaviour on the
reference's asset -- parity with the reference itself stays unpinned."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import pyoracle  # noqa: E402
from toyrenderer_amd import gltf_lite  # noqa: E402


def cull(scene, view, flags):
    inst = scene.instances.copy()
    pyoracle.update_instance_consts(scene.nodes, scene.primToNode, inst)
    sc = dict(scene.as_oracle()); sc["instances"] = inst
    hzb = pyoracle.HzbTexture(*view.hzb_dims)
    depth = np.zeros((view.renderH, view.renderW), np.float32) if flags & 2 else None
    return inst, pyoracle.frame(sc, view.as_dict(), hzb, depth, cullingFlags=flags, maxGroups=65535, record_capacity=65535)


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/resources/cornell.gltf"
    scene = gltf_lite.load(src)
    cam = scene.cameras[0]
    view = gltf_lite.view_of(cam, (1920, 1080))
    out = dict(instances=scene.instances, meshData=scene.meshData, meshlets=scene.meshlets, opaqueIds=scene.opaqueIds,
               alphaMaskIds=scene.alphaMaskIds, nodes=scene.nodes, primToNode=scene.primToNode,
               vertices=scene.vertices, meshletVertexIds=scene.meshletVertexIds, meshletTriangles=scene.meshletTriangles,
               camera=np.array([*cam.position, *cam.orientation, cam.yfov, cam.znear, cam.aspect], np.float64))
    for flags in (5, 7):
        inst, ref = cull(scene, view, flags)
        out[f"world_{flags}"] = inst["m_WorldMatrix"]
        for s in (0, 1):
            if ref.passRan[s]:
                out[f"f{flags}_s{s}_records"] = ref.records[s].view(np.uint32).reshape(-1, 3)
                out[f"f{flags}_s{s}_visMask"] = ref.visMask[s]
                out[f"f{flags}_s{s}_visibleList"] = ref.visibleList[s]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cornell_scene.npz"), **out)
    print("primitives", len(scene.instances), "meshlets", len(scene.meshlets), "cameras", len(scene.cameras))
    for flags in (5, 7):
        print("flags", flags, "records", len(out[f"f{flags}_s0_records"]), "visible", len(out[f"f{flags}_s0_visibleList"]))


if __name__ == "__main__":
    main()
