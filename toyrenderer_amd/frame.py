"""Python driver of one visibility frame over the C ABI (include/trhip.h).

Mirrors, call for call, the C++ host side (toyrenderer_amd/csrc/host/BasePassRenderers.cpp, itself
the re-authoring of the reference's source/BasePassRenderers.cpp:223-588): same buffers, same
clears, same constants, same dispatch order.  It exists so that tests and bench.py can drive the
kernels without the C++ RenderGraph layer in between (kernel-level parity tests, profiling); the
drop-in path for a C++ application is the host library.
"""
from __future__ import annotations

import numpy as np

from . import interop as I
from . import rhi
from .rhi import CB, PUSH, SAMPLER, SRV, TEX_SRV, TEX_UAV, UAV

SLOT_NAMES = ("early_opaque", "late_opaque", "early_alphamask", "late_alphamask")


def culling_frustum(view_to_clip: np.ndarray) -> np.ndarray:
    """BasePassRenderers.cpp:557-563 (host-side, float32; XMVector4Normalize = v / sqrt(dot4))."""
    P = np.asarray(view_to_clip, np.float32)
    fx = (P[:, 3] + P[:, 0]).astype(np.float32)
    fy = (P[:, 3] + P[:, 1]).astype(np.float32)

    def length4(v):
        acc = np.float32(v[0] * v[0])
        for i in (1, 2, 3):
            acc = I.fmaf(v[i], v[i], acc)
        return np.sqrt(acc, dtype=np.float32)
    lx, ly = length4(fx), length4(fy)
    return np.array([fx[0] / lx, fx[2] / lx, fy[1] / ly, fy[2] / ly], np.float32)


class GpuScene:
    """Scene buffers resident in HBM (Scene.h:152-162, Graphic.h:137-143)."""

    def __init__(self, dev: rhi.Device, instances, meshData, meshlets, opaqueIds, alphaMaskIds, num_meshlets: int | None = None):
        self.dev = dev
        self.numInstances = len(instances)
        self.instances = dev.buffer_from(instances, "Instance Consts Buffer")
        self.meshData = dev.buffer_from(meshData, "GlobalMeshDataBuffer", uav=False)
        if meshlets is None:
            self.meshlets = dev.create_buffer(num_meshlets * 32, "GlobalMeshletDataBuffer", stride=32, uav=False)
        else:
            self.meshlets = dev.buffer_from(meshlets, "GlobalMeshletDataBuffer", uav=False, min_bytes=32)
        self.opaqueIds = dev.buffer_from(np.asarray(opaqueIds, np.uint32), "OpaqueInstanceIDsBuffer", uav=False)
        self.alphaMaskIds = dev.buffer_from(np.asarray(alphaMaskIds, np.uint32), "AlphaMaskInstanceIDsBuffer", uav=False)
        self.numOpaque, self.numAlphaMask = len(opaqueIds), len(alphaMaskIds)
        self.vertices = self.meshletVertexIds = self.meshletTriangles = None

    def set_geometry(self, vertices, meshletVertexIds, meshletTriangles):
        """The buffers the mesh shader reads (basepass.hlsl t1, t5, t6): lets the frame rasterise its own depth
        (FrameDriver(raster_depth=True)) instead of taking a synthetic depth image."""
        self.vertices = self.dev.buffer_from(np.ascontiguousarray(vertices, I.RawVertexFormat), "GlobalVertexBuffer", uav=False, min_bytes=20)
        self.meshletVertexIds = self.dev.buffer_from(np.ascontiguousarray(meshletVertexIds, np.uint32), "GlobalMeshletVertexIdxOffsetsBuffer", uav=False)
        self.meshletTriangles = self.dev.buffer_from(np.ascontiguousarray(meshletTriangles, np.uint32), "GlobalMeshletIndicesBuffer", uav=False)

    def release(self):
        for b in (self.instances, self.meshData, self.meshlets, self.opaqueIds, self.alphaMaskIds, self.vertices, self.meshletVertexIds, self.meshletTriangles):
            if b is not None:
                b.release()


class FrameDriver:
    """BasePassRenderer (Setup + RenderBasePass) over rhi.  One command list per frame."""

    def __init__(self, dev: rhi.Device, scene: GpuScene, view, *, record_capacity: int, list_capacity: int | None = None,
                 culling_flags: int = 7, force_mesh_lod: int = -1, freeze_culling_camera: bool = False, alloc=None,
                 shard_late=None, raster_depth: bool = False):
        """alloc(nbytes, name, stride, indirect) -> rhi.Buffer or None: lets the caller own the memory of the
        output buffers (e.g. torch tensors handed to RCCL, gather.py); None -> device allocation.
        shard_late(hip_stream, late_count_ptr, shard_info_ptr, bucket, phase): multi-GPU hook, called while the
        frame is submitted: phase 0 after each early instance cull, phase 1 before each late one (include/trhost.h)."""
        self.raster_depth = bool(raster_depth)       # depth = the visible meshlets rasterised ("basepass_MS_Main_depth"), cleared per frame
        assert not self.raster_depth or scene.vertices is not None, "raster_depth needs GpuScene.set_geometry()"
        self.shard_late = shard_late
        self.shardInfo = [dev.create_buffer(8, f"ShardLateInfo{b}") for b in (0, 1)] if shard_late is not None else None
        self.dev, self.scene, self.view = dev, scene, view
        self.flags = culling_flags & 7
        self.force_mesh_lod = force_mesh_lod
        self.freeze = freeze_culling_camera
        self.record_capacity = int(record_capacity)
        self.list_capacity = int(list_capacity if list_capacity is not None else record_capacity * 32)
        hw, hh = I.hzb_dims(view.renderW, view.renderH)
        self.hzb_w, self.hzb_h = hw, hh
        self.hzb_mips = I.compute_nb_mips(hw, hh)
        # GBufferRenderer::Initialize (BasePassRenderers.cpp:596-616): HZB cleared to far = 0
        self.hzb = dev.create_texture(hw, hh, self.hzb_mips, rhi.FORMAT_R16_FLOAT, "HZB")
        self.depth = dev.create_texture(view.renderW, view.renderH, 1, rhi.FORMAT_R32_FLOAT, "Depth Buffer")
        init = dev.create_command_list()
        init.open(); init.clear_texture_f32(self.hzb, 0.0); init.clear_texture_f32(self.depth, 0.0); init.close()
        dev.execute(init); dev.wait_idle(); init.release()
        # BasePassRenderer::Setup (:223-296); one set of outputs per pass slot (DESIGN.md "Outputs")
        n = max(scene.numInstances, 1)

        def mk(nbytes, name, stride=4, indirect=False):
            b = alloc(nbytes, name, stride, indirect) if alloc is not None else None
            return b if b is not None else dev.create_buffer(nbytes, name, stride=stride, indirect=indirect)
        # slots 2,3 (alpha-mask lists) are only materialised when the scene has alpha-mask primitives
        slots = 4 if scene.numAlphaMask else 2
        self.records = [mk(12 * self.record_capacity, f"MeshletAmplificationDataBuffer{s}", 12) for s in range(slots)]
        self.dispatchArgs = [mk(16, f"MeshletDispatchArgumentsBuffer{s}", 16, True) for s in range(slots)]
        self.visMask = [mk(4 * self.record_capacity, f"MeshletVisibilityMaskBuffer{s}") for s in range(slots)]
        self.visibleList = [mk(4 * max(self.list_capacity, 1), f"VisibleMeshletListBuffer{s}") for s in range(slots)]
        self.drawArgs = [mk(12, f"VisibleMeshletDrawArgsBuffer{s}", 12, True) for s in range(slots)]
        self.num_slots = slots
        self.lateArgs = dev.create_buffer(12, "LateCullDispatchIndirectArgs", stride=12, indirect=True)
        self.lateCount = dev.create_buffer(4, "LateCullInstanceCountBuffer")
        self.lateIds = dev.create_buffer(4 * n, "LateCullInstanceIDsBuffer")
        self.spdAtomic = dev.create_buffer(24, "SPD Global Atomic Buffer", stride=24)
        self.dummy = dev.create_buffer(16, "DummyUIntStructuredBuffer")
        self.cl = dev.create_command_list()
        self.ran = [False] * 4

    # ---- per-frame constants (BasePassRenderers.cpp:334-347, 445-458, 551-563) ------------------
    def _cull_consts(self, nb: int) -> np.ndarray:
        v = self.view
        k = np.zeros(1, I.GPUCullingPassConstants)
        occ = bool(self.flags & 2)
        k["m_NbInstances"] = nb
        k["m_CullingFlags"] = self.flags
        k["m_HZBDimensions"] = (self.hzb_w, self.hzb_h) if occ else (1, 1)
        k["m_Frustum"] = culling_frustum(v.viewToClip)
        k["m_WorldToView"] = v.worldToView
        k["m_PrevWorldToView"] = v.prevWorldToView
        k["m_NearPlane"] = v.nearPlane
        k["m_P00"] = v.viewToClip[0, 0]
        k["m_P11"] = v.viewToClip[1, 1]
        k["m_ForcedMeshLOD"] = self.force_mesh_lod if self.force_mesh_lod >= 0 else I.kInvalidMeshLOD
        k["m_MeshLODTarget"] = np.float32(np.float32(2.0) / v.viewToClip[1, 1]) * np.float32(np.float32(1.0) / np.float32(v.renderH))
        return k

    def _basepass_consts(self, alpha_mask: bool) -> np.ndarray:
        v = self.view
        k = np.zeros(1, I.BasePassConstants)
        occ = bool(self.flags & 2)
        k["m_WorldToView"] = v.worldToView
        k["m_WorldToClip"] = I.world_to_clip(v.worldToView, v.viewToClip)
        k["m_Frustum"] = culling_frustum(v.viewToClip)
        k["m_CullingFlags"] = (self.flags & ~4) if alpha_mask else self.flags   # :436-442 (Q8)
        k["m_HZBDimensions"] = (self.hzb_w, self.hzb_h) if occ else (1, 1)
        k["m_P00"] = v.viewToClip[0, 0]
        k["m_P11"] = v.viewToClip[1, 1]
        k["m_NearPlane"] = v.nearPlane
        k["m_OutputResolution"] = (v.renderW, v.renderH)
        return k

    # ---- BasePassRenderer::GPUCulling (:298-404) ------------------------------------------------
    def _gpu_culling(self, cl, slot: int, late: bool, alpha_mask: bool):
        sc = self.scene
        nb = sc.numAlphaMask if alpha_mask else sc.numOpaque
        if nb == 0:
            return False
        occ = bool(self.flags & 2)
        late_args = self.lateArgs if occ else self.dummy
        late_count = self.lateCount if occ else self.dummy
        late_ids = self.lateIds if occ else self.dummy
        cl.clear_buffer_u32(self.dispatchArgs[slot], 0)                                   # :325
        if not late and occ:                                                              # :327-331
            cl.clear_buffer_u32(late_count, 0)
            cl.clear_buffer_u32(late_ids, 0)
        cb = cl.constant_buffer(self._cull_consts(nb), "GPUCullingPassConstants")         # :336-349
        bindings = [CB(0, cb), SRV(0, sc.instances), SRV(1, sc.alphaMaskIds if alpha_mask else sc.opaqueIds),
                    SRV(2, sc.meshData), UAV(0, self.records[slot]), UAV(1, self.dispatchArgs[slot]),
                    UAV(2, late_count), UAV(3, late_ids), SAMPLER(0)]
        if occ:
            bindings.append(TEX_SRV(3, self.hzb))
        name = f"gpuculling_CS_GPUCulling LATE_CULL={int(late)}"
        if not late:
            cl.dispatch(name, bindings, ((nb + 31) // 32, 1, 1))                          # :367-375
            if occ:                                                                       # :377-389
                cl.dispatch("gpuculling_CS_BuildLateCullIndirectArgs", [SRV(0, late_count), UAV(0, late_args)], (1, 1, 1))
                if self.shard_late is not None:                                           # multi-GPU only (trhost.h)
                    b = int(alpha_mask)
                    cl.host_callback(lambda stream, c=late_count.ptr, i=self.shardInfo[b].ptr, b=b: self.shard_late(stream, c, i, b, 0))
        elif occ:
            if self.shard_late is not None:                                               # multi-GPU only (trhost.h)
                bucket = int(alpha_mask)
                info = self.shardInfo[bucket]
                cl.host_callback(lambda stream, c=late_count.ptr, i=info.ptr, b=bucket: self.shard_late(stream, c, i, b, 1))
                bindings.append(SRV(4, info))
            cl.dispatch_indirect(name, bindings, late_args)                               # :392-402
        else:
            return False
        return True

    # ---- BasePassRenderer::RenderInstances (:406-503), cull half --------------------------------
    def _render_instances(self, cl, slot: int, late: bool, alpha_mask: bool):
        sc = self.scene
        nb = sc.numAlphaMask if alpha_mask else sc.numOpaque
        if nb == 0:
            return
        occ = bool(self.flags & 2)
        cb = cl.constant_buffer(self._basepass_consts(alpha_mask), "BasePassConstants")
        bindings = [CB(0, cb), SRV(0, sc.instances), SRV(2, sc.meshData), SRV(4, sc.meshlets), SRV(7, self.records[slot]),
                    UAV(0, self.visMask[slot]), UAV(1, self.visibleList[slot]), UAV(2, self.drawArgs[slot]), SAMPLER(4)]
        if occ:
            bindings.append(TEX_SRV(8, self.hzb))
        cl.dispatch_indirect(f"basepass_AS_Main LATE_CULL={int(late)}", bindings, self.dispatchArgs[slot])   # :497-502
        if self.raster_depth:                                                            # the mesh + pixel stage of the same draw: depth only
            b = [CB(0, cb), SRV(0, sc.instances), SRV(1, sc.vertices), SRV(2, sc.meshData), SRV(4, sc.meshlets), SRV(5, sc.meshletVertexIds),
                 SRV(6, sc.meshletTriangles), SRV(7, self.records[slot]), SRV(9, self.visibleList[slot]), TEX_UAV(0, self.depth, 0)]
            cl.dispatch_indirect("basepass_MS_Main_depth", b, self.drawArgs[slot])

    # ---- BasePassRenderer::GenerateHZB (:505-542) + SPD::Execute (FFXHelpers.cpp:36-115) --------
    def _generate_hzb(self, cl):
        if self.freeze:
            return
        k = np.zeros(1, I.MinMaxDownsampleConsts)
        k["m_OutputDimensions"] = (self.hzb_w, self.hzb_h)
        k["m_bDownsampleMax"] = 0
        cl.dispatch("minmaxdownsample_CS_Main", [PUSH(0), TEX_SRV(0, self.depth), TEX_UAV(0, self.hzb, 0), SAMPLER(0)],
                    ((self.hzb_w + 7) // 8, (self.hzb_h + 7) // 8, 1), push=k)
        cl.clear_buffer_u32(self.spdAtomic, 0)
        spd = np.zeros(8, np.uint32)
        spd[0] = self.hzb_mips - 1
        spd[1] = ((self.hzb_w + 63) // 64) * ((self.hzb_h + 63) // 64)
        b = [PUSH(0), TEX_SRV(0, self.depth), UAV(0, self.spdAtomic), TEX_UAV(1, self.hzb, min(6, self.hzb_mips - 1)), TEX_UAV(2, self.hzb, 0)]
        b += [TEX_UAV(3 + i, self.hzb, i + 1) for i in range(self.hzb_mips - 1)]
        cl.dispatch("ffx_spd_downsample_pass_CS FFX_SPD_OPTION_DOWNSAMPLE_FILTER=1", b,
                    ((self.hzb_w + 63) // 64, (self.hzb_h + 63) // 64, 1), push=spd)

    # ---- BasePassRenderer::RenderBasePass (:544-588) --------------------------------------------
    def record(self):
        cl = self.cl
        cl.open()
        occ = bool(self.flags & 2)
        self.ran = [False] * 4

        if self.raster_depth:
            cl.clear_texture_f32(self.depth, 0.0)                                        # depth cleared to far at the start of the base pass

        def do(slot, late, am):
            self.ran[slot] = self._gpu_culling(cl, slot, late, am)
            if self.ran[slot]:
                self._render_instances(cl, slot, late, am)
        do(0, False, False)
        if occ:
            self._generate_hzb(cl)
            do(1, True, False)
            do(2, False, True)
            do(3, True, True)
            self._generate_hzb(cl)
        else:
            do(2, False, True)
        cl.close()
        return cl

    def run(self):
        self.dev.execute(self.cl)

    # ---- read-back (tests) ------------------------------------------------------------------------
    def results(self):
        self.dev.wait_idle()
        out = {}
        for s in range(4):
            if s >= self.num_slots or not self.ran[s]:
                out[s] = None
                continue
            args = self.dispatchArgs[s].download(np.uint32, 4)
            G = int(min(args[0], args[3], self.record_capacity))
            draw = self.drawArgs[s].download(np.uint32, 3)
            V = int(min(draw[0], self.list_capacity))
            out[s] = dict(dispatchArgs=args[:3].copy(), validRecords=int(args[3]),
                          records=self.records[s].download(I.MeshletAmplificationData, G),
                          visMask=self.visMask[s].download(np.uint32, G),
                          visibleList=self.visibleList[s].download(np.uint32, V), drawArgs=draw)
        out["lateCount"] = int(self.lateCount.download(np.uint32, 1)[0])
        out["lateArgs"] = self.lateArgs.download(np.uint32, 3)
        return out

    def release(self):
        self.cl.release()
        for lst in (self.records, self.dispatchArgs, self.visMask, self.visibleList, self.drawArgs):
            for b in lst:
                b.release()
        for b in (self.lateArgs, self.lateCount, self.lateIds, self.spdAtomic, self.dummy, *(self.shardInfo or ())):
            b.release()
        self.hzb.release(); self.depth.release()
