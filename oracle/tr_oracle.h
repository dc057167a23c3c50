/*
 * tr_oracle.h -- CPU ORACLE for the ToyRenderer meshlet-visibility hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (toyrenderer_amd/) never
 * includes, links or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference (lawfuyang/ToyRenderer) ships no tests, golden vectors
 * or fixtures for this path and cannot be built here (Windows/D3D12/DXC, empty submodules;
 * SURVEY.md 8c).  This file is a plain-C restatement of the reference HLSL, function by
 * function, each citing the reference file:line it follows.  It is pinned only by
 * (a) hand-derived known-answer vectors (tests/golden/kat_*.json) and (b) an independent
 * numpy restatement (oracle/np_oracle.py) written from SURVEY.md section 10.
 *
 * Arithmetic convention (SURVEY.md 8.2 -- the build's choice, used verbatim by the HIP
 * kernels): IEEE-754 binary32, round-to-nearest-even, no implicit contraction
 * (-ffp-contract=off); the *explicit* fmaf() calls written below are part of the
 * convention (a matrix/dot product is an FMA chain, as DXC's FMad lowering of mul()/dot());
 * '/' and sqrtf are correctly rounded; floor(log2(x)) is exponent extraction; fp16 HZB
 * texels are written round-to-nearest-even.
 */
#ifndef TR_ORACLE_H_
#define TR_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- wire formats: reference source/shaders/ShaderInterop.h ---------------------- */

typedef struct { float m[4][4]; } OrcMatrix; /* row-major, row vectors: p' = p * M */

/* ShaderInterop.h:70-77 (144 B) */
typedef struct {
    OrcMatrix m_WorldMatrix;
    OrcMatrix m_PrevWorldMatrix;
    uint32_t m_MeshDataIdx;
    uint32_t m_MaterialDataIdx;
    float PAD0[2];
} OrcBasePassInstanceConstants;

/* ShaderInterop.h:174-180 (16 B) */
typedef struct {
    uint32_t m_MeshletDataBufferIdx;
    uint32_t m_NumMeshlets;
    float m_Error;
    uint32_t PAD0;
} OrcMeshLODData;

/* ShaderInterop.h:182-189 (156 B) */
typedef struct {
    float m_BoundingSphere[4];
    OrcMeshLODData m_MeshLODDatas[8];
    uint32_t m_NumLODs;
    uint32_t m_GlobalVertexBufferIdx;
    uint32_t m_GlobalIndexBufferIdx;
} OrcMeshData;

/* ShaderInterop.h:191-198 (32 B) */
typedef struct {
    float m_BoundingSphere[4];
    uint32_t m_ConeAxisAndCutoff;
    uint32_t m_MeshletVertexIDsBufferIdx;
    uint32_t m_MeshletIndexIDsBufferIdx;
    uint32_t m_VertexAndTriangleCount;
} OrcMeshletData;

/* ShaderInterop.h:207-212 (12 B) */
typedef struct {
    uint32_t m_InstanceConstIdx;
    uint32_t m_MeshLOD;
    uint32_t m_MeshletGroupOffset;
} OrcMeshletAmplificationData;

/* ShaderInterop.h:131-144 (180 B) */
typedef struct {
    uint32_t m_NbInstances;
    uint32_t m_CullingFlags;
    uint32_t m_HZBDimensions[2];
    float m_Frustum[4];
    OrcMatrix m_WorldToView;
    OrcMatrix m_PrevWorldToView;
    float m_NearPlane;
    float m_P00;
    float m_P11;
    uint32_t m_ForcedMeshLOD;
    float m_MeshLODTarget;
} OrcGPUCullingPassConstants;

/* ShaderInterop.h:49-68 (256 B) */
typedef struct {
    OrcMatrix m_WorldToClip;
    OrcMatrix m_PrevWorldToClip;
    OrcMatrix m_WorldToView;
    float m_Frustum[4];
    uint32_t m_HZBDimensions[2];
    float m_P00;
    float m_P11;
    float m_NearPlane;
    uint32_t m_CullingFlags;
    uint32_t m_DebugMode;
    uint32_t PAD0;
    uint32_t m_OutputResolution[2];
    uint32_t m_bVisualizeMinMipTilesOnAlbedoOutput;
    uint32_t m_bWriteSamplerFeedback;
} OrcBasePassConstants;

/* ShaderInterop.h:220-227 (48 B) */
typedef struct {
    uint32_t m_ParentNodeIdx;
    float m_Position[3];
    float m_Rotation[4];
    float m_Scale[3];
    uint32_t PAD0;
} OrcNodeLocalTransform;

enum {
    ORC_NUM_THREADS_PER_WAVE = 32,        /* ShaderInterop.h:6  */
    ORC_MAX_THREAD_GROUPS = 65535,        /* ShaderInterop.h:7  */
    ORC_CULL_FRUSTUM = 1,                 /* ShaderInterop.h:15 */
    ORC_CULL_OCCLUSION = 2,               /* ShaderInterop.h:16 */
    ORC_CULL_CONE = 4,                    /* ShaderInterop.h:17 */
    ORC_MAX_LODS = 8,                     /* ShaderInterop.h:23 */
    ORC_INVALID_LOD = 0xFF,               /* ShaderInterop.h:24 */
    ORC_MAX_MIPS = 16
};

/* HZB: R16_FLOAT mip chain in one linear allocation; mip k is row-major
 * max(w>>k,1) x max(h>>k,1) texels starting at texel offset mipOffset[k].
 * (GraphicConstants.h:28 kHZBFormat, BasePassRenderers.cpp:596-606.) */
typedef struct {
    uint32_t width, height, mips, pad;
    const uint16_t* texels;
    uint64_t mipOffset[ORC_MAX_MIPS];
} OrcHZB;

/* ---- scalar primitives (exported for known-answer tests) -------------------------- */

uint16_t orc_f32_to_f16(float f);          /* RNE, Q10 */
float    orc_f16_to_f32(uint16_t h);
uint32_t orc_hzb_layout(uint32_t w, uint32_t h, uint64_t* mipOffset, uint64_t* totalTexels); /* returns mips */

int   orc_frustum_cull(const float c[3], float r, const float frustum[4]);          /* 1 = visible */
int   orc_occlusion_cull(const float c[3], float r, float nearPlane, float P00, float P11, const OrcHZB* hzb); /* 1 = visible */
int   orc_cone_cull(const float c[3], float r, const float axis[3], float cutoff);  /* 1 = backfacing (cull) */
float orc_sample_hzb_min(const OrcHZB* hzb, float u, float v, float level);
int   orc_hzb_level(float width, float height, uint32_t mips);
float orc_max_scale(const OrcMatrix* world);
void  orc_sphere_to_world(const OrcMatrix* world, const float sphere[4], float out[4]);
void  orc_to_view(const float p[3], const OrcMatrix* worldToView, float out[3]);
void  orc_unpack_cone_view(uint32_t packed, const OrcMatrix* world, const OrcMatrix* worldToView, float axisOut[3], float* cutoffOut);
uint32_t orc_select_lod(const OrcMeshData* mesh, const float cv[3], float r, const OrcMatrix* world,
                        uint32_t forcedLOD, float lodTarget);
void  orc_make_world_matrix(const float pos[3], const float rot[4], const float scale[3], OrcMatrix* out);
void  orc_culling_frustum(const OrcMatrix* viewToClip, float out[4]);               /* BasePassRenderers.cpp:557-563 */

/* ---- passes ----------------------------------------------------------------------- */

/* CS_GPUCulling (gpuculling.hlsl:87-180) in canonical order = ascending dispatch thread id.
 * dispatchArgs[3] must have been cleared by the caller (BasePassRenderers.cpp:325).
 * lateCull=0: thread count = ceil(nbInstances/32)*32, ids from primitiveIds, appends to late list.
 * lateCull=1: thread count = lateDispatchArgsX*32 (Q1), ids from lateIds[0..*lateCount).
 * validRecords (optional) receives the number of leading records that are defined (Q2). */
void orc_instance_cull(const OrcGPUCullingPassConstants* k, int lateCull,
                       const OrcBasePassInstanceConstants* instances,
                       const uint32_t* primitiveIds,
                       const OrcMeshData* meshData,
                       const OrcHZB* hzb,
                       OrcMeshletAmplificationData* records,
                       uint32_t dispatchArgs[3],
                       uint32_t* lateCount, uint32_t* lateIds,
                       uint32_t lateDispatchArgsX,
                       uint32_t maxGroups,
                       uint32_t* validRecords);

/* Test helper: HZB level and bilinear footprint origin of each sphere's occlusion lookup (see tr_oracle.c). */
void orc_occlusion_footprints(const float* centres, const float* radii, uint32_t n, const OrcMatrix* worldToView,
                              float P00, float P11, uint32_t hzbW, uint32_t hzbH, uint32_t mips, int32_t* out);

/* ShaderInterop.h:249-261 (124 B) */
typedef struct {
    uint32_t m_NumProbes;
    float m_CameraOrigin[3];
    float m_Frustum[4];
    OrcMatrix m_WorldToView;
    uint32_t m_HZBDimensions[2];
    float m_P00, m_P11, m_NearPlane, m_ProbeRadius;
    uint32_t m_bHideInactiveProbes;
} OrcGIProbeVisualizationUpdateConsts;

/* CS_VisualizeGIProbesCulling (giprobevisualization.hlsl:16-69), the second consumer of FrustumCull / OcclusionCull.
 * The reference reads a probe's state and world position from the RTXGI-DDGI volume (:29-38, DDGILoadProbeState /
 * DDGIGetProbeWorldPosition: SDK code in an empty submodule); here both are INPUT arrays (positions: 3 floats per
 * probe, states: one float per probe, RTXGI_DDGI_PROBE_STATE_INACTIVE = 1).  Everything from :40 on is restated:
 * view transform, z flip, frustum test, occlusion test, compaction -- in ascending probe order (the reference appends
 * with InterlockedAdd, i.e. in no defined order).  drawArgs = DrawIndexedIndirectArguments (5 words): m_InstanceCount
 * (word 1) is incremented per visible probe from its incoming value, which is also where the appended entries start.
 * Returns the number of probes appended. */
uint32_t orc_gi_probe_cull(const OrcGIProbeVisualizationUpdateConsts* k, const float* probePositions, const float* probeStates,
                           const OrcHZB* hzb, float* outPositions, uint32_t* drawArgs, uint32_t* outInstanceToProbe);

/* CS_BuildLateCullIndirectArgs (gpuculling.hlsl:182-195), Q1: divides by 64. */
void orc_build_late_args(uint32_t lateCount, uint32_t out[3]);

/* AS_Main (basepass.hlsl:40-122), groups [groupBegin, groupEnd) in ascending order.
 * visMask[g] bit k = lane k visible.  If visibleList != NULL, entries ((g<<5)|lane) are written
 * starting at visibleList[*listCursor] and *listCursor is advanced.  Returns meshlets tested
 * (lanes with meshletIdx < numMeshlets). */
uint64_t orc_meshlet_cull(const OrcBasePassConstants* k,
                          const OrcBasePassInstanceConstants* instances,
                          const OrcMeshData* meshData,
                          const OrcMeshletData* meshlets,
                          const OrcMeshletAmplificationData* records,
                          uint32_t groupBegin, uint32_t groupEnd,
                          const OrcHZB* hzb,
                          uint32_t* visMask,
                          uint32_t* visibleList, uint64_t* listCursor);

/* ShaderInterop.h:278-283 (20 B): what the mesh shader fetches per vertex. */
typedef struct {
    float m_Position[3];
    uint32_t m_PackedNormal;
    uint16_t m_TexCoord[2];
} OrcRawVertexFormat;

/* Depth of the visible meshlets of one pass slot: stand-in for MS_Main (basepass.hlsl:124-188: vertex fetch through the
 * meshlet vertex-id / packed-triangle buffers, position * world * worldToClip) + the fixed-function rasteriser and
 * depth test, which have no source to restate -> CONVENTION, parity unpinned:
 *   clip = mulPoint(mulPoint(p, World), WorldToClip) with w = the 4th column's chain; a triangle with a vertex at
 *   w <= nearPlane is dropped (no near clipping); ndc = clip.xy / w, depth = clip.z / w (reverse-Z, far = 0);
 *   screen = (ndc.x * 0.5 + 0.5) * W, (-ndc.y * 0.5 + 0.5) * H as fma; samples at pixel centres; a sample is covered
 *   iff the three edge functions (fma(dx, py, -(dy * px)) form, oriented by the sign of the area) are >= 0 (both
 *   windings, ties included); depth = (e0 d0 + e1 d1 + e2 d2) / (e0 + e1 + e2) as an fma chain and one division;
 *   the buffer keeps the maximum (GREATER test, reverse-Z).  depth[] is max-merged in place. */
void orc_raster_depth(const OrcBasePassConstants* k,
                      const OrcBasePassInstanceConstants* instances, const OrcMeshData* meshData,
                      const OrcMeshletData* meshlets, const OrcRawVertexFormat* vertices,
                      const uint32_t* meshletVertexIds, const uint32_t* meshletTriangles,
                      const OrcMeshletAmplificationData* records, const uint32_t* visibleList, uint32_t numVisible,
                      float* depth);

/* minmaxdownsample CS_Main (minmaxdownsample.hlsl:10-35) + 2x2-min mip chain (FidelityFX SPD,
 * FFXHelpers.cpp:36-115; SPD source absent -> convention, parity unpinned). */
void orc_hzb_build(const float* depth, uint32_t depthW, uint32_t depthH,
                   uint16_t* texels, uint32_t hzbW, uint32_t hzbH, uint32_t mips, const uint64_t* mipOffset);

/* CS_UpdateInstanceConstsAndBuildTLAS (updateinstanceconsts.hlsl:11-36; TLAS part is OOS). */
void orc_update_instance_consts(const OrcNodeLocalTransform* nodes, const uint32_t* primToNode,
                                OrcBasePassInstanceConstants* instances, uint32_t numInstances);

/* ---- whole frame: BasePassRenderer::RenderBasePass (BasePassRenderers.cpp:544-588) --- */

typedef struct {
    /* scene */
    const OrcBasePassInstanceConstants* instances;
    const OrcMeshData* meshData;
    const OrcMeshletData* meshlets;
    const uint32_t* opaqueIds;   uint32_t numOpaque;
    const uint32_t* alphaMaskIds; uint32_t numAlphaMask;
    /* view */
    OrcMatrix worldToView, prevWorldToView, viewToClip;
    float nearPlane;
    uint32_t renderHeight;
    /* toggles (Scene.h:128-132) */
    uint32_t cullingFlags;      /* bits 0..2 */
    int32_t  forceMeshLOD;      /* <0 : automatic */
    uint32_t freezeCullingCamera;
    uint32_t maxGroups;         /* 65535 in the reference */
    /* HZB (in: previous frame's; rebuilt in place from depth twice per frame) */
    uint16_t* hzbTexels; uint32_t hzbW, hzbH, hzbMips; uint64_t hzbMipOffset[ORC_MAX_MIPS];
    const float* depth; uint32_t depthW, depthH;
    uint32_t threads;           /* >1: static contiguous partition, outputs concatenated in order */
    /* Multi-GPU checker only (not in the reference): this frame culls ONE SHARD of the instance list.
     * The late dispatch size rule (Q1) then applies to the whole scene's late list, of which this
     * shard's entries start at shardLateBase[bucket] (bucket 0 opaque, 1 alpha mask):
     * threads = clamp(ceil(shardLateTotal/64)*32 - shardLateBase, 0, own late count). */
    uint32_t shardLate;
    uint32_t shardLateBase[2], shardLateTotal[2];
    /* rasterDepth != 0: `depth` is an OUTPUT of the frame instead of an input: cleared to 0 (far) at frame
     * start, and every pass that ran rasterises its visible meshlets into it (orc_raster_depth) before the next
     * HZB build, as the mesh + pixel stages of the same draw do (basepass.hlsl:124-205). */
    uint32_t rasterDepth;
    OrcMatrix worldToClip;      /* BasePassConstants::m_WorldToClip (BasePassRenderers.cpp:447) */
    const OrcRawVertexFormat* vertices;
    const uint32_t* meshletVertexIds;
    const uint32_t* meshletTriangles;
} OrcFrameDesc;

/* Outputs per pass slot: 0 early-opaque, 1 late-opaque, 2 early-alphamask, 3 late-alphamask. */
typedef struct {
    OrcMeshletAmplificationData* records[4]; uint32_t recordCapacity;
    uint32_t dispatchArgs[4][3];
    uint32_t validRecords[4];
    uint32_t* visMask[4];
    uint32_t* visibleList[4]; uint64_t listCapacity;
    uint32_t drawArgs[4][3];
    uint32_t lateCount[2];        /* after the early pass of opaque / alphamask */
    uint32_t* lateIds[2];
    uint32_t lateArgs[2][3];
    uint64_t meshletsTested[4];
    uint32_t passRan[4];
} OrcFrameOut;

void orc_frame(OrcFrameDesc* d, OrcFrameOut* o);

#ifdef __cplusplus
}
#endif
#endif
