// Scene.h -- the part of the reference's Scene / View (source/Scene.h:44-179, source/Scene.cpp) that
// feeds the visibility path: culling matrices, culling toggles, the GPU buffers the passes bind, and
// the per-frame pass schedule.  Camera controls, animation, ImGui, TLAS and lighting are out of scope.
#pragma once

#include <cstdint>
#include <memory>
#include <vector>

#include "MathUtilities.h"
#include "RenderGraph.h"
#include "nvrhi_lite.h"
#include "tf_lite.h"

// Scene.h:44-74
class View
{
public:
    void Update();                                   // Scene.cpp:109-145

    float m_ZNearP = 0.1f;                           // Scene.h:50
    float m_FOV = 0.785398163f;                      // 45 deg, Scene.h:52
    float m_AspectRatio = 16.0f / 9.0f;

    Matrix m_WorldToView{}, m_PrevWorldToView{};
    Matrix m_ViewToClip{}, m_PrevViewToClip{};
    Matrix m_CullingWorldToView{}, m_CullingPrevWorldToView{};   // frozen by m_bFreezeCullingCamera

    // The reference derives m_WorldToView from eye/orientation with DirectXMath (absent here) in
    // View::Update; this build takes the camera matrix from the application instead.
    void SetCamera(const Matrix& worldToView) { m_PendingWorldToView = worldToView; m_bHasPending = true; }
    // Tests need an explicit previous-frame matrix for frame 0.
    void SetPrevCamera(const Matrix& prevWorldToView) { m_WorldToView = prevWorldToView; }
    bool m_bUseExplicitProjection = false;           // keep m_ViewToClip as set by the application

private:
    Matrix m_PendingWorldToView{};
    bool m_bHasPending = false;
};

class Scene
{
public:
    void Initialize();
    void PostSceneLoad();                            // Scene.cpp:662-681
    void Update();                                   // Scene.cpp:468-521
    void Shutdown();

    // Scene content as flat arrays = what SceneLoading.cpp:203-224,1016-1088 produces
    // (global mesh / meshlet buffers) plus the primitive table of Scene.cpp:282-362.
    void LoadFromArrays(const void* instances, uint32_t numInstances,
                        const void* meshData, uint32_t numMeshes,
                        const void* meshlets, uint64_t numMeshlets,
                        const uint32_t* opaqueIds, uint32_t numOpaque,
                        const uint32_t* alphaMaskIds, uint32_t numAlphaMask);
    void LoadNodes(const void* nodeLocalTransforms, uint32_t numNodes, const uint32_t* primitiveToNode);

    View m_View;

    // Scene.h:128-132
    bool m_bEnableFrustumCulling = true;
    bool m_bEnableOcclusionCulling = true;
    bool m_bEnableMeshletConeCulling = true;
    bool m_bFreezeCullingCamera = false;
    int32_t m_ForceMeshLOD = -1;
    bool m_bUpdateInstanceTransforms = false;        // run UpdateInstanceConstsRenderer (animated scenes)
    // multi-GPU (not in the reference): the instances whose transforms this rank updates every frame -- the contiguous range its
    // id lists cover (trhost_set_instance_update_range); the whole (replicated) table by default
    uint32_t m_InstanceUpdateFirst = 0, m_InstanceUpdateCount = 0xFFFFFFFFu;

    uint32_t m_NumPrimitives = 0;
    std::vector<uint32_t> m_OpaquePrimitiveIDs, m_AlphaMaskPrimitiveIDs;
    std::vector<uint8_t> m_NodeLocalTransforms;      // NodeLocalTransform[] (host copy, uploaded every frame)
    uint32_t m_NumNodes = 0;
    bool m_bNodeLocalTransformsDirty = false;        // the host copy changed since UpdateInstanceConstsRenderer last uploaded it

    // Scene.h:152-162
    nvrhi::BufferHandle m_InstanceConstsBuffer;
    nvrhi::BufferHandle m_OpaqueInstanceIDsBuffer, m_AlphaMaskInstanceIDsBuffer;
    nvrhi::BufferHandle m_NodeLocalTransformsBuffer, m_PrimitiveIDToNodeIDBuffer;
    nvrhi::TextureHandle m_HZB;
    // stand-in for the rasteriser's output (RenderInstances' pixel work is out of scope): the depth
    // image that GenerateHZB consumes is copied from here into the transient depth buffer.
    nvrhi::TextureHandle m_SyntheticDepth;
    // Instead of the stand-in: every pass rasterises the depth of its visible meshlets ("basepass_MS_Main_depth", the
    // compute replacement of MS_Main + depth test) into the depth buffer, cleared at the start of the base pass.
    bool m_bRasterDepth = false;
    // `<scene>_CachedData.bin` version 3 (SceneLoading.cpp:57-79 layout, :706-781 LoadCachedData): meshes, meshlets and
    // the mesh-shader geometry come from the file, instances and id lists from the caller (the glTF side of the reference).
    void LoadCachedData(const char* path, const void* instances, uint32_t numInstances, const uint32_t* opaqueIds, uint32_t numOpaque,
                        const uint32_t* alphaMaskIds, uint32_t numAlphaMask);
    void LoadGeometry(const void* vertices, uint64_t numVertices, const uint32_t* meshletVertexIds, uint64_t numVertexIds,
                      const uint32_t* meshletTriangles, uint64_t numTriangles);

    // GI debug view (GIRenderer.cpp:598-808): the probes' world positions and states as the DDGI volume would give them
    // (inputs here: the RTXGI SDK is absent), culled every frame by GIDebugRenderer when m_bShowGIProbes is set.
    void LoadGIProbes(const float* positions, const float* states, uint32_t numProbes, float probeRadius, bool hideInactive);
    nvrhi::BufferHandle m_GIProbePositionsBuffer, m_GIProbeStatesBuffer;
    uint32_t m_NumGIProbes = 0;
    float m_GIProbeRadius = 0.1f;
    bool m_bHideInactiveGIProbes = false, m_bShowGIProbes = false;
    uint32_t m_GIProbeSphereIndexCount = 2880;       // index count of CommonResources' unit sphere mesh (draw side, out of scope)

    std::shared_ptr<RenderGraph> m_RenderGraph;
    tf::Executor m_Executor{ 4 };                    // Engine.cpp:19,110-116 (default 12 workers)
};
#define g_Scene (Graphic::GetInstance().m_Scene)
