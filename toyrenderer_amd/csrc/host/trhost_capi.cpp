// trhost_capi.cpp -- a small C entry-point set over the C++ host mirror so that Python (tests,
// bench.py) can drive the SAME code path a C++ application would: Graphic::Initialize ->
// Scene::LoadFromArrays -> Graphic::Update per frame (Scene::Update -> RenderGraph -> renderers ->
// AddComputePass -> C ABI -> HIP kernels).  Declared in include/trhost.h.
#include <cstring>
#include <exception>
#include <string>

#include "../../../include/trhost.h"
#include "Graphic.h"
#include "RenderGraph.h"
#include "Scene.h"
#include "VisibilityOutputs.h"
#include "../ShaderInterop.h"

namespace
{
thread_local std::string tl_error;
bool s_Initialized = false;

template <typename F> int guarded(F&& f)
{
    try {
        f();
        return 0;
    } catch (const std::exception& e) {
        tl_error = e.what();
        return -1;
    }
}

Matrix toMatrix(const float* m)
{
    Matrix r;
    std::memcpy(r.m, m, sizeof r.m);
    return r;
}
} // namespace

extern "C" {

const char* trhost_last_error(void) { return tl_error.c_str(); }

int trhost_initialize(int device_index, uint32_t render_width, uint32_t render_height, void* external_hip_stream)
{
    if (s_Initialized) { tl_error = "trhost_initialize: already initialized (call trhost_shutdown first)"; return -1; }
    int rc = guarded([&] { g_Graphic.Initialize(device_index, Vector2U{ render_width, render_height }, external_hip_stream); });
    s_Initialized = rc == 0;
    return rc;
}

void trhost_shutdown(void)
{
    if (!s_Initialized) return;
    (void)guarded([&] {
        ShardExchangeDestroy();
        ReleaseVisibilityPassBuffers();
        ReleaseGIProbeCullBuffers();
        g_Graphic.Shutdown();
    });
    s_Initialized = false;
}

int trhost_load_scene(const void* instances, uint32_t num_instances, const void* mesh_data, uint32_t num_meshes,
                      const void* meshlets, uint64_t num_meshlets, const uint32_t* opaque_ids, uint32_t num_opaque,
                      const uint32_t* alpha_mask_ids, uint32_t num_alpha_mask)
{
    return guarded([&] {
        g_Scene->LoadFromArrays(instances, num_instances, mesh_data, num_meshes, meshlets, num_meshlets, opaque_ids, num_opaque, alpha_mask_ids, num_alpha_mask);
        g_Graphic.PostSceneLoad();
    });
}

int trhost_upload_meshlets(uint64_t first_meshlet, const void* meshlets, uint64_t count)
{
    return guarded([&] {
        check(g_Graphic.m_GlobalMeshletDataBuffer);
        nvrhi::throwIfFailed(trhip_buffer_upload(g_Graphic.m_GlobalMeshletDataBuffer->native(), first_meshlet * sizeof(interop::MeshletData), meshlets,
                                                 count * sizeof(interop::MeshletData)), "trhost_upload_meshlets");
    });
}

int trhost_load_scene_cached(const char* cached_data_path, const void* instances, uint32_t num_instances, const uint32_t* opaque_ids, uint32_t num_opaque,
                             const uint32_t* alpha_mask_ids, uint32_t num_alpha_mask)
{
    return guarded([&] {
        check(cached_data_path && instances);
        g_Scene->LoadCachedData(cached_data_path, instances, num_instances, opaque_ids, num_opaque, alpha_mask_ids, num_alpha_mask);
        g_Graphic.PostSceneLoad();
    });
}

int trhost_load_geometry(const void* vertices, uint64_t num_vertices, const uint32_t* meshlet_vertex_ids, uint64_t num_vertex_ids,
                         const uint32_t* meshlet_triangles, uint64_t num_triangles)
{
    return guarded([&] {
        check(g_Scene && (vertices || !num_vertices) && (meshlet_vertex_ids || !num_vertex_ids) && (meshlet_triangles || !num_triangles));
        g_Scene->LoadGeometry(vertices, num_vertices, meshlet_vertex_ids, num_vertex_ids, meshlet_triangles, num_triangles);
    });
}

int trhost_set_raster_depth(int enable)
{
    return guarded([&] {
        check(!enable || g_Graphic.m_GlobalVertexBuffer);      // trhost_load_geometry first
        g_Scene->m_bRasterDepth = enable != 0;
    });
}

int trhost_download_depth(float* depth, uint64_t bytes)
{
    return guarded([&] {
        nvrhi::TextureHandle t = GetLastDepthBuffer();
        check(t);
        nvrhi::throwIfFailed(trhip_texture_download(t->native(), 0, depth, bytes), "trhost_download_depth");
    });
}

int trhost_load_nodes(const void* node_local_transforms, uint32_t num_nodes, const uint32_t* primitive_to_node)
{
    return guarded([&] { g_Scene->LoadNodes(node_local_transforms, num_nodes, primitive_to_node); });
}

int trhost_set_node_transforms(const void* node_local_transforms, uint32_t num_nodes)
{
    return guarded([&] {
        check(num_nodes == g_Scene->m_NumNodes);
        std::memcpy(g_Scene->m_NodeLocalTransforms.data(), node_local_transforms, (size_t)num_nodes * sizeof(interop::NodeLocalTransform));
        g_Scene->m_bNodeLocalTransformsDirty = true;
    });
}

int trhost_set_instance_update_range(uint32_t first, uint32_t count)
{
    return guarded([&] { g_Scene->m_InstanceUpdateFirst = first; g_Scene->m_InstanceUpdateCount = count; });
}

int trhost_set_camera(const float* world_to_view, const float* prev_world_to_view, const float* view_to_clip, float near_plane)
{
    return guarded([&] {
        View& v = g_Scene->m_View;
        if (prev_world_to_view) v.SetPrevCamera(toMatrix(prev_world_to_view));
        v.SetCamera(toMatrix(world_to_view));
        if (view_to_clip) { v.m_ViewToClip = toMatrix(view_to_clip); v.m_bUseExplicitProjection = true; }
        v.m_ZNearP = near_plane;
    });
}

int trhost_set_culling(int frustum, int occlusion, int cone, int freeze_culling_camera, int force_mesh_lod)
{
    return guarded([&] {
        g_Scene->m_bEnableFrustumCulling = frustum != 0;
        g_Scene->m_bEnableOcclusionCulling = occlusion != 0;
        g_Scene->m_bEnableMeshletConeCulling = cone != 0;
        g_Scene->m_bFreezeCullingCamera = freeze_culling_camera != 0;
        g_Scene->m_ForceMeshLOD = force_mesh_lod;
    });
}

int trhost_set_gpu_timers(int enable)
{
    return guarded([&] { g_Graphic.m_bEnableGPUTimers = enable != 0; });
}

int trhost_set_limits(uint32_t max_meshlet_groups, uint64_t max_transient_resource_bytes)
{
    return guarded([&] {
        if (max_meshlet_groups) g_Graphic.m_MaxMeshletGroups = max_meshlet_groups;
        if (max_transient_resource_bytes) RenderGraph::ms_MaxHeapBlockSize = max_transient_resource_bytes;
    });
}

int trhost_upload_depth(const float* depth, uint32_t width, uint32_t height)
{
    return guarded([&] {
        check(g_Scene->m_SyntheticDepth);
        nvrhi::throwIfFailed(trhip_texture_upload(g_Scene->m_SyntheticDepth->native(), 0, depth, (uint64_t)width * height * 4), "trhost_upload_depth");
    });
}

int trhost_upload_hzb_mip(uint32_t mip, const uint16_t* texels, uint64_t bytes)
{
    return guarded([&] { nvrhi::throwIfFailed(trhip_texture_upload(g_Scene->m_HZB->native(), mip, texels, bytes), "trhost_upload_hzb_mip"); });
}

int trhost_download_hzb_mip(uint32_t mip, uint16_t* texels, uint64_t bytes)
{
    return guarded([&] { nvrhi::throwIfFailed(trhip_texture_download(g_Scene->m_HZB->native(), mip, texels, bytes), "trhost_download_hzb_mip"); });
}

int trhost_hzb_info(uint32_t* width, uint32_t* height, uint32_t* mips)
{
    return guarded([&] {
        const nvrhi::TextureDesc& d = g_Scene->m_HZB->getDesc();
        if (width) *width = d.width;
        if (height) *height = d.height;
        if (mips) *mips = d.mipLevels;
    });
}

int trhost_frame(void) { return guarded([&] { g_Graphic.Update(); }); }

int trhost_wait_idle(void) { return guarded([&] { g_Graphic.m_NVRHIDevice->waitForIdle(); }); }

void* trhost_device(void) { return s_Initialized ? (void*)g_Graphic.m_NVRHIDevice->native() : nullptr; }

int trhost_pass_buffers(uint32_t slot, trhost_pass_buffers_t* out)
{
    return guarded([&] {
        VisibilityPassBuffers b;
        check(GetVisibilityPassBuffers(slot, &b));
        std::memset(out, 0, sizeof *out);
        out->ran = b.m_bRan ? 1 : 0;
        auto h = [](const nvrhi::BufferHandle& x) { return x ? (void*)x->native() : nullptr; };
        out->records = h(b.m_MeshletAmplificationDataBuffer);
        out->dispatch_args = h(b.m_MeshletDispatchArgumentsBuffer);
        out->vis_mask = h(b.m_MeshletVisibilityMaskBuffer);
        out->visible_list = h(b.m_VisibleMeshletListBuffer);
        out->draw_args = h(b.m_VisibleMeshletDrawArgsBuffer);
        out->late_count = h(b.m_LateCullInstanceCountBuffer);
        out->late_args = h(b.m_LateCullDispatchIndirectArgsBuffer);
    });
}

int trhost_instance_buffer(void** buffer)
{
    return guarded([&] { *buffer = g_Scene->m_InstanceConstsBuffer ? (void*)g_Scene->m_InstanceConstsBuffer->native() : nullptr; });
}

int trhost_load_gi_probes(const float* positions, const float* states, uint32_t num_probes, float probe_radius, int hide_inactive)
{
    return guarded([&] {
        check(num_probes == 0 || (positions && states));
        g_Scene->LoadGIProbes(positions, states, num_probes, probe_radius, hide_inactive != 0);
    });
}

int trhost_gi_probe_buffers(void** positions, void** draw_args, void** instance_to_probe)
{
    return guarded([&] {
        check(positions && draw_args && instance_to_probe);
        nvrhi::BufferHandle p, a, i;
        if (!GetGIProbeCullBuffers(&p, &a, &i)) throw nvrhi::Error("no GI probe culling dispatch has been recorded");
        *positions = p->native(); *draw_args = a->native(); *instance_to_probe = i->native();
    });
}

int trhost_scene_list_sizes(uint32_t* num_opaque, uint32_t* num_alpha_mask)
{
    return guarded([&] {
        check(num_opaque && num_alpha_mask);
        *num_opaque = (uint32_t)g_Scene->m_OpaquePrimitiveIDs.size();
        *num_alpha_mask = (uint32_t)g_Scene->m_AlphaMaskPrimitiveIDs.size();
    });
}

int trhost_set_shard_late_exchange(trhost_shard_late_fn fn, void* user)
{
    return guarded([&] { SetShardLateExchange(fn, user); });
}

int trhost_exchange_create(const trhost_exchange_desc* desc)
{
    return guarded([&] { check(desc); ShardExchangeCreate(*desc); });
}
int trhost_exchange_run(void) { return guarded([&] { ShardExchangeRun(); }); }
int trhost_exchange_wait(void) { return guarded([&] { ShardExchangeWait(); }); }
int trhost_exchange_outputs(uint32_t pass_slot, void** records, void** masks, void** list, void** args)
{
    return guarded([&] { check(records && masks && list && args); ShardExchangeOutputs(pass_slot, records, masks, list, args); });
}
int trhost_exchange_destroy(void) { return guarded([&] { ShardExchangeDestroy(); }); }

int trhost_set_renderer_queue(const char* renderer_name, int queue)
{
    return guarded([&] {
        check(renderer_name && (queue == 0 || queue == 1));
        for (IRenderer* r : IRenderer::ms_AllRenderers)
            if (r->m_Name == renderer_name) { r->m_Queue = queue ? nvrhi::CommandQueue::Compute : nvrhi::CommandQueue::Graphics; return; }
        throw nvrhi::Error(std::string("no renderer named ") + renderer_name);
    });
}

int trhost_render_graph_frame_stats(uint32_t* compute_queue_passes, uint32_t* cross_queue_waits, uint64_t* transient_bytes, uint64_t* aliased_bytes)
{
    return guarded([&] {
        const RenderGraph::FrameStats& f = g_Scene->m_RenderGraph->GetFrameStats();
        if (compute_queue_passes) *compute_queue_passes = f.m_NumComputeQueuePasses;
        if (cross_queue_waits) *cross_queue_waits = f.m_NumCrossQueueWaits;
        if (transient_bytes) *transient_bytes = f.m_TransientBytes;
        if (aliased_bytes) *aliased_bytes = f.m_AliasedBytes;
    });
}

int trhost_render_graph_stats(uint32_t* num_heaps, uint64_t* bytes_reserved, uint64_t* bytes_used, uint32_t* num_passes)
{
    return guarded([&] {
        const RenderGraph& rg = *g_Scene->m_RenderGraph;
        uint64_t reserved = 0, used = 0;
        for (const RenderGraph::Heap& h : rg.GetHeaps()) { reserved += h.m_Heap->getDesc().capacity; used += h.m_Used; }
        if (num_heaps) *num_heaps = (uint32_t)rg.GetHeaps().size();
        if (bytes_reserved) *bytes_reserved = reserved;
        if (bytes_used) *bytes_used = used;
        if (num_passes) *num_passes = (uint32_t)rg.GetNumPasses();
    });
}

int trhost_renderer_times(const char* renderer_name, float* cpu_ms, float* gpu_ms)
{
    return guarded([&] {
        if (std::string(renderer_name) == "<frame>") {       // host time of the last frame: recording / submission
            if (cpu_ms) *cpu_ms = g_Graphic.m_LastRecordMs;
            if (gpu_ms) *gpu_ms = g_Graphic.m_LastSubmitMs;
            return;
        }
        for (IRenderer* r : IRenderer::ms_AllRenderers)
            if (r->m_Name == renderer_name) {
                if (cpu_ms) *cpu_ms = r->m_CPUFrameTime;
                if (gpu_ms) *gpu_ms = r->m_GPUFrameTime;
                return;
            }
        throw nvrhi::Error(std::string("unknown renderer ") + renderer_name);
    });
}

int trhost_heap_sim(uint64_t heap_size, const int64_t* ops, uint32_t num_ops, uint64_t* results, uint64_t* used, uint64_t* peak, uint32_t* num_blocks)
{
    return guarded([&] {
        RenderGraph::Heap heap;                       // no device heap attached: the free-list logic only
        heap.m_Blocks.push_back({ heap_size, false });
        for (uint32_t i = 0; i < num_ops; ++i) {
            if (ops[i] > 0) {
                results[i] = heap.Allocate((uint64_t)ops[i]);
            } else {
                const uint32_t ref = (uint32_t)(-ops[i]) - 1;
                check(ref < i);
                heap.Free(results[ref]);
                results[i] = results[ref];
            }
        }
        if (used) *used = heap.m_Used;
        if (peak) *peak = heap.m_Peak;
        if (num_blocks) *num_blocks = (uint32_t)heap.m_Blocks.size();
    });
}

} // extern "C"
