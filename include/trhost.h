/*
 * trhost.h -- C entry points of the C++ host mirror (toyrenderer_amd/csrc/host): the reference's
 * frame loop for the visibility path (Graphic::Update -> Scene::Update -> RenderGraph::AddRenderer /
 * Compile -> UpdateInstanceConstsRenderer / GBufferRenderer -> Graphic::AddComputePass) behind a
 * plain-C facade, for callers that cannot link C++ (Python tests, bench.py).  A C++ application
 * uses the classes directly.
 *
 * All functions return 0 on success, -1 on failure (text: trhost_last_error()).  One context per
 * process (the reference's Graphic / Scene are singletons: Graphic.h:43, Scene.h:179).
 */
#ifndef TRHOST_H_
#define TRHOST_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* trhost_last_error(void);

/* Graphic::Initialize (Graphic.cpp:608-647): device, shaders(kernels), common resources, renderers'
 * Initialize() (HZB creation, BasePassRenderers.cpp:596-616).  external_hip_stream may be NULL. */
int  trhost_initialize(int device_index, uint32_t render_width, uint32_t render_height, void* external_hip_stream);
void trhost_shutdown(void);

/* Scene content as flat arrays in the wire formats of ShaderInterop.h (what SceneLoading.cpp:203-224,
 * 1016-1088 and Scene.cpp:282-362 produce); followed by Graphic::PostSceneLoad. */
int  trhost_load_scene(const void* instances, uint32_t num_instances, const void* mesh_data, uint32_t num_meshes,
                       const void* meshlets, uint64_t num_meshlets, const uint32_t* opaque_ids, uint32_t num_opaque,
                       const uint32_t* alpha_mask_ids, uint32_t num_alpha_mask);
/* Same, with meshes, meshlets and the mesh-shader geometry read from a `<scene>_CachedData.bin` version 3, the
 * reference's mesh-processing cache (SceneLoading.cpp:57-79 layout, :706-781 LoadCachedData); instances and id lists
 * come from the caller (the glTF side).  Fails on another version, a truncated file or dangling ranges. */
int  trhost_load_scene_cached(const char* cached_data_path, const void* instances, uint32_t num_instances, const uint32_t* opaque_ids, uint32_t num_opaque,
                              const uint32_t* alpha_mask_ids, uint32_t num_alpha_mask);
/* Large scenes: pass meshlets = NULL to trhost_load_scene (allocation only) and stream the meshlet
 * buffer in with this call. */
int  trhost_upload_meshlets(uint64_t first_meshlet, const void* meshlets, uint64_t count);
/* Node hierarchy for UpdateInstanceConstsRenderer (BasePassRenderers.cpp:64-104); enables the pass. */
int  trhost_load_nodes(const void* node_local_transforms, uint32_t num_nodes, const uint32_t* primitive_to_node);
int  trhost_set_node_transforms(const void* node_local_transforms, uint32_t num_nodes);
/* Multi-GPU (not in the reference, which updates every instance it renders): UpdateInstanceConstsRenderer rebuilds the
 * transforms of instances [first, first + count) only -- the range a rank's id lists cover; the rest of the replicated
 * instance table is never read by its passes.  Default: the whole table. */
int  trhost_set_instance_update_range(uint32_t first, uint32_t count);

/* View (Scene.cpp:109-145): row-major 4x4, row vectors.  prev_world_to_view / view_to_clip may be NULL
 * (previous = last frame's; projection = RH reverse-Z infinite from fov/aspect/near). */
int  trhost_set_camera(const float* world_to_view, const float* prev_world_to_view, const float* view_to_clip, float near_plane);
/* Scene.h:128-132 toggles; force_mesh_lod < 0 = automatic. */
int  trhost_set_culling(int frustum, int occlusion, int cone, int freeze_culling_camera, int force_mesh_lod);
/* Capacity of the amplification-record buffer (kMaxThreadGroupsPerDimension = 65535 in the reference)
 * and the per-resource cap of the render graph (1 GB in the reference); 0 keeps the current value. */
int  trhost_set_limits(uint32_t max_meshlet_groups, uint64_t max_transient_resource_bytes);
/* Per-renderer GPU timer queries (RenderGraph.cpp:262-281, read back by trhost_renderer_times).  On by default like
 * the reference; each query is two timestamped barrier packets on the stream. */
int  trhost_set_gpu_timers(int enable);

/* Depth image the next frames' GenerateHZB will consume (stand-in for the rasteriser). */
int  trhost_upload_depth(const float* depth, uint32_t width, uint32_t height);
/* Instead of the stand-in: the frame rasterises the depth of its own visible meshlets ("basepass_MS_Main_depth", the
 * compute replacement of MS_Main + depth test, basepass.hlsl:124-188) after every cull pass, from the buffers the mesh
 * shader reads (SceneLoading.cpp:1016-1088): RawVertexFormat vertices (20 B), meshlet vertex ids, packed meshlet
 * triangles.  trhost_download_depth: the depth buffer of the last frame (float32, render resolution), after wait_idle. */
int  trhost_load_geometry(const void* vertices, uint64_t num_vertices, const uint32_t* meshlet_vertex_ids, uint64_t num_vertex_ids,
                          const uint32_t* meshlet_triangles, uint64_t num_triangles);
int  trhost_set_raster_depth(int enable);
int  trhost_download_depth(float* depth, uint64_t bytes);
int  trhost_upload_hzb_mip(uint32_t mip, const uint16_t* texels, uint64_t bytes);
int  trhost_download_hzb_mip(uint32_t mip, uint16_t* texels, uint64_t bytes);
int  trhost_hzb_info(uint32_t* width, uint32_t* height, uint32_t* mips);

/* GI debug view (GIDebugRenderer, GIRenderer.cpp:598-808): probe world positions (3 floats each) and states (1 float each,
 * 1 = RTXGI_DDGI_PROBE_STATE_INACTIVE) as the DDGI volume would provide them; from then on every frame runs
 * "giprobevisualization_CS_VisualizeGIProbesCulling" after the base pass.  num_probes = 0 switches it off.
 * trhost_gi_probe_buffers: trhip_buffer handles of the last dispatch's outputs (positions, DrawIndexedIndirectArguments,
 * instance -> probe index). */
int  trhost_load_gi_probes(const float* positions, const float* states, uint32_t num_probes, float probe_radius, int hide_inactive);
int  trhost_gi_probe_buffers(void** positions, void** draw_args, void** instance_to_probe);

int  trhost_frame(void);       /* Graphic::Update: record every pass, submit, (asynchronous)       */
int  trhost_wait_idle(void);

typedef struct {
    int   ran;
    void* records;        /* trhip_buffer handles (include/trhip.h) of the pass slot's outputs      */
    void* dispatch_args;
    void* vis_mask;
    void* visible_list;
    void* draw_args;
    void* late_count;
    void* late_args;
} trhost_pass_buffers_t;
/* slot: 0 early-opaque, 1 late-opaque, 2 early-alpha-mask, 3 late-alpha-mask */
int  trhost_pass_buffers(uint32_t slot, trhost_pass_buffers_t* out);
int  trhost_instance_buffer(void** buffer);
/* Lengths of this process' opaque / alpha-mask id lists (Scene.cpp:282-362). */
int  trhost_scene_list_sizes(uint32_t* num_opaque, uint32_t* num_alpha_mask);
void* trhost_device(void);     /* the trhip_device in use                                           */

/* Multi-GPU (one process per GPU, instance list sharded; not in the reference).  When set, fn(user, hip_stream,
 * late_count, shard_info, bucket, phase) is called on the thread inside trhost_frame while the frame is submitted,
 * twice per bucket (0 opaque, 1 alpha mask):
 *   phase 0  right after the EARLY instance cull: late_count (device address of this rank's late-list length, 1 x u32)
 *            is final from this point of hip_stream on.  Start the exchange -- typically on another stream, after an
 *            event recorded on hip_stream: all-gather late_count, then trhip_launch_shard_late_info fills shard_info
 *            (2 x u32) with {late entries of the lower ranks, late entries of all ranks}.  It has the whole early
 *            meshlet cull and the HZB build to complete.
 *   phase 1  right before the LATE instance cull: make hip_stream wait for shard_info (event wait).
 * The late dispatch then covers exactly the entries the single-GPU dispatch would (gpuculling.hlsl:182-195).
 * fn = NULL removes the hook. */
typedef void (*trhost_shard_late_fn)(void* user, void* hip_stream, void* late_count, void* shard_info, int bucket, int phase);
int  trhost_set_shard_late_exchange(trhost_shard_late_fn fn, void* user);

/* Native per-frame driver of the multi-GPU exchange (protocol: toyrenderer_amd/gather.py).
 * The collectives are callbacks so that the library needs no RCCL at link time: all-gather `count_words` 32-bit words
 * from `send` of every rank into `recv` (rank-major), enqueued on `hip_stream`; return 0 on success.
 * trhost_rccl_allgather is the ready-made binding: user = void*[2] { address of ncclAllGather, the ncclComm_t }. */
typedef int (*trhost_allgather_fn)(void* user, const void* send, void* recv, uint64_t count_words, void* hip_stream);
int  trhost_rccl_allgather(void* user, const void* send, void* recv, uint64_t count_words, void* hip_stream);
int  trhost_rccl_allreduce_max_u32(void* user, void* words, uint64_t count_words, void* hip_stream);   /* user = void*[2] { address of ncclAllReduce, the ncclComm_t } */
typedef struct trhost_exchange_desc {
    uint32_t world, rank;
    uint32_t slot_groups;            /* groups one rank's shard slot holds (the same on every rank)                 */
    uint32_t group_capacity;         /* whole-scene outputs, groups (0: world * slot_groups)                        */
    uint64_t list_capacity;          /* whole-scene visible list, entries (0: 32 * group_capacity)                  */
    uint32_t pass_slot_mask;         /* bit s: gather pass slot s (0 early-opaque, 1 late-opaque, 2/3 alpha mask)   */
    int      overlap;                /* 1: gather + unpack on their own stream, overlapping the next frame          */
    trhost_allgather_fn slots_allgather; void* slots_user;     /* shard slots, once per frame                       */
    trhost_allgather_fn late_allgather;  void* late_user;      /* late-list lengths, inside the frame (1 word)      */
    /* bit 0: SOME rank holds opaque ids, bit 1: some rank holds alpha-mask ids (an all-reduce of trhost_scene_list_sizes
     * at set-up).  Every rank posts the in-frame late-count collective of exactly these buckets, whether its own
     * list is empty or not (an empty list contributes 0), so the collectives match on all ranks.  0 = this rank's
     * own lists (only right when every rank holds the same kinds of lists). */
    uint32_t list_presence_mask;
    /* 1: every rank rasterises only its shard's visible meshlets (trhost_set_raster_depth), so before each
     * GenerateHZB the depth buffers are combined across ranks with `depth_allreduce_max` (element-wise MAX of the
     * reverse-Z depth words; positive floats order like their bit patterns).  Without the callback the combination
     * exchange + raster depth is rejected: per-rank HZBs would make the late and next-frame culls diverge from the
     * single-GPU frame. */
    int (*depth_allreduce_max)(void* user, void* depth_words, uint64_t count_words, void* hip_stream); void* depth_user;
    /* run entries one rank's shard slot holds (the same on every rank): a run = the consecutive records of one
     * submitted instance, so the number of id-list entries of the largest shard bounds it.  0: slot_groups (always
     * enough).  Slot size = 16 + 4 * slot_runs + slot_groups words. */
    uint32_t slot_runs;
    /* Q2 (gpuculling.hlsl:64-74) made global.  > 0: the group capacity (max_groups) of the single-GPU run this exchange
     * reproduces; every rank must run its passes with that same capacity.  The unpack then cuts the rank-major
     * concatenation in front of the first instance the single-GPU pass would drop (its position follows from the ranks'
     * dispatch counters in the slot headers; protocol: toyrenderer_amd/gather.py) and reports {sum of the counters, 1, 1,
     * validRecords}.  0: a rank that drops groups only raises status bit 8 in the whole-scene arguments. */
    uint32_t global_group_capacity;
} trhost_exchange_desc;
int  trhost_exchange_create(const trhost_exchange_desc* desc);   /* also installs the in-frame late-count hook      */
int  trhost_exchange_run(void);                                  /* after trhost_frame: pack, gather, unpack (async) */
int  trhost_exchange_wait(void);                                 /* until the last run's results are complete; fails if the unpack flagged a
                                                                  * pass slot (status word 7 of its arguments: slot overflow, capacity, header, drop) */
/* trhip_buffer handles of the whole-scene results of a pass slot: records, lane masks, ordered visible list,
 * args (8 words: {G,1,1,G}, {V,1,1}, status bits as in gather.py). */
int  trhost_exchange_outputs(uint32_t pass_slot, void** records, void** masks, void** list, void** args);
int  trhost_exchange_destroy(void);

/* Async compute (RenderGraph.cpp:251 "TODO: compute queue" in the reference): which queue a renderer records for --
 * 0 graphics (default), 1 compute = a second stream; the render graph derives the cross-queue waits from the passes'
 * declared resource accesses.  renderer_name: "UpdateInstanceConstsRenderer", "GBufferRenderer", "GIDebugRenderer". */
int  trhost_set_renderer_queue(const char* renderer_name, int queue);
/* Of the last frame: passes on the compute queue, cross-queue waits placed, bytes of live transient resources and what
 * an allocator that aliases non-overlapping pass lifetimes would need for them. */
int  trhost_render_graph_frame_stats(uint32_t* compute_queue_passes, uint32_t* cross_queue_waits, uint64_t* transient_bytes, uint64_t* aliased_bytes);
int  trhost_render_graph_stats(uint32_t* num_heaps, uint64_t* bytes_reserved, uint64_t* bytes_used, uint32_t* num_passes);
/* renderer_name "<frame>": host milliseconds the last trhost_frame spent recording (cpu_ms) and submitting (gpu_ms). */
int  trhost_renderer_times(const char* renderer_name, float* cpu_ms, float* gpu_ms);

/* Test hook (no GPU): drives RenderGraph::Heap's free-list allocator (RenderGraph.cpp:443-580).
 * ops[i] > 0: Allocate(ops[i]) -> results[i] = offset (UINT64_MAX = no fit);
 * ops[i] < 0: Free(results[-ops[i]-1]). */
int  trhost_heap_sim(uint64_t heap_size, const int64_t* ops, uint32_t num_ops, uint64_t* results, uint64_t* used, uint64_t* peak, uint32_t* num_blocks);

#ifdef __cplusplus
}
#endif
#endif /* TRHOST_H_ */
