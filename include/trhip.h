/*
 * trhip.h -- C ABI of the MI355X (gfx950 / HIP) compute back end that replaces the NVRHI/D3D12
 * compute dispatch under ToyRenderer's GPU-driven visibility path.
 *
 * Plain C: opaque handles, plain pointers and sizes, int status returns (0 = ok, <0 = error,
 * text via trhip_last_error()).  No torch / C++ types cross this boundary.  The reference never
 * returns errors (everything asserts: PCH.h:42 check(), GraphicRHI.cpp:18-38); the C++ wrapper
 * above this ABI (toyrenderer_amd/csrc/host) re-creates the assert-on-failure behaviour.
 *
 * Each entry point cites the reference interface it replaces (paths relative to
 * /root/reference/source; nvrhi = the renderer's RHI, used by the reference as cited).
 *
 * Threading: distinct command lists may be recorded concurrently from
 * different threads; trhip_queue_execute is called from one thread at a time.  Nothing is
 * enqueued on the HIP stream before trhip_queue_execute (Graphic.cpp:786-830).
 */
#ifndef TRHIP_H_
#define TRHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRHIP_ABI_VERSION 1

typedef struct trhip_device_t*  trhip_device;
typedef struct trhip_heap_t*    trhip_heap;
typedef struct trhip_buffer_t*  trhip_buffer;
typedef struct trhip_texture_t* trhip_texture;
typedef struct trhip_cmdlist_t* trhip_cmdlist;
typedef struct trhip_timer_t*   trhip_timer;

enum {
    TRHIP_OK = 0,
    TRHIP_ERR_INVALID = -1,      /* bad argument / binding / shape                       */
    TRHIP_ERR_HIP = -2,          /* a HIP runtime call failed                            */
    TRHIP_ERR_UNKNOWN_SHADER = -3,
    TRHIP_ERR_STATE = -4,        /* e.g. recording into a closed command list            */
    TRHIP_ERR_NO_DEVICE = -5
};

/* nvrhi::Format subset used on the path (GraphicConstants.h:26-28). */
enum { TRHIP_FORMAT_R16_FLOAT = 1, TRHIP_FORMAT_R32_FLOAT = 2 };

/* ---- error / introspection ---------------------------------------------------------------- */
const char* trhip_last_error(void);          /* thread-local text of the last failure          */
uint32_t    trhip_abi_version(void);
/* Kernel registry keyed by the reference's shader-name strings (Graphic.cpp:159-171,246,270-278;
 * ShadersToCompile.txt): "gpuculling_CS_GPUCulling LATE_CULL=0", "... LATE_CULL=1",
 * "gpuculling_CS_BuildLateCullIndirectArgs", "minmaxdownsample_CS_Main",
 * "ffx_spd_downsample_pass_CS FFX_SPD_OPTION_DOWNSAMPLE_FILTER=1|2",
 * "updateinstanceconsts_CS_UpdateInstanceConstsAndBuildTLAS", and the compute replacement of the
 * amplification shader: "basepass_AS_Main LATE_CULL=0|1" (alias "basepass_AS_Main_cull").
 * "basepass_MS_Main_depth" (basepass.hlsl:124-188 + raster + depth test, depth only): b0 BasePassConstants,
 * t0 instances, t1 vertices (RawVertexFormat, 20 B), t2 mesh data, t4 meshlets, t5 meshlet vertex ids,
 * t6 packed meshlet triangles, t7 amplification records, t9 visible list, u0 (texture) R32_FLOAT depth;
 * dispatched indirectly on the visible list's draw args. */
uint32_t    trhip_shader_count(void);
const char* trhip_shader_name(uint32_t index);
int         trhip_shader_exists(const char* name);

/* ---- device: replaces GraphicRHI::CreateDevice / nvrhi::IDevice (GraphicRHI.cpp:56-200) ----- */
int  trhip_device_create(int device_index, trhip_device* out);
/* Same, but submissions go to a caller-owned hipStream_t (e.g. torch's current stream). */
int  trhip_device_create_on_stream(int device_index, void* hip_stream, trhip_device* out);
void trhip_device_destroy(trhip_device dev);
int  trhip_device_wait_idle(trhip_device dev);                 /* nvrhi waitForIdle, Graphic.cpp:790 */
/* Orders the device's stream after everything its lists have put on the back end's internal side stream so far: an event
 * recorded on trhip_device_stream() afterwards covers ALL work of the lists executed before (what another queue waits
 * for: nvrhi queueWaitForCommandList). */
int  trhip_device_join_side_stream(trhip_device dev);
int  trhip_device_info(trhip_device dev, uint32_t* compute_units, uint32_t* wave_size, uint64_t* total_mem);
void* trhip_device_stream(trhip_device dev);                   /* the hipStream_t in use            */

/* ---- memory: replaces nvrhi heaps / createBuffer / createTexture / bind*Memory
 *      (RenderGraph.cpp:137-221,431-441; BasePassRenderers.cpp:236-291,596-616) --------------- */
typedef struct {
    uint64_t    byteSize;
    uint32_t    structStride;
    uint32_t    canHaveUAVs;
    uint32_t    isDrawIndirectArgs;
    uint32_t    isVirtual;          /* 1: no memory until trhip_buffer_bind_memory (placed)    */
    uint32_t    isVolatileConstant; /* nvrhi volatile constant buffer (Graphic.h:66-72)        */
    const char* debugName;
} trhip_buffer_desc;

typedef struct {
    uint32_t    width, height, mipLevels;
    uint32_t    format;             /* TRHIP_FORMAT_*                                          */
    uint32_t    isUAV;
    uint32_t    isVirtual;
    const char* debugName;
} trhip_texture_desc;

int  trhip_heap_create(trhip_device dev, uint64_t bytes, trhip_heap* out);     /* nvrhi createHeap */
void trhip_heap_release(trhip_heap heap);

int  trhip_buffer_create(trhip_device dev, const trhip_buffer_desc* desc, trhip_buffer* out);
/* Wrap caller-owned device memory (e.g. a torch tensor's data_ptr) as a buffer; never freed here. */
int  trhip_buffer_wrap(trhip_device dev, void* device_ptr, const trhip_buffer_desc* desc, trhip_buffer* out);
int  trhip_buffer_memory_requirements(trhip_buffer buf, uint64_t* size, uint64_t* alignment);
int  trhip_buffer_bind_memory(trhip_buffer buf, trhip_heap heap, uint64_t offset);
void trhip_buffer_retain(trhip_buffer buf);
void trhip_buffer_release(trhip_buffer buf);
void* trhip_buffer_device_ptr(trhip_buffer buf);
uint64_t trhip_buffer_size(trhip_buffer buf);

int  trhip_texture_create(trhip_device dev, const trhip_texture_desc* desc, trhip_texture* out);
int  trhip_texture_memory_requirements(trhip_texture tex, uint64_t* size, uint64_t* alignment);
int  trhip_texture_bind_memory(trhip_texture tex, trhip_heap heap, uint64_t offset);
void trhip_texture_retain(trhip_texture tex);
void trhip_texture_release(trhip_texture tex);
void* trhip_texture_device_ptr(trhip_texture tex);
/* Layout in HBM: one linear allocation, mip k row-major max(w>>k,1) x max(h>>k,1) texels at
 * byte offset mip_offset(k); offsets are 256-byte aligned. */
int  trhip_texture_mip_info(trhip_texture tex, uint32_t mip, uint32_t* w, uint32_t* h, uint64_t* byte_offset);
uint64_t trhip_texture_size(trhip_texture tex);

/* Synchronous host access (scene upload, test read-back; the reference's equivalents are
 * writeBuffer on an init command list, SceneLoading.cpp:1016-1088, and no read-back at all). */
int  trhip_buffer_upload(trhip_buffer buf, uint64_t dst_offset, const void* src, uint64_t bytes);
int  trhip_buffer_download(trhip_buffer buf, uint64_t src_offset, void* dst, uint64_t bytes);
int  trhip_texture_upload(trhip_texture tex, uint32_t mip, const void* src, uint64_t bytes);
int  trhip_texture_download(trhip_texture tex, uint32_t mip, void* dst, uint64_t bytes);

/* Contents changed behind the back end's back.  The back end keeps derived, private copies of some bound resources (the
 * instance cull cache, the meshlet cull stream of the buffer bound at t4 of basepass_AS_Main, an HZB's footprint-min table)
 * and rebuilds them when the source's version counter moves.  The counter moves for every write the back end SEES:
 * trhip_*_upload, a UAV use in an executed command list, trhip_*_bind_memory.  A write it cannot see -- through the raw
 * pointer of trhip_buffer_wrap / trhip_*_device_ptr (a torch kernel, hipMemcpy), through a second wrap of the same memory,
 * through another virtual resource aliased on the same heap range -- MUST be followed by this call (any thread; ordered by
 * the caller before the next trhip_queue_execute that reads the resource), or later culls use the old contents.
 * nvrhi has no counterpart: D3D12 shaders read the buffers themselves. */
int  trhip_buffer_mark_written(trhip_buffer buf);
int  trhip_texture_mark_written(trhip_texture tex);

/* ---- command lists: replaces nvrhi::ICommandList (Graphic.cpp:520-606,893-947) --------------- */
typedef enum {
    TRHIP_BIND_CONSTANT_BUFFER = 0,  /* nvrhi::BindingSetItem::ConstantBuffer(slot, buf)        */
    TRHIP_BIND_PUSH_CONSTANTS  = 1,  /* ::PushConstants(slot, bytes); data via push_constants   */
    TRHIP_BIND_STRUCTURED_SRV  = 2,  /* ::StructuredBuffer_SRV(slot, buf)   register(tN)        */
    TRHIP_BIND_STRUCTURED_UAV  = 3,  /* ::StructuredBuffer_UAV(slot, buf)   register(uN)        */
    TRHIP_BIND_TEXTURE_SRV     = 4,  /* ::Texture_SRV(slot, tex)            register(tN)        */
    TRHIP_BIND_TEXTURE_UAV     = 5,  /* ::Texture_UAV(slot, tex, fmt, {baseMip,1,0,1}) (uN)     */
    TRHIP_BIND_SAMPLER         = 6   /* ::Sampler(slot, s): accepted and ignored (sampling is
                                        done in software, see DESIGN.md "HZB sampling")         */
} trhip_binding_type;

typedef struct {
    uint32_t type;       /* trhip_binding_type                                                  */
    uint32_t slot;
    void*    resource;   /* trhip_buffer or trhip_texture (NULL for push constants / sampler)   */
    uint32_t baseMip;    /* Texture_UAV subresource                                             */
    uint32_t reserved;
} trhip_binding;

int  trhip_cmd_create(trhip_device dev, trhip_cmdlist* out);          /* AllocateCommandList    */
void trhip_cmd_release(trhip_cmdlist cl);
int  trhip_cmd_open(trhip_cmdlist cl);                                /* ICommandList::open     */
int  trhip_cmd_close(trhip_cmdlist cl);                               /* ::close                */
/* ::writeBuffer -- the source bytes are copied at record time (nvrhi upload manager semantics).
 * On a volatile constant buffer this sets the version later dispatches in this list see. */
int  trhip_cmd_write_buffer(trhip_cmdlist cl, trhip_buffer buf, uint64_t dst_offset, const void* src, uint64_t bytes);
int  trhip_cmd_clear_buffer_u32(trhip_cmdlist cl, trhip_buffer buf, uint32_t value);   /* ::clearBufferUInt   */
int  trhip_cmd_clear_texture_f32(trhip_cmdlist cl, trhip_texture tex, float value);    /* ::clearTextureFloat */
int  trhip_cmd_copy_buffer(trhip_cmdlist cl, trhip_buffer dst, uint64_t dst_offset, trhip_buffer src, uint64_t src_offset, uint64_t bytes); /* ::copyBuffer */
/* Multi-GPU hook (no counterpart in the reference, which is single-GPU: GraphicRHI.cpp:165).
 * fn(user, hip_stream) is called on the submitting thread while the list is executed, in order with
 * the surrounding commands: whatever fn enqueues on hip_stream runs after everything recorded before
 * it and before everything recorded after it.  fn must not execute lists on / wait for this device. */
typedef void (*trhip_host_fn)(void* user, void* hip_stream);
int  trhip_cmd_host_callback(trhip_cmdlist cl, trhip_host_fn fn, void* user);
/* ::copyTexture, whole mip chain; both textures must have identical dimensions, mips and format. */
int  trhip_cmd_copy_texture(trhip_cmdlist cl, trhip_texture dst, trhip_texture src);
/* ::setComputeState + ::setPushConstants + ::dispatch(gx,gy,gz) (Graphic.cpp:893-947).
 * Group counts keep the reference's meaning (groups of the HLSL [numthreads]); the HIP launch
 * shape behind a shader name is the back end's business. */
int  trhip_cmd_dispatch(trhip_cmdlist cl, const char* shader_name,
                        const trhip_binding* bindings, uint32_t num_bindings,
                        const void* push_constants, uint32_t push_bytes,
                        uint32_t gx, uint32_t gy, uint32_t gz);
/* ::dispatchIndirect(offset): the 3 x u32 group counts are read ON THE DEVICE at execution. */
int  trhip_cmd_dispatch_indirect(trhip_cmdlist cl, const char* shader_name,
                                 const trhip_binding* bindings, uint32_t num_bindings,
                                 const void* push_constants, uint32_t push_bytes,
                                 trhip_buffer args_buffer, uint32_t args_offset_bytes);
int  trhip_cmd_begin_timer(trhip_cmdlist cl, trhip_timer t);          /* ::beginTimerQuery      */
int  trhip_cmd_end_timer(trhip_cmdlist cl, trhip_timer t);            /* ::endTimerQuery        */
int  trhip_cmd_begin_marker(trhip_cmdlist cl, const char* name);      /* ::beginMarker          */
int  trhip_cmd_end_marker(trhip_cmdlist cl);                          /* ::endMarker            */

/* nvrhi executeCommandLists (Graphic.cpp:786-830): enqueue the recorded lists, in order, on the
 * device stream.  Asynchronous; a list may be executed more than once without re-recording. */
int  trhip_queue_execute(trhip_device dev, const trhip_cmdlist* lists, uint32_t num_lists);

/* ---- timer queries: nvrhi::TimerQuery (RenderGraph.cpp:269-285) ----------------------------- */
int  trhip_timer_create(trhip_device dev, trhip_timer* out);
void trhip_timer_release(trhip_timer t);
int  trhip_timer_get_ms(trhip_timer t, float* ms);   /* waits for the end event (getTimerQueryTime) */

/* ---- per-shader GPU profile: PROFILE_GPU_SCOPED in AddComputePass (Graphic.cpp:899) ----------
 * When enabled, every dispatch executed is bracketed by HIP events on the device stream and
 * accumulated under its shader name (sub-kernels under "name#kernel").  Off by default. */
int  trhip_profile_enable(trhip_device dev, int enabled);
/* Bracket only the dispatches accumulated under exactly `name` ("shader#kernel"); NULL or "" = all of them.  One event pair
 * per frame instead of one per launch: the frame keeps its steady-state overlap and clocks while one kernel is timed. */
int  trhip_profile_filter(trhip_device dev, const char* name);
int  trhip_profile_reset(trhip_device dev);
int  trhip_profile_count(trhip_device dev, uint32_t* n);
int  trhip_profile_entry(trhip_device dev, uint32_t index, const char** name, uint64_t* launches, double* total_ms);

/* Streams and events for callers that order work across streams themselves (the multi-GPU exchange runs on its own
 * streams next to the renderer's).  Plain wrappers: hipStreamCreateWithFlags(NonBlocking), hipEventCreateWithFlags
 * (DisableTiming), hipEventRecord, hipStreamWaitEvent, hipStreamSynchronize. */
int  trhip_stream_create(int device_index, void** out_hip_stream);
/* priority_class < 0: the device's highest stream priority, 0: default, > 0: lowest (hipStreamCreateWithPriority).
 * Streams of different priority classes never share a hardware queue; streams of one class may (HIP multiplexes them
 * round robin onto GPU_MAX_HW_QUEUES queues) and then run one after the other. */
int  trhip_stream_create_priority(int device_index, int priority_class, void** out_hip_stream);
void trhip_stream_destroy(void* hip_stream);
int  trhip_stream_synchronize(void* hip_stream);
int  trhip_event_create(int device_index, void** out_hip_event);
void trhip_event_destroy(void* hip_event);
int  trhip_event_record(void* hip_event, void* hip_stream);
int  trhip_stream_wait_event(void* hip_stream, void* hip_event);

/* Multi-GPU late phase.  The reference sizes the late instance cull from the late-list length
 * (gpuculling.hlsl:182-195, Q1: ceil(count/64) groups of 32 threads).  With the instance list sharded
 * over ranks that rule has to see the WHOLE scene's late list: every rank all-gathers its late count,
 * then this kernel writes info = { sum of counts[0..rank), sum of counts[0..world) } on hip_stream.
 * The late "gpuculling_CS_GPUCulling LATE_CULL=1" dispatch reads it from an optional SRV t4 and
 * processes exactly the entries the single-GPU dispatch would. */
int  trhip_launch_shard_late_info(void* hip_stream, const uint32_t* gathered_counts, uint32_t world, uint32_t rank, uint32_t* info);

#ifdef __cplusplus
}
#endif
#endif /* TRHIP_H_ */
