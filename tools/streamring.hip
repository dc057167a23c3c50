// streamring.hip -- the MeshletData stream of the cull kernel in isolation: what the ACCESS PATTERN (tile order vs list
// order), the ring depth, the batch bubble and the VALU work beside it each cost, without the cull arithmetic.
// Build: hipcc --offload-arch=gfx950 -O3 tools/streamring.hip -o tools/streamring ; run on the GPU box.
//
// Model of k_basepass_as.hip::meshletCullKernel: persistent workgroups of 4 waves, G workgroups per CU; a workgroup walks
// windows of 4 x S consecutive 2-KB chunks of a chunk list (a chunk = the two records of a wave step); wave w takes chunk
// 4 s + w at step s; the chunk goes memory -> LDS by two global_load_lds_dwordx4 (nt), D steps ahead, hand-counted vmcnt;
// every lane then reads 16 + 4 bytes of it from LDS and runs V x 16 independent v_fma_f32.  At a window's end the ring
// drains (vmcnt(0)) and the next window's chunk indices are fetched (one dependent load, as the kernel's prologue).
//
// Chunk lists over a 3.2-GB buffer (100 M meshlets x 32 B; an "instance" = 4 KB = 2 chunks; 56 % of the instances kept):
//   linear   every chunk in address order (what tools/membw.hip streams)
//   list     kept instances in address order (TRHIP_AS_NO_PERM=1)
//   tileN    kept instances stable-sorted by a random tile id in [0, N)   (the shipped order: N = 256)
//   random   kept instances shuffled
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

template <int NT>
__device__ __forceinline__ void issue2k(char* slotLds, const char* src, uint32_t lane)
{
    const char* pa = src + 16u * lane;
    const char* pb = pa + 1024;
    const uint32_t ldsOff = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)slotLds;
    if (NT)
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, off nt\n\ts_add_u32 m0, %2, 0x400\n\tglobal_load_lds_dwordx4 %1, off nt"
                     :: "v"(pa), "v"(pb), "s"(ldsOff) : "memory", "m0", "scc");
    else
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, off\n\ts_add_u32 m0, %2, 0x400\n\tglobal_load_lds_dwordx4 %1, off"
                     :: "v"(pa), "v"(pb), "s"(ldsOff) : "memory", "m0", "scc");
}

template <int N> __device__ __forceinline__ void waitVm();
#define WV(n) template <> __device__ __forceinline__ void waitVm<n>() { asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); }
WV(0) WV(2) WV(3) WV(4) WV(5) WV(6) WV(7) WV(8) WV(9) WV(10) WV(11) WV(12) WV(14) WV(15) WV(16)
#undef WV

constexpr int kMaxSteps = 64;

// D ring slots (2 KB each) per wave; S steps per window; V: groups of 16 v_fma_f32 per step.
// LK: the table lookup of the occlusion test beside the stream -- one 2-byte load per lane and step from an 8.5-MB table,
//   the 64 lanes of a step spread over ~20 lines of a region that depends on the chunk (the kernel's footprint table in
//   8 x 8 blocks), issued after 5/6 of the step's VALU work, in the same vmcnt queue as the ring's loads:
//   0 = none; 1 = consumed in the same step behind vmcnt(2) (the shipped kernel); 2 = consumed one step LATER (the value
//   rides through the next step's arithmetic in a register).
template <int D, int NT, int LK>
__global__ __launch_bounds__(256) void ringKernel(const char* buf, const uint32_t* chunks, uint32_t nChunks, int S, int V, const uint16_t* table, float* sink)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* ring = lds + wave * (D * 2048);
    uint32_t* idx = reinterpret_cast<uint32_t*>(lds + 4 * D * 2048) + wave * (kMaxSteps + 16);
    const uint32_t winSize = 4u * S;
    const uint32_t nWin = (nChunks + winSize - 1) / winSize;
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = (float)i;
    const float a = 1.0000001f;
    auto loadIdx = [&](uint32_t win) -> uint32_t {
        const uint64_t e = (uint64_t)win * winSize + 4u * lane + wave;
        return (win < nWin && lane < (uint32_t)S && e < nChunks) ? chunks[e] : 0xFFFFFFFFu;
    };
    uint32_t next = loadIdx(blockIdx.x);
    for (uint32_t win = blockIdx.x; win < nWin; win += gridDim.x) {
        const uint32_t cur = next;
        next = loadIdx(win + gridDim.x);                       // in flight during this window (the kernel's entry prefetch)
        if (lane < (uint32_t)S + D) idx[lane] = lane < (uint32_t)S ? cur : 0xFFFFFFFFu;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        auto src = [&](uint32_t s) -> const char* { const uint32_t c = idx[s]; return buf + (size_t)(c == 0xFFFFFFFFu ? 0u : c) * 2048u; };
#pragma unroll
        for (int k = 0; k < D; ++k) issue2k<NT>(ring + 2048 * k, src(k), lane);
        uint32_t pending = 0;                                   // LK == 2: the previous step's lookup, still in flight
        bool havePending = false;
        for (int s0 = 0; s0 < S; s0 += D) {
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const int s = s0 + k;
                char* slot = ring + 2048 * k;
                // loads outstanding here, oldest first: LK 0/1: {slot s, ..., slot s+D-1} x 2; LK 2: the same with the previous
                // step's lookup in front of the youngest slot
                if (LK == 2 && havePending) waitVm<2 * (D - 1) + 1>(); else waitVm<2 * (D - 1)>();
                const v4f sph = *reinterpret_cast<const v4f*>(slot + lane * 32u);
                const uint32_t cone = *reinterpret_cast<const uint32_t*>(slot + lane * 32u + 16u);
                r[0] += sph.x; r[1] += sph.y; r[2] += sph.z; r[3] += sph.w + (float)cone;
                const int vEarly = LK ? V - V / 6 : V;
                for (int v = 0; v < vEarly; ++v) {
#define M(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r[i]) : "v"(a));
                    M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#undef M
                }
                uint32_t bits = 0;
                if (LK) {
                    // region of the chunk: 64 lines (8 KB) somewhere in the table; lane -> one of ~20 lines of it, any 2-byte entry
                    const uint32_t c = idx[s] == 0xFFFFFFFFu ? 0u : idx[s];
                    const uint32_t region = (c * 2654435761u) >> 12 & 0xFFFFu;                        // 65 536 regions x 128 B = the 8.4-MB table
                    const uint32_t line = (region + ((lane * 11u + (c & 7u)) % 20u) * 3u) & 0xFFFFu;
                    const uint16_t* e = table + line * 64u + (lane & 63u);
                    asm volatile("global_load_ushort %0, %1, off" : "+v"(bits) : "v"(e) : "memory");
                }
                issue2k<NT>(slot, src(s + D), lane);            // past the window: the padding entries (chunk 0), as the kernel does
                if (LK == 1) {
                    asm volatile("s_waitcnt vmcnt(2)" : "+v"(bits) :: "memory");
                    r[4] += (float)bits;
                }
                if (LK == 2) {
                    if (havePending) {                          // the previous step's lookup; younger than it: its step's prefetch, this lookup, this prefetch
                        asm volatile("s_waitcnt vmcnt(5)" : "+v"(pending) :: "memory");
                        r[4] += (float)pending;
                    }
                    pending = bits; havePending = true;
                }
                for (int v = vEarly; v < V; ++v) {
#define M(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r[i]) : "v"(a));
                    M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#undef M
                }
            }
        }
        if (LK == 2) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(pending) :: "memory"); r[4] += (float)pending; }
        waitVm<0>();                                            // the padding prefetches: nothing may land in the ring later
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    waitVm<0>();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += r[i];
    if (acc == 123.456f) sink[0] = acc;
}

// Plain register streaming of the same chunk list (no LDS): 2 x dwordx4 per lane and chunk, U chunks in flight per wave
template <int U>
__global__ __launch_bounds__(256) void regKernel(const char* buf, const uint32_t* chunks, uint32_t nChunks, float* sink)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waveId = (blockIdx.x * 256u + threadIdx.x) >> 6, nWaves = (gridDim.x * 256u) >> 6;
    float acc = 0.f;
    for (uint32_t c0 = waveId * U; c0 < nChunks; c0 += nWaves * U) {
        v4f v[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t c = c0 + u < nChunks ? chunks[c0 + u] : chunks[0];
            const v4f* p = reinterpret_cast<const v4f*>(buf + (size_t)c * 2048u);
            v[u][0] = __builtin_nontemporal_load(p + lane); v[u][1] = __builtin_nontemporal_load(p + 64 + lane);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u][0].x + v[u][0].w + v[u][1].y + v[u][1].z;
    }
    if (acc == 123.456f) sink[0] = acc;
}

static std::vector<uint32_t> makeList(const std::string& kind, uint32_t nInst, double keep, uint32_t seed)
{
    std::mt19937 rng(seed);
    std::vector<uint32_t> inst;
    if (kind == "linear") { inst.resize(nInst); std::iota(inst.begin(), inst.end(), 0u); }
    else {
        std::uniform_real_distribution<double> u(0, 1);
        for (uint32_t i = 0; i < nInst; ++i) if (u(rng) < keep) inst.push_back(i);
        if (kind.rfind("tile", 0) == 0) {
            const uint32_t nt = (uint32_t)atoi(kind.c_str() + 4);
            std::vector<uint32_t> tile(nInst);
            for (auto& t : tile) t = rng() % nt;
            std::stable_sort(inst.begin(), inst.end(), [&](uint32_t x, uint32_t y) { return tile[x] < tile[y]; });
        } else if (kind == "random") std::shuffle(inst.begin(), inst.end(), rng);
    }
    std::vector<uint32_t> chunks;
    chunks.reserve(inst.size() * 2);
    for (uint32_t i : inst) { chunks.push_back(2 * i); chunks.push_back(2 * i + 1); }
    return chunks;
}

static const uint16_t* g_table = nullptr;
template <int D, int NT, int LK = 0>
static double timeRing(const char* buf, const uint32_t* dChunks, uint32_t n, int S, int V, int wgPerCU, int numCUs, float* sink)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t ldsBytes = 4 * D * 2048 + 4 * (kMaxSteps + 16) * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ringKernel<D, NT, LK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((ringKernel<D, NT, LK>), dim3(numCUs * wgPerCU), dim3(256), ldsBytes, 0, buf, dChunks, n, S, V, g_table, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    CK(hipGetLastError());
    return best;
}

template <int U>
static double timeReg(const char* buf, const uint32_t* dChunks, uint32_t n, int wgPerCU, int numCUs, float* sink)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((regKernel<U>), dim3(numCUs * wgPerCU), dim3(256), 0, 0, buf, dChunks, n, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    return best;
}

int main(int argc, char** argv)
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int numCUs = prop.multiProcessorCount;
    const uint32_t nInst = 781250;                              // C3: 781 250 instances x 128 meshlets x 32 B = 3.2 GB
    const size_t bytes = (size_t)nInst * 4096;
    char* buf; float* sink; uint32_t* dChunks;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&dChunks, (size_t)nInst * 2 * 4));
    CK(hipMemset(buf, 1, bytes));
    { uint16_t* t; CK(hipMalloc(&t, 65536 * 128)); CK(hipMemset(t, 0, 65536 * 128)); g_table = t; }
    const bool quick = argc > 1 && !strcmp(argv[1], "lookup");   // only the lookup / bubble rows, tile256 and list
    const char* kinds[] = { "linear", "list", "tile16", "tile256", "tile1024", "random" };
    printf("%s, %d CUs; buffer %.2f GB; GB/s = chunk bytes / best-of-4 time\n", prop.name, numCUs, bytes / 1e9);
    for (const char* kind : kinds) {
        if (quick && strcmp(kind, "tile256") && strcmp(kind, "list")) continue;
        const std::vector<uint32_t> chunks = makeList(kind, nInst, 0.56, 12345u);
        const uint32_t n = (uint32_t)chunks.size();
        CK(hipMemcpy(dChunks, chunks.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        const double gb = (double)n * 2048 / 1e9;
        auto show = [&](const char* what, double ms) { printf("%-9s %-46s %7.3f ms %8.1f GB/s\n", kind, what, ms, gb / ms * 1e3); fflush(stdout); };
        if (quick) {
            show("ring D3 S15 V0  4wg/CU", timeRing<3, 1>(buf, dChunks, n, 15, 0, 4, numCUs, sink));
            show("ring D3 S15 V14 4wg/CU", timeRing<3, 1>(buf, dChunks, n, 15, 14, 4, numCUs, sink));
            show("ring D3 S60 V14 4wg/CU (bubble 4x rarer)", timeRing<3, 1>(buf, dChunks, n, 60, 14, 4, numCUs, sink));
            show("ring D3 S15 V18 4wg/CU", timeRing<3, 1>(buf, dChunks, n, 15, 18, 4, numCUs, sink));
            show("ring D3 S60 V18 4wg/CU (bubble 4x rarer)", timeRing<3, 1>(buf, dChunks, n, 60, 18, 4, numCUs, sink));
            show("ring D3 S15 V18 4wg/CU + lookup, same step", timeRing<3, 1, 1>(buf, dChunks, n, 15, 18, 4, numCUs, sink));
            show("ring D3 S15 V18 4wg/CU + lookup, next step", timeRing<3, 1, 2>(buf, dChunks, n, 15, 18, 4, numCUs, sink));
            show("ring D3 S60 V18 4wg/CU + lookup, same step", timeRing<3, 1, 1>(buf, dChunks, n, 60, 18, 4, numCUs, sink));
            show("ring D3 S60 V18 4wg/CU + lookup, next step", timeRing<3, 1, 2>(buf, dChunks, n, 60, 18, 4, numCUs, sink));
            show("ring D2 S16 V18 4wg/CU + lookup, same step", timeRing<2, 1, 1>(buf, dChunks, n, 16, 18, 4, numCUs, sink));
            show("ring D2 S16 V18 4wg/CU + lookup, next step", timeRing<2, 1, 2>(buf, dChunks, n, 16, 18, 4, numCUs, sink));
            show("ring D2 S16 V18 6wg/CU + lookup, same step", timeRing<2, 1, 1>(buf, dChunks, n, 16, 18, 6, numCUs, sink));
            show("ring D2 S16 V18 6wg/CU + lookup, next step", timeRing<2, 1, 2>(buf, dChunks, n, 16, 18, 6, numCUs, sink));
            show("ring D3 S15 V18 5wg/CU + lookup, same step", timeRing<3, 1, 1>(buf, dChunks, n, 15, 18, 5, numCUs, sink));
            show("ring D3 S15 V18 6wg/CU + lookup, same step", timeRing<3, 1, 1>(buf, dChunks, n, 15, 18, 6, numCUs, sink));
            show("ring D3 S15 V18 6wg/CU + lookup, next step", timeRing<3, 1, 2>(buf, dChunks, n, 15, 18, 6, numCUs, sink));
            show("ring D3 S15 V18 6wg/CU", timeRing<3, 1>(buf, dChunks, n, 15, 18, 6, numCUs, sink));
            show("ring D3 S15 V14 4wg/CU + lookup, same step", timeRing<3, 1, 1>(buf, dChunks, n, 15, 14, 4, numCUs, sink));
            show("ring D3 S15 V14 4wg/CU + lookup, next step", timeRing<3, 1, 2>(buf, dChunks, n, 15, 14, 4, numCUs, sink));
            printf("\n");
            continue;
        }
        show("registers, 2 chunks in flight/wave, 8 wg/CU", timeReg<2>(buf, dChunks, n, 8, numCUs, sink));
        show("registers, 4 chunks in flight/wave, 4 wg/CU", timeReg<4>(buf, dChunks, n, 4, numCUs, sink));
        show("ring D3 S15 V0  4wg/CU (shipped shape)", timeRing<3, 1>(buf, dChunks, n, 15, 0, 4, numCUs, sink));
        show("ring D3 S15 V0  4wg/CU, no nt", timeRing<3, 0>(buf, dChunks, n, 15, 0, 4, numCUs, sink));
        show("ring D3 S60 V0  4wg/CU", timeRing<3, 1>(buf, dChunks, n, 60, 0, 4, numCUs, sink));
        show("ring D2 S16 V0  4wg/CU", timeRing<2, 1>(buf, dChunks, n, 16, 0, 4, numCUs, sink));
        show("ring D4 S16 V0  4wg/CU", timeRing<4, 1>(buf, dChunks, n, 16, 0, 4, numCUs, sink));
        show("ring D6 S18 V0  3wg/CU", timeRing<6, 1>(buf, dChunks, n, 18, 0, 3, numCUs, sink));
        show("ring D8 S16 V0  2wg/CU", timeRing<8, 1>(buf, dChunks, n, 16, 0, 2, numCUs, sink));
        show("ring D3 S15 V0  2wg/CU", timeRing<3, 1>(buf, dChunks, n, 15, 0, 2, numCUs, sink));
        show("ring D3 S15 V0  6wg/CU", timeRing<3, 1>(buf, dChunks, n, 15, 0, 6, numCUs, sink));
        show("ring D3 S15 V8  4wg/CU (+128 fma/step)", timeRing<3, 1>(buf, dChunks, n, 15, 8, 4, numCUs, sink));
        show("ring D3 S15 V14 4wg/CU (+224 fma/step)", timeRing<3, 1>(buf, dChunks, n, 15, 14, 4, numCUs, sink));
        show("ring D3 S15 V20 4wg/CU (+320 fma/step)", timeRing<3, 1>(buf, dChunks, n, 15, 20, 4, numCUs, sink));
        show("ring D4 S16 V14 4wg/CU", timeRing<4, 1>(buf, dChunks, n, 16, 14, 4, numCUs, sink));
        show("ring D6 S18 V14 3wg/CU", timeRing<6, 1>(buf, dChunks, n, 18, 14, 3, numCUs, sink));
        show("ring D3 S15 V14 6wg/CU", timeRing<3, 1>(buf, dChunks, n, 15, 14, 6, numCUs, sink));
        printf("\n");
    }
    return 0;
}
