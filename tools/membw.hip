// membw.hip -- access-pattern calibration for the MeshletData stream (32-byte AoS records).
// Build: hipcc --offload-arch=gfx950 -O3 tools/membw.hip -o tools/membw ; run on the GPU box.
// Prints achieved GB/s (bytes of the whole buffer / time) for each pattern; run under
// rocprofv3 --pmc FETCH_SIZE to calibrate the counter against the known byte count.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

struct Rec { float4 a; uint32_t b, c, d, e; };

// A: fully coalesced float4 stream over the buffer
__global__ __launch_bounds__(256) void kCoalesced(const float4* p, size_t n16, float* out)
{
    float acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
// B: one record per lane: 16 B + 4 B at 32-byte stride (what the cull kernel does)
__global__ __launch_bounds__(256) void kStride20(const Rec* p, size_t n, float* out)
{
    float acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { float4 v = p[i].a; uint32_t b = p[i].b; acc += v.x + v.y + v.z + v.w + (float)b; }
    if (acc == 123.456f) out[0] = acc;
}
// C: one record per lane, all 32 B (two dwordx4)
__global__ __launch_bounds__(256) void kStride32(const Rec* p, size_t n, float* out)
{
    float acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 v = p[i].a; float4 w = *reinterpret_cast<const float4*>(&p[i].b); acc += v.x + v.y + v.z + v.w + w.x + w.y + w.z + w.w; }
    if (acc == 123.456f) out[0] = acc;
}
// D: per wave a 2-KB block read as two coalesced 1-KB instructions (lane l: chunk l and chunk 64+l)
__global__ __launch_bounds__(256) void kWaveBlock(const float4* p, size_t nBlocks2k, float* out)
{
    float acc = 0;
    const size_t wave = (blockIdx.x * 256ull + threadIdx.x) >> 6, lane = threadIdx.x & 63, nw = ((size_t)gridDim.x * 256) >> 6;
    for (size_t b = wave; b < nBlocks2k; b += nw) { float4 v = p[b * 128 + lane]; float4 w = p[b * 128 + 64 + lane]; acc += v.x + v.y + v.z + v.w + w.x + w.y + w.z + w.w; }
    if (acc == 123.456f) out[0] = acc;
}
// E: like B but only the 16-B sphere (stride 32)
__global__ __launch_bounds__(256) void kStride16(const Rec* p, size_t n, float* out)
{
    float acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { float4 v = p[i].a; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}

int main(int argc, char** argv)
{
    size_t bytes = (argc > 1 ? atof(argv[1]) : 1.6) * 1e9;
    bytes &= ~size_t(2047);
    void* buf; float* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 4));
    CK(hipMemset(buf, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grids[] = { 2048, 4096, 8192 };
    for (int grid : grids) {
        for (int k = 0; k < 5; ++k) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0));
                switch (k) {
                case 0: hipLaunchKernelGGL(kCoalesced, dim3(grid), dim3(256), 0, 0, (const float4*)buf, bytes / 16, out); break;
                case 1: hipLaunchKernelGGL(kStride20, dim3(grid), dim3(256), 0, 0, (const Rec*)buf, bytes / 32, out); break;
                case 2: hipLaunchKernelGGL(kStride32, dim3(grid), dim3(256), 0, 0, (const Rec*)buf, bytes / 32, out); break;
                case 3: hipLaunchKernelGGL(kWaveBlock, dim3(grid), dim3(256), 0, 0, (const float4*)buf, bytes / 2048, out); break;
                case 4: hipLaunchKernelGGL(kStride16, dim3(grid), dim3(256), 0, 0, (const Rec*)buf, bytes / 32, out); break;
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const char* names[] = { "coalesced16", "stride32_16+4", "stride32_16+16", "waveblock2k", "stride32_16" };
            printf("grid %5d %-16s %8.3f ms  %8.1f GB/s (buffer bytes / time)\n", grid, names[k], best, bytes / best / 1e6);
        }
    }
    return 0;
}
