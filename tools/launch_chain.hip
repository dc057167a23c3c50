// launch_chain.hip -- what one small dependent kernel costs on this box: plain stream launches vs a hipGraph of the same
// chain, and a chain whose links are grid barriers inside ONE cooperative kernel.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/launch_chain.hip -o /tmp/launch_chain && /tmp/launch_chain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void tiny(uint32_t* p, uint32_t n) { uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1u; }

// one kernel, `links` phases separated by a grid barrier (all blocks resident: grid <= CUs * 2)
__global__ void chained(uint32_t* p, uint32_t n, uint32_t* bar, uint32_t links)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t l = 0; l < links; ++l) {
        if (i < n) p[(i + l * 977u) % n] += 1u;
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const uint32_t target = (l + 1u) * gridDim.x;
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const uint32_t n = 1u << 16;
    uint32_t *p, *bar;
    CK(hipMalloc(&p, n * 4)); CK(hipMemset(p, 0, n * 4)); CK(hipMalloc(&bar, 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int chain = 30, reps = 200;
    for (int blocks : { 1, 256, 1024 }) {
        for (int w = 0; w < 2; ++w) {
            CK(hipStreamSynchronize(s));
            double t = now();
            for (int r = 0; r < reps; ++r)
                for (int k = 0; k < chain; ++k) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(64), 0, s, p, n);
            double sub = now() - t;
            CK(hipStreamSynchronize(s));
            double all = now() - t;
            if (w) printf("stream  blocks %4d: %.2f us per kernel (host submit %.2f us)\n", blocks, all / (reps * chain) * 1e6, sub / (reps * chain) * 1e6);
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int k = 0; k < chain; ++k) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(64), 0, s, p, n);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 2; ++w) {
            CK(hipStreamSynchronize(s));
            double t = now();
            for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
            double sub = now() - t;
            CK(hipStreamSynchronize(s));
            double all = now() - t;
            if (w) printf("graph   blocks %4d: %.2f us per kernel (host submit %.2f us)\n", blocks, all / (reps * chain) * 1e6, sub / (reps * chain) * 1e6);
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    for (int blocks : { 256, 512 }) {
        for (int w = 0; w < 2; ++w) {
            CK(hipMemsetAsync(bar, 0, 4, s));
            CK(hipStreamSynchronize(s));
            double t = now();
            hipLaunchKernelGGL(chained, dim3(blocks), dim3(256), 0, s, p, n, bar, 200u);
            CK(hipStreamSynchronize(s));
            double all = now() - t;
            if (w) printf("barrier blocks %4d x 256 threads: %.2f us per link (one kernel, 200 grid barriers)\n", blocks, all / 200 * 1e6);
        }
    }
    return 0;
}
