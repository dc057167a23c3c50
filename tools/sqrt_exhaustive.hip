// Exhaustive check of candidate fp32 square-root sequences against the compiler's correctly rounded sqrt, over every
// float in [2^-96, FLT_MAX] (the range in which the hot kernel may take a cheaper sequence).  Prints mismatch counts.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/sqrt_exhaustive tools/sqrt_exhaustive.hip && tools/sqrt_exhaustive
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// A: the compiler's algorithm without its range handling: hardware sqrt (<= 1 ulp), then pick among s-1ulp, s, s+1ulp
__device__ __forceinline__ float sqrtA(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    const float ed = fma_(-sd, s, x), eu = fma_(-su, s, x);
    float r = ed <= 0.0f ? sd : s;
    r = eu > 0.0f ? su : r;
    return r;
}

// B: Markstein-style: reciprocal square root, one coupled Newton step, one residual correction
__device__ __forceinline__ float sqrtB(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float g0 = x * y, h0 = 0.5f * y;
    const float r0 = fma_(-h0, g0, 0.5f);
    const float g1 = fma_(g0, r0, g0), h1 = fma_(h0, r0, h0);
    const float d1 = fma_(-g1, g1, x);
    return fma_(d1, h1, g1);
}

// C: B with a second residual correction
__device__ __forceinline__ float sqrtC(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float g0 = x * y, h0 = 0.5f * y;
    const float r0 = fma_(-h0, g0, 0.5f);
    const float g1 = fma_(g0, r0, g0), h1 = fma_(h0, r0, h0);
    const float d1 = fma_(-g1, g1, x);
    const float g2 = fma_(d1, h1, g1);
    const float d2 = fma_(-g2, g2, x);
    return fma_(d2, h1, g2);
}

__global__ void check(uint32_t first, uint64_t count, unsigned long long* bad)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long a = 0, b = 0, c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const float x = __uint_as_float(first + (uint32_t)i);
        const uint32_t ref = __float_as_uint(__builtin_sqrtf(x));
        a += __float_as_uint(sqrtA(x)) != ref;
        b += __float_as_uint(sqrtB(x)) != ref;
        c += __float_as_uint(sqrtC(x)) != ref;
    }
    if (a) atomicAdd(&bad[0], a);
    if (b) atomicAdd(&bad[1], b);
    if (c) atomicAdd(&bad[2], c);
}

int main()
{
    unsigned long long* bad = nullptr;
    hipMalloc(&bad, 3 * sizeof *bad);
    hipMemset(bad, 0, 3 * sizeof *bad);
    const uint32_t first = getenv("SQRT_FIRST") ? (uint32_t)strtoul(getenv("SQRT_FIRST"), nullptr, 16) : 0x0F800000u;   // default 2^-96
    const uint64_t count = 0x7F800000ull - first;       // up to FLT_MAX
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, first, count, bad);
    unsigned long long h[3];
    hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost);
    printf("inputs %llu  mismatches: A (sqrt + neighbour pick) %llu   B (rsq + 1 step + 1 correction) %llu   C (B + 2nd correction) %llu\n",
           (unsigned long long)count, h[0], h[1], h[2]);
    return 0;
}
