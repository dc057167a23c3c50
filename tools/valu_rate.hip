// valu_rate.hip -- what one vector instruction costs a gfx950 SIMD, by instruction kind and by waves per SIMD.
// Settles the issue model behind the cull kernel's "VALU time" (DESIGN.md section 5): is a wave64 v_fma_f32 2 or 4
// cycles of a SIMD, is a packed fp32 instruction one or two of those, what do v_rcp_f32 / v_rsq_f32 cost.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate ; run on the GPU box.
// Every wave issues kIters x 16 INDEPENDENT instructions of one kind (16 accumulators, no dependency closer than 16
// instructions) between two s_memtime stamps; a workgroup of 256 x W threads puts W waves on each SIMD of its CU (one
// workgroup per CU).  Printed: shader cycles per wave-instruction seen by ONE wave, and cycles of the SIMD per
// instruction issued on it (= the former / W): the second is the price in a throughput model.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kIters = 512;

#define REP16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

enum Kind { FMA = 0, PKFMA, RCP, RSQ, MUL, PKMUL, FMAC, CNDMASK, MIX_PK_SCALAR, SQRT, CMP, ADD_U32, BCNT, DEP_FMA, DEP_PKFMA, CMP_SGPR, KINDS };
static const char* kNames[KINDS] = { "v_fma_f32", "v_pk_fma_f32", "v_rcp_f32", "v_rsq_f32", "v_mul_f32", "v_pk_mul_f32", "v_fmac_f32 (VOP2)",
                                     "v_cndmask_b32 (reads vcc)", "v_pk_fma_f32 + v_fma_f32 alternating", "v_sqrt_f32", "v_cmp_lt_f32 (-> vcc)", "v_add_u32", "v_bcnt_u32_b32",
                                     "v_fma_f32, ONE dependent chain", "v_pk_fma_f32, ONE dependent chain", "v_cmp_lt_f32_e64 (-> sgpr pair)" };

template <int K>
__global__ __launch_bounds__(1024) void rateKernel(unsigned long long* stamps, float seed, float* sink)
{
    float r[16];
    v2f p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { r[i] = seed + (float)i + (float)threadIdx.x * 1e-3f; p[i] = v2f{ r[i], r[i] + 0.5f }; }
    const float a = 1.0000001f, b = 1e-9f;
    const v2f a2 = { a, a }, b2 = { b, b };
    uint32_t u = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll 1
    for (int it = 0; it < kIters; ++it) {
        if (K == FMA) {
#define M(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
            REP16(M)
#undef M
        } else if (K == PKFMA) {
#define M(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(a2), "v"(b2));
            REP16(M)
#undef M
        } else if (K == RCP) {
#define M(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
            REP16(M)
#undef M
        } else if (K == RSQ) {
#define M(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(r[i]));
            REP16(M)
#undef M
        } else if (K == SQRT) {
#define M(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(r[i]));
            REP16(M)
#undef M
        } else if (K == MUL) {
#define M(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
            REP16(M)
#undef M
        } else if (K == PKMUL) {
#define M(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(a2));
            REP16(M)
#undef M
        } else if (K == FMAC) {
#define M(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
            REP16(M)
#undef M
        } else if (K == CNDMASK) {
#define M(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a));
            REP16(M)
#undef M
        } else if (K == MIX_PK_SCALAR) {
#define M(i) asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %4, %5" : "+v"(p[i]), "+v"(r[i]) : "v"(a2), "v"(b2), "v"(a), "v"(b));
            REP16(M)
#undef M
        } else if (K == CMP) {
#define M(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(r[i]), "v"(a) : "vcc");
            REP16(M)
#undef M
        } else if (K == ADD_U32) {
#define M(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(u));
            REP16(M)
#undef M
        } else if (K == DEP_FMA) {
#define M(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[0]) : "v"(a), "v"(b));
            REP16(M)
#undef M
        } else if (K == DEP_PKFMA) {
#define M(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(a2), "v"(b2));
            REP16(M)
#undef M
        } else if (K == CMP_SGPR) {
#define M(i) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" :: "v"(r[i]), "v"(a) : "s20", "s21");
            REP16(M)
#undef M
        } else if (K == BCNT) {
#define M(i) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(r[i]) : "v"(u));
            REP16(M)
#undef M
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += r[i] + p[i].x + p[i].y;
    if (acc == 123.456f) sink[0] = acc;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        stamps[2 * w] = t0; stamps[2 * w + 1] = t1;
    }
}

template <int K>
static void run(int wavesPerSimd, unsigned long long* dStamps, float* sink, int numCUs)
{
    const int threads = 256 * std::min(wavesPerSimd, 4);   // 1024 threads = the most a workgroup holds: 4 waves per SIMD
    const int blocksPerCU = wavesPerSimd <= 4 ? 1 : wavesPerSimd / 4;
    const int grid = numCUs * blocksPerCU;
    const int wavesPerBlock = threads / 64;
    std::vector<unsigned long long> h(2ull * grid * wavesPerBlock);
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(rateKernel<K>, dim3(grid), dim3(threads), 0, 0, dStamps, 1.0f, sink);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), dStamps, h.size() * 8, hipMemcpyDeviceToHost));
        // per workgroup: last end - first start; median over the workgroups
        std::vector<double> spans;
        for (int b = 0; b < grid; ++b) {
            unsigned long long s = ~0ull, e = 0;
            for (int w = 0; w < wavesPerBlock; ++w) { s = std::min(s, h[2ull * (b * wavesPerBlock + w)]); e = std::max(e, h[2ull * (b * wavesPerBlock + w) + 1]); }
            spans.push_back((double)(e - s));
        }
        std::sort(spans.begin(), spans.end());
        best = std::min(best, spans[spans.size() / 2]);
    }
    const double instr = (double)kIters * 16 * (K == MIX_PK_SCALAR ? 2 : 1);
    // 8 waves per SIMD = two 1024-thread workgroups per CU, assumed to run side by side (the span is one workgroup's)
    printf("%-40s waves/SIMD %d : %6.2f cycles per wave-instruction seen by a wave, %5.2f SIMD cycles per instruction issued\n",
           kNames[K], wavesPerSimd, best / instr, best / instr / wavesPerSimd);
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int numCUs = prop.multiProcessorCount;
    printf("%s, %d CUs, clock %d MHz\n", prop.name, numCUs, prop.clockRate / 1000);
    unsigned long long* dStamps; float* sink;
    CK(hipMalloc(&dStamps, 2ull * 8 * numCUs * 2 * 16)); CK(hipMalloc(&sink, 4));
    for (int w : { 1, 2, 3, 4 }) {
        run<FMA>(w, dStamps, sink, numCUs);
        run<FMAC>(w, dStamps, sink, numCUs);
        run<MUL>(w, dStamps, sink, numCUs);
        run<PKFMA>(w, dStamps, sink, numCUs);
        run<PKMUL>(w, dStamps, sink, numCUs);
        run<MIX_PK_SCALAR>(w, dStamps, sink, numCUs);
        run<RCP>(w, dStamps, sink, numCUs);
        run<RSQ>(w, dStamps, sink, numCUs);
        run<SQRT>(w, dStamps, sink, numCUs);
        run<CNDMASK>(w, dStamps, sink, numCUs);
        run<CMP>(w, dStamps, sink, numCUs);
        run<ADD_U32>(w, dStamps, sink, numCUs);
        run<BCNT>(w, dStamps, sink, numCUs);
        run<CMP_SGPR>(w, dStamps, sink, numCUs);
        run<DEP_FMA>(w, dStamps, sink, numCUs);
        run<DEP_PKFMA>(w, dStamps, sink, numCUs);
        printf("\n");
    }
    return 0;
}
