// sync_cost.hip -- what a cross-stream dependency costs the MAIN chain of a frame on this box.
// A chain of N dependent kernels of ~12 us on stream A (the frame's main chain); per link, optionally: an event recorded
// on A, a fork (stream B waits for it and runs a short kernel), a join (A waits for B), the same through the kernel's own
// completion signal (hipExtLaunchKernelGGL stop event), or through a flag in memory polled by a one-wave gate kernel.
// Reported: microseconds per link of the main chain over the plain chain.
//   hipcc --offload-arch=gfx950 -O2 tools/sync_cost.hip -o tools/sync_cost && tools/sync_cost
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// busy for `ticks` of the 100 MHz wall clock; the last block to finish (ticket) publishes `value` in *flag (if any)
__global__ void busy(unsigned long long ticks, uint32_t* ticket, uint32_t* flag, uint32_t value)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (flag && threadIdx.x == 0) {
        const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1u) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// first thing on B after a fork: has A's kernel of this link published its value?  (counts violations of the dependency)
__global__ void checkThenBusy(unsigned long long ticks, const uint32_t* flag, uint32_t value, uint32_t* violations)
{
    if (blockIdx.x == 0 && threadIdx.x == 0 && (int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - value) < 0) atomicAdd(violations, 1u);
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
}
// one wave: wait (bounded) until *flag >= value
__global__ void gate(const uint32_t* flag, uint32_t value, uint32_t* timeouts)
{
    uint32_t spins = 0;
    while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - value) < 0) {
        if (++spins > (1u << 22)) { atomicAdd(timeouts, 1u); break; }
        __builtin_amdgcn_s_sleep(8);
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const int N = 300;
    hipStream_t A, B; CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    std::vector<hipEvent_t> e(N), f(N);
    for (int i = 0; i < N; ++i) { CK(hipEventCreateWithFlags(&e[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&f[i], hipEventDisableTiming)); }
    uint32_t* mem; CK(hipMalloc(&mem, 64 * 4)); CK(hipMemset(mem, 0, 64 * 4));
    uint32_t *ticketA = mem, *flagA = mem + 16, *ticketB = mem + 32, *flagB = mem + 48, *timeouts = mem + 8, *violations = mem + 9;
    const unsigned long long longK = 3000, shortK = 300;            // 30 us, 3 us at 100 MHz: the host stays ahead of the GPU in every variant
    const dim3 grid(512), block(64);
    double base = 0;
    uint32_t epoch = 0;
    for (int variant = 0; variant <= 11; ++variant) {
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            const double t0 = now();
            for (int i = 0; i < N; ++i) {
                ++epoch;
                switch (variant) {
                case 0: hipLaunchKernelGGL(busy, grid, block, 0, A, longK, nullptr, nullptr, 0u); break;
                case 1: hipLaunchKernelGGL(busy, grid, block, 0, A, longK, nullptr, nullptr, 0u); CK(hipEventRecord(e[i], A)); break;
                case 2:   // fork
                    hipLaunchKernelGGL(busy, grid, block, 0, A, longK, nullptr, nullptr, 0u); CK(hipEventRecord(e[i], A));
                    CK(hipStreamWaitEvent(B, e[i], 0)); hipLaunchKernelGGL(busy, dim3(64), block, 0, B, shortK, nullptr, nullptr, 0u); break;
                case 3:   // fork, and the join one link later (B's work has long finished when A gets there)
                    if (i) CK(hipStreamWaitEvent(A, f[i - 1], 0));
                    hipLaunchKernelGGL(busy, grid, block, 0, A, longK, nullptr, nullptr, 0u); CK(hipEventRecord(e[i], A));
                    CK(hipStreamWaitEvent(B, e[i], 0)); hipLaunchKernelGGL(busy, dim3(64), block, 0, B, shortK, nullptr, nullptr, 0u); CK(hipEventRecord(f[i], B)); break;
                case 4:   // join only: A waits for an event of B recorded long ago (already complete)
                    if (i) CK(hipStreamWaitEvent(A, f[0], 0));
                    hipLaunchKernelGGL(busy, grid, block, 0, A, longK, nullptr, nullptr, 0u);
                    if (i == 0) { hipLaunchKernelGGL(busy, dim3(64), block, 0, B, shortK, nullptr, nullptr, 0u); CK(hipEventRecord(f[0], B)); } break;
                case 5:   // fork through the kernel's own completion signal (ONE event reused for every fork, and given to TWO launches, as the
                          // back end does; B checks that it sees the SECOND one's result)
                    hipExtLaunchKernelGGL(busy, dim3(8), block, 0, A, nullptr, e[0], 0, shortK, nullptr, nullptr, 0u);
                    hipExtLaunchKernelGGL(busy, grid, block, 0, A, nullptr, e[0], 0, longK, ticketA, flagA, epoch);
                    CK(hipStreamWaitEvent(B, e[0], 0)); hipLaunchKernelGGL(checkThenBusy, dim3(64), block, 0, B, shortK, flagA, epoch, violations); break;
                case 6:   // fork + late join, both through completion signals
                    if (i) CK(hipStreamWaitEvent(A, f[i - 1], 0));
                    hipExtLaunchKernelGGL(busy, grid, block, 0, A, nullptr, e[i], 0, longK, nullptr, nullptr, 0u);
                    CK(hipStreamWaitEvent(B, e[i], 0)); hipExtLaunchKernelGGL(busy, dim3(64), block, 0, B, nullptr, f[i], 0, shortK, nullptr, nullptr, 0u); break;
                case 9:   // fork per link, the join THREE links later: satisfied long before A's queue reaches it, but not when the host submits it
                    if (i >= 3) CK(hipStreamWaitEvent(A, f[i - 3], 0));
                    hipLaunchKernelGGL(busy, grid, block, 0, A, longK, nullptr, nullptr, 0u); CK(hipEventRecord(e[i], A));
                    CK(hipStreamWaitEvent(B, e[i], 0)); hipLaunchKernelGGL(busy, dim3(64), block, 0, B, shortK, nullptr, nullptr, 0u); CK(hipEventRecord(f[i], B)); break;
                case 10:  // the same through completion signals
                    if (i >= 3) CK(hipStreamWaitEvent(A, f[i - 3], 0));
                    hipExtLaunchKernelGGL(busy, grid, block, 0, A, nullptr, e[i], 0, longK, nullptr, nullptr, 0u);
                    CK(hipStreamWaitEvent(B, e[i], 0)); hipExtLaunchKernelGGL(busy, dim3(64), block, 0, B, nullptr, f[i], 0, shortK, nullptr, nullptr, 0u); break;
                case 11:  // only the late join (no fork on A): B runs free, A waits three links later for B's completion signal
                    if (i >= 3) CK(hipStreamWaitEvent(A, f[i - 3], 0));
                    hipLaunchKernelGGL(busy, grid, block, 0, A, longK, nullptr, nullptr, 0u);
                    hipExtLaunchKernelGGL(busy, dim3(64), block, 0, B, nullptr, f[i], 0, shortK, nullptr, nullptr, 0u); break;
                case 7:   // fork through a flag in memory: A's kernel publishes, a gate wave on B polls
                    hipLaunchKernelGGL(busy, grid, block, 0, A, longK, ticketA, flagA, epoch);
                    hipLaunchKernelGGL(gate, dim3(1), dim3(64), 0, B, flagA, epoch, timeouts); hipLaunchKernelGGL(busy, dim3(64), block, 0, B, shortK, nullptr, nullptr, 0u); break;
                case 8:   // fork + late join through flags
                    if (i) hipLaunchKernelGGL(gate, dim3(1), dim3(64), 0, A, flagB, epoch - 1u, timeouts);
                    hipLaunchKernelGGL(busy, grid, block, 0, A, longK, ticketA, flagA, epoch);
                    hipLaunchKernelGGL(gate, dim3(1), dim3(64), 0, B, flagA, epoch, timeouts); hipLaunchKernelGGL(busy, dim3(64), block, 0, B, shortK, ticketB, flagB, epoch); break;
                }
            }
            const double sub = now() - t0;
            CK(hipDeviceSynchronize());
            const double all = now() - t0;
            if (all < best) best = all;
            if (rep == 2) {
                static const char* names[] = { "plain chain on A", "+ hipEventRecord on A per link", "+ fork (record on A, B waits, short kernel on B)",
                    "+ fork and a join one link later (A waits for B's event)", "+ A waits for an already complete event per link",
                    "fork through the kernel's completion signal (hipExtLaunchKernelGGL stop event)", "fork + late join through completion signals",
                    "fork through a memory flag + gate wave on B", "fork + late join through memory flags + gate waves",
                    "fork per link, join three links later (events)", "fork per link, join three links later (completion signals)",
                    "no fork; A waits for B's completion signal of three links ago" };
                if (variant == 0) base = best;
                printf("%-86s %6.2f us per link  (+%.2f over plain; host submit %.2f us per link)\n", names[variant], best / N * 1e6, (best - base) / N * 1e6, sub / N * 1e6);
            }
        }
    }
    uint32_t to = 0; CK(hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost));
    printf("gate timeouts: %u\n", to);
    CK(hipMemcpy(&to, violations, 4, hipMemcpyDeviceToHost));
    printf("dependency violations seen by B behind a completion-signal fork: %u of %d forks x 3 repetitions\n", to, N);
    return 0;
}
