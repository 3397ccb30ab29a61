// tools/gridsync_probe.hip -- what a device-wide barrier between co-resident workgroups costs on this MI355X, and whether
// data handed from one workgroup to another across it arrives: the building block of the persistent bottom solver
// (k_box_bicgstab, kernels.hip).  Every spin loop gives up after a bounded number of polls and raises a flag.
//   hipcc -O3 --offload-arch=gfx950 tools/gridsync_probe.hip -o tools/gridsync_probe && tools/gridsync_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr unsigned SPIN_MAX = 1u << 24;

// MODE 0: fences at agent scope (what __threadfence() gives: L2 write-back + invalidate around the counter)
// MODE 1: data moved with agent-scope relaxed atomics (write-through stores, cache-bypassing loads), the barrier itself only
//         waits for the workgroup's outstanding memory operations (no whole-cache write-back / invalidate)
template <int MODE>
__device__ __forceinline__ bool grid_barrier(unsigned* cnt, unsigned target, unsigned* abort_flag)
{
    if (MODE == 0) __threadfence();
    else __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) expcnt(0) lgkmcnt(0): this wave's write-through stores have been acknowledged
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        if (MODE == 0) {
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > SPIN_MAX || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = false; break; }
            }
        } else {
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > SPIN_MAX || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = false; break; }
            }
        }
        if (!ok) __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __shared__ int s_ok;
    if (threadIdx.x == 0) s_ok = ok;
    __syncthreads();
    if (MODE == 0) __threadfence();
    return s_ok != 0;
}

// every round: each workgroup writes round-stamped values into its slab of buf, barrier, reads the next workgroup's slab and
// checks the stamp (errors counted), barrier.  rounds * 2 barriers per launch.
template <int MODE>
__global__ __launch_bounds__(1024) void k_ring(double* buf, int per, int rounds, unsigned* cnt, unsigned* abort_flag,
                                               unsigned long long* errors)
{
    const int b = blockIdx.x, nb = gridDim.x;
    unsigned target = 0;
    unsigned long long bad = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int i = threadIdx.x; i < per; i += blockDim.x) {
            const double v = (double)(r * 1000003 + b * 1009 + i);
            if (MODE == 0) buf[(long long)b * per + i] = v;
            else __hip_atomic_store(buf + (long long)b * per + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        target += nb;
        if (!grid_barrier<MODE>(cnt, target, abort_flag)) return;
        const int nbx = (b + 1) % nb;
        for (int i = threadIdx.x; i < per; i += blockDim.x) {
            const double want = (double)(r * 1000003 + nbx * 1009 + i);
            const double got = MODE == 0 ? buf[(long long)nbx * per + i]
                                         : __hip_atomic_load(buf + (long long)nbx * per + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (got != want) ++bad;
        }
        target += nb;
        if (!grid_barrier<MODE>(cnt, target, abort_flag)) return;
    }
    if (bad) atomicAdd(errors, bad);
}

template <int MODE>
static void run(int nwg, int threads, int per, int rounds)
{
    double* buf;
    unsigned *cnt, *abortf;
    unsigned long long* err;
    CK(hipMalloc(&buf, sizeof(double) * (size_t)nwg * per));
    CK(hipMalloc(&cnt, 4));
    CK(hipMalloc(&abortf, 4));
    CK(hipMalloc(&err, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    unsigned habort = 0;
    unsigned long long herr = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(cnt, 0, 4));
        CK(hipMemset(abortf, 0, 4));
        CK(hipMemset(err, 0, 8));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_ring<MODE>, dim3(nwg), dim3(threads), 0, 0, buf, per, rounds, cnt, abortf, err);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
        unsigned a;
        unsigned long long e;
        CK(hipMemcpy(&a, abortf, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&e, err, 8, hipMemcpyDeviceToHost));
        habort |= a;
        herr += e;
    }
    printf("{\"mode\": %d, \"workgroups\": %d, \"threads\": %d, \"doubles_per_wg\": %d, \"us_per_barrier\": %.3f, \"errors\": %llu, \"aborted\": %u}\n",
           MODE, nwg, threads, per, best * 1e3 / (2.0 * rounds), herr, habort);
    CK(hipFree(buf)); CK(hipFree(cnt)); CK(hipFree(abortf)); CK(hipFree(err));
}

int main()
{
    const int rounds = 500;
    for (int nwg : {1, 8, 16, 64, 128, 256})
        for (int per : {64, 1024, 4096}) {
            run<0>(nwg, 1024, per, rounds);
            run<1>(nwg, 1024, per, rounds);
        }
    for (int nwg : {16, 64}) {
        run<0>(nwg, 256, 1024, rounds);
        run<1>(nwg, 256, 1024, rounds);
    }
    return 0;
}
