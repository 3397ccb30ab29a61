// tools/bw_probe.hip -- what this MI355X box can stream: plain copies and the fused sweep's stream mix
// (6 read streams + 1 write stream), to put the fused GSRB kernel's rate next to a measured ceiling.
//   hipcc -O3 --offload-arch=gfx950 tools/bw_probe.hip -o tools/bw_probe && tools/bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

template <int NT>
__global__ __launch_bounds__(256) void k_copy(const v2d* __restrict__ a, v2d* __restrict__ o, long long n)
{
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += 256ll * gridDim.x) {
        v2d v = NT & 1 ? __builtin_nontemporal_load(a + i) : a[i];
        if (NT & 2) __builtin_nontemporal_store(v, o + i); else o[i] = v;
    }
}
// contiguous chunk per workgroup (as a k-marching tile reads) instead of a grid stride
template <int NT>
__global__ __launch_bounds__(256) void k_copy_chunk(const v2d* __restrict__ a, v2d* __restrict__ o, long long n)
{
    const long long per = (n + gridDim.x - 1) / gridDim.x;
    const long long b = per * blockIdx.x, e = b + per < n ? b + per : n;
    for (long long i = b + threadIdx.x; i < e; i += 256) {
        v2d v = NT & 1 ? __builtin_nontemporal_load(a + i) : a[i];
        if (NT & 2) __builtin_nontemporal_store(v, o + i); else o[i] = v;
    }
}
__global__ __launch_bounds__(256) void k_read(const v2d* __restrict__ a, double* __restrict__ o, long long n)
{
    v2d s = {0.0, 0.0};
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += 256ll * gridDim.x) s += a[i];
    if (s.x + s.y == 1.2345e300) o[0] = s.x;
}
__global__ __launch_bounds__(256) void k_write(v2d* __restrict__ o, long long n)
{
    const v2d v = {1.0, 2.0};
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += 256ll * gridDim.x) o[i] = v;
}
template <int NT>
__global__ __launch_bounds__(256) void k_mix6(const v2d* __restrict__ a, const v2d* __restrict__ b, const v2d* __restrict__ c,
                                              const v2d* __restrict__ d, const v2d* __restrict__ e, const v2d* __restrict__ f,
                                              v2d* __restrict__ o, long long n)
{
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += 256ll * gridDim.x) {
        v2d v;
        if (NT & 1) v = a[i] + __builtin_nontemporal_load(b + i) + __builtin_nontemporal_load(c + i) + __builtin_nontemporal_load(d + i) +
                        __builtin_nontemporal_load(e + i) + __builtin_nontemporal_load(f + i);
        else v = a[i] + b[i] + c[i] + d[i] + e[i] + f[i];
        if (NT & 2) __builtin_nontemporal_store(v, o + i); else o[i] = v;
    }
}

// NR read streams + one write stream (how does the achievable rate fall with the number of streams?)
struct Ptrs { const v2d* r[6]; };
template <int NR>
__global__ __launch_bounds__(256) void k_mixn(Ptrs P, v2d* __restrict__ o, long long n)
{
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += 256ll * gridDim.x) {
        v2d v = P.r[0][i];
#pragma unroll
        for (int q = 1; q < NR; ++q) v += P.r[q][i];
        o[i] = v;
    }
}
// the same 48 B read + 8 B written per cell, but the six operands of a cell pair interleaved in ONE array (array of structs,
// 96 B per lane): one read stream instead of six
__global__ __launch_bounds__(256) void k_packed6(const v2d* __restrict__ a, v2d* __restrict__ o, long long n)
{
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += 256ll * gridDim.x) {
        const v2d* q = a + 6 * i;
        o[i] = q[0] + q[1] + q[2] + q[3] + q[4] + q[5];
    }
}
// ... and plane-interleaved: for every row of 128 pairs the six operands' rows follow each other (6 x 2 KB), the layout a
// k-marching kernel would stream: each wavefront still issues six fully coalesced 1 KB loads, from one 12 KB run
__global__ __launch_bounds__(256) void k_rowpacked6(const v2d* __restrict__ a, v2d* __restrict__ o, long long n)
{
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += 256ll * gridDim.x) {
        const long long row = i >> 6, lane = i & 63;
        const v2d* q = a + row * 384 + lane;
        o[i] = q[0] + q[64] + q[128] + q[192] + q[256] + q[320];
    }
}

template <class F>
static double time_ms(F f, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv)
{
    const long long cells = argc > 1 ? atoll(argv[1]) : 134217728ll;   // 512^3 doubles = 1 GiB per array
    const long long n = cells / 2;
    v2d* p[7];
    for (auto& q : p) { CK(hipMalloc(&q, cells * 8)); CK(hipMemset(q, 0, cells * 8)); }
    double* o1; CK(hipMalloc(&o1, 8));
    const int reps = 20;
    printf("{\"cells\": %lld, \"results\": [\n", cells);
    bool first = true;
    auto rep = [&](const char* name, int wgs, double bytes, double ms) {
        printf("%s {\"kernel\": \"%s\", \"workgroups\": %d, \"ms\": %.4f, \"GBs\": %.1f}", first ? "" : ",\n", name, wgs, ms, bytes / ms * 1e-6);
        first = false;
        fflush(stdout);
    };
    {
        v2d* big;
        CK(hipMalloc(&big, cells * 8 * 6));
        CK(hipMemset(big, 0, cells * 8 * 6));
        Ptrs P;
        for (int q = 0; q < 6; ++q) P.r[q] = p[q];
        for (int wgs : {1024, 4096}) {
            rep("mix1r1w", wgs, 16.0 * cells, time_ms([&] { hipLaunchKernelGGL(k_mixn<1>, dim3(wgs), dim3(256), 0, 0, P, p[6], n); }, reps));
            rep("mix2r1w", wgs, 24.0 * cells, time_ms([&] { hipLaunchKernelGGL(k_mixn<2>, dim3(wgs), dim3(256), 0, 0, P, p[6], n); }, reps));
            rep("mix3r1w", wgs, 32.0 * cells, time_ms([&] { hipLaunchKernelGGL(k_mixn<3>, dim3(wgs), dim3(256), 0, 0, P, p[6], n); }, reps));
            rep("mix4r1w", wgs, 40.0 * cells, time_ms([&] { hipLaunchKernelGGL(k_mixn<4>, dim3(wgs), dim3(256), 0, 0, P, p[6], n); }, reps));
            rep("mix6r1w_again", wgs, 56.0 * cells, time_ms([&] { hipLaunchKernelGGL(k_mixn<6>, dim3(wgs), dim3(256), 0, 0, P, p[6], n); }, reps));
            rep("packed6r1w", wgs, 56.0 * cells, time_ms([&] { hipLaunchKernelGGL(k_packed6, dim3(wgs), dim3(256), 0, 0, big, p[6], n); }, reps));
            rep("rowpacked6r1w", wgs, 56.0 * cells, time_ms([&] { hipLaunchKernelGGL(k_rowpacked6, dim3(wgs), dim3(256), 0, 0, big, p[6], n); }, reps));
        }
        CK(hipFree(big));
    }
    for (int wgs : {1024, 2048, 4096, 8192, 32768}) {
        rep("copy", wgs, 16.0 * cells, time_ms([&] { hipLaunchKernelGGL(k_copy<0>, dim3(wgs), dim3(256), 0, 0, p[0], p[6], n); }, reps));
        rep("copy_nt_store", wgs, 16.0 * cells, time_ms([&] { hipLaunchKernelGGL(k_copy<2>, dim3(wgs), dim3(256), 0, 0, p[0], p[6], n); }, reps));
        rep("copy_nt_both", wgs, 16.0 * cells, time_ms([&] { hipLaunchKernelGGL(k_copy<3>, dim3(wgs), dim3(256), 0, 0, p[0], p[6], n); }, reps));
        rep("copy_chunk", wgs, 16.0 * cells, time_ms([&] { hipLaunchKernelGGL(k_copy_chunk<0>, dim3(wgs), dim3(256), 0, 0, p[0], p[6], n); }, reps));
        rep("read", wgs, 8.0 * cells, time_ms([&] { hipLaunchKernelGGL(k_read, dim3(wgs), dim3(256), 0, 0, p[0], o1, n); }, reps));
        rep("write", wgs, 8.0 * cells, time_ms([&] { hipLaunchKernelGGL(k_write, dim3(wgs), dim3(256), 0, 0, p[6], n); }, reps));
        rep("mix6r1w", wgs, 56.0 * cells, time_ms([&] { hipLaunchKernelGGL(k_mix6<0>, dim3(wgs), dim3(256), 0, 0, p[0], p[1], p[2], p[3], p[4], p[5], p[6], n); }, reps));
        rep("mix6r1w_nt_store", wgs, 56.0 * cells, time_ms([&] { hipLaunchKernelGGL(k_mix6<2>, dim3(wgs), dim3(256), 0, 0, p[0], p[1], p[2], p[3], p[4], p[5], p[6], n); }, reps));
        rep("mix6r1w_nt_both", wgs, 56.0 * cells, time_ms([&] { hipLaunchKernelGGL(k_mix6<3>, dim3(wgs), dim3(256), 0, 0, p[0], p[1], p[2], p[3], p[4], p[5], p[6], n); }, reps));
    }
    rep("hipMemcpyDtoD", 0, 16.0 * cells, time_ms([&] { CK(hipMemcpyAsync(p[6], p[0], cells * 8, hipMemcpyDeviceToDevice, 0)); }, reps));
    printf("\n]}\n");
    return 0;
}
