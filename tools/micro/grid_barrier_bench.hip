// Micro-benchmark: what does a grid-wide barrier cost on this part, with the grid k_register runs at kitti64
// (255 workgroups of 512 threads, all co-resident)?  Decides whether a persistent LM loop (one launch, a barrier per
// iteration) would beat one launch per iteration (kernel boundary + kernel entry ~4 us).  Every spin is bounded: a
// workgroup that waits too long sets a flag and leaves.     hipcc --offload-arch=gfx950 -O3 -o grid_barrier_bench grid_barrier_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(512) void k_barriers(unsigned* counter, unsigned* failed, double* rows, int iters, int payload)
{
    const unsigned nb = gridDim.x;
    __shared__ unsigned s_ok;
    for (int it = 0; it < iters; it++) {
        if (payload && threadIdx.x < 28) {                   // a partial row per workgroup, like k_register's
            rows[((size_t)(it & 1) * nb + blockIdx.x) * 28 + threadIdx.x] = (double)(it + blockIdx.x);
        }
        __threadfence();                                     // release: the row is visible device-wide
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(counter, 1u);
            const unsigned target = nb * (unsigned)(it + 1);
            unsigned ok = 0;
            for (int spin = 0; spin < 2000000; spin++) {
                if (__atomic_load_n(counter, __ATOMIC_RELAXED) >= target) { ok = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (!ok) atomicAdd(failed, 1u);
            s_ok = ok;
        }
        __syncthreads();
        if (!s_ok) return;
        __threadfence();                                     // acquire
        if (payload) {                                       // every workgroup reads every row (the fused close does)
            double s = 0.0;
            if (threadIdx.x < 28) for (unsigned b = 0; b < nb; b++) s += rows[((size_t)(it & 1) * nb + b) * 28 + threadIdx.x];
            if (s == -1.0) rows[0] = s;
        }
    }
}

// the same with a two-level arrival (groups of 32 workgroups, the last of a group reports to the grid counter) and a
// release flag per iteration that the last arriver sets and everybody else polls: 32 + 8 atomics on a line instead of 255
template <bool FENCE>
__global__ __launch_bounds__(512) void k_barriers_tree(unsigned* group_counters, unsigned* grid_counter, unsigned* flag, unsigned* failed, int iters)
{
    const unsigned nb = gridDim.x, ngroups = (nb + 31) / 32, grp = blockIdx.x / 32;
    const unsigned in_group = (grp == ngroups - 1) ? nb - grp * 32 : 32;
    __shared__ unsigned s_ok;
    for (int it = 0; it < iters; it++) {
        if (FENCE) __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned g = atomicAdd(&group_counters[grp * 32], 1u) + 1u;          // (counters 128 bytes apart)
            if (g == in_group * (unsigned)(it + 1)) {
                const unsigned t = atomicAdd(grid_counter, 1u) + 1u;
                if (t == ngroups * (unsigned)(it + 1)) __atomic_store_n(flag, (unsigned)(it + 1), __ATOMIC_RELAXED);
            }
            unsigned ok = 0;
            for (int spin = 0; spin < 2000000; spin++) {
                if (__atomic_load_n(flag, __ATOMIC_RELAXED) >= (unsigned)(it + 1)) { ok = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (!ok) atomicAdd(failed, 1u);
            s_ok = ok;
        }
        __syncthreads();
        if (!s_ok) return;
        if (FENCE) __threadfence();
    }
}

// no fences at all: the rows themselves travel through device-scope atomics (stores and loads that are served memory-side),
// arrival after the stores have completed, two-level arrival + release flag as above; every workgroup reads every row
__global__ __launch_bounds__(512) void k_barriers_atomic_rows(unsigned* group_counters, unsigned* grid_counter, unsigned* flag, unsigned* failed,
                                                             double* rows, int iters, double* sink)
{
    const unsigned nb = gridDim.x, ngroups = (nb + 31) / 32, grp = blockIdx.x / 32;
    const unsigned in_group = (grp == ngroups - 1) ? nb - grp * 32 : 32;
    __shared__ unsigned s_ok;
    double acc = 0.0;
    for (int it = 0; it < iters; it++) {
        double* slot = rows + (size_t)(it & 1) * nb * 28;
        if (threadIdx.x < 28) __hip_atomic_store(&slot[(size_t)blockIdx.x * 28 + threadIdx.x], (double)(it + blockIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): the stores have been acknowledged
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned g = atomicAdd(&group_counters[grp * 32], 1u) + 1u;
            if (g == in_group * (unsigned)(it + 1)) {
                const unsigned t = atomicAdd(grid_counter, 1u) + 1u;
                if (t == ngroups * (unsigned)(it + 1)) __hip_atomic_store(flag, (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            unsigned ok = 0;
            for (int spin = 0; spin < 2000000; spin++) {
                if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)(it + 1)) { ok = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (!ok) atomicAdd(failed, 1u);
            s_ok = ok;
        }
        __syncthreads();
        if (!s_ok) return;
        // 16 row groups x 32 columns like the fused close: 16 loads in flight per lane
        const int col = threadIdx.x & 31, grp2 = threadIdx.x >> 5;
        if (col < 28) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) { const unsigned b = grp2 + u * 16; v[u] = (b < nb) ? __hip_atomic_load(&slot[(size_t)b * 28 + col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0; }
#pragma unroll
            for (int u = 0; u < 16; u++) acc += v[u];
        }
    }
    if (acc == -1.0) *sink = acc;
}

__global__ __launch_bounds__(512) void k_empty(double* rows, int it)
{
    if (threadIdx.x < 28) rows[(size_t)blockIdx.x * 28 + threadIdx.x] = (double)it;
}

int main()
{
    const int nb = 255, iters = 2000;
    unsigned *d_counter, *d_failed; double* d_rows;
    hipMalloc(&d_counter, 4); hipMalloc(&d_failed, 4); hipMalloc(&d_rows, sizeof(double) * 2 * nb * 28);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int payload = 0; payload < 2; payload++) {
        hipMemset(d_counter, 0, 4); hipMemset(d_failed, 0, 4);
        hipLaunchKernelGGL(k_barriers, dim3(nb), dim3(512), 0, 0, d_counter, d_failed, d_rows, 10, payload);   // warm-up
        hipDeviceSynchronize();
        hipMemset(d_counter, 0, 4);
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(k_barriers, dim3(nb), dim3(512), 0, 0, d_counter, d_failed, d_rows, iters, payload);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        unsigned failed = 0; hipMemcpy(&failed, d_failed, 4, hipMemcpyDeviceToHost);
        printf("grid barrier, %d workgroups x 512 threads%s: %.2f us per iteration (%u workgroups gave up)\n", nb,
               payload ? ", row written before / all rows read after" : "", 1e3 * ms / iters, failed);
    }
    for (int fence = 1; fence >= 0; fence--) {
        unsigned *d_g, *d_flag; hipMalloc(&d_g, 4 * 32 * 16); hipMalloc(&d_flag, 4);
        hipMemset(d_g, 0, 4 * 32 * 16); hipMemset(d_flag, 0, 4); hipMemset(d_counter, 0, 4); hipMemset(d_failed, 0, 4);
        hipEventRecord(a, 0);
        if (fence) hipLaunchKernelGGL(k_barriers_tree<true>, dim3(nb), dim3(512), 0, 0, d_g, d_counter, d_flag, d_failed, iters);
        else       hipLaunchKernelGGL(k_barriers_tree<false>, dim3(nb), dim3(512), 0, 0, d_g, d_counter, d_flag, d_failed, iters);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        unsigned failed = 0; hipMemcpy(&failed, d_failed, 4, hipMemcpyDeviceToHost);
        printf("grid barrier, two-level arrival + release flag, %s: %.2f us per iteration (%u workgroups gave up)\n",
               fence ? "device-scope fences before and after" : "NO fences (synchronisation only: data would not be visible)", 1e3 * ms / iters, failed);
    }
    {
        unsigned *d_g, *d_flag; hipMalloc(&d_g, 4 * 32 * 16); hipMalloc(&d_flag, 4);
        hipMemset(d_g, 0, 4 * 32 * 16); hipMemset(d_flag, 0, 4); hipMemset(d_counter, 0, 4); hipMemset(d_failed, 0, 4);
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(k_barriers_atomic_rows, dim3(nb), dim3(512), 0, 0, d_g, d_counter, d_flag, d_failed, d_rows, iters, d_rows);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        unsigned failed = 0; hipMemcpy(&failed, d_failed, 4, hipMemcpyDeviceToHost);
        printf("grid barrier without fences, rows stored and loaded with device-scope atomics, every workgroup reads every row: %.2f us per iteration (%u workgroups gave up)\n", 1e3 * ms / iters, failed);
    }
    // the alternative: one (empty) launch per iteration, back to back in one stream
    hipEventRecord(a, 0);
    for (int it = 0; it < iters; it++) hipLaunchKernelGGL(k_empty, dim3(nb), dim3(512), 0, 0, d_rows, it);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    printf("one launch per iteration (row written, nothing else): %.2f us per iteration\n", 1e3 * ms / iters);
    return 0;
}
