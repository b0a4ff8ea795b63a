// s2m_icp.hip — pcl::IterativeClosestPoint (PCL 1.10 [ext]: registration/impl/icp.hpp,
// correspondence_estimation.hpp, transformation_estimation_svd.hpp -> Eigen::umeyama,
// default_convergence_criteria.hpp, Registration::getFitnessScore) on gfx950, see s2m_icp.hpp.
//
// Per iteration: (1) exact nearest target point of every (already transformed) source point - a tiled
// brute force, target slices staged through LDS, the winner of each source point settled across slices
// by one 64-bit atomicMin on (d2 bits << 32 | target index), i.e. ties go to the lower index like the
// oracle; loop-closure submaps are 1e3..1e5 points, so the whole search is a few 1e8..1e9 distance
// evaluations spread over ~1000 workgroups; (2) one pass over the correspondences within the distance
// limit accumulates count, sum d2, the two centroids and the raw cross moments in fp64 (17 numbers); the
// host turns them into Eigen::umeyama's mean / covariance, does the 3x3 SVD, the pose composition and
// the convergence state machine exactly as PCL does; (3) the source cloud is transformed in place by the
// incremental transform, as PCL transforms input_transformed.  fp32 distances use the operation order
// of the oracle ((dx*dx + dy*dy) + dz*dz, -ffp-contract=off), so correspondences are identical to it.
#include "s2m_icp.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <new>

#include "s2m_host_math.hpp"

namespace s2m {

namespace {

constexpr int kNnThreads = 256;
constexpr int kNnTile = 1024;                       // target points per LDS tile
constexpr unsigned long long kNoMatch = ~0ull;

struct Buf {
    void*  p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        const size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

__global__ __launch_bounds__(256) void k_icp_load(const unsigned char* __restrict__ pts, size_t stride, int n, float4* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = reinterpret_cast<const float*>(pts + (size_t)i * stride);
    out[i] = make_float4(p[0], p[1], p[2], 0.0f);
}

__global__ __launch_bounds__(256) void k_icp_reset(unsigned long long* __restrict__ best, int n, double* __restrict__ sums)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) best[i] = kNoMatch;
    if (i < 17) sums[i] = 0.0;
}

// grid (source blocks, target slices): 256 source points against target points [slice*len, (slice+1)*len)
__global__ __launch_bounds__(kNnThreads) void k_icp_nn(const float4* __restrict__ cur, int n_src, const float4* __restrict__ tgt, int n_tgt,
                                                       int slice_len, unsigned long long* __restrict__ best)
{
    __shared__ float4 tile[kNnTile];
    const int i = blockIdx.x * kNnThreads + threadIdx.x;
    const bool valid = i < n_src;
    const float4 p = valid ? cur[i] : make_float4(0, 0, 0, 0);
    const bool fin = valid && isfinite(p.x) && isfinite(p.y) && isfinite(p.z);
    float bd = INFINITY;
    int bi = -1;
    const int j0 = blockIdx.y * slice_len, j1 = min(j0 + slice_len, n_tgt);
    for (int base = j0; base < j1; base += kNnTile) {
        const int m = min(kNnTile, j1 - base);
        __syncthreads();
        for (int k = threadIdx.x; k < m; k += kNnThreads) tile[k] = tgt[base + k];
        __syncthreads();
        if (fin) {
            int k = 0;
            for (; k + 4 <= m; k += 4) {
                const float4 a = tile[k], b = tile[k + 1], c = tile[k + 2], d = tile[k + 3];
                const float ax = p.x - a.x, ay = p.y - a.y, az = p.z - a.z, bx = p.x - b.x, by = p.y - b.y, bz = p.z - b.z;
                const float cx = p.x - c.x, cy = p.y - c.y, cz = p.z - c.z, dx = p.x - d.x, dy = p.y - d.y, dz = p.z - d.z;
                const float da = (ax * ax + ay * ay) + az * az, db = (bx * bx + by * by) + bz * bz;
                const float dc = (cx * cx + cy * cy) + cz * cz, dd = (dx * dx + dy * dy) + dz * dz;
                if (da < bd) { bd = da; bi = base + k; }
                if (db < bd) { bd = db; bi = base + k + 1; }
                if (dc < bd) { bd = dc; bi = base + k + 2; }
                if (dd < bd) { bd = dd; bi = base + k + 3; }
            }
            for (; k < m; k++) {
                const float4 a = tile[k];
                const float ax = p.x - a.x, ay = p.y - a.y, az = p.z - a.z;
                const float da = (ax * ax + ay * ay) + az * az;
                if (da < bd) { bd = da; bi = base + k; }
            }
        }
    }
    if (fin && bi >= 0)       // d2 >= 0: its bit pattern orders like the value; NaN distances (non-finite targets) never win above
        atomicMin(&best[i], ((unsigned long long)__float_as_uint(bd) << 32) | (unsigned)bi);
}

// count, sum d2, centroids and raw cross moments of the correspondences with d2 <= max_d2 (fp64)
__global__ __launch_bounds__(256) void k_icp_sums(const float4* __restrict__ cur, int n_src, const float4* __restrict__ tgt,
                                                  const unsigned long long* __restrict__ best, float max_d2_f, double max_d2,
                                                  double* __restrict__ sums)
{
    __shared__ double sh[4][17];
    double a[17];
#pragma unroll
    for (int k = 0; k < 17; k++) a[k] = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_src; i += gridDim.x * blockDim.x) {
        const unsigned long long key = best[i];
        if (key == kNoMatch) continue;
        const float d2 = __uint_as_float((unsigned)(key >> 32));
        if (!((double)d2 <= max_d2)) continue;                       // determineCorrespondences: distance[0] > max_dist_sqr -> skip
        const float4 s = cur[i], t = tgt[(unsigned)key];
        a[0] += 1.0; a[1] += (double)d2;
        a[2] += s.x; a[3] += s.y; a[4] += s.z; a[5] += t.x; a[6] += t.y; a[7] += t.z;
        a[8]  += (double)t.x * s.x; a[9]  += (double)t.x * s.y; a[10] += (double)t.x * s.z;
        a[11] += (double)t.y * s.x; a[12] += (double)t.y * s.y; a[13] += (double)t.y * s.z;
        a[14] += (double)t.z * s.x; a[15] += (double)t.z * s.y; a[16] += (double)t.z * s.z;
    }
    (void)max_d2_f;
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a[k] += __shfl_down(a[k], off, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 17; k++) sh[wave][k] = a[k];
    }
    __syncthreads();
    if (threadIdx.x < 17) atomicAdd(&sums[threadIdx.x], sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

struct Mat34 { float m[12]; };
__global__ __launch_bounds__(256) void k_icp_transform(float4* __restrict__ cur, int n, Mat34 T)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = cur[i];
    cur[i] = make_float4(T.m[0] * p.x + T.m[1] * p.y + T.m[2]  * p.z + T.m[3],
                         T.m[4] * p.x + T.m[5] * p.y + T.m[6]  * p.z + T.m[7],
                         T.m[8] * p.x + T.m[9] * p.y + T.m[10] * p.z + T.m[11], 0.0f);
}

}  // namespace

struct IcpWorkspace {
    Buf cur, tgt, best, sums;
    double* h_sums = nullptr;            // pinned, 17 doubles
};

IcpWorkspace* icp_create()
{
    IcpWorkspace* w = new (std::nothrow) IcpWorkspace();
    if (!w) return nullptr;
    if (w->sums.ensure(sizeof(double) * 17) != hipSuccess || hipHostMalloc((void**)&w->h_sums, sizeof(double) * 17) != hipSuccess) {
        icp_destroy(w);
        return nullptr;
    }
    return w;
}

void icp_destroy(IcpWorkspace* w)
{
    if (!w) return;
    Buf* bufs[] = { &w->cur, &w->tgt, &w->best, &w->sums };
    for (Buf* b : bufs) if (b->p) (void)hipFree(b->p);
    if (w->h_sums) (void)hipHostFree(w->h_sums);
    delete w;
}

#define ICP_TRY(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return e__; } while (0)

namespace {

hipError_t nearest(IcpWorkspace* w, hipStream_t stream, int n_src, int n_tgt)
{
    const int sb = (n_src + kNnThreads - 1) / kNnThreads;
    int slices = (1024 + sb - 1) / sb;                                   // ~1000 workgroups in flight
    slices = std::max(1, std::min(slices, (n_tgt + kNnTile - 1) / kNnTile));
    const int slice_len = (((n_tgt + slices - 1) / slices) + kNnTile - 1) / kNnTile * kNnTile;
    slices = (n_tgt + slice_len - 1) / slice_len;
    hipLaunchKernelGGL(k_icp_reset, dim3((std::max(n_src, 17) + 255) / 256), dim3(256), 0, stream, w->best.as<unsigned long long>(), n_src,
                       w->sums.as<double>());
    hipLaunchKernelGGL(k_icp_nn, dim3(sb, slices), dim3(kNnThreads), 0, stream, (const float4*)w->cur.as<float4>(), n_src,
                       (const float4*)w->tgt.as<float4>(), n_tgt, slice_len, w->best.as<unsigned long long>());
    return hipGetLastError();
}

hipError_t sums(IcpWorkspace* w, hipStream_t stream, int n_src, double max_d2)
{
    hipLaunchKernelGGL(k_icp_sums, dim3(std::min((n_src + 255) / 256, 256)), dim3(256), 0, stream, (const float4*)w->cur.as<float4>(), n_src,
                       (const float4*)w->tgt.as<float4>(), (const unsigned long long*)w->best.as<unsigned long long>(), 0.0f, max_d2,
                       w->sums.as<double>());
    ICP_TRY(hipGetLastError());
    ICP_TRY(hipMemcpyAsync(w->h_sums, w->sums.p, sizeof(double) * 17, hipMemcpyDeviceToHost, stream));
    return hipStreamSynchronize(stream);
}

}  // namespace

hipError_t icp_align(IcpWorkspace* w, hipStream_t stream, const unsigned char* d_src, size_t n_src_, const unsigned char* d_tgt,
                     size_t n_tgt_, size_t stride, const IcpParams& prm, IcpResult* res)
{
    const int n_src = (int)n_src_, n_tgt = (int)n_tgt_;
    for (int i = 0; i < 16; i++) res->T[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    res->converged = 0; res->iterations = 0; res->fitness = DBL_MAX;
    if (n_src == 0 || n_tgt == 0) return hipSuccess;
    ICP_TRY(w->cur.ensure(sizeof(float4) * n_src_)); ICP_TRY(w->tgt.ensure(sizeof(float4) * n_tgt_));
    ICP_TRY(w->best.ensure(sizeof(unsigned long long) * n_src_));
    hipLaunchKernelGGL(k_icp_load, dim3((n_src + 255) / 256), dim3(256), 0, stream, d_src, stride, n_src, w->cur.as<float4>());
    hipLaunchKernelGGL(k_icp_load, dim3((n_tgt + 255) / 256), dim3(256), 0, stream, d_tgt, stride, n_tgt, w->tgt.as<float4>());
    ICP_TRY(hipGetLastError());

    // icp.hpp computeTransformation + default_convergence_criteria.hpp hasConverged
    const double max_d2 = prm.max_corr_dist * prm.max_corr_dist;
    const double rot_thr = 1.0 - prm.trans_eps, trl_thr = prm.trans_eps, mse_abs = 1e-12, mse_rel = prm.fit_eps;
    const int max_similar = 0;                                            // max_iterations_similar_transforms_
    double prev_mse = DBL_MAX;
    int it = 0, conv = 0, similar = 0;
    for (;;) {
        ICP_TRY(nearest(w, stream, n_src, n_tgt));
        ICP_TRY(sums(w, stream, n_src, max_d2));
        const double* S = w->h_sums;
        const double cnt = S[0];
        if (cnt < 3.0) { conv = 0; break; }                               // min_number_correspondences_
        const double mse = S[1] / cnt;
        float ms[3], mt[3], sg[9];
        for (int d = 0; d < 3; d++) { ms[d] = (float)(S[2 + d] / cnt); mt[d] = (float)(S[5 + d] / cnt); }
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) sg[r * 3 + c] = (float)(S[8 + r * 3 + c] / cnt - (S[5 + r] / cnt) * (S[2 + c] / cnt));
        float T[16];
        host_umeyama(ms, mt, sg, T);
        Mat34 T34;
        memcpy(T34.m, T, sizeof(float) * 12);
        hipLaunchKernelGGL(k_icp_transform, dim3((n_src + 255) / 256), dim3(256), 0, stream, w->cur.as<float4>(), n_src, T34);
        ICP_TRY(hipGetLastError());
        float F[16];
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) {
                float a = 0;
                for (int k = 0; k < 4; k++) a += T[r * 4 + k] * res->T[k * 4 + c];
                F[r * 4 + c] = a;
            }
        memcpy(res->T, F, sizeof(F));
        ++it;
        int is_similar = 0;
        if (it >= prm.max_iter) { conv = 1; break; }                      // CONVERGENCE_CRITERIA_ITERATIONS
        const double cos_angle = 0.5 * ((double)T[0] + (double)T[5] + (double)T[10] - 1.0);
        const double tsq = (double)T[3] * T[3] + (double)T[7] * T[7] + (double)T[11] * T[11];
        if (cos_angle >= rot_thr && tsq <= trl_thr) { if (similar >= max_similar) { conv = 1; break; } is_similar = 1; }
        if (std::fabs(mse - prev_mse) < mse_abs) { if (similar >= max_similar) { conv = 1; break; } is_similar = 1; }
        if (std::fabs(mse - prev_mse) / prev_mse < mse_rel) { if (similar >= max_similar) { conv = 1; break; } is_similar = 1; }
        similar = is_similar ? similar + 1 : 0;
        prev_mse = mse;
    }
    // getFitnessScore(): the original source under the final transform, mean squared nearest distance
    hipLaunchKernelGGL(k_icp_load, dim3((n_src + 255) / 256), dim3(256), 0, stream, d_src, stride, n_src, w->cur.as<float4>());
    Mat34 F34;
    memcpy(F34.m, res->T, sizeof(float) * 12);
    hipLaunchKernelGGL(k_icp_transform, dim3((n_src + 255) / 256), dim3(256), 0, stream, w->cur.as<float4>(), n_src, F34);
    ICP_TRY(nearest(w, stream, n_src, n_tgt));
    ICP_TRY(sums(w, stream, n_src, DBL_MAX));
    res->fitness = w->h_sums[0] > 0 ? w->h_sums[1] / w->h_sums[0] : DBL_MAX;
    res->converged = conv; res->iterations = it;
    return hipSuccess;
}

}  // namespace s2m
