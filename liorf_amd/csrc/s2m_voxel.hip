// s2m_voxel.hip — pcl::VoxelGrid<PointXYZI> and transformPointCloud on gfx950 (see s2m_voxel.hpp).
//
// The filter follows PCL's own passes (voxel_grid.hpp, PCL 1.10): bounding box of the finite
// points -> integer voxel index per point -> sort by index -> one centroid per run. A dense
// per-voxel table is not an option (a 200 m x 200 m x 30 m sweep at 0.4 m is 19 M voxels for
// ~1e5 points), so the runs come from a stable LSD radix sort of (index, point) pairs (hipCUB, a
// plain library sort) and everything around it is hand-written: the sort being stable fixes the
// summation order inside a voxel to ascending point index, which is what the CPU oracle defines,
// so centroids are bit-identical to it. Compiled with -ffp-contract=off like the rest.
#include "s2m_voxel.hpp"

#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstring>
#include <new>

namespace s2m {

namespace {

constexpr uint32_t kInvalidKey = 0xffffffffu;       // non-finite points sort behind every voxel

struct VoxSetup {
    int32_t  min_b[3];
    int32_t  mul1, mul2;
    float    inv_leaf;
    int32_t  leaf_too_small;
    int32_t  n_valid;
    int32_t  n_out;
    int32_t  n_long;                                 // voxels with a long run of points, handed to k_vox_centroid_long
    uint32_t mm[6];                                  // ordered-uint min xyz, max xyz
};

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

__device__ __forceinline__ uint32_t f2ord(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

__global__ void k_vox_reset(VoxSetup* s)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int d = 0; d < 3; d++) { s->mm[d] = 0xffffffffu; s->mm[3 + d] = 0u; }
        s->n_valid = 0; s->n_out = 0; s->leaf_too_small = 0; s->n_long = 0;
    }
}

// getMinMax3D over the points whose three coordinates are all finite, and their count.
__global__ __launch_bounds__(256) void k_vox_bbox(const unsigned char* __restrict__ pts, size_t stride, int n, VoxSetup* s)
{
    __shared__ float smn[4][3], smx[4][3];
    __shared__ int scnt[4];
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    int cnt = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float* p = reinterpret_cast<const float*>(pts + (size_t)i * stride);
        const float x = p[0], y = p[1], z = p[2];
        if (isfinite(x) && isfinite(y) && isfinite(z)) {
            mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);
            mn[1] = fminf(mn[1], y); mx[1] = fmaxf(mx[1], y);
            mn[2] = fminf(mn[2], z); mx[2] = fmaxf(mx[2], z);
            cnt++;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int d = 0; d < 3; d++) {
            mn[d] = fminf(mn[d], __shfl_down(mn[d], off, 64));
            mx[d] = fmaxf(mx[d], __shfl_down(mx[d], off, 64));
        }
        cnt += __shfl_down(cnt, off, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 3; d++) { smn[wave][d] = mn[d]; smx[wave][d] = mx[d]; }
        scnt[wave] = cnt;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int d = threadIdx.x;
        const float a = fminf(fminf(smn[0][d], smn[1][d]), fminf(smn[2][d], smn[3][d]));
        const float b = fmaxf(fmaxf(smx[0][d], smx[1][d]), fmaxf(smx[2][d], smx[3][d]));
        if (a <= b) { atomicMin(&s->mm[d], f2ord(a)); atomicMax(&s->mm[3 + d], f2ord(b)); }
    } else if (threadIdx.x == 3) {
        const int c = scnt[0] + scnt[1] + scnt[2] + scnt[3];
        if (c) atomicAdd(&s->n_valid, c);
    }
}

// min_b_/div_b_/divb_mul_ and the "leaf size too small" test of applyFilter, on one thread.
__global__ void k_vox_setup(VoxSetup* s, float leaf)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float inv = 1.0f / leaf;
    s->inv_leaf = inv;
    if (s->n_valid == 0) { s->min_b[0] = s->min_b[1] = s->min_b[2] = 0; s->mul1 = s->mul2 = 1; return; }
    float mn[3], mx[3];
    for (int d = 0; d < 3; d++) { mn[d] = ord2f(s->mm[d]); mx[d] = ord2f(s->mm[3 + d]); }
    // (int64) of a float product: clamp before converting, the comparison below only needs "> INT32_MAX"
    double ext[3];
    for (int d = 0; d < 3; d++) {
        const float e = (mx[d] - mn[d]) * inv;
        ext[d] = (e < 9.0e18f) ? (double)(long long)e + 1.0 : 9.0e18;
    }
    if (ext[0] * ext[1] * ext[2] > 2147483647.0) { s->leaf_too_small = 1; }
    int div_b[3];
    for (int d = 0; d < 3; d++) {
        const float lo = floorf(mn[d] * inv), hi = floorf(mx[d] * inv);
        const int ilo = (int)fminf(fmaxf(lo, -2147483520.0f), 2147483520.0f);
        const int ihi = (int)fminf(fmaxf(hi, -2147483520.0f), 2147483520.0f);
        s->min_b[d] = ilo;
        div_b[d] = ihi - ilo + 1;
    }
    s->mul1 = div_b[0];
    s->mul2 = div_b[0] * div_b[1];
}

__global__ __launch_bounds__(256) void k_vox_keys(const unsigned char* __restrict__ pts, size_t stride, int n,
                                                  const VoxSetup* __restrict__ s, uint32_t* __restrict__ keys,
                                                  int32_t* __restrict__ vals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = reinterpret_cast<const float*>(pts + (size_t)i * stride);
    const float x = p[0], y = p[1], z = p[2];
    uint32_t key = kInvalidKey;
    if (isfinite(x) && isfinite(y) && isfinite(z)) {
        const float inv = s->inv_leaf;
        const int ijk0 = (int)(floorf(x * inv) - (float)s->min_b[0]);
        const int ijk1 = (int)(floorf(y * inv) - (float)s->min_b[1]);
        const int ijk2 = (int)(floorf(z * inv) - (float)s->min_b[2]);
        key = (uint32_t)(ijk0 + ijk1 * s->mul1 + ijk2 * s->mul2);
        if (key == kInvalidKey) key = kInvalidKey - 1u;          // unreachable unless the leaf is too small
    }
    keys[i] = key;
    vals[i] = i;
}

__global__ __launch_bounds__(256) void k_vox_heads(const uint32_t* __restrict__ keys, int n, uint8_t* __restrict__ flags)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = keys[i];
    flags[i] = (k != kInvalidKey && (i == 0 || keys[i - 1] != k)) ? 1 : 0;
}

// One lane per voxel: CentroidPoint<PointXYZI> over the run [heads[v], heads[v+1]) in sorted order.
constexpr int kLongRun = 96;          // a voxel with more points than this is summed by a whole wave (k_vox_centroid_long)

__device__ __forceinline__ void write_centroid(unsigned char* __restrict__ out, size_t out_stride, int v, float sx, float sy, float sz, float si, float cnt)
{
    float* o = reinterpret_cast<float*>(out + (size_t)v * out_stride);
    o[0] = sx / cnt; o[1] = sy / cnt; o[2] = sz / cnt;
    const int words = (int)(out_stride >> 2);
    if (words > 3) o[3] = 1.0f;
    if (words > 4) o[4] = si / cnt;
    for (int k = 5; k < words; k++) o[k] = 0.0f;
}

__global__ __launch_bounds__(256) void k_vox_centroid(const unsigned char* __restrict__ pts, size_t stride,
                                                      const int32_t* __restrict__ vals, const int32_t* __restrict__ heads,
                                                      VoxSetup* __restrict__ s, int32_t* __restrict__ long_list, unsigned char* __restrict__ out,
                                                      size_t out_stride, int cap)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_out = s->n_out;
    if (v >= n_out || v >= cap) return;
    const int first = heads[v];
    const int last = (v + 1 < n_out) ? heads[v + 1] : s->n_valid;
    if (last - first > kLongRun) { long_list[atomicAdd(&s->n_long, 1)] = v; return; }   // (order of the list does not matter: a voxel, an output record)
    float sx = 0.0f, sy = 0.0f, sz = 0.0f, si = 0.0f;
    // The sums are sequential by definition (fp32, ascending point index: the oracle's order), the loads are not: eight points'
    // records are requested at once and added one after the other (a lane walking its run one dependent gather at a time was 60 %
    // of extractCloud: 21 points per voxel on average, hundreds where key frames overlap).
    const bool has_i = stride >= 20;
    int j = first;
    for (; j + 8 <= last; j += 8) {
        int id[8];
#pragma unroll
        for (int u = 0; u < 8; u++) id[u] = vals[j + u];
        float x[8], y[8], z[8], w[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float* p = reinterpret_cast<const float*>(pts + (size_t)id[u] * stride);
            x[u] = p[0]; y[u] = p[1]; z[u] = p[2]; w[u] = has_i ? p[4] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) { sx += x[u]; sy += y[u]; sz += z[u]; si += w[u]; }
    }
    for (; j < last; j++) {
        const float* p = reinterpret_cast<const float*>(pts + (size_t)vals[j] * stride);
        sx += p[0]; sy += p[1]; sz += p[2];
        if (has_i) si += p[4];
    }
    write_centroid(out, out_stride, v, sx, sy, sz, si, (float)(last - first));
}

// The long runs (a voxel where many key frames overlap holds hundreds of points; one near the sensor thousands): a wave per
// voxel.  64 records at a time are gathered by the 64 lanes (one round trip) into LDS, and lanes 0..3 each add up one
// component - x, y, z, intensity - point after point in the run's order: the same sequential fp32 sums as a single lane
// would form, without 64 dependent gathers in a row.
__global__ __launch_bounds__(256) void k_vox_centroid_long(const unsigned char* __restrict__ pts, size_t stride,
                                                           const int32_t* __restrict__ vals, const int32_t* __restrict__ heads,
                                                           const VoxSetup* __restrict__ s, const int32_t* __restrict__ long_list,
                                                           unsigned char* __restrict__ out, size_t out_stride)
{
    constexpr int kChunk = 256;                      // records per round trip: four per lane
    __shared__ __attribute__((aligned(16))) float comp[4][4][kChunk];   // [wave][component][point of the chunk]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_long = s->n_long, n_out = s->n_out, n_valid = s->n_valid;
    const bool has_i = stride >= 20;
    auto lds_sync = []() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    for (int item = blockIdx.x * 4 + wave; item < n_long; item += gridDim.x * 4) {
        const int v = long_list[item];
        const int first = heads[v];
        const int last = (v + 1 < n_out) ? heads[v + 1] : n_valid;
        float sum = 0.0f;                             // lanes 0..3: the running sum of component `lane`
        float rx[4], ry[4], rz[4], rw[4];
        auto fetch = [&](int c) {                     // this lane's four records of the chunk that starts at c
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = c + 64 * u + lane;
                rx[u] = ry[u] = rz[u] = rw[u] = 0.0f;
                if (j < last) {
                    const float* p = reinterpret_cast<const float*>(pts + (size_t)vals[j] * stride);
                    rx[u] = p[0]; ry[u] = p[1]; rz[u] = p[2]; rw[u] = has_i ? p[4] : 0.0f;
                }
            }
        };
        fetch(first);
        for (int c = first; c < last; c += kChunk) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                comp[wave][0][64 * u + lane] = rx[u]; comp[wave][1][64 * u + lane] = ry[u];
                comp[wave][2][64 * u + lane] = rz[u]; comp[wave][3][64 * u + lane] = rw[u];
            }
            lds_sync();
            if (c + kChunk < last) fetch(c + kChunk);    // the next chunk's gathers are in flight behind this chunk's sums
            const int m = min(kChunk, last - c);
            if (lane < 4) {
                // sixteen values per group of LDS reads (four 16-byte reads in flight), added one after the other
                const float* col = comp[wave][lane];
                const float4* col4 = reinterpret_cast<const float4*>(col);
                int k = 0;
                for (; k + 16 <= m; k += 16) {
                    const float4 a = col4[k / 4], b = col4[k / 4 + 1], c4 = col4[k / 4 + 2], d = col4[k / 4 + 3];
                    sum += a.x; sum += a.y; sum += a.z; sum += a.w; sum += b.x; sum += b.y; sum += b.z; sum += b.w;
                    sum += c4.x; sum += c4.y; sum += c4.z; sum += c4.w; sum += d.x; sum += d.y; sum += d.z; sum += d.w;
                }
                for (; k < m; k++) sum += col[k];
            }
            lds_sync();
        }
        const float sx = __shfl(sum, 0, 64), sy = __shfl(sum, 1, 64), sz = __shfl(sum, 2, 64), si = __shfl(sum, 3, 64);
        if (lane == 0) write_centroid(out, out_stride, v, sx, sy, sz, si, (float)(last - first));
    }
}

__global__ __launch_bounds__(256) void k_copy_records(const unsigned char* __restrict__ in, size_t stride, int n,
                                                      unsigned char* __restrict__ out, size_t out_stride)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* p = reinterpret_cast<const uint32_t*>(in + (size_t)i * stride);
    uint32_t* o = reinterpret_cast<uint32_t*>(out + (size_t)i * out_stride);
    const int wi = (int)(stride >> 2), wo = (int)(out_stride >> 2);
    for (int k = 0; k < wo; k++) o[k] = k < wi ? p[k] : 0u;
}

// transformPointCloud (:320-327): three products and three adds per row, as written. Frame f owns output
// records [offsets[f], offsets[f+1]) and reads its own source buffer src[f].
__global__ __launch_bounds__(256) void k_transform_frames(const unsigned char* const* __restrict__ src, size_t stride,
                                                          const int32_t* __restrict__ offsets, const float* __restrict__ T,
                                                          int n_frames, unsigned char* __restrict__ out, size_t out_stride)
{
    const int n = offsets[n_frames];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int lo = 0, hi = n_frames - 1;                 // frame f with offsets[f] <= i < offsets[f+1]
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (offsets[mid] <= i) lo = mid; else hi = mid - 1;
    }
    const float* t = T + 12 * lo;
    const float* p = reinterpret_cast<const float*>(src[lo] + (size_t)(i - offsets[lo]) * stride);
    const float x = p[0], y = p[1], z = p[2];
    float* o = reinterpret_cast<float*>(out + (size_t)i * out_stride);
    o[0] = t[0] * x + t[1] * y + t[2] * z + t[3];
    o[1] = t[4] * x + t[5] * y + t[6] * z + t[7];
    o[2] = t[8] * x + t[9] * y + t[10] * z + t[11];
    const int words = (int)(out_stride >> 2);
    if (words > 3) o[3] = 1.0f;
    if (words > 4) o[4] = (stride >= 20) ? p[4] : 0.0f;
    for (int k = 5; k < words; k++) o[k] = 0.0f;
}

}  // namespace

struct VoxWorkspace {
    Buf setup, keys_a, keys_b, vals_a, vals_b, flags, heads, cub_tmp, frame_tab, long_list;
    VoxSetup* h_setup = nullptr;          // pinned
};

VoxWorkspace* vox_create()
{
    VoxWorkspace* w = new (std::nothrow) VoxWorkspace();
    if (!w) return nullptr;
    if (w->setup.ensure(sizeof(VoxSetup)) != hipSuccess ||
        hipHostMalloc((void**)&w->h_setup, sizeof(VoxSetup)) != hipSuccess) { vox_destroy(w); return nullptr; }
    return w;
}

void vox_destroy(VoxWorkspace* w)
{
    if (!w) return;
    Buf* bufs[] = { &w->setup, &w->keys_a, &w->keys_b, &w->vals_a, &w->vals_b, &w->flags, &w->heads, &w->cub_tmp, &w->frame_tab, &w->long_list };
    for (Buf* b : bufs) if (b->p) (void)hipFree(b->p);
    if (w->h_setup) (void)hipHostFree(w->h_setup);
    delete w;
}

#define VOX_TRY(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return e__; } while (0)

hipError_t vox_downsample(VoxWorkspace* w, hipStream_t stream, const unsigned char* d_in, size_t n, size_t stride,
                          float leaf, unsigned char* d_out, size_t out_stride, size_t cap, VoxResult* res)
{
    *res = VoxResult{};
    if (n == 0) return hipSuccess;
    const int ni = (int)n;
    VOX_TRY(w->keys_a.ensure(4 * n)); VOX_TRY(w->keys_b.ensure(4 * n));
    VOX_TRY(w->vals_a.ensure(4 * n)); VOX_TRY(w->vals_b.ensure(4 * n));
    VOX_TRY(w->flags.ensure(n));      VOX_TRY(w->heads.ensure(4 * n));
    VOX_TRY(w->long_list.ensure(4 * (n / kLongRun + 1)));          // (a long run holds more than kLongRun points)
    VoxSetup* s = w->setup.as<VoxSetup>();
    uint32_t* keys_a = w->keys_a.as<uint32_t>(); uint32_t* keys_b = w->keys_b.as<uint32_t>();
    int32_t* vals_a = w->vals_a.as<int32_t>();   int32_t* vals_b = w->vals_b.as<int32_t>();
    uint8_t* flags = w->flags.as<uint8_t>();     int32_t* heads = w->heads.as<int32_t>();

    size_t tmp_sort = 0, tmp_sel = 0;
    VOX_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_sort, keys_a, keys_b, vals_a, vals_b, ni, 0, 32, stream));
    hipcub::CountingInputIterator<int32_t> counting(0);
    VOX_TRY(hipcub::DeviceSelect::Flagged(nullptr, tmp_sel, counting, flags, heads, &s->n_out, ni, stream));
    const size_t tmp_bytes = tmp_sort > tmp_sel ? tmp_sort : tmp_sel;
    VOX_TRY(w->cub_tmp.ensure(tmp_bytes));

    const int nb = (ni + 255) / 256;
    hipLaunchKernelGGL(k_vox_reset, dim3(1), dim3(64), 0, stream, s);
    hipLaunchKernelGGL(k_vox_bbox, dim3(nb < 1024 ? nb : 1024), dim3(256), 0, stream, d_in, stride, ni, s);
    hipLaunchKernelGGL(k_vox_setup, dim3(1), dim3(64), 0, stream, s, leaf);
    hipLaunchKernelGGL(k_vox_keys, dim3(nb), dim3(256), 0, stream, d_in, stride, ni, (const VoxSetup*)s, keys_a, vals_a);
    VOX_TRY(hipGetLastError());
    size_t tb = tmp_bytes;
    VOX_TRY(hipcub::DeviceRadixSort::SortPairs(w->cub_tmp.p, tb, keys_a, keys_b, vals_a, vals_b, ni, 0, 32, stream));
    hipLaunchKernelGGL(k_vox_heads, dim3(nb), dim3(256), 0, stream, (const uint32_t*)keys_b, ni, flags);
    tb = tmp_bytes;
    VOX_TRY(hipcub::DeviceSelect::Flagged(w->cub_tmp.p, tb, counting, flags, heads, &s->n_out, ni, stream));
    // launched for the worst case (one voxel per point); lanes past n_out exit on the device-side count
    hipLaunchKernelGGL(k_vox_centroid, dim3(nb), dim3(256), 0, stream, d_in, stride, (const int32_t*)vals_b,
                       (const int32_t*)heads, s, w->long_list.as<int32_t>(), d_out, out_stride, (int)(cap < n ? cap : n));
    {   // the long runs, a wave each (the grid walks the list the kernel above left)
        const int nl = ni / kLongRun + 1;
        hipLaunchKernelGGL(k_vox_centroid_long, dim3(nl < 4096 ? (nl + 3) / 4 : 1024), dim3(256), 0, stream, d_in, stride, (const int32_t*)vals_b,
                           (const int32_t*)heads, (const VoxSetup*)s, (const int32_t*)w->long_list.as<int32_t>(), d_out, out_stride);
    }
    VOX_TRY(hipGetLastError());
    VOX_TRY(hipMemcpyAsync(w->h_setup, s, sizeof(VoxSetup), hipMemcpyDeviceToHost, stream));
    VOX_TRY(hipStreamSynchronize(stream));
    if (w->h_setup->leaf_too_small) {              // PCL warns and hands the input through unchanged
        res->leaf_too_small = 1;
        res->n_out = n;
        if (cap >= n) {
            hipLaunchKernelGGL(k_copy_records, dim3(nb), dim3(256), 0, stream, d_in, stride, ni, d_out, out_stride);
            VOX_TRY(hipGetLastError());
            VOX_TRY(hipStreamSynchronize(stream));
        }
        return hipSuccess;
    }
    res->n_out = (size_t)w->h_setup->n_out;
    return hipSuccess;
}

hipError_t vox_transform_frames(VoxWorkspace* w, hipStream_t stream, const unsigned char* const* h_src, size_t stride,
                                const int32_t* h_offsets, const float* h_T, int n_frames,
                                unsigned char* d_out, size_t out_stride)
{
    if (n_frames <= 0) return hipSuccess;
    const int n = h_offsets[n_frames];
    if (n <= 0) return hipSuccess;
    // one table: source pointers | offsets | transforms (each part 16-byte aligned)
    const size_t ptr_bytes = (sizeof(void*) * (size_t)n_frames + 15) & ~(size_t)15;
    const size_t off_bytes = (sizeof(int32_t) * (size_t)(n_frames + 1) + 15) & ~(size_t)15;
    const size_t t_bytes = sizeof(float) * 12 * (size_t)n_frames;
    VOX_TRY(w->frame_tab.ensure(ptr_bytes + off_bytes + t_bytes));
    unsigned char* tab = w->frame_tab.as<unsigned char>();
    // the caller keeps the three host arrays alive until its next synchronisation of `stream`
    VOX_TRY(hipMemcpyAsync(tab, h_src, sizeof(void*) * (size_t)n_frames, hipMemcpyHostToDevice, stream));
    VOX_TRY(hipMemcpyAsync(tab + ptr_bytes, h_offsets, sizeof(int32_t) * (size_t)(n_frames + 1), hipMemcpyHostToDevice, stream));
    VOX_TRY(hipMemcpyAsync(tab + ptr_bytes + off_bytes, h_T, t_bytes, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_transform_frames, dim3((n + 255) / 256), dim3(256), 0, stream,
                       (const unsigned char* const*)tab, stride, (const int32_t*)(tab + ptr_bytes),
                       (const float*)(tab + ptr_bytes + off_bytes), n_frames, d_out, out_stride);
    return hipGetLastError();
}

}  // namespace s2m
