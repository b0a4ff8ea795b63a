// s2m_kernels.hpp — hand-written HIP kernels (gfx950 / CDNA4, wave64) of the scan-to-map
// registration path. Compiled with -ffp-contract=off: every fp32 expression that mirrors the
// reference is evaluated as written (one rounding per operation), and HIP's default
// correctly-rounded fp32 divide/sqrt keeps the per-correspondence arithmetic bit-identical
// to an x86-64 (no-FMA) build of the reference.
//
// Reference citations are file:line into jimmyshe/liorf (src/mapOptmization.cpp unless named).
#pragma once
#include <float.h>
#include <math.h>
#include <string.h>
#include "s2m_types.h"

#if defined(__HIPCC__)
#define S2M_HD __host__ __device__
#else
#define S2M_HD
#endif

namespace s2m {

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
// Pointers fetched from the DevCtx block in memory are generic to the compiler, and generic
// (flat_*) loads return out of order, forcing vmcnt(0)+lgkmcnt(0) at every use. G() states
// that they point to global memory so that global_load/global_store with counted waits are used.
template <typename T> using gptr = __attribute__((address_space(1))) T*;
template <typename T> __device__ __forceinline__ gptr<T> G(T* p) { return (gptr<T>)p; }
typedef float v4f __attribute__((ext_vector_type(4)));
// The DevCtx block of a scan is written by tiny kernels between the loops and only read by the registration kernels: seen
// through the constant address space its fields - the buffer pointers above all - come by scalar loads into scalar
// registers (a generic pointer makes every one of them a vector load into a pair of vector registers, because some store
// of the kernel might alias it: sixteen registers per lane, and spills in the 128-register builds).
typedef const __attribute__((address_space(4))) DevCtx* CtxP;
__device__ __forceinline__ CtxP ctx_const(const DevCtx* p) { return (CtxP)p; }
__device__ __forceinline__ GridDesc grid_of(CtxP cp)
{
    GridDesc g;
    g.ox = cp->g.ox; g.oy = cp->g.oy; g.oz = cp->g.oz; g.inv_e = cp->g.inv_e; g.e = cp->g.e;
    g.nx = cp->g.nx; g.ny = cp->g.ny; g.nz = cp->g.nz; g.ncells = cp->g.ncells;
    return g;
}
__device__ __forceinline__ uint32_t f2ord(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ int cell_coord(float v, float o, float inv_e, int n)
{
    float f = floorf((v - o) * inv_e);
    f = fminf(fmaxf(f, 0.0f), (float)(n - 1));      // NaN -> 0, +-inf -> edge cells
    return (int)f;
}

__device__ __forceinline__ void swap_if(bool c, float& a, float& b)
{
    float t = a; a = c ? b : a; b = c ? t : b;
}
__device__ __forceinline__ void swap_if(bool c, int& a, int& b)
{
    int t = a; a = c ? b : a; b = c ? t : b;
}

// ------------------------------------------------------------------------------------------
// sinf / cosf as glibc >= 2.28 computes them (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, s_sincosf.h;
// S. Nagy's implementation [ext]): the reference builds its transform with std::sin/std::cos of floats
// (pcl::getTransformation, :348-351, and the LM trig, :1170-1175), i.e. with the host's libm, and glibc's
// sinf is not correctly rounded - near 0.3 rad it is one ulp off the correctly rounded value for 3 % of
// the arguments.  One ulp in the transform is enough to flip a marginal correspondence, so the launches
// that rebuild the transform on the device use the same arithmetic: argument reduction by a scaled
// float-to-int conversion, degree-7 / degree-8 polynomials in fp64, one rounding to fp32 at the end.
// Checked against glibc 2.35's sinf/cosf on 2e7 random arguments in [-4, 4]: no mismatch, with or without
// FMA contraction of the polynomial; tests/test_parity_gpu.py::test_device_trig_is_the_hosts compares the
// device results with the test host's libm.  |x| >= 120 does not occur for a pose angle and takes the fp64
// library function.
// ------------------------------------------------------------------------------------------
struct SincosfTable { double sign[4]; double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; };
S2M_HD inline const SincosfTable& sincosf_table(int negate)
{
    static constexpr SincosfTable T[2] = {
        { { 1.0, -1.0, -1.0, 1.0 }, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5,
          -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13 },
        { { 1.0, -1.0, -1.0, 1.0 }, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5,
          0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13 } };
    return T[negate];
}
S2M_HD inline uint32_t abstop12(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    return (u >> 20) & 0x7ffu;
}
S2M_HD inline float sinf_poly(double x, double x2, const SincosfTable& p, int n)
{
    if ((n & 1) == 0) {
        const double x3 = x * x2, s1 = p.s2 + x2 * p.s3, x7 = x3 * x2, s = x + x3 * p.s1;
        return (float)(s + x7 * s1);
    }
    const double x4 = x2 * x2, c2 = p.c3 + x2 * p.c4, c1 = p.c0 + x2 * p.c1, x6 = x4 * x2, c = c1 + x4 * p.c2;
    return (float)(c + x6 * c2);
}
// which = 0: sinf(y), 1: cosf(y)
S2M_HD inline float glibc_sincosf(float y, int which)
{
    double x = (double)y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {                      // |y| < pi/4
        if (abstop12(y) < abstop12(0x1p-12f)) return which ? 1.0f : y;
        return sinf_poly(x, x * x, sincosf_table(0), which);
    }
    if (abstop12(y) < abstop12(120.0f)) {
        const SincosfTable& p0 = sincosf_table(0);
        const double r = x * p0.hpi_inv;                                // reduce_fast: quadrant in bits 24..31
        const int n = ((int32_t)r + 0x800000) >> 24;
        x = x - (double)n * p0.hpi;
        const double sgn = p0.sign[n & 3];
        return sinf_poly(x * sgn, x * x, sincosf_table((n & 2) ? 1 : 0), n ^ which);
    }
    return which ? (float)cos(x) : (float)sin(x);
}

// Both at once for the launches that rebuild the transform: one argument reduction, and the two polynomials (independent
// chains) side by side; glibc's sinf and cosf share the reduction and differ only in which polynomial they return.
S2M_HD inline void glibc_sincosf_both(float y, float& sn, float& cs)
{
    double x = (double)y;
    int n = 0;
    int tbl = 0;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {                      // |y| < pi/4
        if (abstop12(y) < abstop12(0x1p-12f)) { sn = y; cs = 1.0f; return; }
    } else if (abstop12(y) < abstop12(120.0f)) {
        const SincosfTable& p0 = sincosf_table(0);
        const double r = x * p0.hpi_inv;
        n = ((int32_t)r + 0x800000) >> 24;
        x = x - (double)n * p0.hpi;
        x = x * p0.sign[n & 3];
        tbl = (n & 2) ? 1 : 0;
    } else { sn = (float)sin(x); cs = (float)cos(x); return; }
    const SincosfTable& p = sincosf_table(tbl);
    const double x2 = x * x;
    const float a = sinf_poly(x, x2, p, 0), b = sinf_poly(x, x2, p, 1);
    sn = (n & 1) ? b : a;
    cs = (n & 1) ? a : b;
}

// atanf as glibc computes it (sysdeps/ieee754/flt-32/s_atanf.c: the fdlibm algorithm in fp32 [ext]; no FMA
// variant exists for it): SCManager's xy2theta (include/Scancontext.cpp:23-36) calls atan on a float, i.e. the
// host's atanf, and a sector index is the ceiling of the angle - one ulp decides on which side of a sector
// boundary a point falls.  Checked against glibc 2.35's atanf on 2e7 random arguments: no mismatch; the
// GPU tests compare whole descriptors with the oracle, which calls the host's libm.
S2M_HD inline float glibc_atanf(float x)
{
    const float atanhi[4] = { 4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f };
    const float atanlo[4] = { 5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f };
    const float aT[11] = { 3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                           6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f };
    uint32_t u;
    memcpy(&u, &x, 4);
    const int32_t hx = (int32_t)u, ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {                              // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;               // NaN
        return (hx > 0) ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {                               // |x| < 0.4375
        if (ix < 0x31000000) return x;                   // |x| < 2^-29
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {                           // |x| < 1.1875
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else                 { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else                 { id = 3; x = -1.0f / x; }
        }
    }
    const float z = x * x, w = z * z;
    const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return (hx < 0) ? -r : r;
}

// ------------------------------------------------------------------------------------------
// index build: bounding box, cell histogram with per-point rank, exclusive scan, scatter
// ------------------------------------------------------------------------------------------
// Bounding box of the finite coordinates, as ordered uints (min x, y, z, max x, y, z).  Every workgroup leaves its own box in
// `part` (8 words each), k_bbox_fold puts them together: adds from every workgroup on one address are served one after the other on
// this part (~6 ns each) - six of them from each of 782 workgroups were 15 of this kernel's 20 us at 200 k points.
__global__ __launch_bounds__(256) void k_bbox(const unsigned char* __restrict__ pts, size_t stride, int n, uint32_t* __restrict__ part)
{
    __shared__ float smn[4][3], smx[4][3];
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float* p = reinterpret_cast<const float*>(pts + (size_t)i * stride);
#pragma unroll
        for (int d = 0; d < 3; d++) {
            float v = p[d];
            if (isfinite(v)) { mn[d] = fminf(mn[d], v); mx[d] = fmaxf(mx[d], v); }
        }
    }
#pragma unroll
    for (int d = 0; d < 3; d++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[d] = fminf(mn[d], __shfl_down(mn[d], off, 64));
            mx[d] = fmaxf(mx[d], __shfl_down(mx[d], off, 64));
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 3; d++) { smn[wave][d] = mn[d]; smx[wave][d] = mx[d]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int d = threadIdx.x;
        const float a = fminf(fminf(smn[0][d], smn[1][d]), fminf(smn[2][d], smn[3][d]));
        const float b = fmaxf(fmaxf(smx[0][d], smx[1][d]), fmaxf(smx[2][d], smx[3][d]));
        part[8 * blockIdx.x + d] = a <= b ? f2ord(a) : 0xffffffffu;          // (no finite value: the neutral elements)
        part[8 * blockIdx.x + 3 + d] = a <= b ? f2ord(b) : 0u;
    }
}
// mm[0..2] = min, mm[3..5] = max over the workgroups' boxes (0xffffffff / 0 where no coordinate was finite)
__global__ __launch_bounds__(256) void k_bbox_fold(const uint32_t* __restrict__ part, int nparts, uint32_t* __restrict__ mm)
{
    __shared__ uint32_t slo[4][3], shi[4][3];
    uint32_t lo[3] = { 0xffffffffu, 0xffffffffu, 0xffffffffu }, hi[3] = { 0u, 0u, 0u };
    for (int b = threadIdx.x; b < nparts; b += 256) {
#pragma unroll
        for (int d = 0; d < 3; d++) { lo[d] = min(lo[d], part[8 * b + d]); hi[d] = max(hi[d], part[8 * b + 3 + d]); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int d = 0; d < 3; d++) {
            lo[d] = min(lo[d], (uint32_t)__shfl_down((int)lo[d], off, 64));
            hi[d] = max(hi[d], (uint32_t)__shfl_down((int)hi[d], off, 64));
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 3; d++) { slo[wave][d] = lo[d]; shi[wave][d] = hi[d]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int d = threadIdx.x;
        mm[d] = min(min(slo[0][d], slo[1][d]), min(slo[2][d], slo[3][d]));
        mm[3 + d] = max(max(shi[0][d], shi[1][d]), max(shi[2][d], shi[3][d]));
    }
}

// linear cell id: x-fastest rows (map grid: a query's 3 x-neighbours are one contiguous run)
__device__ __forceinline__ int lin_rows(const GridDesc& g, int cx, int cy, int cz)
{
    return (cz * g.ny + cy) * g.nx + cx;
}
// Scan locality order (ordering only, never results): lidar returns are binned on a
// log-polar grid around the sensor (128 azimuth x 128 log-range bins; cell size grows with
// range like the return spacing does, so cells hold a few points each and the rank atomics
// do not pile up on the dense near-field cells), numbered along a Hilbert curve so that 64
// consecutive sorted points form a compact patch; a rigid transform keeps it compact.
constexpr int kPolarCells = 16384;
// Hilbert curve index of (x, y) on a 128 x 128 grid: consecutive indices are always adjacent
// cells (a Z-order curve jumps across the grid at every power-of-two boundary, and a wave that
// straddles a jump gets a huge bounding box).
__device__ __forceinline__ uint32_t hilbert128(uint32_t x, uint32_t y)
{
    uint32_t d = 0;
#pragma unroll
    for (uint32_t s = 64; s > 0; s >>= 1) {
        const uint32_t rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
        d += s * s * ((3u * rx) ^ ry);
        if (ry == 0) {
            if (rx == 1) { x = 127u - x; y = 127u - y; }
            const uint32_t t = x; x = y; y = t;
        }
    }
    return d;
}
__device__ __forceinline__ int polar_cell(float x, float y)
{
    const float rho = sqrtf(x * x + y * y);
    float az = atan2f(y, x);                                  // [-pi, pi]
    int ab = (int)((az + 3.14159265f) * (128.0f / 6.2831853f));
    ab = min(max(ab, 0), 127);
    float lr = (__log2f(fmaxf(rho, 0.5f)) + 1.0f) * (128.0f / 9.23f);   // 0.5 m .. 300 m
    int rb = (rho == rho) ? min(max((int)lr, 0), 127) : 0;
    if (!(az == az)) ab = 0;
    return (int)hilbert128((uint32_t)ab, (uint32_t)rb);
}

// Histogram with per-point rank, DETERMINISTIC: the position of a scan point in the locality order depends on the
// input alone, never on the order in which atomics happen to arrive - so the wave partition, the summation order of the
// normal equations and with them every bit of a registration's result are reproducible from run to run.
//   * inside a wave, points of the same cell are ranked by lane (match loop over the distinct cells of the wave);
//   * the 16 waves of a workgroup add their cell counts to the workgroup's LDS histogram one after the other;
//   * workgroup b writes its histogram as row b of a table; k_polar_prefix turns every column into an exclusive prefix
//     over the workgroups (rank of the workgroup's first point in that cell) and the cell totals.
// (Device-scope returning atomics, the obvious alternative, are also served memory-side on this multi-die part: 120 k
// of them cost 27 us.)
constexpr int kPolarBlock = 1024;
__global__ __launch_bounds__(kPolarBlock) void k_polar_count(const PrepTable tbl)
{
    const PrepSlot& ps = tbl.s[blockIdx.y];
    if ((int)blockIdx.x >= ps.npb) return;
    const unsigned char* __restrict__ pts = ps.pts; const size_t stride = ps.stride; const int n = ps.n;
    int32_t* __restrict__ cell_of = ps.cell_of; int32_t* __restrict__ rank_of = ps.rank_of; int32_t* __restrict__ block_hist = ps.block_hist;
    __shared__ int32_t hist[kPolarCells];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int k = t; k < kPolarCells; k += kPolarBlock) hist[k] = 0;
    const int i = blockIdx.x * kPolarBlock + t;
    int c = -1;
    if (i < n) {
        const float* p = reinterpret_cast<const float*>(pts + (size_t)i * stride);
        c = polar_cell(p[0], p[1]);
    }
    // rank among the lanes of this wave that share the cell, the group's size and its first lane
    int r_in = 0, cnt = 0, leader = lane;
    unsigned long long todo = __ballot(c >= 0);
    while (todo) {
        const int first = (int)__builtin_ctzll(todo);
        const int v = __builtin_amdgcn_readlane(c, first);
        const unsigned long long same = __ballot(c == v);
        if (c == v) { r_in = __popcll(same & ((1ull << lane) - 1ull)); cnt = __popcll(same); leader = first; }
        todo &= ~same;
    }
    int base = 0;
    for (int w = 0; w < kPolarBlock / 64; w++) {
        __syncthreads();
        if (wave == w && c >= 0 && leader == lane) { base = hist[c]; hist[c] = base + cnt; }     // distinct cells per leader: no conflict
    }
    base = __shfl(base, leader, 64);
    __syncthreads();
    if (i < n) { cell_of[i] = c; rank_of[i] = base + r_in; }
    int32_t* row = block_hist + (size_t)blockIdx.x * kPolarCells;
    for (int k = t; k < kPolarCells; k += kPolarBlock) row[k] = hist[k];
}

// Column-wise exclusive prefix of the workgroup histograms over the workgroups, in place, and the cell totals.
// 64 columns x 16 segments of workgroups per 1024 threads: consecutive lanes read consecutive columns.
__global__ __launch_bounds__(1024) void k_polar_prefix(const PrepTable tbl)
{
    const PrepSlot& ps = tbl.s[blockIdx.y];
    if (ps.n <= 0) return;                                // an empty slot of a batch (its row of the table holds no buffers)
    int32_t* __restrict__ H = ps.block_hist; const int nb = ps.npb; int32_t* __restrict__ counts = ps.counts;
    __shared__ int32_t seg[16][64];
    const int lane = threadIdx.x & 63, s = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int per = (nb + 15) / 16, b0 = min(s * per, nb), b1 = min(b0 + per, nb);
    int32_t sum = 0;
    for (int b = b0; b < b1; b++) sum += H[(size_t)b * kPolarCells + col];
    seg[s][lane] = sum;
    __syncthreads();
    int32_t run = 0, total = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) { const int32_t v = seg[q][lane]; run += (q < s) ? v : 0; total += v; }
    if (s == 0) counts[col] = total;
    for (int b = b0; b < b1; b++) {
        const size_t at = (size_t)b * kPolarCells + col;
        const int32_t v = H[at];
        H[at] = run;
        run += v;
    }
}

// exclusive scan of the 16384 polar cell counts by one workgroup (16 per thread); the counts are
// zeroed as they are read, ready for the next scan
__global__ __launch_bounds__(1024) void k_polar_scan(const PrepTable tbl)
{
    const PrepSlot& ps = tbl.s[blockIdx.y];
    if (ps.n <= 0) return;                                // an empty slot of a batch
    int32_t* __restrict__ counts = ps.counts; int32_t* __restrict__ start = ps.cell_start;
    __shared__ int32_t wsum[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int32_t v[16];
    int32_t sum = 0;
    int4* c4 = reinterpret_cast<int4*>(counts + t * 16);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int4 q = c4[k];
        c4[k] = make_int4(0, 0, 0, 0);
        v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
        sum += q.x + q.y + q.z + q.w;
    }
    int32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int32_t woff = 0;
    for (int w = 0; w < wave; w++) woff += wsum[w];
    int32_t run = woff + incl - sum;
    int4* s4 = reinterpret_cast<int4*>(start + t * 16);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int4 q;
        q.x = run; run += v[4 * k];
        q.y = run; run += v[4 * k + 1];
        q.z = run; run += v[4 * k + 2];
        q.w = run; run += v[4 * k + 3];
        s4[k] = q;
    }
}

__global__ void k_bin_count(const unsigned char* __restrict__ pts, size_t stride, int n, GridDesc g,
                            int32_t* __restrict__ cell_of, int32_t* __restrict__ rank_of,
                            int32_t* __restrict__ counts)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = reinterpret_cast<const float*>(pts + (size_t)i * stride);
    int cx = cell_coord(p[0], g.ox, g.inv_e, g.nx);
    int cy = cell_coord(p[1], g.oy, g.inv_e, g.ny);
    int cz = cell_coord(p[2], g.oz, g.inv_e, g.nz);
    int c = lin_rows(g, cx, cy, cz);
    cell_of[i] = c;
    rank_of[i] = atomicAdd(&counts[c], 1);
}

// exclusive scan, 1024 elements per 256-thread workgroup
__global__ __launch_bounds__(256) void k_scan_local(const int32_t* __restrict__ in, int32_t* __restrict__ out,
                                                    int32_t* __restrict__ block_sums, int n)
{
    __shared__ int32_t wsum[4];
    const int base = blockIdx.x * 1024 + threadIdx.x * 4;
    int32_t v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = (base + k < n) ? in[base + k] : 0;
    int32_t t = v[0] + v[1] + v[2] + v[3];
    int32_t incl = t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int32_t o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int32_t woff = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) if (w < wave) woff += wsum[w];
    int32_t excl = woff + incl - t;
#pragma unroll
    for (int k = 0; k < 4; k++) { if (base + k < n) out[base + k] = excl; excl += v[k]; }
    if (threadIdx.x == 255) block_sums[blockIdx.x] = woff + incl;
}

// in-place exclusive scan of the block sums by one workgroup (any nb)
__global__ __launch_bounds__(1024) void k_scan_sums(int32_t* __restrict__ sums, int nb)
{
    __shared__ int32_t wsum[16];
    __shared__ int32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < nb; base += 1024) {
        int i = base + threadIdx.x;
        int32_t t = (i < nb) ? sums[i] : 0;
        int32_t incl = t;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            int32_t o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int32_t woff = 0;
        for (int w = 0; w < wave; w++) woff += wsum[w];
        int32_t carry = carry_s;
        if (i < nb) sums[i] = carry + woff + incl - t;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + incl;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_scan_add(int32_t* __restrict__ out, const int32_t* __restrict__ block_sums,
                                                  int n, int total)
{
    const int base = blockIdx.x * 1024 + threadIdx.x * 4;
    const int32_t off = block_sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; k++) if (base + k < n) out[base + k] += off;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = total;
}

// k_scan_sums and k_scan_add in one launch for grids of up to a few thousand workgroups: every workgroup adds up the sums of the
// workgroups before it for itself (block_sums as k_scan_local left them)
__global__ __launch_bounds__(256) void k_scan_add_fold(int32_t* __restrict__ out, const int32_t* __restrict__ block_sums, int n, int total)
{
    __shared__ int32_t wsum[4];
    int32_t before = 0;
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += 256) before += block_sums[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = before;
    __syncthreads();
    const int32_t off = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    const int base = blockIdx.x * 1024 + threadIdx.x * 4;
#pragma unroll
    for (int k = 0; k < 4; k++) if (base + k < n) out[base + k] += off;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = total;
}

__global__ void k_scatter_map(const unsigned char* __restrict__ pts, size_t stride, int n,
                              const int32_t* __restrict__ cell_of, const int32_t* __restrict__ rank_of,
                              const int32_t* __restrict__ cell_start, float4* __restrict__ map_sorted)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = reinterpret_cast<const float*>(pts + (size_t)i * stride);
    int pos = cell_start[cell_of[i]] + rank_of[i];
    map_sorted[pos] = make_float4(p[0], p[1], p[2], __int_as_float(i));
}

__global__ void k_scatter_scan(const PrepTable tbl)
{
    const PrepSlot& ps = tbl.s[blockIdx.y];
    const unsigned char* __restrict__ pts = ps.pts; const size_t stride = ps.stride; const int n = ps.n;
    const int32_t* __restrict__ cell_of = ps.cell_of; const int32_t* __restrict__ rank_of = ps.rank_of;
    const int32_t* __restrict__ cell_start = ps.cell_start; const int32_t* __restrict__ block_hist = ps.block_hist;
    float* __restrict__ qx = ps.qx; float* __restrict__ qy = ps.qy; float* __restrict__ qz = ps.qz;
    int32_t* __restrict__ qperm = ps.qperm; float4* __restrict__ cert = ps.cert; int4* __restrict__ aux = ps.aux;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = reinterpret_cast<const float*>(pts + (size_t)i * stride);
    const int c = cell_of[i];
    // first point of the cell + points of the cell in earlier workgroups of k_polar_count + rank inside the workgroup
    int pos = cell_start[c] + block_hist[(size_t)(i / kPolarBlock) * kPolarCells + c] + rank_of[i];
    if ((unsigned)pos >= (unsigned)n) return;            // cannot happen while the histogram is consistent
    qx[pos] = p[0]; qy[pos] = p[1]; qz[pos] = p[2]; qperm[pos] = i;
    cert[pos] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);     // a new scan: no certificate (slack 0), no neighbour tuple, no plane
    aux[pos] = make_int4(0, 0, 0, 0);
}

// Work-proportional wave assignment.  The sorted scan is cut into chunks of 64 points; a chunk
// whose points spread over a large box (sparse far field, tall structures) would make its wave
// stage and sweep a large map tile, and the slowest wave sets the kernel time.  Such chunks are
// given to 2, 4 or 8 waves (32 / 16 / 8 points each: tighter boxes, run in parallel).  The extent is
// measured in the lidar frame - a rigid transform does not change it.
__global__ __launch_bounds__(256) void k_chunk_parts(const PrepTable tbl)
{
    const PrepSlot& ps = tbl.s[blockIdx.y];
    float* __restrict__ qx = ps.qx; float* __restrict__ qy = ps.qy; float* __restrict__ qz = ps.qz; int32_t* __restrict__ qperm = ps.qperm;
    const int n = ps.n, n_chunks = ps.n_chunks, base_parts = ps.base_parts; int32_t* __restrict__ parts = ps.chunk_parts;
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    const int i = c * 64 + lane;
    const bool v = i < n;
    float x = v ? qx[i] : 0.0f, y = v ? qy[i] : 0.0f, z = v ? qz[i] : 0.0f;
    int perm = v ? qperm[i] : 0;
    const bool f = v && fabsf(x) < 3.0e38f && fabsf(y) < 3.0e38f && fabsf(z) < 3.0e38f;
    float mn[3] = { f ? x : INFINITY, f ? y : INFINITY, f ? z : INFINITY };
    float mx[3] = { f ? x : -INFINITY, f ? y : -INFINITY, f ? z : -INFINITY };
#pragma unroll
    for (int d = 0; d < 3; d++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[d] = fminf(mn[d], __shfl_xor(mn[d], off, 64));
            mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], off, 64));
        }
    }
    // Order inside the chunk: the polar cells ignore height and the rank inside a cell is arbitrary, so
    // the 64 points of a chunk are in no useful order, and cutting the chunk into 2..8 waves would leave
    // every part as large as the whole.  Sort the chunk along a 3-D Morton curve over its own bounding
    // box (cubic cells, 16 along the longest axis; bitonic network over the wave, ties by scan index:
    // deterministic), so that every part is a compact blob whatever the sensor's heading is.  Lanes past
    // the end of the scan and non-finite points sort last.
    if (mn[0] <= mx[0]) {
        const float emax = fmaxf(fmaxf(mx[0] - mn[0], mx[1] - mn[1]), mx[2] - mn[2]);
        const float sc = emax > 0.0f ? 15.99f / emax : 0.0f;
        auto spread4 = [](uint32_t b) { return (b & 1u) | ((b & 2u) << 2) | ((b & 4u) << 4) | ((b & 8u) << 6); };   // bit k -> bit 3k
        uint32_t key = 0x7fffffffu;
        if (f) {
            const uint32_t ix = (uint32_t)((x - mn[0]) * sc), iy = (uint32_t)((y - mn[1]) * sc), iz = (uint32_t)((z - mn[2]) * sc);
            key = spread4(min(ix, 15u)) | (spread4(min(iy, 15u)) << 1) | (spread4(min(iz, 15u)) << 2);
        }
        int ord = v ? perm : 0x7fffffff;
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                const uint32_t okey = (uint32_t)__shfl_xor((int)key, j, 64);
                const int oord = __shfl_xor(ord, j, 64), operm = __shfl_xor(perm, j, 64);
                const float ox = __shfl_xor(x, j, 64), oy = __shfl_xor(y, j, 64), oz = __shfl_xor(z, j, 64);
                const bool lower = (lane & j) == 0, asc = (lane & k) == 0;
                const bool mine_first = (key < okey) || (key == okey && ord < oord);     // (key, ord) pairs of real points are distinct
                const bool keep = (lower == asc) ? mine_first : !mine_first;
                if (!keep) { key = okey; ord = oord; perm = operm; x = ox; y = oy; z = oz; }
            }
        }
        if (v) { qx[i] = x; qy[i] = y; qz[i] = z; qperm[i] = perm; }
    }
    if (lane == 0) {
        int p = 1;
        if (mn[0] <= mx[0]) {
            // cells a wave's box would span (1 m cells, +1 halo each side, bound radius)
            const float vol = (mx[0] - mn[0] + 3.0f) * (mx[1] - mn[1] + 3.0f) * (mx[2] - mn[2] + 3.0f);
            p = vol > 700.0f ? 4 : (vol > 180.0f ? 2 : 1);
        }
        // a small scan leaves most of the GPU idle: cut every chunk finer (base_parts) and large ones finer still
        parts[c] = min(8, max(p, base_parts) * ((p > 1 && base_parts > 1) ? 2 : 1));
    }
}

// one workgroup: exclusive scan of parts[] and emission of the wave table {first point, count}.
// The table has room for every chunk unsplit plus a budget of extra waves; if the wishes exceed
// it they are scaled back uniformly (cap 8 -> 4 -> 2 -> 1), so the table never overflows.
// `factor` (optional, consumed and zeroed): extra split asked for by k_wave_density; `st` (optional)
// receives the new wave count as well.
__device__ __forceinline__ void chunk_table_body(const int32_t* __restrict__ parts, int n, int n_chunks, int capacity,
                                                 int2* __restrict__ table, int32_t* __restrict__ n_waves_out,
                                                 int32_t* __restrict__ factor, DevState* __restrict__ st)
{
    __shared__ int32_t wsum[16];
    __shared__ int32_t carry_s, tot_s[3], base_s, pol_s[9];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // One workgroup per CU runs measurably faster than two (steady launch 25 vs 27.7 us at 120k points): the
    // density wishes are dropped if they alone would push the grid over one workgroup on each of the 256 CUs.
    constexpr int kOnePerCu = 256 * (kBlock / 64);
    bool use_factor = factor != nullptr;
    if (use_factor) {
        int tb = 0, tf = 0;
        for (int c = threadIdx.x; c < n_chunks; c += 1024) { tb += min(8, parts[c]); tf += min(8, parts[c] * max(factor[c], 1)); }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { tb += __shfl_xor(tb, off, 64); tf += __shfl_xor(tf, off, 64); }
        if (threadIdx.x == 0) { tot_s[0] = 0; base_s = 0; }
        __syncthreads();
        if (lane == 0) { atomicAdd(&base_s, tb); atomicAdd(&tot_s[0], tf); }
        __syncthreads();
        use_factor = !(base_s <= kOnePerCu && tot_s[0] > kOnePerCu);
        __syncthreads();
    }
    // pass A: total waves wished for under each way of scaling the wishes back, mildest first.  The density wishes are the ones
    // that count - a launch lasts as long as its slowest wave, and the slowest are the waves whose box streams thousands of map
    // points - so the chunks that asked for the finest cut keep it longest: 0 every wish; 1 / 2 density wishes below 4 / below 8
    // dropped; 3..5 as 2 with the other chunks capped at 4, 2, 1; 6..8 as 2 with every chunk capped at 4, 2, 1.
    // (Scaling every wish back alike left the heaviest chunks of a dense map at two parts: dense1m launch 0, slowest wave 223 us.)
    constexpr int kPolicies = 9;
    auto wish_under = [&](int c, int pol) {
        const int f = use_factor ? max(factor[c], 1) : 1;
        const bool heavy = f >= 8;
        const int fe = (pol == 0) ? f : ((pol == 1) ? (f >= 4 ? f : 1) : (heavy ? f : 1));
        const int w = min(8, parts[c] * fe);
        const int cap = (pol <= 2) ? 8 : ((pol <= 5) ? (heavy ? 8 : (32 >> pol)) : (256 >> pol));      // 3,4,5 -> 4,2,1; 6,7,8 -> 4,2,1
        return min(w, cap);
    };
    int tp[kPolicies];
#pragma unroll
    for (int k = 0; k < kPolicies; k++) tp[k] = 0;
    for (int c = threadIdx.x; c < n_chunks; c += 1024) {
#pragma unroll
        for (int k = 0; k < kPolicies; k++) tp[k] += wish_under(c, k);
    }
#pragma unroll
    for (int k = 0; k < kPolicies; k++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tp[k] += __shfl_xor(tp[k], off, 64);
    }
    if (threadIdx.x == 0) carry_s = 0;
    if (threadIdx.x < kPolicies) pol_s[threadIdx.x] = 0;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < kPolicies; k++) atomicAdd(&pol_s[k], tp[k]);
    }
    __syncthreads();
    // (Until the last session of round 4 a scan whose unsplit chunks fitted the 4 096 co-resident waves gave up every split before
    // it gave up that - measured on the round-2 kernel, 262 144 points = 4 096 chunks exactly: 20.5 k -> 22.2 k LM iterations/s.
    // With the rebuilt search the opposite holds: ouster128's launches 0-3 272 -> 222 us with its scattered and dense chunks split,
    // launch 1 below launch 0 at last, and the 30-launch loop no slower, 0.70 ms either way - the second entry some waves take in
    // a steady launch costs less than the stragglers of the cold ones.  The table's own capacity is the only limit now.)
    const int cap_eff = capacity;
    int pol = kPolicies - 1;
#pragma unroll
    for (int k = kPolicies - 2; k >= 0; k--) if (pol_s[k] <= cap_eff) pol = k;
    auto wish = [&](int c) { return wish_under(c, pol); };
    constexpr int pcap = 8;
    // pass B: scan and emit
    for (int base = 0; base < n_chunks; base += 1024) {
        const int c = base + threadIdx.x;
        const int p = (c < n_chunks) ? min(wish(c), pcap) : 0;
        int incl = p;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; w++) woff += wsum[w];
        const int carry = carry_s;
        const int off0 = carry + woff + incl - p;            // first wave slot of chunk c
        if (c < n_chunks) {
            const int per = 64 / p;
            for (int j = 0; j < p; j++) {
                const int start = c * 64 + j * per;
                if (off0 + j < capacity) table[off0 + j] = make_int2(start, max(0, min(per, n - start)));
            }
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int nw = min(carry_s, capacity);
        *n_waves_out = nw;
        if (st) st->n_waves = nw;
    }
    if (factor) for (int c = threadIdx.x; c < n_chunks; c += 1024) factor[c] = 0;   // every wish above was read before the last barrier
}

// per s2m_set_scan: extent-based parts only
__global__ __launch_bounds__(1024) void k_chunk_table(const PrepTable tbl)
{
    const PrepSlot& ps = tbl.s[blockIdx.y];
    if (ps.n <= 0) return;                                // an empty slot of a batch
    chunk_table_body(ps.chunk_parts, ps.n, ps.n_chunks, ps.capacity, ps.wave_table, ps.n_waves, nullptr, nullptr);
}

// per scan before launch 0, inside the captured loop: everything comes from the DevCtx block
__global__ __launch_bounds__(1024) void k_chunk_table_density(const SlotTable tbl)
{
    DevCtx* __restrict__ cp = const_cast<DevCtx*>(tbl.ctx[blockIdx.y]);
    DevState* __restrict__ st = tbl.st[blockIdx.y];
    if (!cp->density_pending) return;                     // once per scan (the flag is uniform: read before the barrier below)
    __syncthreads();
    if (threadIdx.x == 0) cp->density_pending = 0;
    chunk_table_body(cp->chunk_parts, cp->n_q, cp->n_chunks, cp->table_cap, cp->wave_table_rw, cp->n_waves_rw,
                     cp->chunk_factor, st);
}

// ------------------------------------------------------------------------------------------
// per-correspondence arithmetic
// ------------------------------------------------------------------------------------------
// 5x3 least-squares plane  A x = -1  by column-pivoted Householder QR in fp32: the
// algorithm behind `matA0.colPivHouseholderQr().solve(matB0)` (:1104, Eigen 3.3), with every
// index resolved at compile time so the 5x3 matrix lives in registers.
__device__ __forceinline__ void plane_fit_5x3(float (&qr)[5][3], float (&x)[3])
{
    float hC[3], nU[3], nD[3];
    int perm[3] = { 0, 1, 2 };
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 5; i++) s += qr[i][k] * qr[i][k];
        nD[k] = sqrtf(s); nU[k] = nD[k];
    }
    float maxn = nU[0];
    if (nU[1] > maxn) maxn = nU[1];
    if (nU[2] > maxn) maxn = nU[2];
    const float th = maxn * FLT_EPSILON;
    const float threshold_helper = (th * th) / 5.0f;
    const float norm_downdate_threshold = sqrtf(FLT_EPSILON);
    int np = 3;

#pragma unroll
    for (int k = 0; k < 3; k++) {
        int big = k; float bign = nU[k];
#pragma unroll
        for (int j = k + 1; j < 3; j++) if (nU[j] > bign) { bign = nU[j]; big = j; }
        const float big_sq = bign * bign;
        if (np == 3 && big_sq < threshold_helper * (float)(5 - k)) np = k;
#pragma unroll
        for (int j = k + 1; j < 3; j++) {
            const bool c = (big == j);
#pragma unroll
            for (int i = 0; i < 5; i++) swap_if(c, qr[i][k], qr[i][j]);
            swap_if(c, nU[k], nU[j]); swap_if(c, nD[k], nD[j]); swap_if(c, perm[k], perm[j]);
        }
        float tailSq = 0.0f;
#pragma unroll
        for (int i = k + 1; i < 5; i++) tailSq += qr[i][k] * qr[i][k];
        const float c0 = qr[k][k];
        float beta, tau;
        if (tailSq <= FLT_MIN) {
            tau = 0.0f; beta = c0;
#pragma unroll
            for (int i = k + 1; i < 5; i++) qr[i][k] = 0.0f;
        } else {
            beta = sqrtf(c0 * c0 + tailSq);
            if (c0 >= 0.0f) beta = -beta;
            const float den = c0 - beta;
#pragma unroll
            for (int i = k + 1; i < 5; i++) qr[i][k] = qr[i][k] / den;
            tau = (beta - c0) / beta;
        }
        qr[k][k] = beta; hC[k] = tau;
        if (tau != 0.0f) {
#pragma unroll
            for (int j = k + 1; j < 3; j++) {
                float tmp = 0.0f;
#pragma unroll
                for (int i = k + 1; i < 5; i++) tmp += qr[i][k] * qr[i][j];
                tmp += qr[k][j];
                qr[k][j] -= tau * tmp;
#pragma unroll
                for (int i = k + 1; i < 5; i++) qr[i][j] -= (tau * qr[i][k]) * tmp;
            }
        }
#pragma unroll
        for (int j = k + 1; j < 3; j++) {
            if (nU[j] != 0.0f) {
                float temp = fabsf(qr[k][j]) / nU[j];
                temp = (1.0f + temp) * (1.0f - temp);
                temp = temp < 0.0f ? 0.0f : temp;
                const float r = nU[j] / nD[j];
                const float temp2 = temp * (r * r);
                if (temp2 <= norm_downdate_threshold) {
                    float s = 0.0f;
#pragma unroll
                    for (int i = k + 1; i < 5; i++) s += qr[i][j] * qr[i][j];
                    nD[j] = sqrtf(s); nU[j] = nD[j];
                } else {
                    nU[j] *= sqrtf(temp);
                }
            }
        }
    }

    x[0] = x[1] = x[2] = 0.0f;
    if (np == 0) return;
    float c[5] = { -1.0f, -1.0f, -1.0f, -1.0f, -1.0f };              // matB0.fill(-1) (:1094)
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (k < np) {
            const float tau = hC[k];
            if (tau != 0.0f) {
                float tmp = 0.0f;
#pragma unroll
                for (int i = k + 1; i < 5; i++) tmp += qr[i][k] * c[i];
                tmp += c[k];
                c[k] -= tau * tmp;
#pragma unroll
                for (int i = k + 1; i < 5; i++) c[i] -= (tau * qr[i][k]) * tmp;
            }
        }
    }
#pragma unroll
    for (int i = 2; i >= 0; i--) {
        if (i < np) {
            if (c[i] != 0.0f) {
                c[i] /= qr[i][i];
#pragma unroll
                for (int r = 0; r < i; r++) c[r] -= c[i] * qr[r][i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        if (i < np) {
#pragma unroll
            for (int t = 0; t < 3; t++) if (perm[i] == t) x[t] = c[i];
        }
    }
}

// one row of matA / matB (:1216-1234); sc = srx,crx,sry,cry,srz,crz (:1170-1175)
__device__ __forceinline__ void jacobian_row(const float (&sc)[6], float px, float py, float pz,
                                             const float (&cf)[4], float (&row)[6], float& rhs)
{
    const float srx = sc[0], crx = sc[1], sry = sc[2], cry = sc[3], srz = sc[4], crz = sc[5];
    const float arx = (-srx * cry * px - (srx * sry * srz + crx * crz) * py + (crx * srz - srx * sry * crz) * pz) * cf[0]
                    + (crx * cry * px - (srx * crz - crx * sry * srz) * py + (crx * sry * crz + srx * srz) * pz) * cf[1];
    const float ary = (-crx * sry * px + crx * cry * srz * py + crx * cry * crz * pz) * cf[0]
                    + (-srx * sry * px + srx * sry * srz * py + srx * cry * crz * pz) * cf[1]
                    + (-cry * px - sry * srz * py - sry * crz * pz) * cf[2];
    const float arz = ((crx * sry * crz + srx * srz) * py + (srx * crz - crx * sry * srz) * pz) * cf[0]
                    + ((-crx * srz + srx * sry * crz) * py + (-srx * sry * srz - crx * crz) * pz) * cf[1]
                    + (cry * crz * py - cry * srz * pz) * cf[2];
    row[0] = arz; row[1] = ary; row[2] = arx; row[3] = cf[0]; row[4] = cf[1]; row[5] = cf[2];
    rhs = -cf[3];
}

// ------------------------------------------------------------------------------------------
// shared constants and wave-level helpers of the registration kernel (s2m_register.hpp)
// ------------------------------------------------------------------------------------------
constexpr int kTilePts = 512;        // points per wave tile (8 KiB) after filtering; a tile slot fits the low 9 bits of a sweep key
constexpr int kTileRaw = 1024;       // unfiltered points of one 64-row group a wave is willing to stream through the filter
constexpr int kRowMax = 256;         // boxes with more rows go straight to the gather path
constexpr float kSlabMargin = 1e-3f; // covers the fp32 rounding of the cell binning of a local map (<= 5e-5 at 200 m extent; s2m_register.hpp adds 1e-6 per metre)
constexpr uint64_t kKeyInf = ((uint64_t)0x7f800000u << 32) | 0x7fffffffu;

// LDS written by this wave is read back by other lanes of the same wave: DS operations of one
// wave execute in order, so only the compiler has to be kept from reordering them.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// wave64 reductions / scan on the DPP cross-lane path (VALU rate, no LDS round trips):
// row_shr 1,2,4,8 inside the 16-lane rows, row_bcast:15 into rows 1 and 3, row_bcast:31 into
// rows 2 and 3; lane 63 then holds the result.  All 64 lanes must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float old, float src)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_min_f32(float v)      // NaN-free inputs
{
    const float ID = INFINITY;
    v = fminf(v, dpp_f32<0x111, 0xf>(ID, v)); v = fminf(v, dpp_f32<0x112, 0xf>(ID, v));
    v = fminf(v, dpp_f32<0x114, 0xf>(ID, v)); v = fminf(v, dpp_f32<0x118, 0xf>(ID, v));
    v = fminf(v, dpp_f32<0x142, 0xa>(ID, v)); v = fminf(v, dpp_f32<0x143, 0xc>(ID, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_f32(float v)
{
    const float ID = -INFINITY;
    v = fmaxf(v, dpp_f32<0x111, 0xf>(ID, v)); v = fmaxf(v, dpp_f32<0x112, 0xf>(ID, v));
    v = fmaxf(v, dpp_f32<0x114, 0xf>(ID, v)); v = fmaxf(v, dpp_f32<0x118, 0xf>(ID, v));
    v = fmaxf(v, dpp_f32<0x142, 0xa>(ID, v)); v = fmaxf(v, dpp_f32<0x143, 0xc>(ID, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_max_i32(int v)
{
    constexpr int ID = (int)0x80000000;
    v = max(v, dpp_i32<0x111, 0xf>(ID, v)); v = max(v, dpp_i32<0x112, 0xf>(ID, v));
    v = max(v, dpp_i32<0x114, 0xf>(ID, v)); v = max(v, dpp_i32<0x118, 0xf>(ID, v));
    v = max(v, dpp_i32<0x142, 0xa>(ID, v)); v = max(v, dpp_i32<0x143, 0xc>(ID, v));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t u)
{
    constexpr int ID = -1;                                   // 0xffffffff
    int v = (int)u;
    auto mn = [](int a, int b) { return (int)min((uint32_t)a, (uint32_t)b); };
    v = mn(v, dpp_i32<0x111, 0xf>(ID, v)); v = mn(v, dpp_i32<0x112, 0xf>(ID, v));
    v = mn(v, dpp_i32<0x114, 0xf>(ID, v)); v = mn(v, dpp_i32<0x118, 0xf>(ID, v));
    v = mn(v, dpp_i32<0x142, 0xa>(ID, v)); v = mn(v, dpp_i32<0x143, 0xc>(ID, v));
    return (uint32_t)__builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t u)
{
    int v = (int)u;
    v |= dpp_i32<0x111, 0xf>(0, v); v |= dpp_i32<0x112, 0xf>(0, v);
    v |= dpp_i32<0x114, 0xf>(0, v); v |= dpp_i32<0x118, 0xf>(0, v);
    v |= dpp_i32<0x142, 0xa>(0, v); v |= dpp_i32<0x143, 0xc>(0, v);
    return (uint32_t)__builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_incl_scan_i32(int v)
{
    v += dpp_i32<0x111, 0xf>(0, v); v += dpp_i32<0x112, 0xf>(0, v);
    v += dpp_i32<0x114, 0xf>(0, v); v += dpp_i32<0x118, 0xf>(0, v);
    v += dpp_i32<0x142, 0xa>(0, v); v += dpp_i32<0x143, 0xc>(0, v);
    return v;
}

// (dy,dz) of the 9 rows of a 3x3x3 neighbourhood, centre row first so that the 5th-best bound
// tightens early: (0,0),(-1,0),(1,0),(0,-1),(0,1),(-1,-1),(1,-1),(-1,1),(1,1), 2 bits each
__device__ __forceinline__ constexpr int run_dy(int k) { return (int)((139617u >> (2 * k)) & 3u) - 1; }
__device__ __forceinline__ constexpr int run_dz(int k) { return (int)((164373u >> (2 * k)) & 3u) - 1; }

// ------------------------------------------------------------------------------------------
// 6x6 algebra of LMOptimization (:1237-1271)
// ------------------------------------------------------------------------------------------
// cv::eigen (:1248): max-pivot Jacobi in fp32, eigenvalues descending, eigenvectors as rows.
// A and V are LDS scratch of the calling lane (data-dependent indexing).
__device__ void eigen6_sym(float (*A)[6], float (*V)[6], float* W, int* indR, int* indC)
{
    constexpr int N = 6;
    const float eps = FLT_EPSILON;
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) V[i][j] = (i == j) ? 1.0f : 0.0f;
    for (int k = 0; k < N; k++) {
        W[k] = A[k][k];
        if (k < N - 1) {
            int m = k + 1; float mv = fabsf(A[k][m]);
            for (int i = k + 2; i < N; i++) { float v = fabsf(A[k][i]); if (mv < v) { mv = v; m = i; } }
            indR[k] = m;
        }
        if (k > 0) {
            int m = 0; float mv = fabsf(A[0][k]);
            for (int i = 1; i < k; i++) { float v = fabsf(A[i][k]); if (mv < v) { mv = v; m = i; } }
            indC[k] = m;
        }
    }
    for (int it = 0; it < N * N * 30; it++) {
        int k = 0, l; float mv = fabsf(A[0][indR[0]]);
        for (int i = 1; i < N - 1; i++) { float v = fabsf(A[i][indR[i]]); if (mv < v) { mv = v; k = i; } }
        l = indR[k];
        for (int i = 1; i < N; i++) { float v = fabsf(A[indC[i]][i]); if (mv < v) { mv = v; k = indC[i]; l = i; } }
        const float p = A[k][l];
        if (fabsf(p) <= eps) break;
        const float y = (W[l] - W[k]) * 0.5f;
        float t = fabsf(y) + hypotf(p, y);
        float s = hypotf(p, t);
        const float c = t / s;
        s = p / s; t = (p / t) * p;
        if (y < 0.0f) { s = -s; t = -t; }
        A[k][l] = 0.0f;
        W[k] -= t; W[l] += t;
#define S2M_ROT(v0, v1) do { const float a0 = (v0), b0 = (v1); (v0) = a0 * c - b0 * s; (v1) = a0 * s + b0 * c; } while (0)
        for (int i = 0; i < k; i++)     S2M_ROT(A[i][k], A[i][l]);
        for (int i = k + 1; i < l; i++) S2M_ROT(A[k][i], A[i][l]);
        for (int i = l + 1; i < N; i++) S2M_ROT(A[k][i], A[l][i]);
        for (int i = 0; i < N; i++)     S2M_ROT(V[k][i], V[l][i]);
#undef S2M_ROT
        for (int j = 0; j < 2; j++) {
            const int idx = j == 0 ? k : l;
            if (idx < N - 1) {
                int m = idx + 1; float mv2 = fabsf(A[idx][m]);
                for (int i = idx + 2; i < N; i++) { float v = fabsf(A[idx][i]); if (mv2 < v) { mv2 = v; m = i; } }
                indR[idx] = m;
            }
            if (idx > 0) {
                int m = 0; float mv2 = fabsf(A[0][idx]);
                for (int i = 1; i < idx; i++) { float v = fabsf(A[i][idx]); if (mv2 < v) { mv2 = v; m = i; } }
                indC[idx] = m;
            }
        }
    }
    for (int k = 0; k < N - 1; k++) {
        int m = k;
        for (int i = k + 1; i < N; i++) if (W[m] < W[i]) m = i;
        if (k != m) {
            float tw = W[m]; W[m] = W[k]; W[k] = tw;
            for (int i = 0; i < N; i++) { float tv = V[m][i]; V[m][i] = V[k][i]; V[k][i] = tv; }
        }
    }
}

// cv::Mat::inv() (:1263): LU with partial pivoting applied to [A | I], fp32. A is destroyed.
__device__ bool inv6_lu(float (*A)[6], float (*B)[6])
{
    constexpr int N = 6;
    const float eps = FLT_EPSILON * 10.0f;
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) B[i][j] = (i == j) ? 1.0f : 0.0f;
    for (int i = 0; i < N; i++) {
        int k = i;
        for (int j = i + 1; j < N; j++) if (fabsf(A[j][i]) > fabsf(A[k][i])) k = j;
        if (fabsf(A[k][i]) < eps) {
            for (int a = 0; a < N; a++) for (int b = 0; b < N; b++) B[a][b] = 0.0f;
            return false;
        }
        if (k != i) {
            for (int j = i; j < N; j++) { float t = A[i][j]; A[i][j] = A[k][j]; A[k][j] = t; }
            for (int j = 0; j < N; j++) { float t = B[i][j]; B[i][j] = B[k][j]; B[k][j] = t; }
        }
        const float d = -1.0f / A[i][i];
        for (int j = i + 1; j < N; j++) {
            const float alpha = A[j][i] * d;
            for (int m = i + 1; m < N; m++) A[j][m] += alpha * A[i][m];
            for (int m = 0; m < N; m++) B[j][m] += alpha * B[i][m];
        }
    }
    for (int i = N - 1; i >= 0; i--)
        for (int j = 0; j < N; j++) {
            float s = B[i][j];
            for (int k = i + 1; k < N; k++) s -= A[i][k] * B[k][j];
            B[i][j] = s / A[i][i];
        }
    return true;
}

__device__ __forceinline__ void store_trace(gptr<s2m_iter_trace> dst, const s2m_iter_trace& tr)
{
    dst->n_sel = tr.n_sel; dst->stepped = tr.stepped; dst->deltaR = tr.deltaR; dst->deltaT = tr.deltaT;
#pragma unroll
    for (int k = 0; k < 6; k++) { dst->delta[k] = tr.delta[k]; dst->pose[k] = tr.pose[k]; }
}

// cv::solve(matAtA, matAtB, matX, DECOMP_QR) (:1240) spread over 7 lanes of one wave: lane j < 6
// owns column j of AtA, lane 6 the right-hand side.  Every number goes through exactly the operations,
// in the order, of OpenCV's hal::QR32f (un-pivoted Householder QR in fp32 with unit-length reflectors,
// the reflector kept as v/v[0] below the diagonal and re-expanded for the right-hand side, back
// substitution; OpenCV >= 3.3).  The code is branch-free and LDS-free: all lanes run the reflector
// arithmetic on their own column and lane l's result is broadcast with v_readlane.  The chain is bound
// by instruction issue (a wave64 instruction occupies the SIMD for 4 cycles however few lanes matter,
// a correctly rounded fp32 divide is ~10 instructions), so the independent divides of a reflector are
// spread over the lanes - lane i divides element i - and the quotients v[i]/v[0] (needed by lane l for
// storage and by lane 6 for the rhs) are computed once.  Returns x in every lane.
__device__ __forceinline__ float lane_bcast(float v, int src_lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

// element `idx` (0..5, per lane) of a register array as a select chain.  Left alone, the compiler recognises the chain
// as a dynamically indexed array and goes through scratch memory (a store and a dependent load: ~0.2 us each, twelve of
// them in the solve); the empty asm statements keep the selects apart.
__device__ __forceinline__ float pick6(const float (&a)[6], int idx)
{
    float r = a[0];
#pragma unroll
    for (int i = 1; i < 6; i++) {
        r = (idx == i) ? a[i] : r;
        asm volatile("" : "+v"(r));
    }
    return r;
}

__device__ __forceinline__ bool solve6_qr_lanes(int lane, float (&col)[6], float (&x)[6])
{
    const int li = min(lane, 5);
#pragma unroll
    for (int l = 0; l < 6; l++) {
        // column l lives in lane l: bring its tail and the two norms to every lane
        float vl[6], u[6];
        float nrm = 0.0f;
#pragma unroll
        for (int i = 0; i < 6 - l; i++) { vl[i] = col[l + i]; nrm += vl[i] * vl[i]; }
        const float tmpV = vl[0];
        vl[0] = vl[0] + (vl[0] >= 0.0f ? 1.0f : -1.0f) * sqrtf(nrm);
        nrm = sqrtf(nrm + vl[0] * vl[0] - tmpV * tmpV);
        nrm = lane_bcast(nrm, l);
#pragma unroll
        for (int i = 0; i < 6; i++) vl[i] = (i < 6 - l) ? lane_bcast(vl[i], l) : 0.0f;
        // the 6-l divides by the norm, one per lane, then the 5-l quotients v[i]/v[0], one per lane
        const float q = pick6(vl, li) / nrm;
#pragma unroll
        for (int i = 0; i < 6 - l; i++) vl[i] = lane_bcast(q, i);
        const float hf = vl[0] * vl[0];
        const float qu = pick6(vl, li) / vl[0];
        u[0] = 1.0f;
#pragma unroll
        for (int i = 1; i < 6 - l; i++) u[i] = lane_bcast(qu, i);
        // columns l..5: A -= 2 v (v^T A); the rhs (lane 6) gets the same reflector rebuilt from its stored
        // form, col -= ((2 u) (u^T col)) hf.  One dot product per lane, on the vector that lane needs.
        const bool is_col = lane >= l && lane < 6, is_rhs = lane == 6;
        float w[6], dot = 0.0f;
#pragma unroll
        for (int i = l; i < 6; i++) { w[i - l] = is_rhs ? u[i - l] : vl[i - l]; dot += w[i - l] * col[i]; }
#pragma unroll
        for (int i = l; i < 6; i++) {
            const float t = 2.0f * w[i - l] * dot;
            const float c = col[i] - (is_rhs ? t * hf : t);
            col[i] = (is_col || is_rhs) ? c : col[i];
        }
#pragma unroll
        for (int i = 1; i < 6 - l; i++) col[l + i] = (lane == l) ? u[i] : col[l + i];
    }
    // back substitution on wave-uniform values: R[i][j] = column j's entry i, b = lane 6's column
    bool ok = true;
    float b[6];
#pragma unroll
    for (int i = 0; i < 6; i++) b[i] = lane_bcast(col[i], 6);
#pragma unroll
    for (int i = 5; i >= 0; i--) {
#pragma unroll
        for (int j = 5; j > i; j--) b[i] -= b[j] * lane_bcast(col[i], j);
        const float d = lane_bcast(col[i], i);
        if (fabsf(d) < FLT_EPSILON * 10.0f) ok = false;
        b[i] /= d;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) x[i] = ok ? b[i] : 0.0f;
    return ok;
}

// Is every eigenvalue of the symmetric 6x6 A safely above `thresh`?  Cholesky of A - s*I in
// fp64 with s = thresh + margin, margin = 1e-5 * trace(A) (far above the fp32 Jacobi's error of a
// few ulp of the largest eigenvalue, far below any eigenvalue that matters): if it succeeds,
// cv::eigen would report all six eigenvalues >= thresh and isDegenerate stays false (:1251-1262).
__device__ bool all_eigen_above(const float* A, float thresh)
{
    double tr = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) tr += (double)A[i * 6 + i];
    if (!(tr > 0.0) || !(tr < 1.0e30)) return false;
    const double sft = (double)thresh + 1e-5 * tr;
    double L[6][6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double d = (double)A[j * 6 + j] - sft;
#pragma unroll
        for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k];
        if (!(d > 1e-9 * tr)) return false;
        const double dj = sqrt(d);
        L[j][j] = dj;
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            double v = (double)A[i * 6 + j];
#pragma unroll
            for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k];
            L[i][j] = v / dj;
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------------
// The rest of LMOptimization (:1177-1292) after the per-point work: second stage of the
// AtA/AtB reduction (fixed order: bitwise reproducible for a given workgroup partition and
// workgroup size), the 6x6 solve, the iteration-0 degeneracy analysis, the pose update and the
// convergence test.  Two callers share it:
//   * k_finalize (one workgroup of 1024) closes iteration 0, which needs the degeneracy
//     analysis, and the last iteration of a scan;
//   * k_register closes iteration L-1 in the prologue of launch L >= 2: every workgroup reduces
//     the previous launch's partial sums and solves the 6x6 system for itself (same numbers in
//     the same order, so every workgroup arrives at the same pose), and workgroup 0 records
//     the outcome.  That removes a kernel boundary and a single-workgroup kernel from every
//     iteration of the loop (:1304-1315); the few microseconds of redundant algebra overlap
//     with the loads of the scan points and their priors.
// Loop state is double-buffered by launch parity so that nothing read in a launch is written
// in it: launch L uses the pose in pose2[L & 1] and writes its partial sums to slot L & 1.
// ------------------------------------------------------------------------------------------
struct LmShared {
    double part[kFinThreads / 32][32];
    double tot[32];
    float  AtA[36], AtB[6];
    float  eA[6][6], eV[6][6], eVi[6][6], eV2[6][6], eW[6];      // iteration-0 analysis only
    int    eR[6], eC[6];
};

// Closes iteration `iter`: matAtA / matAtB / laserCloudSelNum from the partial sums of launch `iter`
// (:1182-1239), then solve, project, update, test (:1177-1180, :1240-1292).  `pose0` is the pose launch
// `iter` ran with, `nb_act` the number of workgroups that wrote partial sums.  All NT threads of the
// workgroup call it; after the first barrier wave 0 works alone (wave-level LDS ordering only) and
// publishes {pose[6], ended} in `s_out` (8 floats of LDS that outlive `sh`); all threads return the
// updated pose in pose_out and whether the loop has ended (converged with early exit on, or fewer
// than min_corr correspondences).  kFull adds the iteration-0 degeneracy analysis on wave 1.
// `writer` records the outcome in DevState / trace.  ne_only = normal equations only (observation hook).
// `early` / `late`: work of the caller that does not depend on the pose and hides behind the serial part - called by every
// thread once the partial sums are in (everything requested before the call has arrived by then), and again just before
// the last barrier (wave 0: after the solve; the other waves: at once, they would only wait there).
struct NoHook { __device__ __forceinline__ void operator()() const {} };
template <int NT, bool kFull, typename Early = NoHook, typename Late = NoHook>
__device__ __forceinline__ bool lm_close_iteration(CtxP cp, gptr<DevState> st, int nb_act, int iter,
                                                   bool writer, bool ne_only, const float (&pose0)[6], int degen0,
                                                   LmShared& sh, float* s_out, float (&pose_out)[6],
                                                   unsigned long long* stamps = nullptr,     // diagnostics: 5 wall-clock stamps
                                                   Early early = Early(), Late late = Late())
{
    constexpr int NG = NT / 32;                         // row groups
    const auto trace = G(cp->trace);
    const int t = threadIdx.x, col = t & 31, grp = t >> 5, lane = t & 63, wave = t >> 6;
    double s = 0.0;
    if (col < kAcc) {
        const int nbk = cp->nblocks;
        const auto P = G((const double*)cp->partials) + (size_t)(iter & 1) * (size_t)nbk * kAcc;
        const int nb = min(nbk, nb_act);                // active workgroups
        for (int b0 = grp; b0 < nb; b0 += 16 * NG) {    // 16 independent loads in flight per lane
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) { const int b = b0 + u * NG; v[u] = (b < nb) ? P[(size_t)b * kAcc + col] : 0.0; }
#pragma unroll
            for (int u = 0; u < 16; u++) s += v[u];
        }
    }
    sh.part[grp][col] = s;
    __syncthreads();
    if (stamps) stamps[0] = wall_clock64();
    early();

    int n_sel = 0;
    if (wave == 0) {
        if (t < kAcc) {
            double v = 0.0;
#pragma unroll
            for (int g2 = 0; g2 < NG; g2++) v += sh.part[g2][t];
            sh.tot[t] = v;
        }
        wave_lds_sync();
        // fp32 matrices (:1184-1186): entry (a, b), a <= b, is sum number a*6 - a(a-1)/2 + (b-a)
        n_sel = (int)sh.tot[27];
        if (t < 36) {
            const int a = min(t / 6, t % 6), b = max(t / 6, t % 6);
            const float v = (float)sh.tot[a * 6 - (a * (a - 1)) / 2 + (b - a)];
            sh.AtA[t] = v;
            if (writer) st->AtA[t] = v;
        } else if (t < 42) {
            const float v = (float)sh.tot[21 + (t - 36)];
            sh.AtB[t - 36] = v;
            if (writer) st->AtB[t - 36] = v;
        }
        if (writer && t == 0) st->n_sel_last = n_sel;
        wave_lds_sync();
    }
    if (stamps) stamps[1] = wall_clock64();
    if (ne_only) return false;

    if (kFull && iter == 0) {                           // :1242-1264, wave 1 lane 0
        __syncthreads();
        if (wave == 1 && lane == 0 && (int)sh.tot[27] >= cp->min_corr) {
            int degenerate = 0;
            if (!all_eigen_above(sh.AtA, cp->eig_thresh)) {
                // the full restatement: cv::eigen, the row-zeroing loop, matP = matV.inv() * matV2
                for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) sh.eA[i][j] = sh.AtA[i * 6 + j];
                eigen6_sym(sh.eA, sh.eV, sh.eW, sh.eR, sh.eC);
                for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) sh.eV2[i][j] = sh.eV[i][j];
                for (int i = 5; i >= 0; i--) {
                    if (sh.eW[i] < cp->eig_thresh) { for (int j = 0; j < 6; j++) sh.eV2[i][j] = 0.0f; degenerate = 1; }
                    else break;
                }
                for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) sh.eA[i][j] = sh.eV[i][j];
                inv6_lu(sh.eA, sh.eVi);
                for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) {      // matP = matV.inv() * matV2
                    double a = 0.0;
                    for (int k = 0; k < 6; k++) a += (double)sh.eVi[i][k] * (double)sh.eV2[k][j];
                    st->matP[i * 6 + j] = (float)a;
                }
            }
            // not degenerate: matP is never read before the next scan's iteration 0 rewrites it
            st->isDegenerate = degenerate;
            __threadfence_block();
        }
        __syncthreads();                                // thread 0 reads what wave 1 decided
    }

    if (wave == 0) {
        if (n_sel < cp->min_corr) {                     // :1178-1180: false, pose unchanged
            if (t == 0) {
                if (writer) {
                    s2m_iter_trace tr;
                    tr.n_sel = n_sel; tr.stepped = 0; tr.deltaR = 0.0f; tr.deltaT = 0.0f;
#pragma unroll
                    for (int k = 0; k < 6; k++) { tr.pose[k] = pose0[k]; tr.delta[k] = 0.0f; st->pose[k] = pose0[k]; }
                    st->stalled = 1; st->done = 1;      // the remaining iterations repeat this no-op
                    st->iters_run = cp->max_iter;
                    store_trace(trace + iter, tr);
                }
#pragma unroll
                for (int k = 0; k < 6; k++) s_out[k] = pose0[k];
                s_out[6] = 1.0f;
            }
        } else {
            float colv[6], X[6];
#pragma unroll
            for (int i = 0; i < 6; i++) colv[i] = (lane < 6) ? sh.AtA[i * 6 + lane] : ((lane == 6) ? sh.AtB[i] : 0.0f);
            solve6_qr_lanes(lane, colv, X);             // :1240, result in every lane
            if (stamps) stamps[2] = wall_clock64();
            if (t == 0) {                               // projection, pose update, convergence test (:1266-1292)
                float pose[6];
#pragma unroll
                for (int k = 0; k < 6; k++) pose[k] = pose0[k];
                if ((kFull && iter == 0) ? (st->isDegenerate != 0) : (degen0 != 0)) {   // :1266-1271 (iteration 0: just decided by wave 1)
                    float X2[6];
#pragma unroll
                    for (int k = 0; k < 6; k++) X2[k] = X[k];
                    for (int i = 0; i < 6; i++) {
                        double a = 0.0;
                        for (int k = 0; k < 6; k++) a += (double)st->matP[i * 6 + k] * (double)X2[k];
                        X[i] = (float)a;
                    }
                }
#pragma unroll
                for (int k = 0; k < 6; k++) { pose[k] += X[k]; s_out[k] = pose[k]; }        // :1273-1278
                const double r0 = (double)(X[0] * 57.29578f), r1 = (double)(X[1] * 57.29578f), r2 = (double)(X[2] * 57.29578f);
                const float deltaR = (float)sqrt(r0 * r0 + r1 * r1 + r2 * r2);          // :1280-1283
                const double t0 = (double)(X[3] * 100), t1 = (double)(X[4] * 100), t2 = (double)(X[5] * 100);
                const float deltaT = (float)sqrt(t0 * t0 + t1 * t1 + t2 * t2);          // :1284-1287
                const bool conv = ((double)deltaR < cp->conv_deg) && ((double)deltaT < cp->conv_cm);   // :1289
                s_out[6] = (conv && cp->early_exit) ? 1.0f : 0.0f;                      // break (:1313-1314)
                if (writer) {
                    s2m_iter_trace tr;
                    tr.n_sel = n_sel; tr.stepped = 1; tr.deltaR = deltaR; tr.deltaT = deltaT;
#pragma unroll
                    for (int k = 0; k < 6; k++) { tr.delta[k] = X[k]; tr.pose[k] = pose[k]; }
                    // launch iter+1 rebuilds its transform from this pose (k_register prologue)
#pragma unroll
                    for (int k = 0; k < 6; k++) { st->pose[k] = pose[k]; st->pose2[(iter + 1) & 1][k] = pose[k]; }
                    st->T_valid = 0;
                    store_trace(trace + iter, tr);
                    st->iters_run = iter + 1;
                    if (conv && !st->converged) st->converged = 1;
                    if (conv && cp->early_exit) st->done = 1;
                }
            }
        }
    }
    if (stamps) stamps[3] = wall_clock64();
    late();
    __syncthreads();
    if (stamps) stamps[4] = wall_clock64();
#pragma unroll
    for (int k = 0; k < 6; k++) pose_out[k] = s_out[k];
    return s_out[6] != 0.0f;
}

// transPointAssociateToMap (:1069-1072) and the LM trig (:1170-1175) from a pose: lanes 0..2 of the calling wave take one
// angle each (glibc's sinf / cosf arithmetic, see glibc_sincosf: what the host's libm would return); result in every lane.
__device__ __forceinline__ void build_transform(const float (&pose)[6], int lane, float (&T)[12], float (&sc6)[6])
{
    const float ang = (lane == 0) ? pose[2] : ((lane == 1) ? pose[1] : pose[0]);   // lane 0: yaw, 1: pitch, 2+: roll
    float snf, csf;
    glibc_sincosf_both(ang, snf, csf);
    const float B = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(snf), 0)), A = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(csf), 0));
    const float D = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(snf), 1)), C = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(csf), 1));
    const float F = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(snf), 2)), E = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(csf), 2));
    const float DE = D * E, DF = D * F;
    T[0] = A * C; T[1] = A * DF - B * E; T[2]  = B * F + A * DE; T[3]  = pose[3];
    T[4] = B * C; T[5] = A * E + B * DF; T[6]  = B * DE - A * F; T[7]  = pose[4];
    T[8] = -D;    T[9] = C * F;          T[10] = C * E;          T[11] = pose[5];
    sc6[0] = B; sc6[1] = A; sc6[2] = D; sc6[3] = C; sc6[4] = F; sc6[5] = E;
}

// ------------------------------------------------------------------------------------------
// Density-aware re-split of the wave table, once per scan before launch 0.  k_chunk_parts can only
// look at the extent of a chunk in the lidar frame; how many map points fall into a wave's box is
// known once the initial guess is.  All workgroups start together, so a launch lasts as long as its
// slowest wave, and the slowest waves are the few whose box covers a dense part of the map (they
// stage and sweep a large tile, or overflow it and fall back to the gather path).  One wave per
// table entry transforms its points with the initial guess and sums the cell ranges of its box rows
// (what launch 0 will stream); only the heavy tail asks for its chunk to be cut 2, 4 or 8 times
// finer (the chunk is sorted along its longest axis, so the parts are compact), and
// k_chunk_table_density rebuilds the table.  Partition only: results do not depend on it beyond the
// summation order of the normal equations.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_wave_density(const SlotTable tbl, int raw_limit)
{
    const CtxP cp = ctx_const(tbl.ctx[blockIdx.y]);
    const auto st = G((const DevState*)tbl.st[blockIdx.y]);
    const int lane = threadIdx.x & 63;
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (!cp->density_pending || wv >= st->n_waves) return;   // the table of a scan is re-split once, at its first optimisation
    const auto tb = G((const int2*)cp->wave_table);
    const int2 e = make_int2(tb[wv].x, tb[wv].y);
    const int i = e.x + lane;
    const bool valid = lane < e.y && i < cp->n_q;
    float sx = 0.0f, sy = 0.0f, sz = 0.0f;
    if (valid) {
        const float px = G(cp->qx)[i], py = G(cp->qy)[i], pz = G(cp->qz)[i];
        sx = ((st->T[0] * px + st->T[1] * py) + st->T[2]  * pz) + st->T[3];
        sy = ((st->T[4] * px + st->T[5] * py) + st->T[6]  * pz) + st->T[7];
        sz = ((st->T[8] * px + st->T[9] * py) + st->T[10] * pz) + st->T[11];
    }
    const bool fin = valid && (fabsf(sx) < 3.0e38f) && (fabsf(sy) < 3.0e38f) && (fabsf(sz) < 3.0e38f);
    const float mnx = wave_min_f32(fin ? sx : INFINITY), mxx = wave_max_f32(fin ? sx : -INFINITY);
    const float mny = wave_min_f32(fin ? sy : INFINITY), mxy = wave_max_f32(fin ? sy : -INFINITY);
    const float mnz = wave_min_f32(fin ? sz : INFINITY), mxz = wave_max_f32(fin ? sz : -INFINITY);
    if (!(mnx <= mxx)) return;
    const GridDesc g = grid_of(cp);
    const auto cell_start = G(cp->cell_start);
    const int bx0 = max(cell_coord(mnx, g.ox, g.inv_e, g.nx) - 1, 0), bx1 = min(cell_coord(mxx, g.ox, g.inv_e, g.nx) + 1, g.nx - 1);
    const int by0 = max(cell_coord(mny, g.oy, g.inv_e, g.ny) - 1, 0), by1 = min(cell_coord(mxy, g.oy, g.inv_e, g.ny) + 1, g.ny - 1);
    const int bz0 = max(cell_coord(mnz, g.oz, g.inv_e, g.nz) - 1, 0), bz1 = min(cell_coord(mxz, g.oz, g.inv_e, g.nz) + 1, g.nz - 1);
    const int nyb = by1 - by0 + 1, nzb = bz1 - bz0 + 1;
    if (nyb * nzb > kRowMax) return;                      // scattered points: the gather path's business, a finer cut does not help
    int raw = 0;
    for (int r = lane; r < nyb * nzb; r += 64) {
        const int zq = r / nyb;
        const int gcell = ((bz0 + zq) * g.ny + by0 + (r - zq * nyb)) * g.nx;
        raw += cell_start[gcell + bx1 + 1] - cell_start[gcell + bx0];
    }
    raw = __builtin_amdgcn_readlane(wave_incl_scan_i32(raw), 63);
    int f = raw <= raw_limit ? 1 : (raw <= 2 * raw_limit ? 2 : (raw <= 4 * raw_limit ? 4 : 8));
    f = min(f, max(e.y / 8, 1));                          // at least 8 points per wave
    if (lane == 0 && f > 1) atomicMax(&cp->chunk_factor[e.x >> 6], f);
}

#include "s2m_register.hpp"

// ------------------------------------------------------------------------------------------
// k_finalize: one workgroup closing iteration `iter` on its own (lm_close_iteration above):
// iteration 0 with the degeneracy analysis, and the last iteration of a scan (or every iteration
// when the grid is too large for the fused form).  mode 1 = normal equations only (observation hook).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kFinThreads) void k_finalize(const SlotTable tbl, int iter, int mode)
{
    const CtxP cp = ctx_const(tbl.ctx[blockIdx.y]);
    const auto st = G(tbl.st[blockIdx.y]);
    // loop state, fetched up front (the state block is a kernel argument: no pointer chase through DevCtx)
    const int done0 = st->done, degen0 = st->isDegenerate;
    const int nb_act = (st->n_waves + cp->wpb - 1) / cp->wpb;
    float pose0[6];
#pragma unroll
    for (int k = 0; k < 6; k++) pose0[k] = st->pose2[iter & 1][k];
    if (mode == 0 && done0) return;
    __shared__ LmShared sh;
    __shared__ float s_out[8];
    // the worklists of the search kernel start empty behind every close of its own (the certify kernel of the next launch appends)
    if (mode == 0 && threadIdx.x == 0 && cp->wl_count) { cp->wl_count[0] = 0; cp->wl_count[1] = 0; }
    float pose[6];
    const bool ended = lm_close_iteration<kFinThreads, true>(cp, st, nb_act, iter, true, mode == 1, pose0, degen0, sh, s_out, pose);
    // the transform of the new pose, once, for every workgroup of the next registration launch (which would otherwise
    // rebuild it - a microsecond of dependent trig arithmetic in each of them)
    if (mode == 0 && !ended && threadIdx.x < 64) {
        float T[12], sc6[6];
        build_transform(pose, (int)threadIdx.x, T, sc6);
        if (threadIdx.x == 0 && !st->stalled) {
#pragma unroll
            for (int k = 0; k < 12; k++) st->T[k] = T[k];
#pragma unroll
            for (int k = 0; k < 6; k++) st->sc[k] = sc6[k];
            st->T_valid = 1;
        }
    }
}

// Observation hook: the device's sinf / cosf of n arguments (tests compare them with the host's libm).
__global__ __launch_bounds__(256) void k_debug_sincos(const float* __restrict__ x, int n, float* __restrict__ s, float* __restrict__ c,
                                                      float* __restrict__ a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { s[i] = glibc_sincosf(x[i], 0); c[i] = glibc_sincosf(x[i], 1); if (a) a[i] = glibc_atanf(x[i]); }
}

// Parameter blocks travel as kernel arguments (copied at launch), so the host never has to
// keep a staging buffer alive or synchronise to update them.
__global__ void k_set_ctx(DevCtx* dst, DevCtx v) { if (threadIdx.x == 0 && blockIdx.x == 0) *dst = v; }
// The wave count of the resident scan (left by k_chunk_table) is folded into the state block here.
// the DevCtx block and the loop state of a scan in one launch (s2m_optimize: both change with every scan)
__global__ void k_set_ctx_state(DevCtx* cdst, DevCtx c, DevState* dst, DevState v, const int32_t* n_waves)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { *cdst = c; v.n_waves = n_waves ? *n_waves : 0; *dst = v; }
}

// the same for the scan slots of a batch: one launch for all of them (blockIdx.x = slot)
__global__ void k_set_ctxs(const CtxTable t) { if (threadIdx.x == 0 && t.dst[blockIdx.x]) *t.dst[blockIdx.x] = t.v[blockIdx.x]; }
__global__ __launch_bounds__(256) void k_init_states(const StateInitTable t)
{
    const StateInit& si = t.s[blockIdx.x];
    if (!si.dst) return;
    float* w = reinterpret_cast<float*>(si.dst);
    for (int k = threadIdx.x; k < (int)(sizeof(DevState) / 4); k += blockDim.x) w[k] = 0.0f;
    __syncthreads();
    if (threadIdx.x == 0) {
        DevState* d = si.dst;
        for (int k = 0; k < 6; k++) { d->pose[k] = si.pose[k]; d->pose2[0][k] = si.pose[k]; d->sc[k] = si.sc[k]; }
        for (int k = 0; k < 12; k++) d->T[k] = si.T[k];
        for (int k = 0; k < 36; k++) d->matP[k] = si.matP[k];
        d->isDegenerate = si.isDegenerate;
        d->T_valid = 1;
        d->n_waves = si.n_waves ? *si.n_waves : 0;
    }
}

__global__ void k_set_state(DevState* dst, DevState v, const int32_t* n_waves)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { v.n_waves = n_waves ? *n_waves : 0; *dst = v; }
}

// ------------------------------------------------------------------------------------------
// ScanContext descriptor (include/Scancontext.cpp:151-211): max-z polar histogram, 20 rings x
// 60 sectors, LDS bins per workgroup merged with global atomicMax on an order-preserving
// integer image of fp32 z; ring key = row mean.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sc_polar_max(const unsigned char* __restrict__ pts, size_t stride, int n,
                                                      uint32_t* __restrict__ bins /*[1200], 0 = empty*/)
{
    __shared__ uint32_t lb[S2M_SC_NUM_RING * S2M_SC_NUM_SECTOR];
    for (int k = threadIdx.x; k < S2M_SC_NUM_RING * S2M_SC_NUM_SECTOR; k += blockDim.x) lb[k] = 0u;
    __syncthreads();
    const double kdeg = 180.0 / M_PI;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float* p = reinterpret_cast<const float*>(pts + (size_t)i * stride);
        const float x = p[0], y = p[1];
        const float z = (float)((double)p[2] + 2.0);                       // LIDAR_HEIGHT (:168)
        const float rng = sqrtf(x * x + y * y);                            // :171
        float th;                                                          // xy2theta (:23-36)
        if ((x >= 0) & (y >= 0))      th = (float)(kdeg * (double)glibc_atanf(y / x));
        else if ((x < 0) & (y >= 0))  th = (float)(180.0 - (kdeg * (double)glibc_atanf(y / (-x))));
        else if ((x < 0) & (y < 0))   th = (float)(180.0 + (kdeg * (double)glibc_atanf(y / x)));
        else if ((x >= 0) & (y < 0))  th = (float)(360.0 - (kdeg * (double)glibc_atanf((-y) / x)));
        else th = NAN;
        if ((double)rng > 80.0) continue;                                  // PC_MAX_RADIUS (:175)
        if (!(z == z)) continue;                                           // NaN z never wins desc < z
        const double rv = ceil(((double)rng / 80.0) * 20.0);               // :178
        const double sv = ceil(((double)th / 360.0) * 60.0);               // :179
        int ring = (rv != rv) ? 1 : (int)fmin(fmax(rv, 1.0), 20.0);
        int sect = (sv != sv) ? 1 : (int)fmin(fmax(sv, 1.0), 60.0);
        atomicMax(&lb[(ring - 1) * S2M_SC_NUM_SECTOR + (sect - 1)], f2ord(z));
    }
    __syncthreads();
    for (int k = threadIdx.x; k < S2M_SC_NUM_RING * S2M_SC_NUM_SECTOR; k += blockDim.x)
        if (lb[k]) atomicMax(&bins[k], lb[k]);
}

__global__ __launch_bounds__(64) void k_sc_finish(const uint32_t* __restrict__ bins, double* __restrict__ desc,
                                                  double* __restrict__ ringkey)
{
    // one wave; lane r < 20 owns ring r
    const int r = threadIdx.x;
    if (r >= S2M_SC_NUM_RING) return;
    double s = 0.0;
    for (int k = 0; k < S2M_SC_NUM_SECTOR; k++) {
        const uint32_t u = bins[r * S2M_SC_NUM_SECTOR + k];
        double v = 0.0;                                                    // NO_POINT -> 0 (:187-190)
        if (u) {
            const uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
            const float z = __uint_as_float(b);
            // desc starts at -1000 and takes z only if -1000 < z (:182); -1000 -> 0 afterwards
            v = ((double)z > -1000.0) ? (double)z : 0.0;
        }
        desc[r * S2M_SC_NUM_SECTOR + k] = v;
        s += v;
    }
    ringkey[r] = s / 60.0;                                                 // row mean (:198-211)
}

// ------------------------------------------------------------------------------------------
// ScanContext matching (reference include/Scancontext.cpp:69-148,
// 214-344).  The SCManager's containers live on the device as three arrays indexed by key-frame
// number: descriptors [n][20*60] fp64 (ring-major), ring keys [n][20] fp32 (eig2stdvec, :62-66),
// sector keys [n][60] fp64.  Sizes are tiny (a descriptor is 9.6 KB), so one workgroup does a whole
// detectLoopClosureID(); the batched kernel runs one workgroup per candidate.  All sums run left to
// right in fp64 like the oracle's restatement, so distances are bit-identical to it.
// ------------------------------------------------------------------------------------------
constexpr int kScThreads = 256;
constexpr int kScShifts = 7;           // 1 + 2 * round(0.5 * SEARCH_RATIO * 60) (:122-129, Scancontext.h:93)

struct ScDetectOut {                   // mirrors s2m_sc_match + loop id / yaw
    double  min_dist;
    int32_t loop_id, nn_idx, nn_align;
    float   yaw_diff_rad;
    int32_t cand_idx[3];
    float   cand_d2[3];
};

// makeAndSaveScancontextAndKeys (:236-250): descriptor + ring key (as k_sc_finish left them) into slot `idx`
__global__ __launch_bounds__(kScThreads) void k_sc_append(const double* __restrict__ desc_ring, double* __restrict__ store_desc,
                                                          float* __restrict__ store_ring, double* __restrict__ store_sector, int idx)
{
    constexpr int NR = S2M_SC_NUM_RING, NS = S2M_SC_NUM_SECTOR;
    double* d = store_desc + (size_t)idx * NR * NS;
    for (int k = threadIdx.x; k < NR * NS; k += kScThreads) d[k] = desc_ring[k];
    if (threadIdx.x < NR) store_ring[(size_t)idx * NR + threadIdx.x] = (float)desc_ring[NR * NS + threadIdx.x];
    if (threadIdx.x < NS) {                                                // makeSectorkeyFromScancontext (:214-227)
        double a = 0.0;
        for (int r = 0; r < NR; r++) a += desc_ring[r * NS + threadIdx.x];
        store_sector[(size_t)idx * NS + threadIdx.x] = a / (double)NR;
    }
}

struct ScShared {
    double v1[S2M_SC_NUM_SECTOR], v2[S2M_SC_NUM_SECTOR]; // the two sector keys, staged
    double vnorm[S2M_SC_NUM_SECTOR];                     // fastAlignUsingVkey: ||vkey1 - circshift(vkey2, s)||
    double sim[kScShifts][S2M_SC_NUM_SECTOR];            // per (shift, sector): cosine similarity
    int    eff[kScShifts][S2M_SC_NUM_SECTOR];            // ... and whether the sector pair counts
    double dist[kScShifts];
    int    shifts[kScShifts];
    double dist_out;                                     // the result: distance and the shift it was found at
    int    shift_out;
};

// distanceBtnScanContext(query, cand[c]) (:116-148) for NC candidates side by side by one workgroup: every phase of the
// reference's function is spread over (candidate, shift) or (candidate, shift, sector) tasks, every sum is taken by one thread
// from left to right like the oracle's.  `cand` lives in LDS; results in sh[c].dist_out / .shift_out, valid after the call.
template <int NC>
__device__ __forceinline__ void sc_distance_block(const double* __restrict__ store_desc, const double* __restrict__ store_sector,
                                                  int q, const int* cand, ScShared* sh)
{
    constexpr int NR = S2M_SC_NUM_RING, NS = S2M_SC_NUM_SECTOR;
    const int t = threadIdx.x;
    const double* __restrict__ sc1 = store_desc + (size_t)q * NR * NS;
    __syncthreads();                                     // the scratch may still be read from a previous call
    for (int task = t; task < NC * NS; task += kScThreads) {
        const int c = task / NS, j = task - c * NS;
        sh[c].v1[j] = store_sector[(size_t)q * NS + j];
        sh[c].v2[j] = store_sector[(size_t)cand[c] * NS + j];
    }
    __syncthreads();
    for (int task = t; task < NC * NS; task += kScThreads) {              // fastAlignUsingVkey (:94-113), one shift per task
        const int c = task / NS, s = task - c * NS;
        double a = 0.0;
        for (int j = 0; j < NS; j++) {
            int j2 = j - s; j2 += (j2 < 0) ? NS : 0;
            const double d = sh[c].v1[j] - sh[c].v2[j2];
            a += d * d;
        }
        sh[c].vnorm[s] = sqrt(a);
    }
    __syncthreads();
    if (t < NC) {
        ScShared& m = sh[t];
        int a0 = 0;
        double best = 10000000;
        for (int s = 0; s < NS; s++) if (m.vnorm[s] < best) { a0 = s; best = m.vnorm[s]; }
        // search space a0, a0 +- 1..3 (mod 60), ascending (:122-129)
        int sp[kScShifts];
        sp[0] = a0;
        for (int ii = 1; ii <= (kScShifts - 1) / 2; ii++) { sp[2 * ii - 1] = (a0 + ii + NS) % NS; sp[2 * ii] = (a0 - ii + NS) % NS; }
        for (int a = 1; a < kScShifts; a++) {            // insertion sort
            const int v = sp[a];
            int b = a - 1;
            while (b >= 0 && sp[b] > v) { sp[b + 1] = sp[b]; b--; }
            sp[b + 1] = v;
        }
        for (int k = 0; k < kScShifts; k++) m.shifts[k] = sp[k];
    }
    __syncthreads();
    for (int task = t; task < NC * kScShifts * NS; task += kScThreads) {  // distDirectSC (:69-91), one sector pair per task
        const int c = task / (kScShifts * NS), rem = task - c * (kScShifts * NS);
        const int k = rem / NS, j = rem - k * NS;
        int j2 = j - sh[c].shifts[k]; j2 += (j2 < 0) ? NS : 0;            // circshift (:39-59)
        const double* __restrict__ sc2 = store_desc + (size_t)cand[c] * NR * NS;
        double n1 = 0.0, n2 = 0.0, dot = 0.0;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const double a = sc1[r * NS + j], b = sc2[r * NS + j2];
            n1 += a * a; n2 += b * b; dot += a * b;
        }
        n1 = sqrt(n1); n2 = sqrt(n2);
        const bool skip = (n1 == 0) | (n2 == 0);
        sh[c].eff[k][j] = skip ? 0 : 1;
        sh[c].sim[k][j] = skip ? 0.0 : dot / (n1 * n2);
    }
    __syncthreads();
    if (t < NC * kScShifts) {
        const int c = t / kScShifts, k = t - c * kScShifts;
        int num_eff = 0;
        double sum = 0;
        for (int j = 0; j < NS; j++) if (sh[c].eff[k][j]) { sum = sum + sh[c].sim[k][j]; num_eff = num_eff + 1; }
        sh[c].dist[k] = 1.0 - sum / (double)num_eff;      // 0/0 = NaN when no sector counts, as in the reference
    }
    __syncthreads();
    if (t < NC) {
        ScShared& m = sh[t];
        int argmin_shift = 0;
        double min_sc_dist = 10000000;
        for (int k = 0; k < kScShifts; k++) if (m.dist[k] < min_sc_dist) { argmin_shift = m.shifts[k]; min_sc_dist = m.dist[k]; }
        m.dist_out = min_sc_dist; m.shift_out = argmin_shift;
    }
    __syncthreads();
}

// distanceBtnScanContext of descriptor `query` against cand[0..m): one workgroup per candidate
__global__ __launch_bounds__(kScThreads) void k_sc_distance_batch(const double* __restrict__ store_desc, const double* __restrict__ store_sector,
                                                                  int query, const int32_t* __restrict__ cand, double* __restrict__ dist,
                                                                  int32_t* __restrict__ shift)
{
    __shared__ ScShared sh[1];
    __shared__ int s_cand[1];
    if (threadIdx.x == 0) s_cand[0] = cand[blockIdx.x];
    __syncthreads();
    sc_distance_block<1>(store_desc, store_sector, query, s_cand, sh);
    if (threadIdx.x == 0) { dist[blockIdx.x] = sh[0].dist_out; shift[blockIdx.x] = sh[0].shift_out; }
}

// detectLoopClosureID (:253-344) for the newest descriptor (index n_total - 1) against the ring keys
// [0, n_search) (the contents of the reference's kd-tree at its last rebuild).  One workgroup: every thread keeps the three
// nearest of its share of the keys, the lists are merged by a butterfly inside each wave and by one thread over the four waves
// (order: distance, then index - the lower index wins a tie, whoever held it), then the three candidates are compared side by
// side.  `out` may be pinned host memory (the result is 48 bytes: written where the host reads it, no copy behind the kernel).
__device__ __forceinline__ bool sc_near_less(float da, int ia, float db, int ib) { return da < db || (da == db && ia < ib); }
__device__ __forceinline__ void sc_near_insert(float (&bd)[3], int (&bi)[3], float d, int i)
{
    if (sc_near_less(d, i, bd[2], bi[2])) {
        bd[2] = d; bi[2] = i;
        if (sc_near_less(bd[2], bi[2], bd[1], bi[1])) { const float a = bd[1]; bd[1] = bd[2]; bd[2] = a; const int b2 = bi[1]; bi[1] = bi[2]; bi[2] = b2; }
        if (sc_near_less(bd[1], bi[1], bd[0], bi[0])) { const float a = bd[0]; bd[0] = bd[1]; bd[1] = a; const int b2 = bi[0]; bi[0] = bi[1]; bi[1] = b2; }
    }
}

__global__ __launch_bounds__(kScThreads) void k_sc_detect(const double* __restrict__ store_desc, const float* __restrict__ store_ring,
                                                          const double* __restrict__ store_sector, int n_total, int n_search,
                                                          ScDetectOut* __restrict__ out)
{
    constexpr int NR = S2M_SC_NUM_RING;
    static_assert(NR % 4 == 0, "ring keys are read four floats at a time");
    __shared__ ScShared sh[3];
    __shared__ float s_d2[kScThreads / 64][3];
    __shared__ int   s_ix[kScThreads / 64][3];
    __shared__ int   s_cand[3];
    const int t = threadIdx.x, q = n_total - 1;
    // exact 3-NN over the fp32 ring keys, accumulation order of nanoflann's L2_Adaptor (groups of four);
    // ties go to the lower index
    float bd[3] = { INFINITY, INFINITY, INFINITY };
    int bi[3] = { 0x7fffffff, 0x7fffffff, 0x7fffffff };
    float4 qk[NR / 4];
#pragma unroll
    for (int g = 0; g < NR / 4; g++) qk[g] = reinterpret_cast<const float4*>(store_ring + (size_t)q * NR)[g];
    for (int i = t; i < n_search; i += kScThreads) {
        const float4* b = reinterpret_cast<const float4*>(store_ring + (size_t)i * NR);      // (rows are 80 bytes: 16-byte aligned)
        float4 bk[NR / 4];
#pragma unroll
        for (int g = 0; g < NR / 4; g++) bk[g] = b[g];
        float result = 0.0f;
#pragma unroll
        for (int g = 0; g < NR / 4; g++) {
            const float d0 = qk[g].x - bk[g].x, d1 = qk[g].y - bk[g].y, d2 = qk[g].z - bk[g].z, d3 = qk[g].w - bk[g].w;
            result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        }
        if (result < bd[2]) {                             // i ascends per thread: an equal distance stays behind
            bd[2] = result; bi[2] = i;
            if (bd[2] < bd[1]) { const float a = bd[1]; bd[1] = bd[2]; bd[2] = a; const int b2 = bi[1]; bi[1] = bi[2]; bi[2] = b2; }
            if (bd[1] < bd[0]) { const float a = bd[0]; bd[0] = bd[1]; bd[1] = a; const int b2 = bi[0]; bi[0] = bi[1]; bi[1] = b2; }
        }
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {              // both partners end with the same three
        float pd[3]; int pi[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { pd[k] = __shfl_xor(bd[k], off, 64); pi[k] = __shfl_xor(bi[k], off, 64); }
#pragma unroll
        for (int k = 0; k < 3; k++) sc_near_insert(bd, bi, pd[k], pi[k]);
    }
    if ((t & 63) == 0)
        for (int k = 0; k < 3; k++) { s_d2[t >> 6][k] = bd[k]; s_ix[t >> 6][k] = bi[k]; }
    __syncthreads();
    if (t == 0) {
        float fd[3] = { INFINITY, INFINITY, INFINITY };
        int fi[3] = { 0x7fffffff, 0x7fffffff, 0x7fffffff };
        for (int u = 0; u < kScThreads / 64; u++)
            for (int k = 0; k < 3; k++) sc_near_insert(fd, fi, s_d2[u][k], s_ix[u][k]);
        for (int k = 0; k < 3; k++) {
            const bool have = fi[k] != 0x7fffffff;        // fewer keys than candidates: index 0 (:289 zero-initialised)
            s_cand[k] = have ? fi[k] : 0;
            out->cand_idx[k] = s_cand[k];
            out->cand_d2[k] = have ? fd[k] : 0.0f;
        }
    }
    __syncthreads();
    sc_distance_block<3>(store_desc, store_sector, q, s_cand, sh);
    if (t == 0) {
        double min_dist = 10000000;
        int nn_align = 0, nn_idx = 0;
        for (int c = 0; c < 3; c++)                        // :302-316
            if (sh[c].dist_out < min_dist) { min_dist = sh[c].dist_out; nn_align = sh[c].shift_out; nn_idx = s_cand[c]; }
        out->min_dist = min_dist; out->nn_idx = nn_idx; out->nn_align = nn_align;
        out->loop_id = (min_dist < 0.3) ? nn_idx : -1;    // SC_DIST_THRES (Scancontext.h:95)
        out->yaw_diff_rad = (float)((double)(float)((double)nn_align * (360.0 / 60.0)) * M_PI / 180.0);   // deg2rad(float) (:17-20, :338)
    }
}

}  // namespace s2m
