// s2m_voxel.hip — pcl::VoxelGrid<PointXYZI> and transformPointCloud on gfx950 (see s2m_voxel.hpp).
//
// The filter follows PCL's own passes (voxel_grid.hpp, PCL 1.10): bounding box of the finite
// points -> integer voxel index per point -> sort by index -> one centroid per run. A dense
// per-voxel table is not an option (a 200 m x 200 m x 30 m sweep at 0.4 m is 19 M voxels for
// ~1e5 points), so the runs come from a STABLE least-significant-digit radix sort of (index, point)
// pairs, three passes of 11 bits, hand-written (k_rs_*, round 4; rounds 1-3 used hipCUB's): a block
// takes 4 096 keys, every wave a contiguous quarter of them; a key's place among equal digits is
// (blocks before) + (waves before in the block) + (rounds before in the wave) + (lanes before in the
// round, by an 11-ballot match) - a pure function of the input, no atomics between waves. The sort
// being stable fixes the summation order inside a voxel to ascending point index, which is what the
// CPU oracle defines, so centroids are bit-identical to it. Run starts (where the sorted key changes)
// are compacted in two kernels (k_heads_*). Compiled with -ffp-contract=off like the rest.
#include "s2m_voxel.hpp"

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
    int32_t  n_vlong;                                // ... with a very long one, handed to k_vox_centroid_long's workgroup role
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

// getMinMax3D over the points whose three coordinates are all finite, and their count: every workgroup leaves its own box and
// count in `part` (8 words each), k_vox_setup puts them together.  (Adds from every workgroup on one address are served one after
// the other, ~6 ns each on this part: 7 000 of them were most of this kernel.)
__global__ __launch_bounds__(256) void k_vox_bbox(const unsigned char* __restrict__ pts, size_t stride, int n, uint32_t* __restrict__ part)
{
    __shared__ float smn[4][3], smx[4][3];
    __shared__ int scnt[4];
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    int cnt = 0;
    const int step = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += 8 * step) {       // eight records in flight per lane
        float x[8], y[8], z[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            // (unconditional, clamped to the last record: a load under a lane mask of its own is waited for before the next)
            const float* p = reinterpret_cast<const float*>(pts + (size_t)min(i + u * step, n - 1) * stride);
            x[u] = p[0]; y[u] = p[1]; z[u] = p[2];
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (i + u * step < n && isfinite(x[u]) && isfinite(y[u]) && isfinite(z[u])) {
                mn[0] = fminf(mn[0], x[u]); mx[0] = fmaxf(mx[0], x[u]);
                mn[1] = fminf(mn[1], y[u]); mx[1] = fmaxf(mx[1], y[u]);
                mn[2] = fminf(mn[2], z[u]); mx[2] = fmaxf(mx[2], z[u]);
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
    uint32_t* mine = part + 8 * blockIdx.x;
    if (threadIdx.x < 3) {
        const int d = threadIdx.x;
        mine[d] = f2ord(fminf(fminf(smn[0][d], smn[1][d]), fminf(smn[2][d], smn[3][d])));          // (+inf / -inf where the workgroup saw no finite point)
        mine[3 + d] = f2ord(fmaxf(fmaxf(smx[0][d], smx[1][d]), fmaxf(smx[2][d], smx[3][d])));
    } else if (threadIdx.x == 3) {
        mine[6] = (uint32_t)(scnt[0] + scnt[1] + scnt[2] + scnt[3]);
    }
}

__global__ __launch_bounds__(64) void k_vox_setup(VoxSetup* s, float leaf, const uint32_t* __restrict__ part, int nparts)
{
    uint32_t lo[3] = { 0xffffffffu, 0xffffffffu, 0xffffffffu }, hi[3] = { 0u, 0u, 0u };
    int cnt = 0;
    for (int b = threadIdx.x; b < nparts; b += 64) {
        const uint32_t* q = part + 8 * b;
#pragma unroll
        for (int d = 0; d < 3; d++) { lo[d] = min(lo[d], q[d]); hi[d] = max(hi[d], q[3 + d]); }
        cnt += (int)q[6];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int d = 0; d < 3; d++) { lo[d] = min(lo[d], (uint32_t)__shfl_down((int)lo[d], off, 64)); hi[d] = max(hi[d], (uint32_t)__shfl_down((int)hi[d], off, 64)); }
        cnt += __shfl_down(cnt, off, 64);
    }
    if (threadIdx.x != 0) return;
    for (int d = 0; d < 3; d++) { s->mm[d] = lo[d]; s->mm[3 + d] = hi[d]; }
    s->n_valid = cnt; s->n_out = 0; s->leaf_too_small = 0; s->n_long = 0; s->n_vlong = 0;
    const float inv = 1.0f / leaf;
    s->inv_leaf = inv;
    if (cnt == 0) { s->min_b[0] = s->min_b[1] = s->min_b[2] = 0; s->mul1 = s->mul2 = 1; return; }
    float mn[3], mx[3];
    for (int d = 0; d < 3; d++) { mn[d] = ord2f(lo[d]); mx[d] = ord2f(hi[d]); }
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

// the voxel index of a point as the sort's key (non-finite points: kInvalidKey, behind every voxel)
__device__ __forceinline__ uint32_t voxel_key(float x, float y, float z, const VoxSetup* __restrict__ s)
{
    uint32_t key = kInvalidKey;
    if (isfinite(x) && isfinite(y) && isfinite(z)) {
        const float inv = s->inv_leaf;
        const int ijk0 = (int)(floorf(x * inv) - (float)s->min_b[0]);
        const int ijk1 = (int)(floorf(y * inv) - (float)s->min_b[1]);
        const int ijk2 = (int)(floorf(z * inv) - (float)s->min_b[2]);
        key = (uint32_t)(ijk0 + ijk1 * s->mul1 + ijk2 * s->mul2);
        if (key == kInvalidKey) key = kInvalidKey - 1u;          // unreachable unless the leaf is too small
    }
    return key;
}

// ------------------------------------------------------------------------------------------
// stable LSD radix sort of (key, point index) pairs: 11 bits per pass
// ------------------------------------------------------------------------------------------
constexpr int kRsBits = 11, kRsBins = 1 << kRsBits;      // digits per pass
constexpr int kRsTile = 4096;                            // keys per block: 4 waves x 16 rounds x 64 lanes
constexpr int kRsRounds = kRsTile / 256;

// the lanes of the wave that hold the same digit (of the lanes in `live`): one ballot per digit bit
__device__ __forceinline__ unsigned long long rs_peers(uint32_t digit, unsigned long long live)
{
    unsigned long long peers = live;
#pragma unroll
    for (int b = 0; b < kRsBits; b++) {
        const bool bit = ((digit >> b) & 1u) != 0u;
        const unsigned long long m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

// digit histogram of every block's tile: hist[block][digit]
// FROM_POINTS: the first pass makes the keys on the way (voxel index of every point, written to `keys_out`; the value that
// travels with a key is the point's position, supplied by the first scatter).
template <bool FROM_POINTS>
__global__ __launch_bounds__(256) void k_rs_hist(const uint32_t* __restrict__ keys, int n, int shift, int32_t* __restrict__ hist,
                                                 const unsigned char* __restrict__ pts, size_t stride, const VoxSetup* __restrict__ s,
                                                 uint32_t* __restrict__ keys_out)
{
    __shared__ int32_t h[kRsBins];
    for (int b = threadIdx.x; b < kRsBins; b += 256) h[b] = 0;
    __syncthreads();
    const int base = blockIdx.x * kRsTile;
    uint32_t key[kRsRounds];
    if (FROM_POINTS) {
        float x[kRsRounds], y[kRsRounds], z[kRsRounds];
#pragma unroll
        for (int r = 0; r < kRsRounds; r++) {             // (unconditional, clamped: all loads in flight at once)
            const float* p = reinterpret_cast<const float*>(pts + (size_t)min(base + r * 256 + (int)threadIdx.x, n - 1) * stride);
            x[r] = p[0]; y[r] = p[1]; z[r] = p[2];
        }
#pragma unroll
        for (int r = 0; r < kRsRounds; r++) {
            const int i = base + r * 256 + threadIdx.x;
            key[r] = voxel_key(x[r], y[r], z[r], s);
            if (i < n) keys_out[i] = key[r];
        }
    } else {
#pragma unroll
        for (int r = 0; r < kRsRounds; r++) {             // all loads in flight before the first add
            const int i = base + r * 256 + threadIdx.x;
            key[r] = i < n ? keys[i] : 0u;
        }
    }
#pragma unroll
    for (int r = 0; r < kRsRounds; r++) {
        const int i = base + r * 256 + threadIdx.x;
        if (i < n) atomicAdd(&h[(key[r] >> shift) & (kRsBins - 1)], 1);       // (counts: the order of the adds does not matter)
    }
    __syncthreads();
    for (int b = threadIdx.x; b < kRsBins; b += 256) hist[(size_t)blockIdx.x * kRsBins + b] = h[b];
}

// Exclusive prefix of every digit's counts over the blocks, in place, and the digit's total.  A workgroup takes 16 digits (64
// contiguous bytes of every block's row); its 16 groups of 16 lanes share the blocks out in contiguous ranges, add their range
// up, exchange the sums through LDS and walk the range a second time to leave the prefixes (loads independent of one another).
constexpr int kScanDigits = 16;
__global__ __launch_bounds__(256) void k_rs_scan_bins(int32_t* __restrict__ hist, int nblk, int32_t* __restrict__ bin_tot)
{
    __shared__ int32_t part[16][kScanDigits];
    const int g = threadIdx.x >> 4, b = threadIdx.x & 15;
    const int bin = blockIdx.x * kScanDigits + b;
    const int per = (nblk + 15) / 16;
    const int r0 = g * per, r1 = min(r0 + per, nblk);
    int32_t* col = hist + bin;
    int32_t sum = 0;
    int r = r0;
    for (; r + 4 <= r1; r += 4) {
        const int32_t v0 = col[(size_t)r * kRsBins], v1 = col[(size_t)(r + 1) * kRsBins], v2 = col[(size_t)(r + 2) * kRsBins], v3 = col[(size_t)(r + 3) * kRsBins];
        sum += v0 + v1 + v2 + v3;
    }
    for (; r < r1; r++) sum += col[(size_t)r * kRsBins];
    part[g][b] = sum;
    __syncthreads();
    int32_t run = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { const int32_t p = part[k][b]; run += k < g ? p : 0; total += p; }
    if (g == 0) bin_tot[bin] = total;
    r = r0;
    for (; r + 4 <= r1; r += 4) {
        const int32_t v0 = col[(size_t)r * kRsBins], v1 = col[(size_t)(r + 1) * kRsBins], v2 = col[(size_t)(r + 2) * kRsBins], v3 = col[(size_t)(r + 3) * kRsBins];
        col[(size_t)r * kRsBins] = run; run += v0;
        col[(size_t)(r + 1) * kRsBins] = run; run += v1;
        col[(size_t)(r + 2) * kRsBins] = run; run += v2;
        col[(size_t)(r + 3) * kRsBins] = run; run += v3;
    }
    for (; r < r1; r++) { const int32_t v = col[(size_t)r * kRsBins]; col[(size_t)r * kRsBins] = run; run += v; }
}

// Scatter.  Wave w of a block owns keys [base + 1024 w, base + 1024 (w + 1)), 64 consecutive ones per round, so the order of the
// input is (block, wave, round, lane) and a key's place among the keys of its digit is the sum of four counts taken in that
// order: the blocks before (hist, scanned), the waves before in this block (wcnt, scanned below), the rounds before in this
// wave (the wave's running counter) and the lanes before in this round (the match).  vals_in == nullptr: the value is the
// key's own position (first pass).
__global__ __launch_bounds__(256) void k_rs_scatter(const uint32_t* __restrict__ keys_in, const int32_t* __restrict__ vals_in, int n, int shift,
                                                    int nblk, const int32_t* __restrict__ hist, const int32_t* __restrict__ bin_tot,
                                                    uint32_t* __restrict__ keys_out, int32_t* __restrict__ vals_out)
{
    __shared__ int32_t wcnt[4][kRsBins];                  // per wave: counts, then running local positions
    __shared__ int32_t bfirst[kRsBins];                   // first output position of every digit (exclusive scan of the bin totals, by every block for itself)
    __shared__ int32_t wsum4[4];
    __shared__ uint32_t lkeys[kRsTile];                   // the block's keys (and values) in digit order, before they are written out in runs
    __shared__ int32_t lvals[kRsTile];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // everything this workgroup reads from memory is asked for at once: the bin totals, its own column of the scanned histogram,
    // its keys and values (one round trip instead of three)
    static_assert(kRsBins == 8 * 256, "eight bins per thread");
    int32_t t8[8], h8[8];
#pragma unroll
    for (int k = 0; k < 8; k++) t8[k] = bin_tot[8 * tid + k];
#pragma unroll
    for (int k = 0; k < 8; k++) h8[k] = hist[(size_t)blockIdx.x * kRsBins + 8 * tid + k];
    const int base = blockIdx.x * kRsTile + wave * (kRsTile / 4);
    uint32_t key[kRsRounds];
    int32_t val[kRsRounds];
#pragma unroll
    for (int r = 0; r < kRsRounds; r++) {
        const int i = base + r * 64 + lane;
        key[r] = i < n ? keys_in[i] : 0xffffffffu;
        val[r] = i < n ? (vals_in ? vals_in[i] : i) : 0;
    }
    for (int b = tid; b < 4 * kRsBins; b += 256) (&wcnt[0][0])[b] = 0;
    {
        int32_t sum = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) sum += t8[k];
        int32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int32_t o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
        if (lane == 63) wsum4[wave] = incl;
        __syncthreads();
        int32_t run = incl - sum;
        for (int w = 0; w < wave; w++) run += wsum4[w];
#pragma unroll
        for (int k = 0; k < 8; k++) { bfirst[8 * tid + k] = run; run += t8[k]; }
    }
    // phase A: this wave's digit counts (its own row of wcnt: no other wave touches it).  Per round and lane the match is kept
    // for phase B: lanes of the same digit before this one (6 bits), the group's size (7 bits), its first lane (6 bits).
    uint32_t match[kRsRounds];
#pragma unroll
    for (int r = 0; r < kRsRounds; r++) {
        const int i = base + r * 64 + lane;
        const bool live = i < n;
        const uint32_t d = (key[r] >> shift) & (kRsBins - 1);
        const unsigned long long peers = rs_peers(d, __ballot(live));
        const uint32_t before = __popcll(peers & ((1ull << lane) - 1ull)), size = __popcll(peers);
        match[r] = before | (size << 6) | ((uint32_t)(__ffsll((long long)peers) - 1) << 13);
        if (live && before == 0u) atomicAdd(&wcnt[wave][d], (int32_t)size);      // the first lane of every digit group (LDS adds: in order, nothing to wait for)
    }
    __syncthreads();
    // counts -> this block's keys in digit order (local positions), and where each digit's run goes in the output:
    //   lfirst[b]  = keys of this block with a smaller digit          (block scan of the block's digit counts)
    //   wcnt[w][b] = lfirst[b] + keys of digit b in the waves before w (first local position of (wave, digit))
    //   bfirst[b] <- (first output position of digit b for this block) - lfirst[b]: output position = that + local position
    {
        int32_t c8[8], sum = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int b = 8 * tid + k;
            c8[k] = wcnt[0][b] + wcnt[1][b] + wcnt[2][b] + wcnt[3][b];
            sum += c8[k];
        }
        int32_t incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int32_t o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
        __syncthreads();                                  // (wsum4 is read above by every wave before it is written again)
        if (lane == 63) wsum4[wave] = incl;
        __syncthreads();
        int32_t lfirst = incl - sum;
        for (int w = 0; w < wave; w++) lfirst += wsum4[w];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int b = 8 * tid + k;
            int32_t run = lfirst;
#pragma unroll
            for (int w = 0; w < 4; w++) { const int32_t c = wcnt[w][b]; wcnt[w][b] = run; run += c; }
            bfirst[b] = bfirst[b] + h8[k] - lfirst;
            lfirst += c8[k];
        }
    }
    __syncthreads();
    // phase B: the same rounds in the same order, every key to its LOCAL place (LDS): the block's keys in digit order.  The
    // first lane of a digit group moves the wave's counter on by the group's size and hands what it found to the group (a
    // returning LDS add: the rounds' adds on one counter are served in the order they were issued).
#pragma unroll
    for (int r = 0; r < kRsRounds; r++) {
        const int i = base + r * 64 + lane;
        const bool live = i < n;
        const uint32_t d = (key[r] >> shift) & (kRsBins - 1);
        const uint32_t before = match[r] & 63u, size = (match[r] >> 6) & 127u, lead = match[r] >> 13;
        int32_t first = 0;
        if (live && before == 0u) first = atomicAdd(&wcnt[wave][d], (int32_t)size);
        first = __shfl(first, (int)lead, 64);
        if (live) {
            const int32_t pos = first + (int32_t)before;
            lkeys[pos] = key[r];
            lvals[pos] = val[r];
        }
    }
    __syncthreads();
    // write-out: consecutive threads take consecutive local positions - runs of equal digits go to consecutive output positions
    const int nk = min(kRsTile, n - blockIdx.x * kRsTile);
    for (int t = tid; t < nk; t += 256) {
        const uint32_t k = lkeys[t];
        const int32_t pos = bfirst[(k >> shift) & (kRsBins - 1)] + t;
        keys_out[pos] = k;
        vals_out[pos] = lvals[t];
    }
}

// ------------------------------------------------------------------------------------------
// run starts
// ------------------------------------------------------------------------------------------
// A run starts where the sorted key changes (non-finite points carry kInvalidKey and sort behind every voxel).
__device__ __forceinline__ bool run_starts(uint32_t k, uint32_t k_before, int i) { return k != kInvalidKey && (i == 0 || k_before != k); }

// number of run starts in every block's tile of the sorted keys (wave w of a block: keys [base + 1024 w, base + 1024 (w + 1)))
__global__ __launch_bounds__(256) void k_heads_count(const uint32_t* __restrict__ keys, int n, int32_t* __restrict__ blk_cnt)
{
    __shared__ int32_t wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = blockIdx.x * kRsTile + wave * (kRsTile / 4);
    uint32_t k[kRsRounds], kb[kRsRounds];
#pragma unroll
    for (int r = 0; r < kRsRounds; r++) {             // (unconditional loads, clamped: all in flight at once)
        const int i = base + r * 64 + lane;
        k[r] = keys[min(i, n - 1)];
        kb[r] = keys[max(min(i, n - 1) - 1, 0)];
    }
    int32_t c = 0;
#pragma unroll
    for (int r = 0; r < kRsRounds; r++) { const int i = base + r * 64 + lane; c += (i < n && run_starts(k[r], kb[r], i)) ? 1 : 0; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if (lane == 0) wsum[wave] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// The positions of the run starts, in order.  Every block adds up the counts of the blocks before it for itself (a few hundred
// numbers), every wave those of the waves before it; the last block leaves the total.
__global__ __launch_bounds__(256) void k_heads_write(const uint32_t* __restrict__ keys, int n, const int32_t* __restrict__ blk_cnt, int nblk,
                                                     int32_t* __restrict__ heads, int32_t* __restrict__ n_out)
{
    __shared__ int32_t wsum[4], bsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = blockIdx.x * kRsTile + wave * (kRsTile / 4);
    uint32_t k[kRsRounds], kb[kRsRounds];
#pragma unroll
    for (int r = 0; r < kRsRounds; r++) {
        const int i = base + r * 64 + lane;
        k[r] = keys[min(i, n - 1)];
        kb[r] = keys[max(min(i, n - 1) - 1, 0)];
    }
    int32_t before = 0;                               // run starts in the blocks before this one
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += 256) before += blk_cnt[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
    unsigned long long m[kRsRounds];
    int32_t mine = 0;
#pragma unroll
    for (int r = 0; r < kRsRounds; r++) {
        const int i = base + r * 64 + lane;
        m[r] = __ballot(i < n && run_starts(k[r], kb[r], i));
        mine += __popcll(m[r]);
    }
    if (lane == 0) { wsum[wave] = mine; bsum[wave] = before; }
    __syncthreads();
    int32_t off = bsum[0] + bsum[1] + bsum[2] + bsum[3];
    for (int w = 0; w < wave; w++) off += wsum[w];
#pragma unroll
    for (int r = 0; r < kRsRounds; r++) {
        if ((m[r] >> lane) & 1ull) heads[off + __popcll(m[r] & ((1ull << lane) - 1ull))] = base + r * 64 + lane;
        off += __popcll(m[r]);
    }
    if ((int)blockIdx.x == nblk - 1 && threadIdx.x == 255) *n_out = off;      // (wave 3 of the last block: everything before it, and its own)
}

constexpr int kLongRun = 40;          // a voxel with more points than this is summed by a whole wave (k_vox_centroid_long)
constexpr int kVeryLongRun = 1024;    // ... than this by a workgroup (same kernel)

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
                                                      VoxSetup* __restrict__ s, int32_t* __restrict__ long_list, int32_t* __restrict__ vlong_list,
                                                      unsigned char* __restrict__ out, size_t out_stride, int cap)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_out = s->n_out;
    if (v >= n_out || v >= cap) return;
    const int first = heads[v];
    const int last = (v + 1 < n_out) ? heads[v + 1] : s->n_valid;
    {   // long and very long runs go to their lists (order does not matter: a voxel, an output record); one add per wave and list -
        // thousands of returning adds on one address, one per lane, took longer than the sums
        const int len = last - first, lane = threadIdx.x & 63;
        const bool vl = len > kVeryLongRun, lg = !vl && len > kLongRun;
        const unsigned long long mv = __ballot(vl), ml = __ballot(lg), below = (1ull << lane) - 1ull;
        if (mv) {
            const int lead = __ffsll((long long)mv) - 1;
            int base = 0;
            if (lane == lead) base = atomicAdd(&s->n_vlong, __popcll(mv));
            base = __shfl(base, lead, 64);
            if (vl) vlong_list[base + __popcll(mv & below)] = v;
        }
        if (ml) {
            const int lead = __ffsll((long long)ml) - 1;
            int base = 0;
            if (lane == lead) base = atomicAdd(&s->n_long, __popcll(ml));
            base = __shfl(base, lead, 64);
            if (lg) long_list[base + __popcll(ml & below)] = v;
        }
        if (vl || lg) return;
    }
    float sx = 0.0f, sy = 0.0f, sz = 0.0f, si = 0.0f;
    // The sums are sequential by definition (fp32, ascending point index: the oracle's order), the loads are not: eight points'
    // records are requested at once and added one after the other, and the eight positions after them are on their way while
    // that happens - one round trip per eight points (a lane walking its run one dependent gather at a time was 60 % of
    // extractCloud: 21 points per voxel on average, hundreds where key frames overlap).
    // Every load is unconditional (positions past the run's end are clamped to its last point and their values not added): a
    // load under a lane mask of its own is waited for before the next one is issued.
    const int wofs = stride >= 20 ? 4 : 0;           // (xyz-only records: intensity reads as 0)
    int id[8];
#pragma unroll
    for (int u = 0; u < 8; u++) id[u] = vals[min(first + u, last - 1)];
    for (int j = first; j < last; j += 8) {
        const int cnt = last - j;                     // (8 or more: all eight)
        float x[8], y[8], z[8], w[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float* p = reinterpret_cast<const float*>(pts + (size_t)id[u] * stride);
            x[u] = p[0]; y[u] = p[1]; z[u] = p[2]; w[u] = p[wofs];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) id[u] = vals[min(j + 8 + u, last - 1)];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const bool on = u < cnt;
            sx = on ? sx + x[u] : sx; sy = on ? sy + y[u] : sy; sz = on ? sz + z[u] : sz; si = on ? si + w[u] : si;
        }
    }
    if (wofs == 0) si = 0.0f;
    write_centroid(out, out_stride, v, sx, sy, sz, si, (float)(last - first));
}

// sum + col[0] + col[1] + ... + col[m - 1], added one after the other in that order; the next sixteen values are on their way out of
// LDS while sixteen are added (col: 16-byte aligned, readable up to m rounded up to 16, plus kAddAhead).
constexpr int kAddAhead = 48;
__device__ __forceinline__ float add_run(const float* col, int m, float sum)
{
    const float4* col4 = reinterpret_cast<const float4*>(col);
#define S2M_ADD16(q0, q1, q2, q3) \
    sum += q0.x; sum += q0.y; sum += q0.z; sum += q0.w; sum += q1.x; sum += q1.y; sum += q1.z; sum += q1.w; \
    sum += q2.x; sum += q2.y; sum += q2.z; sum += q2.w; sum += q3.x; sum += q3.y; sum += q3.z; sum += q3.w;
    int k = 0;
    if (m >= 16) {
        // two groups of sixteen take turns: one is added while the other is read (the scheduling barriers keep the reads in
        // front of the adds - left alone, the compiler moves them behind and every group waits for its own reads)
        float4 a0 = col4[0], a1 = col4[1], a2 = col4[2], a3 = col4[3];
        for (; k + 32 <= m; k += 32) {
            const float4 b0 = col4[k / 4 + 4], b1 = col4[k / 4 + 5], b2 = col4[k / 4 + 6], b3 = col4[k / 4 + 7];
            __builtin_amdgcn_sched_barrier(0);
            S2M_ADD16(a0, a1, a2, a3)
            __builtin_amdgcn_sched_barrier(0);
            a0 = col4[k / 4 + 8]; a1 = col4[k / 4 + 9]; a2 = col4[k / 4 + 10]; a3 = col4[k / 4 + 11];
            __builtin_amdgcn_sched_barrier(0);
            S2M_ADD16(b0, b1, b2, b3)
            __builtin_amdgcn_sched_barrier(0);
        }
        if (k + 16 <= m) { S2M_ADD16(a0, a1, a2, a3) k += 16; }
    }
#undef S2M_ADD16
    for (; k < m; k++) sum += col[k];
    return sum;
}

constexpr int kWaveChunk = 256;                      // wave per voxel: records per round trip, four per lane
constexpr int kBlockPer = 4, kBlockRound = 256 * kBlockPer;   // workgroup per voxel: records per round, four per thread
constexpr int kCentroidLds = 2 * 4 * (kBlockRound + kAddAhead);      // floats: the workgroup role's two buffers (the wave role needs half)

__device__ __forceinline__ void centroid_wave_role(float* __restrict__ lds, int block, int nblocks, const unsigned char* __restrict__ pts, size_t stride,
                                                   const int32_t* __restrict__ vals, const int32_t* __restrict__ heads,
                                                   const VoxSetup* __restrict__ s, const int32_t* __restrict__ long_list,
                                                   unsigned char* __restrict__ out, size_t out_stride)
{
    constexpr int kChunk = kWaveChunk;
    static_assert(4 * 4 * (kChunk + kAddAhead) <= kCentroidLds, "the wave role's rows fit the kernel's LDS");
    float (*comp)[4][kChunk + kAddAhead] = reinterpret_cast<float (*)[4][kChunk + kAddAhead]>(lds);   // [wave][component][point of the chunk] (add_run reads ahead)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_long = s->n_long, n_out = s->n_out, n_valid = s->n_valid;
    const int wofs = stride >= 20 ? 4 : 0;           // (xyz-only records: intensity reads as 0)
    auto lds_sync = []() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    for (int item = block * 4 + wave; item < n_long; item += nblocks * 4) {
        const int v = long_list[item];
        const int first = heads[v];
        const int last = (v + 1 < n_out) ? heads[v + 1] : n_valid;
        float sum = 0.0f;                             // lanes 0..3: the running sum of component `lane`
        float rx[4], ry[4], rz[4], rw[4];
        auto fetch = [&](int c) {                     // this lane's four records of the chunk that starts at c
#pragma unroll
            for (int u = 0; u < 4; u++) {
                // (unconditional loads, positions past the run's end clamped to its last point and never added: a load under a
                // lane mask of its own is waited for before the next is issued)
                const float* p = reinterpret_cast<const float*>(pts + (size_t)vals[min(c + 64 * u + lane, last - 1)] * stride);
                rx[u] = p[0]; ry[u] = p[1]; rz[u] = p[2]; rw[u] = p[wofs];
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
            if (lane < 4) sum = add_run(comp[wave][lane], m, sum);
            lds_sync();
        }
        const float sx = __shfl(sum, 0, 64), sy = __shfl(sum, 1, 64), sz = __shfl(sum, 2, 64), si = wofs ? __shfl(sum, 3, 64) : 0.0f;
        if (lane == 0) write_centroid(out, out_stride, v, sx, sy, sz, si, (float)(last - first));
    }
}

// The very long runs (a voxel next to the sensor, seen from every key frame: thousands of points - 9 700 in the chain benchmark).
// Their sums are one chain of dependent adds, ~2 ns each; what can be hidden is everything else: the four waves of a workgroup
// gather 1 024 records per round into registers while lanes 0..3 of wave 0 add the round before out of LDS (two buffers).
__device__ __forceinline__ void centroid_block_role(float* __restrict__ lds, int block, int nblocks, const unsigned char* __restrict__ pts, size_t stride,
                                                    const int32_t* __restrict__ vals, const int32_t* __restrict__ heads,
                                                    const VoxSetup* __restrict__ s, const int32_t* __restrict__ vlong_list,
                                                    unsigned char* __restrict__ out, size_t out_stride)
{
    constexpr int kPer = kBlockPer, kRound = kBlockRound;
    float (*comp)[4][kRound + kAddAhead] = reinterpret_cast<float (*)[4][kRound + kAddAhead]>(lds);      // [buffer][component][point of the round]
    const int tid = threadIdx.x;
    const int n_vlong = s->n_vlong, n_out = s->n_out, n_valid = s->n_valid;
    const int wofs = stride >= 20 ? 4 : 0;           // (xyz-only records: intensity reads as 0)
    for (int item = block; item < n_vlong; item += nblocks) {
        const int v = vlong_list[item];
        const int first = heads[v];
        const int last = (v + 1 < n_out) ? heads[v + 1] : n_valid;
        float sum = 0.0f;                             // wave 0, lanes 0..3: the running sum of component `lane`
        float rx[kPer], ry[kPer], rz[kPer], rw[kPer];
        auto fetch = [&](int c) {                     // this thread's records of the round that starts at c
#pragma unroll
            for (int u = 0; u < kPer; u++) {
                const float* p = reinterpret_cast<const float*>(pts + (size_t)vals[min(c + 256 * u + tid, last - 1)] * stride);
                rx[u] = p[0]; ry[u] = p[1]; rz[u] = p[2]; rw[u] = p[wofs];
            }
        };
        auto store = [&](int b) {
#pragma unroll
            for (int u = 0; u < kPer; u++) {
                comp[b][0][256 * u + tid] = rx[u]; comp[b][1][256 * u + tid] = ry[u];
                comp[b][2][256 * u + tid] = rz[u]; comp[b][3][256 * u + tid] = rw[u];
            }
        };
        fetch(first);
        store(0);
        __syncthreads();
        int b = 0;
        for (int c = first; c < last; c += kRound) {
            const bool more = c + kRound < last;
            if (more) fetch(c + kRound);              // the next round's gathers are in flight behind this round's sums
            if (tid < 4) sum = add_run(comp[b][tid], min(kRound, last - c), sum);
            if (more) store(b ^ 1);
            __syncthreads();                          // (also: buffer b is free again - for the round after the next, or the next voxel)
            b ^= 1;
        }
        if (tid < 4) comp[0][0][tid] = sum;
        __syncthreads();
        if (tid == 0) write_centroid(out, out_stride, v, comp[0][0][0], comp[0][0][1], comp[0][0][2], wofs ? comp[0][0][3] : 0.0f, (float)(last - first));
        __syncthreads();
    }
}

// One launch for both: the first `nvb` workgroups take the very long runs (they are the tail of the stage and start first), the
// others the long ones, a wave each.
__global__ __launch_bounds__(256) void k_vox_centroid_long(const unsigned char* __restrict__ pts, size_t stride,
                                                           const int32_t* __restrict__ vals, const int32_t* __restrict__ heads,
                                                           const VoxSetup* __restrict__ s, const int32_t* __restrict__ long_list,
                                                           const int32_t* __restrict__ vlong_list, int nvb,
                                                           unsigned char* __restrict__ out, size_t out_stride)
{
    __shared__ __attribute__((aligned(16))) float lds[kCentroidLds];
    if ((int)blockIdx.x < nvb) centroid_block_role(lds, blockIdx.x, nvb, pts, stride, vals, heads, s, vlong_list, out, out_stride);
    else centroid_wave_role(lds, blockIdx.x - nvb, gridDim.x - nvb, pts, stride, vals, heads, s, long_list, out, out_stride);
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
    const float w = p[stride >= 20 ? 4 : 0];
    const float ox = t[0] * x + t[1] * y + t[2] * z + t[3];
    const float oy = t[4] * x + t[5] * y + t[6] * z + t[7];
    const float oz = t[8] * x + t[9] * y + t[10] * z + t[11];
    const float oi = stride >= 20 ? w : 0.0f;
    if (out_stride == 32) {                        // the library's own record: two 16-byte stores
        float4* o4 = reinterpret_cast<float4*>(out + (size_t)i * 32);
        o4[0] = make_float4(ox, oy, oz, 1.0f);
        o4[1] = make_float4(oi, 0.0f, 0.0f, 0.0f);
        return;
    }
    float* o = reinterpret_cast<float*>(out + (size_t)i * out_stride);
    o[0] = ox; o[1] = oy; o[2] = oz;
    const int words = (int)(out_stride >> 2);
    if (words > 3) o[3] = 1.0f;
    if (words > 4) o[4] = oi;
    for (int k = 5; k < words; k++) o[k] = 0.0f;
}

}  // namespace

struct VoxWorkspace {
    Buf setup, keys_a, keys_b, vals_a, vals_b, heads, rs_hist, rs_tot, frame_tab, long_list;
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
    Buf* bufs[] = { &w->setup, &w->keys_a, &w->keys_b, &w->vals_a, &w->vals_b, &w->heads, &w->rs_hist, &w->rs_tot, &w->frame_tab, &w->long_list };
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
    VOX_TRY(w->heads.ensure(4 * n));
    const size_t n_long_max = n / kLongRun + 1, n_vlong_max = n / kVeryLongRun + 1;          // (a long run holds more than kLongRun points)
    VOX_TRY(w->long_list.ensure(4 * (n_long_max + n_vlong_max)));
    VoxSetup* s = w->setup.as<VoxSetup>();
    uint32_t* keys_a = w->keys_a.as<uint32_t>(); uint32_t* keys_b = w->keys_b.as<uint32_t>();
    int32_t* vals_a = w->vals_a.as<int32_t>();   int32_t* vals_b = w->vals_b.as<int32_t>();
    int32_t* heads = w->heads.as<int32_t>();

    const int nblk = (ni + kRsTile - 1) / kRsTile;
    VOX_TRY(w->rs_hist.ensure(sizeof(int32_t) * (size_t)kRsBins * (size_t)nblk));          // (k_vox_bbox borrows 8 words per workgroup of its own first: 8 ceil(n / 256) <= 2 048 ceil(n / 4 096))
    VOX_TRY(w->rs_tot.ensure(sizeof(int32_t) * (size_t)(kRsBins > nblk ? kRsBins : nblk)));
    int32_t* rs_hist = w->rs_hist.as<int32_t>(); int32_t* rs_tot = w->rs_tot.as<int32_t>();

    const int nb = (ni + 255) / 256;
    const int nbb = nb < 512 ? nb : 512;
    hipLaunchKernelGGL(k_vox_bbox, dim3(nbb), dim3(256), 0, stream, d_in, stride, ni, reinterpret_cast<uint32_t*>(rs_hist));    // (rs_hist: free until the sort)
    hipLaunchKernelGGL(k_vox_setup, dim3(1), dim3(64), 0, stream, s, leaf, reinterpret_cast<const uint32_t*>(rs_hist), nbb);
    VOX_TRY(hipGetLastError());
    // stable sort of (voxel index, point) by voxel index: three 11-bit passes, a -> b -> a -> b (the first takes the point's position as its value)
    for (int pass = 0; pass < 3; pass++) {
        const uint32_t* kin = (pass & 1) ? keys_b : keys_a; uint32_t* kout = (pass & 1) ? keys_a : keys_b;
        const int32_t* vin = pass == 0 ? nullptr : ((pass & 1) ? vals_b : vals_a); int32_t* vout = (pass & 1) ? vals_a : vals_b;
        const int shift = pass * kRsBits;
        if (pass == 0) hipLaunchKernelGGL(k_rs_hist<true>, dim3(nblk), dim3(256), 0, stream, kin, ni, shift, rs_hist, d_in, stride, (const VoxSetup*)s, keys_a);
        else hipLaunchKernelGGL(k_rs_hist<false>, dim3(nblk), dim3(256), 0, stream, kin, ni, shift, rs_hist, d_in, stride, (const VoxSetup*)s, (uint32_t*)nullptr);
        hipLaunchKernelGGL(k_rs_scan_bins, dim3(kRsBins / kScanDigits), dim3(256), 0, stream, rs_hist, nblk, rs_tot);
        hipLaunchKernelGGL(k_rs_scatter, dim3(nblk), dim3(256), 0, stream, kin, vin, ni, shift, nblk, (const int32_t*)rs_hist, (const int32_t*)rs_tot, kout, vout);
    }
    VOX_TRY(hipGetLastError());
    // run starts: the positions where the sorted key changes, in order, and their number (rs_tot is free again: block counts)
    hipLaunchKernelGGL(k_heads_count, dim3(nblk), dim3(256), 0, stream, (const uint32_t*)keys_b, ni, rs_tot);
    hipLaunchKernelGGL(k_heads_write, dim3(nblk), dim3(256), 0, stream, (const uint32_t*)keys_b, ni, (const int32_t*)rs_tot, nblk, heads, &s->n_out);
    VOX_TRY(hipGetLastError());
    // launched for the worst case (one voxel per point); lanes past n_out exit on the device-side count
    hipLaunchKernelGGL(k_vox_centroid, dim3(nb), dim3(256), 0, stream, d_in, stride, (const int32_t*)vals_b,
                       (const int32_t*)heads, s, w->long_list.as<int32_t>(), w->long_list.as<int32_t>() + n_long_max, d_out, out_stride, (int)(cap < n ? cap : n));
    {   // the long runs, a wave each, and the very long ones, a workgroup each (the grid walks the lists the kernel above left)
        const int nl = ni / kLongRun + 1;
        const int nwb = nl < 4096 ? (nl + 3) / 4 : 1024;
        const int nvb = (int)n_vlong_max < 256 ? (int)n_vlong_max : 256;
        hipLaunchKernelGGL(k_vox_centroid_long, dim3(nvb + nwb), dim3(256), 0, stream, d_in, stride, (const int32_t*)vals_b,
                           (const int32_t*)heads, (const VoxSetup*)s, (const int32_t*)w->long_list.as<int32_t>(),
                           (const int32_t*)(w->long_list.as<int32_t>() + n_long_max), nvb, d_out, out_stride);
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
