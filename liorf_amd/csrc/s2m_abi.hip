// s2m_abi.hip — host side of the gfx950 scan-to-map path and its C ABI (include/liorf_s2m.h).
//
// Owns the device memory (grow-only, sized by the actual point counts), builds the map's
// uniform-grid index and the scan's locality order on the device, and runs the whole
// <= max_iter LM loop as one captured hipGraph (enqueue_loop) so that a scan costs one graph
// launch and one synchronisation.  The voxel-grid stages that feed the path live in s2m_voxel.hip.  There is no CPU fallback: without a
// gfx950 device every entry point fails with S2M_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "s2m_host_math.hpp"
#include "s2m_kernels.hpp"
#include "s2m_voxel.hpp"
#include "s2m_icp.hpp"

using namespace s2m;

namespace {

struct DevBuf {
    void*  p = nullptr;
    size_t cap = 0;
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

inline uint32_t host_f2ord(float f) { uint32_t u; memcpy(&u, &f, 4); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
inline float host_ord2f(uint32_t o) { uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o; float f; memcpy(&f, &u, 4); return f; }

}  // namespace

struct s2m_context {
    s2m_params prm{};
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;

    // map side
    DevBuf raw_map, map_sorted, m_counts, m_cell_start, m_cell_of, m_rank_of;
    // scan side
    DevBuf raw_scan, qx, qy, qz, qperm, npos, front, cert, aux, plane_cache, plane_alt, npos_alt, chunk_parts, chunk_factor, wave_table, n_waves, q_counts, q_cell_start, q_cell_of, q_rank_of, q_block_hist;
    // shared
    DevBuf block_sums, partials, state, trace, dctx, mm, dbg_idx5, dbg_d2, dbg_flag, dbg_coeff, dbg_clk, sc_bins, sc_out;
    // voxel-grid stages that feed the path (section 8(f) F1/F2): staging for host clouds, transformed key frames, filtered clouds
    DevBuf vox_in, vox_out, frames_xf, scan_ds, map_ds;
    // ScanContext store (SCManager's containers, section 8(f) F3): descriptors, fp32 ring keys, sector keys, by key-frame index
    DevBuf sc_store_desc, sc_store_ring, sc_store_sector, sc_cand, sc_res;
    size_t sc_n = 0, sc_cap = 0, sc_n_search = 0;
    int    sc_counter = 0;             // tree_making_period_conter (include/Scancontext.cpp:270-283)
    VoxWorkspace* vox = nullptr;
    IcpWorkspace* icp = nullptr;       // ICP loop-closure alignment (section 8(f) F4)
    DevBuf icp_src, icp_tgt;           // staging of host clouds

    DevCtx hctx{};
    bool ctx_dirty = true;
    size_t n_m = 0, n_q = 0;
    bool have_scan = false;

    // pinned host staging
    DevState* h_state = nullptr;       // [2]: [0] upload, [1] download
    s2m_iter_trace* h_trace = nullptr; // [kMaxIter], directly behind h_state[1]: state and trace come back in one copy
    uint32_t* h_mm = nullptr;          // [6]
    double* h_sc = nullptr;            // [1200 + 20]

    // state that persists across scans in the reference node (:139-140)
    int persist_degenerate = 0;
    float persist_matP[36] = { 0 };

    // batch of scans against this handle's map (s2m_optimize_batch): one child context per scan slot.  A child owns its scan-side
    // buffers, loop state and trace, borrows the parent's map index and runs on the parent's stream; inside the captured batch
    // graph every child's loop is a branch of its own.
    s2m_context* parent = nullptr;
    std::vector<s2m_context*> kids;
    std::vector<hipStream_t> branch_streams;
    std::vector<hipEvent_t> branch_events;
    std::vector<hipEvent_t> prep_events;  // a slot's scan preparation runs on the slot's branch stream, next to the other slots': done when this fires
    std::vector<char> prep_pending;
    hipEvent_t ev_fork = nullptr, ev_prep = nullptr;
    // loop state + trace of every slot in one block (device) with a pinned mirror: one copy brings all of a batch's results back
    DevBuf kid_states;
    unsigned char* h_kid_states = nullptr;
    bool slot_mode = false;            // (a slot running a loop of its own on its own stream: s2m_slot_optimize_*)
    bool state_borrowed = false;       // (a slot: `state` and `h_state` point into the parent's blocks)
    std::map<std::vector<int>, hipGraphExec_t> batch_graphs;
    std::vector<int> batch_live;          // the slots of the batch in flight that run a loop
    bool batch_seg_pending = false;       // the batch in flight was issued as its first range of launches only (early exit on)
    unsigned long long map_epoch = 0;     // bumped by every s2m_set_map: children re-adopt the index when it changed
    unsigned long long adopted_epoch = 0;

    std::map<long long, hipGraphExec_t> graphs;
    std::vector<hipEvent_t> iter_events;
    bool use_graph = true;
    bool fuse_solve = true;            // env S2M_NO_FUSE=1 keeps one k_finalize per iteration (A/B measurements)
    int  fuse_max_blocks = 0;          // largest grid that closes iterations inside k_register (env S2M_FUSE_MAX overrides)
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_c = nullptr, ev_d = nullptr, ev_up = nullptr;
    float t_optimize_ms = 0, t_set_map_ms = 0, t_set_scan_ms = 0;
    int base_parts = 1;
    int  seg_iters = 8;                // with early exit on, the loop is issued as launches 0..seg-1 and, only if those did not converge, the rest (env S2M_SEGMENT, 0 = one piece)
    bool seg_pending = false;          // the launch in flight was the first range only
    hipEvent_t ev_a2 = nullptr, ev_b2 = nullptr;
    int  split_mode = -1;              // env S2M_SPLIT: 1 = every loop runs certify + search kernels, 0 = every loop the fused kernel, 2 (and the default)
                                       // = a lockstep batch whose iterations are closed by k_finalize runs the fused kernel up to launch split_from
                                       // and k_certify_lean + the search kernel from there on; everything else the fused kernel
    bool tune_env = false;             // S2M_TUNE given: the thresholds below are not derived from the workgroup shape
    bool lean_certify = true;          // env S2M_LEAN=0: the certify role by the general kernel even where the 64-register one applies
    int  batch_entries = 1;            // env S2M_BATCH_ENTRIES: wave-table entries per wave in the scan slots of a batch (fewer, longer-running workgroups)
    int  batch_minw = 4;               // env S2M_BATCH_MINW=4: the search / fused kernel of batch slots in the 128-register build
    bool close_in_search = false;      // env S2M_CLOSE_IN_SEARCH=1: late split iterations without a k_finalize launch - ONE search workgroup per slot walks the
                                       // worklist and closes the iteration (1 % faster on the benchmark batch, but a slot with several deferred workgroups then
                                       // works them off one after the other: 70 us in a launch that had three)
    int  search_grid = 8;              // env S2M_SEARCH_GRID: workgroups per slot of a late search launch
    int  split_from = 8;               // env S2M_SPLIT_FROM: first launch that runs certify + search under S2M_SPLIT=2
    bool lockstep = true;              // env S2M_LOCKSTEP=0: the scans of a batch as parallel branches of the graph instead of one grid row each (A/B measurements)
    bool big_blocks = true;            // env S2M_BIG_BLOCKS=0: 8-wave workgroups whatever the scan size (A/B measurements)
    int density_raw = 320;             // box points above which a wave asks for a finer cut (env S2M_DENSITY_RAW, 0 = off)
    bool opt_pending = false;
    bool scan_timing_pending = false;
    int pending_skipped = 0;
    float pending_pose_in[6] = { 0 };
    s2m_iter_trace last_trace[kMaxIter];
    int last_trace_n = 0;
};

namespace {

int fail(s2m_context* h, int code, const char* what, hipError_t e = hipSuccess)
{
    if (h) {
        h->err = what;
        if (e != hipSuccess) { h->err += ": "; h->err += hipGetErrorString(e); }
    }
    return code;
}

#define S2M_HIP(h, call)                                                     \
    do {                                                                     \
        hipError_t e__ = (call);                                             \
        if (e__ != hipSuccess) return fail((h), S2M_ERR_HIP, #call, e__);   \
    } while (0)

int ensure(s2m_context* h, DevBuf& b, size_t bytes)
{
    if (bytes <= b.cap) return S2M_OK;
    size_t want = bytes + bytes / 4 + 256;          // grow-only with slack
    if (b.p) { S2M_HIP(h, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    S2M_HIP(h, hipMalloc(&b.p, want));
    b.cap = want;
    h->ctx_dirty = true;
    return S2M_OK;
}

// the partial rows of a launch (two slots, by launch parity) and, behind them, the search kernel's worklist
int ensure_rows(s2m_context* h, int nblocks)
{
    const size_t rows = sizeof(double) * 2 * kAcc * (size_t)nblocks;
    int rc = ensure(h, h->partials, rows + 64 + sizeof(int32_t) * 2 * (size_t)nblocks);
    if (rc) return rc;
    h->hctx.partials = h->partials.as<double>();
    h->hctx.wl_count = reinterpret_cast<int32_t*>(h->partials.as<unsigned char>() + rows);
    h->hctx.wl_items = h->hctx.wl_count + 16;
    h->ctx_dirty = true;
    return S2M_OK;
}

int upload_ctx(s2m_context* h)
{
    if (!h->ctx_dirty) return S2M_OK;
    hipLaunchKernelGGL(k_set_ctx, dim3(1), dim3(64), 0, h->stream, h->dctx.as<DevCtx>(), h->hctx);
    S2M_HIP(h, hipGetLastError());
    h->ctx_dirty = false;
    return S2M_OK;
}

// exclusive scan of counts[0..n) into out[0..n], out[n] = total
int device_exclusive_scan(s2m_context* h, const int32_t* counts, int32_t* out, int n, int total)
{
    const int nb = (n + 1023) / 1024;
    int rc = ensure(h, h->block_sums, sizeof(int32_t) * (size_t)(nb + 1));
    if (rc) return rc;
    int32_t* sums = h->block_sums.as<int32_t>();
    hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(256), 0, h->stream, counts, out, sums, n);
    if (nb <= 4096) hipLaunchKernelGGL(k_scan_add_fold, dim3(nb), dim3(256), 0, h->stream, out, (const int32_t*)sums, n, total);
    else {
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, h->stream, sums, nb);
        hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(256), 0, h->stream, out, (const int32_t*)sums, n, total);
    }
    S2M_HIP(h, hipGetLastError());
    return S2M_OK;
}

// bounding box of the finite points (device reduction + 24-byte readback)
int device_bbox(s2m_context* h, const unsigned char* d_pts, size_t stride, int n, float mn[3], float mx[3])
{
    const int blocks = std::min((n + 255) / 256, 1024);
    int rc = ensure(h, h->mm, 64 + sizeof(uint32_t) * 8 * 1024);          // mm[0..5], then 8 words per workgroup from word 16 on
    if (rc) return rc;
    uint32_t* part = h->mm.as<uint32_t>() + 16;
    hipLaunchKernelGGL(k_bbox, dim3(blocks), dim3(256), 0, h->stream, d_pts, stride, n, part);
    hipLaunchKernelGGL(k_bbox_fold, dim3(1), dim3(256), 0, h->stream, (const uint32_t*)part, blocks, h->mm.as<uint32_t>());
    S2M_HIP(h, hipGetLastError());
    S2M_HIP(h, hipMemcpyAsync(h->h_mm, h->mm.p, 24, hipMemcpyDeviceToHost, h->stream));
    S2M_HIP(h, hipStreamSynchronize(h->stream));
    for (int d = 0; d < 3; d++) {
        if (h->h_mm[d] == 0xffffffffu || h->h_mm[3 + d] == 0u) { mn[d] = 0.0f; mx[d] = 0.0f; }   // no finite value
        else { mn[d] = host_ord2f(h->h_mm[d]); mx[d] = host_ord2f(h->h_mm[3 + d]); }
    }
    return S2M_OK;
}

int check_records(s2m_context* h, const void* pts, size_t n, size_t stride)
{
    if (!h) return S2M_ERR_INVALID_ARG;
    if (n > 0 && !pts) return fail(h, S2M_ERR_INVALID_ARG, "null point buffer");
    if (stride < 12 || (stride & 3)) return fail(h, S2M_ERR_INVALID_ARG, "stride_bytes must be >= 12 and a multiple of 4");
    if ((reinterpret_cast<uintptr_t>(pts) & 3) != 0) return fail(h, S2M_ERR_INVALID_ARG, "point buffer must be 4-byte aligned");
    if (n > (size_t)0x3fffffff) return fail(h, S2M_ERR_CAPACITY, "too many points");
    return S2M_OK;
}

int set_map_build(s2m_context* h, const void* pts, size_t n, size_t stride, bool on_device);

// The index is rebuilt in place: once the rebuild has started the old index is gone, so a failure on the way leaves the handle
// with NO map (n_m = 0, the reference's "no key poses yet" state, :1297), a new map epoch (batch slots drop what they adopted)
// and no certificates - never the old point count over half-rewritten buffers.
int set_map_impl(s2m_context* h, const void* pts, size_t n, size_t stride, bool on_device)
{
    int rc = check_records(h, pts, n, stride);
    if (rc) return rc;
    rc = set_map_build(h, pts, n, stride, on_device);
    if (rc != S2M_OK && n > 0) {
        const std::string why = h->err;
        h->map_epoch++;
        h->n_m = 0; h->hctx.n_m = 0; h->ctx_dirty = true;
        if (h->have_scan && h->n_q > 0 && h->cert.p) {
            (void)hipMemsetAsync(h->cert.p, 0, sizeof(float4) * h->n_q, h->stream);
            (void)hipMemsetAsync(h->aux.p, 0, sizeof(int4) * h->n_q, h->stream);
        }
        (void)upload_ctx(h);
        (void)hipStreamSynchronize(h->stream);
        h->err = why;
    }
    return rc;
}

int set_map_build(s2m_context* h, const void* pts, size_t n, size_t stride, bool on_device)
{
    int rc;
    S2M_HIP(h, hipSetDevice(h->device));
    S2M_HIP(h, hipEventRecord(h->ev_c, h->stream));
    if (n == 0) { h->map_epoch++; h->n_m = 0; h->hctx.n_m = 0; h->ctx_dirty = true; h->t_set_map_ms = 0; return upload_ctx(h); }

    const unsigned char* d_pts;
    if (on_device) d_pts = static_cast<const unsigned char*>(pts);
    else {
        if ((rc = ensure(h, h->raw_map, n * stride))) return rc;
        S2M_HIP(h, hipMemcpyAsync(h->raw_map.p, pts, n * stride, hipMemcpyHostToDevice, h->stream));
        d_pts = h->raw_map.as<unsigned char>();
    }
    float mn[3], mx[3];
    if ((rc = device_bbox(h, d_pts, stride, (int)n, mn, mx))) return rc;

    // cell edge: every map point with fp32 d2 < gate_sq of a query lies in the 3x3x3 block
    // around the query's cell once E >= sqrt(gate_sq) plus a margin that covers the fp32
    // rounding of (v - o) * inv_e (<= 2.4e-5 cells at |v - o| <= 200).
    // The edge is 2.5 % above the gate: a query's 3x3x3 cells then cover a ball of that much more than the gate around it,
    // which is what lets a point with fewer than 5 neighbours inside the gate prove "and none just outside it either" and
    // keep that verdict over the following launches (s2m_register.hpp, tier A) instead of searching again.
    double E = std::sqrt(h->prm.gate_sq) * 1.025;
    GridDesc g{};
    for (int attempt = 0; attempt < 32; attempt++) {
        const float Ef = (float)E;
        g.inv_e = 1.0f / Ef; g.e = Ef;
        g.ox = mn[0] - Ef; g.oy = mn[1] - Ef; g.oz = mn[2] - Ef;
        double nx = std::floor(((double)mx[0] - g.ox) * g.inv_e) + 2.0;
        double ny = std::floor(((double)mx[1] - g.oy) * g.inv_e) + 2.0;
        double nz = std::floor(((double)mx[2] - g.oz) * g.inv_e) + 2.0;
        if (nx * ny * nz <= (double)(1 << 27) && nx < 65536 && ny < 65536 && nz < 65536) {
            g.nx = (int)nx; g.ny = (int)ny; g.nz = (int)nz; g.ncells = g.nx * g.ny * g.nz;
            break;
        }
        E *= 2.0;     // coarser cells stay correct (only more candidates per query)
        g.ncells = 0;
    }
    if (g.ncells <= 0) return fail(h, S2M_ERR_CAPACITY, "map extent too large for the search grid");

    if ((rc = ensure(h, h->map_sorted, sizeof(float4) * n))) return rc;
    if ((rc = ensure(h, h->m_counts, sizeof(int32_t) * ((size_t)g.ncells + 1)))) return rc;
    if ((rc = ensure(h, h->m_cell_start, sizeof(int32_t) * ((size_t)g.ncells + 1)))) return rc;
    if ((rc = ensure(h, h->m_cell_of, sizeof(int32_t) * n))) return rc;
    if ((rc = ensure(h, h->m_rank_of, sizeof(int32_t) * n))) return rc;

    S2M_HIP(h, hipMemsetAsync(h->m_counts.p, 0, sizeof(int32_t) * ((size_t)g.ncells + 1), h->stream));
    const int nb = ((int)n + 255) / 256;
    hipLaunchKernelGGL(k_bin_count, dim3(nb), dim3(256), 0, h->stream, d_pts, stride, (int)n, g,
                       h->m_cell_of.as<int32_t>(), h->m_rank_of.as<int32_t>(), h->m_counts.as<int32_t>());
    S2M_HIP(h, hipGetLastError());
    if ((rc = device_exclusive_scan(h, h->m_counts.as<int32_t>(), h->m_cell_start.as<int32_t>(), g.ncells, (int)n))) return rc;
    hipLaunchKernelGGL(k_scatter_map, dim3(nb), dim3(256), 0, h->stream, d_pts, stride, (int)n,
                       (const int32_t*)h->m_cell_of.as<int32_t>(), (const int32_t*)h->m_rank_of.as<int32_t>(),
                       (const int32_t*)h->m_cell_start.as<int32_t>(), h->map_sorted.as<float4>());
    S2M_HIP(h, hipGetLastError());

    if (h->have_scan && h->n_q > 0 && h->cert.p) {           // tuples and certificates of the old map are meaningless now
        S2M_HIP(h, hipMemsetAsync(h->cert.p, 0, sizeof(float4) * h->n_q, h->stream));
        S2M_HIP(h, hipMemsetAsync(h->aux.p, 0, sizeof(int4) * h->n_q, h->stream));
    }
    h->map_epoch++;
    h->n_m = n;                                       // committed only now; a failure above leaves no map at all (set_map_impl)
    h->hctx.n_m = (int32_t)n;
    h->hctx.g = g;
    h->hctx.map_sorted = h->map_sorted.as<float4>();
    h->hctx.cell_start = h->m_cell_start.as<int32_t>();
    h->ctx_dirty = true;
    S2M_HIP(h, hipEventRecord(h->ev_d, h->stream));
    if ((rc = upload_ctx(h))) return rc;
    S2M_HIP(h, hipStreamSynchronize(h->stream));     // the caller's buffer is free again
    S2M_HIP(h, hipEventElapsedTime(&h->t_set_map_ms, h->ev_c, h->ev_d));
    return S2M_OK;
}

// s2m_set_scan in three steps, so that the scan slots of a batch share the launches of the middle one:
//   scan_slot_prepare   host side: sizes, buffers, the upload of a host source, the DevCtx fields - and the slot's row of the table
//   launch_scan_prep    the six ordering kernels, one grid row per slot (blockIdx.y)
//   scan_slot_finish    bookkeeping
int scan_slot_prepare(s2m_context* h, const void* pts, size_t n, size_t stride, bool on_device, PrepSlot* ps)
{
    int rc = check_records(h, pts, n, stride);
    if (rc) return rc;
    memset(ps, 0, sizeof(*ps));
    h->n_q = n; h->have_scan = true;
    h->hctx.n_q = (int32_t)n;
    // wave table capacity: every 64-point chunk plus a 50 % budget of extra waves for split chunks
    const int n_chunks = (int)((n + 63) / 64);
    constexpr int NW = kBlock / 64;
    // how finely to cut when the scan alone cannot fill the GPU (~2048 co-resident waves): 1, 2, 4 or 8
    int base_parts = 1;
    while (base_parts < 8 && n_chunks * base_parts * 2 <= 2048) base_parts *= 2;
    h->base_parts = base_parts;
    int nblocks = (n_chunks * base_parts + n_chunks / 2 + NW - 1) / NW;
    nblocks = ((nblocks + kBlocksQuantum - 1) / kBlocksQuantum) * kBlocksQuantum;
    if (nblocks == 0) nblocks = kBlocksQuantum;
    const int table_cap = nblocks * NW;                   // wave-table entries: every chunk plus the budget for split chunks
    // more entries than one 8-wave workgroup per CU holds: 16-wave workgroups, one per CU (s2m_types.h, kBigWaves)
    const int wpb = (h->big_blocks && n_chunks * base_parts > (kMaxBlocks / 2) * NW) ? kBigWaves : NW;
    // the grid stays co-resident - one workgroup per CU in either shape (the 8-wave shape of a single scan is built for 2 waves
    // per SIMD: 256 registers per lane, no scratch; the scan slots of a batch run the 128-register build, two per CU) - and waves
    // loop over the table
    nblocks = std::min((table_cap + wpb - 1) / wpb, (wpb == NW && h->parent) ? kMaxBlocks : kMaxBlocks / 2);
    if (h->parent && h->batch_entries > 1) nblocks = std::max((nblocks + h->batch_entries - 1) / h->batch_entries, 1);   // a batch slot: several entries per wave
    h->hctx.wpb = wpb;
    // cutting a dense first-launch pass to 32 lanes pays in the 8-wave shape (kitti64: launch 0 89 -> 69 us); the 128-register
    // build of the 16-wave shape loses more to the extra passes than it gains (ouster128 160 -> 182 us, dense1m 318 -> 345 us)
    h->hctx.nblocks = nblocks;
    h->hctx.table_cap = table_cap;
    h->ctx_dirty = true;
    if ((rc = ensure_rows(h, nblocks))) return rc;
    if (n == 0) return S2M_OK;

    const unsigned char* d_pts;
    if (on_device) d_pts = static_cast<const unsigned char*>(pts);
    else {
        if ((rc = ensure(h, h->raw_scan, n * stride))) return rc;
        S2M_HIP(h, hipMemcpyAsync(h->raw_scan.p, pts, n * stride, hipMemcpyHostToDevice, h->stream));
        S2M_HIP(h, hipEventRecord(h->ev_up, h->stream));   // waited for in scan_slot_finish: the caller's buffer is free when the call returns
        d_pts = h->raw_scan.as<unsigned char>();
    }
    // locality order of the scan: log-polar Z-order bins (k_polar_count), no host round trip
    if ((rc = ensure(h, h->qx, sizeof(float) * n))) return rc;
    if ((rc = ensure(h, h->qy, sizeof(float) * n))) return rc;
    if ((rc = ensure(h, h->qz, sizeof(float) * n))) return rc;
    if ((rc = ensure(h, h->qperm, sizeof(int32_t) * n))) return rc;
    if ((rc = ensure(h, h->npos, sizeof(int32_t) * 5 * n))) return rc;
    if ((rc = ensure(h, h->plane_cache, sizeof(float4) * n))) return rc;
    if ((rc = ensure(h, h->plane_alt, sizeof(float4) * n))) return rc;
    if ((rc = ensure(h, h->npos_alt, sizeof(int32_t) * 5 * n))) return rc;
    if ((rc = ensure(h, h->cert, sizeof(float4) * n))) return rc;           // cert and aux are reset by k_scatter_scan
    if ((rc = ensure(h, h->aux, sizeof(int4) * n))) return rc;
    if ((rc = ensure(h, h->front, sizeof(float4) * kNbrCap * n))) return rc;
    if ((rc = ensure(h, h->q_cell_start, sizeof(int32_t) * kPolarCells))) return rc;
    if ((rc = ensure(h, h->q_cell_of, sizeof(int32_t) * n))) return rc;
    if ((rc = ensure(h, h->q_rank_of, sizeof(int32_t) * n))) return rc;
    const int npb = ((int)n + kPolarBlock - 1) / kPolarBlock;
    if ((rc = ensure(h, h->q_block_hist, sizeof(int32_t) * (size_t)kPolarCells * (size_t)npb))) return rc;
    if ((rc = ensure(h, h->chunk_parts, sizeof(int32_t) * (size_t)(n_chunks + 1)))) return rc;
    {   // density wishes: all zero between uses (k_chunk_table_density clears what it consumes)
        const void* before = h->chunk_factor.p;
        if ((rc = ensure(h, h->chunk_factor, sizeof(int32_t) * (size_t)(n_chunks + 1)))) return rc;
        if (h->chunk_factor.p != before) S2M_HIP(h, hipMemsetAsync(h->chunk_factor.p, 0, h->chunk_factor.cap, h->stream));
    }
    if ((rc = ensure(h, h->wave_table, sizeof(int2) * (size_t)table_cap))) return rc;
    if ((rc = ensure(h, h->n_waves, 64))) return rc;

    ps->pts = d_pts; ps->stride = stride; ps->n = (int32_t)n; ps->n_chunks = n_chunks; ps->base_parts = base_parts;
    ps->capacity = table_cap; ps->npb = npb;
    ps->cell_of = h->q_cell_of.as<int32_t>(); ps->rank_of = h->q_rank_of.as<int32_t>(); ps->block_hist = h->q_block_hist.as<int32_t>();
    ps->counts = h->q_counts.as<int32_t>(); ps->cell_start = h->q_cell_start.as<int32_t>();
    ps->qx = h->qx.as<float>(); ps->qy = h->qy.as<float>(); ps->qz = h->qz.as<float>(); ps->qperm = h->qperm.as<int32_t>();
    ps->cert = h->cert.as<float4>(); ps->aux = h->aux.as<int4>();
    ps->chunk_parts = h->chunk_parts.as<int32_t>(); ps->wave_table = h->wave_table.as<int2>(); ps->n_waves = h->n_waves.as<int32_t>();

    h->hctx.qx = h->qx.as<float>(); h->hctx.qy = h->qy.as<float>(); h->hctx.qz = h->qz.as<float>();
    h->hctx.qperm = h->qperm.as<int32_t>();
    h->hctx.wave_table = h->wave_table.as<int2>();
    h->hctx.n_waves = h->n_waves.as<int32_t>();
    h->hctx.wave_table_rw = h->wave_table.as<int2>();
    h->hctx.n_waves_rw = h->n_waves.as<int32_t>();
    h->hctx.chunk_parts = h->chunk_parts.as<int32_t>();
    h->hctx.chunk_factor = h->chunk_factor.as<int32_t>();
    h->hctx.n_chunks = n_chunks;
    h->hctx.density_pending = 1;
    h->hctx.npos = h->npos.as<int32_t>();
    h->hctx.cert = h->cert.as<float4>();
    h->hctx.aux = h->aux.as<int4>();
    h->hctx.front = h->front.as<float4>();
    h->hctx.plane_cache = h->plane_cache.as<float4>();
    h->hctx.plane_alt = h->plane_alt.as<float4>();
    h->hctx.npos_alt = h->npos_alt.as<int32_t>();
    h->ctx_dirty = true;
    return S2M_OK;
}

// the ordering kernels of `nslots` prepared slots (rows with n = 0 do nothing), on `stream`
void launch_scan_prep(hipStream_t stream, const PrepTable& t, int nslots)
{
    int npb = 0, nb = 0, ncb = 0;
    for (int k = 0; k < nslots; k++) {
        npb = std::max(npb, (int)t.s[k].npb);
        nb = std::max(nb, (t.s[k].n + 255) / 256);
        ncb = std::max(ncb, (t.s[k].n_chunks + 3) / 4);
    }
    if (npb == 0) return;
    hipLaunchKernelGGL(k_polar_count, dim3(npb, nslots), dim3(kPolarBlock), 0, stream, t);
    hipLaunchKernelGGL(k_polar_prefix, dim3(kPolarCells / 64, nslots), dim3(1024), 0, stream, t);
    hipLaunchKernelGGL(k_polar_scan, dim3(1, nslots), dim3(1024), 0, stream, t);
    hipLaunchKernelGGL(k_scatter_scan, dim3(nb, nslots), dim3(256), 0, stream, t);
    hipLaunchKernelGGL(k_chunk_parts, dim3(ncb, nslots), dim3(256), 0, stream, t);
    hipLaunchKernelGGL(k_chunk_table, dim3(1, nslots), dim3(1024), 0, stream, t);
}

int set_scan_impl(s2m_context* h, const void* pts, size_t n, size_t stride, bool on_device)
{
    S2M_HIP(h, hipSetDevice(h->device));
    S2M_HIP(h, hipEventRecord(h->ev_c, h->stream));
    PrepTable t;
    int rc = scan_slot_prepare(h, pts, n, stride, on_device, &t.s[0]);
    if (rc) return rc;
    if (n == 0) { h->t_set_scan_ms = 0; h->scan_timing_pending = false; return upload_ctx(h); }
    launch_scan_prep(h->stream, t, 1);
    S2M_HIP(h, hipGetLastError());
    S2M_HIP(h, hipEventRecord(h->ev_d, h->stream));
    h->scan_timing_pending = true;
    // (the DevCtx block goes to the device with the next launch that needs it: upload_ctx / push_state)
    // A host source (pageable or pinned) has been copied by the time this returns; the ordering kernels behind the copy
    // keep running.  A device-resident source is read asynchronously and must outlive the next synchronising call.
    if (!on_device) S2M_HIP(h, hipEventSynchronize(h->ev_up));
    return S2M_OK;
}

void fill_state(s2m_context* h, DevState* s, const float pose[6])
{
    memset(s, 0, sizeof(*s));
    memcpy(s->pose, pose, 24);
    memcpy(s->pose2[0], pose, 24);       // launch 0 runs with slot 0
    host_pose_to_transform(pose, s->T, s->sc);
    s->T_valid = 1;
    memcpy(s->matP, h->persist_matP, sizeof(s->matP));
    s->isDegenerate = h->persist_degenerate;
}

int push_state(s2m_context* h, const float pose[6])
{
    DevState s;
    fill_state(h, &s, pose);
    if (h->ctx_dirty) {
        hipLaunchKernelGGL(k_set_ctx_state, dim3(1), dim3(64), 0, h->stream, h->dctx.as<DevCtx>(), h->hctx, h->state.as<DevState>(), s,
                           (const int32_t*)h->n_waves.as<int32_t>());
        h->ctx_dirty = false;
    } else
        hipLaunchKernelGGL(k_set_state, dim3(1), dim3(64), 0, h->stream, h->state.as<DevState>(), s,
                           (const int32_t*)h->n_waves.as<int32_t>());
    S2M_HIP(h, hipGetLastError());
    return S2M_OK;
}

// The LM loop (:1304-1315) as a launch sequence.  One iteration L is either one launch of the fused kernel R(L), or -
// split - the certify kernel C(L) followed by the search kernel S(L) for the workgroups C put on its worklist (launch 0 of a
// scan, where nothing is certified yet: S alone, over every workgroup).  When the whole grid is co-resident iterations
// 1..n-2 are closed inside the prologue of the following R / C (solve_prev): every workgroup repeats the small solve, and
// a kernel boundary plus a one-workgroup kernel disappear from every iteration:
//   R0 F0 R1 R2' R3' ... R(n-1)' F(n-1)        (' = closes the iteration before it)
//   S0 F0 C1 S1 C2' S2 ... C(n-1)' S(n-1) F(n-1)
// The plain form  R0 F0 R1 F1 ...  remains for A/B measurements (S2M_NO_FUSE=1).
// Several scan slots (a batch) advance in lockstep: every launch has one grid row per slot.
constexpr int kFuseMaxBlocks = 512;
static_assert(sizeof(CtxTable) <= 4096 && sizeof(StateInitTable) <= 4096 && sizeof(PrepTable) <= 4096 && sizeof(SlotTable) <= 4096,
              "kernel arguments are limited to 4 KB");

// what one loop launches over: the scan slots (one for a single scan), the widest grid among them, their common workgroup shape
struct LoopShape {
    SlotTable tbl;
    int nslots = 1;
    int nblocks = 0;        // grid.x of the registration kernels: the largest DevCtx::nblocks among the slots
    int table_cap = 0;      // the largest wave-table capacity among the slots (grid of k_wave_density)
    int wpb = kBlock / 64;  // waves per workgroup (DevCtx::wpb, the same for every slot)
    bool batch = false;     // scan slots of a batch
    bool split = false;     // C + S per iteration instead of R
    bool late = false;      // the split iterations of this loop come late (few rows on the worklist): small search grid
    bool split_auto = false; // split if the loop's iterations are closed by k_finalize and the lean certify kernel applies
};

LoopShape shape_of(s2m_context* h)
{
    LoopShape sh;
    memset(&sh.tbl, 0, sizeof(sh.tbl));
    sh.tbl.ctx[0] = h->dctx.as<DevCtx>();
    sh.tbl.st[0] = h->state.as<DevState>();
    sh.nslots = 1; sh.nblocks = h->hctx.nblocks; sh.table_cap = h->hctx.table_cap; sh.wpb = h->hctx.wpb;
    sh.batch = h->parent != nullptr && !h->slot_mode;
    sh.split = h->split_mode == 1;
    return sh;
}

template <bool HOOK, int NW, int MINW, int MODE, int CNW>
inline void launch_k(hipStream_t s, const LoopShape& sh, int L, int flags, int grid_x = 0)
{
    hipLaunchKernelGGL((k_register<HOOK, NW, MINW, MODE, CNW>), dim3(grid_x > 0 ? std::min(grid_x, sh.nblocks) : sh.nblocks, sh.nslots), dim3(NW * 64), 0, s,
                       sh.tbl, L, flags);
}

// one fused launch R(L) in the workgroup shape of the scan (DevCtx::wpb); `hook`: the observation variant
inline void launch_fused(s2m_context* h, const LoopShape& sh, bool hook, int L, int solve_prev)
{
    constexpr int NW = kBlock / 64;
    const int fl = solve_prev ? kFlagSolvePrev : 0;
    if (sh.wpb == kBigWaves) {
        if (hook) launch_k<true, kBigWaves, 4, kFused, kBigWaves>(h->stream, sh, L, fl);
        else      launch_k<false, kBigWaves, 4, kFused, kBigWaves>(h->stream, sh, L, fl);
    } else if (sh.batch && h->batch_minw == 4 && !hook) {
        launch_k<false, NW, 4, kFused, NW>(h->stream, sh, L, fl);          // two workgroups per CU (spills in the search path)
    } else {
        if (hook) launch_k<true, NW, 2, kFused, NW>(h->stream, sh, L, fl);
        else      launch_k<false, NW, 2, kFused, NW>(h->stream, sh, L, fl);
    }
}

// the search kernel S(L): over the worklist C(L) left, or (all) over every workgroup
// (small_grid: late in a loop the list is short - a few workgroups per slot walk it instead of one workgroup per row)
inline void launch_search(s2m_context* h, const LoopShape& sh, int L, bool all, bool small_grid = false)
{
    constexpr int NW = kBlock / 64;
    const bool close_after = small_grid && !all && h->close_in_search;     // one workgroup per slot walks the list and closes the iteration
    const int fl = (all ? kFlagAll : 0) | (close_after ? kFlagCloseAfter : 0);
    const int gx = close_after ? 1 : ((small_grid && !all) ? h->search_grid : 0);
    if (sh.wpb == kBigWaves) launch_k<false, NW, 2, kSearch, kBigWaves>(h->stream, sh, L, fl, gx);
    else if (sh.batch && h->batch_minw == 4) launch_k<false, NW, 4, kSearch, NW>(h->stream, sh, L, fl, gx);
    else                     launch_k<false, NW, 2, kSearch, NW>(h->stream, sh, L, fl, gx);
}

inline void launch_certify(s2m_context* h, const LoopShape& sh, int L, int solve_prev, bool fused_loop)
{
    constexpr int NW = kBlock / 64;
    const int fl = solve_prev ? kFlagSolvePrev : 0;
    if (!fused_loop && sh.wpb == NW && h->lean_certify)      // iterations closed by k_finalize: the 64-register kernel
        hipLaunchKernelGGL((k_certify_lean<NW, 1, kCertifyLeanWaves>), dim3(sh.nblocks, sh.nslots), dim3(NW * 64), 0, h->stream, sh.tbl, L);
    else if (sh.wpb == kBigWaves) launch_k<false, kBigWaves, kCertifyWavesBig, kCertify, kBigWaves>(h->stream, sh, L, fl);
    else                     launch_k<false, NW, kCertifyWaves, kCertify, NW>(h->stream, sh, L, fl);
}

// iteration L of the loop: R(L), or C(L) S(L)
inline void launch_iteration(s2m_context* h, const LoopShape& sh, int L, int solve_prev, bool fused_loop = true)
{
    if (!sh.split) { launch_fused(h, sh, false, L, solve_prev); return; }
    if (L == 0) { launch_search(h, sh, 0, true); return; }
    launch_certify(h, sh, L, solve_prev, fused_loop);
    launch_search(h, sh, L, false, sh.late);
}

inline void launch_finalize(s2m_context* h, const LoopShape& sh, int L, int mode)
{
    hipLaunchKernelGGL(k_finalize, dim3(1, sh.nslots), dim3(kFinThreads), 0, h->stream, sh.tbl, L, mode);
}

inline void launch_density(s2m_context* h, const LoopShape& sh)
{
    // re-split the wave table for the map density at the initial guess (the transform k_set_state just stored);
    // both kernels take everything from the DevCtx block, so the captured graph stays valid from scan to scan
    hipLaunchKernelGGL(k_wave_density, dim3((sh.table_cap + 3) / 4, sh.nslots), dim3(256), 0, h->stream, sh.tbl, h->density_raw);
    hipLaunchKernelGGL(k_chunk_table_density, dim3(1, sh.nslots), dim3(1024), 0, h->stream, sh.tbl);
}

// Launches L0 .. L1-1 of the loop.  A range that starts after launch 0 begins like launch 1 does (transform rebuilt from the
// pose the k_finalize before it stored), and a range that ends before the last launch closes its last iteration with a
// k_finalize of its own: with early exit on, the loop is issued in two ranges and the second only if the first did not converge.
// `events`, if given, holds 2*n events recorded around every iteration's launches; with `coarse` only four pairs are recorded - around
// launch 0, launch 1, the run of back-to-back launches 2 .. n-2 (slots 4, 5) and launch n-1 (slots 6, 7) - so that the event
// packets do not break up the loop's back-to-back dispatch.
void enqueue_loop(s2m_context* h, const LoopShape& sh_in, hipEvent_t* events, bool coarse = false, int L0 = 0, int L1 = -1)
{
    const int n = h->prm.max_iter;
    if (L1 < 0) L1 = n;
    const bool fuse = h->fuse_solve && sh_in.nblocks * sh_in.nslots <= h->fuse_max_blocks;     // the whole grid co-resident (see above)
    LoopShape sh = sh_in;
    // S2M_SPLIT=2: from launch `split_from` on (the first launches search most points: there the certify kernel is only one
    // more launch in front of the search) the iterations of a loop closed by k_finalize run certify (lean) + search
    const bool split_late = sh.split_auto && !fuse && sh.wpb == kBlock / 64 && h->lean_certify;   // (with early exit on these launches are the second range: issued only for scans that need them)
    if (h->density_raw > 0 && L0 == 0) launch_density(h, sh);
    for (int L = L0; L < L1; L++) {
        const int slot = !coarse ? 2 * L : (L == 0 ? 0 : (L == 1 ? 2 : (L == n - 1 ? 6 : 4)));
        const bool open = events && (!coarse || L <= 2 || L == n - 1), close = events && (!coarse || L <= 1 || L >= n - 2);
        if (open) (void)hipEventRecord(events[slot], h->stream);
        sh.split = sh_in.split || (split_late && L >= h->split_from);
        sh.late = !sh_in.split && sh.split;
        launch_iteration(h, sh, L, (fuse && L >= 2 && L != L0) ? 1 : 0, fuse);
        if (close) (void)hipEventRecord(events[slot + 1], h->stream);
        if ((!fuse || L == 0 || L == L1 - 1) && !(sh.late && h->close_in_search)) launch_finalize(h, sh, L, 0);   // (late split: the search launch closes)
    }
}

// part 0: the whole loop; 1: launches 0 .. seg-1; 2: launches seg .. max_iter-1
int get_graph(s2m_context* h, int nblocks, int part, hipGraphExec_t* out)
{
    // table_cap fixes both grids in the captured loop: k_register's (nblocks) and k_wave_density's
    const LoopShape sh = shape_of(h);
    const long long key = (((long long)h->hctx.table_cap * 4 + part) * 256 + (part ? h->seg_iters : 0)) * 4 + (sh.split ? 1 : 0) + (sh.wpb == kBigWaves ? 2 : 0);
    auto it = h->graphs.find(key);
    if (it != h->graphs.end()) { *out = it->second; return S2M_OK; }
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    S2M_HIP(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    enqueue_loop(h, sh, nullptr, false, part == 2 ? h->seg_iters : 0, part == 1 ? h->seg_iters : -1);
    // loop state + trace come back as the last node of the graph (into the pinned mirror: its address never changes): a copy
    // issued behind the graph starts ~10 us after the graph's last kernel
    (void)hipMemcpyAsync(&h->h_state[1], h->state.p, sizeof(DevState) + sizeof(s2m_iter_trace) * h->prm.max_iter, hipMemcpyDeviceToHost, h->stream);
    hipError_t e = hipStreamEndCapture(h->stream, &graph);
    if (e != hipSuccess || !graph) return fail(h, S2M_ERR_HIP, "hipStreamEndCapture", e);
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return fail(h, S2M_ERR_HIP, "hipGraphInstantiate", e);
    h->graphs[key] = exec;
    *out = exec;
    return S2M_OK;
}

int launch_loop(s2m_context* h, int part = 0)
{
    const int nblocks = h->hctx.nblocks;
    hipEvent_t e0 = part == 2 ? h->ev_a2 : h->ev_a, e1 = part == 2 ? h->ev_b2 : h->ev_b;
    if (h->use_graph) {
        hipGraphExec_t exec = nullptr;
        int rc = get_graph(h, nblocks, part, &exec);
        if (rc == S2M_OK) {
            S2M_HIP(h, hipEventRecord(e0, h->stream));
            S2M_HIP(h, hipGraphLaunch(exec, h->stream));
            S2M_HIP(h, hipEventRecord(e1, h->stream));
            h->hctx.density_pending = 0;                  // cleared on the device by k_chunk_table_density: keep the host copy in step
            return S2M_OK;
        }
        h->use_graph = false;       // capture unsupported here: fall back to plain launches
    }
    S2M_HIP(h, hipEventRecord(e0, h->stream));
    enqueue_loop(h, shape_of(h), nullptr, false, part == 2 ? h->seg_iters : 0, part == 1 ? h->seg_iters : -1);
    S2M_HIP(h, hipGetLastError());
    S2M_HIP(h, hipEventRecord(e1, h->stream));
    S2M_HIP(h, hipMemcpyAsync(&h->h_state[1], h->state.p, sizeof(DevState) + sizeof(s2m_iter_trace) * h->prm.max_iter, hipMemcpyDeviceToHost, h->stream));
    h->hctx.density_pending = 0;
    return S2M_OK;
}

// the parameters the kernels read, into the host copy of the DevCtx block
void params_to_ctx(s2m_context* h, const s2m_params& prm)
{
    h->hctx.gate_f = nextafterf((float)prm.gate_sq, INFINITY);
    h->hctx.gate_r = nextafterf(sqrtf((float)prm.gate_sq), 0.0f);
    h->hctx.gate_sq = prm.gate_sq; h->hctx.plane_tol = prm.plane_tol; h->hctx.weight_scale = prm.weight_scale;
    h->hctx.weight_min = prm.weight_min; h->hctx.conv_deg = prm.conv_deg; h->hctx.conv_cm = prm.conv_cm;
    h->hctx.eig_thresh = prm.eig_thresh; h->hctx.min_corr = prm.min_corr; h->hctx.max_iter = prm.max_iter;
    h->hctx.early_exit = prm.early_exit;
    h->ctx_dirty = true;
}

constexpr size_t kDsStride = 32;       // filtered clouds are kept as pcl::PointXYZI records

int check_leaf(s2m_context* h, float leaf)
{
    if (!(leaf > 0.0f) || !std::isfinite(leaf)) return fail(h, S2M_ERR_INVALID_ARG, "leaf size must be positive and finite");
    return S2M_OK;
}

// VoxelGrid of a device cloud into `dst` (grown to hold one record per input point, the worst case).
int voxel_into(s2m_context* h, const unsigned char* d_in, size_t n, size_t stride, float leaf, DevBuf& dst, VoxResult* res)
{
    int rc = ensure(h, dst, kDsStride * (n ? n : 1));
    if (rc) return rc;
    hipError_t e = vox_downsample(h->vox, h->stream, d_in, n, stride, leaf, dst.as<unsigned char>(), kDsStride, n, res);
    if (e != hipSuccess) return fail(h, S2M_ERR_HIP, "voxel grid filter", e);
    return S2M_OK;
}

// strided device records -> caller's host buffer (min(n, cap) records)
int download_records(s2m_context* h, const DevBuf& src, size_t n, void* out, size_t out_stride, size_t cap)
{
    const size_t m = n < cap ? n : cap;
    if (m == 0 || !out) return S2M_OK;
    S2M_HIP(h, hipMemcpy2DAsync(out, out_stride, src.p, kDsStride, out_stride < kDsStride ? out_stride : kDsStride, m,
                                hipMemcpyDeviceToHost, h->stream));
    S2M_HIP(h, hipStreamSynchronize(h->stream));
    return S2M_OK;
}

// ---- ScanContext store ---------------------------------------------------------------------------
constexpr size_t kScDesc = (size_t)S2M_SC_NUM_RING * S2M_SC_NUM_SECTOR;

// grow-with-copy (ensure() alone would drop the contents)
int sc_reserve(s2m_context* h, size_t want)
{
    if (want <= h->sc_cap) return S2M_OK;
    size_t cap = h->sc_cap ? h->sc_cap : 256;
    while (cap < want) cap *= 2;
    struct Part { DevBuf* b; size_t elem; } parts[3] = { { &h->sc_store_desc, sizeof(double) * kScDesc },
                                                         { &h->sc_store_ring, sizeof(float) * S2M_SC_NUM_RING },
                                                         { &h->sc_store_sector, sizeof(double) * S2M_SC_NUM_SECTOR } };
    for (Part& p : parts) {
        void* np = nullptr;
        S2M_HIP(h, hipMalloc(&np, p.elem * cap));
        if (h->sc_n) S2M_HIP(h, hipMemcpyAsync(np, p.b->p, p.elem * h->sc_n, hipMemcpyDeviceToDevice, h->stream));
        S2M_HIP(h, hipStreamSynchronize(h->stream));
        if (p.b->p) S2M_HIP(h, hipFree(p.b->p));
        p.b->p = np; p.b->cap = p.elem * cap;
    }
    h->sc_cap = cap;
    return S2M_OK;
}

// descriptor + ring key are in h->sc_out (device): append them as key frame sc_n
int sc_append_from_out(s2m_context* h)
{
    int rc = sc_reserve(h, h->sc_n + 1);
    if (rc) return rc;
    hipLaunchKernelGGL(k_sc_append, dim3(1), dim3(kScThreads), 0, h->stream, (const double*)h->sc_out.as<double>(),
                       h->sc_store_desc.as<double>(), h->sc_store_ring.as<float>(), h->sc_store_sector.as<double>(), (int)h->sc_n);
    S2M_HIP(h, hipGetLastError());
    h->sc_n++;
    return S2M_OK;
}

// SCManager::makeScancontext + ring key of a host cloud into h->sc_out (device)
int sc_build_descriptor(s2m_context* h, const void* pts, size_t n, size_t stride_bytes)
{
    S2M_HIP(h, hipMemsetAsync(h->sc_bins.p, 0, sizeof(uint32_t) * kScDesc, h->stream));
    if (n > 0) {
        int rc = ensure(h, h->raw_scan, n * stride_bytes);
        if (rc) return rc;
        S2M_HIP(h, hipMemcpyAsync(h->raw_scan.p, pts, n * stride_bytes, hipMemcpyHostToDevice, h->stream));
        const int blocks = std::min((int)((n + 255) / 256), 1024);
        hipLaunchKernelGGL(k_sc_polar_max, dim3(blocks), dim3(256), 0, h->stream,
                           (const unsigned char*)h->raw_scan.as<unsigned char>(), stride_bytes, (int)n, h->sc_bins.as<uint32_t>());
    }
    hipLaunchKernelGGL(k_sc_finish, dim3(1), dim3(64), 0, h->stream, (const uint32_t*)h->sc_bins.as<uint32_t>(),
                       h->sc_out.as<double>(), h->sc_out.as<double>() + kScDesc);
    S2M_HIP(h, hipGetLastError());
    return S2M_OK;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

const char* s2m_version(void) { return "liorf_amd s2m 0.1 (gfx950)"; }

int s2m_default_params(s2m_params* p)
{
    if (!p) return S2M_ERR_INVALID_ARG;
    memset(p, 0, sizeof(*p));
    p->struct_size = (uint32_t)sizeof(s2m_params);
    p->device_id = 0; p->stream = nullptr;
    p->k_neighbors = 5; p->gate_sq = 1.0; p->plane_tol = 0.2; p->weight_scale = 0.9; p->weight_min = 0.1;
    p->min_corr = 50; p->min_feats = 30; p->max_iter = 30; p->eig_thresh = 100.0f;
    p->conv_deg = 0.05; p->conv_cm = 0.05; p->z_tol = FLT_MAX; p->rot_tol = FLT_MAX;
    p->imu_type = 0; p->imu_rpy_weight = 0.01f; p->early_exit = 1;
    return S2M_OK;
}

int s2m_create(const s2m_params* p, s2m_handle* out)
{
    if (!out) return S2M_ERR_INVALID_ARG;
    *out = nullptr;
    s2m_params prm;
    if (p) {
        if (p->struct_size != sizeof(s2m_params)) return S2M_ERR_INVALID_ARG;
        prm = *p;
    } else s2m_default_params(&prm);
    if (prm.k_neighbors != 5 || prm.max_iter < 1 || prm.max_iter > kMaxIter || !(prm.gate_sq > 0.0)) return S2M_ERR_INVALID_ARG;

    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || prm.device_id < 0 || prm.device_id >= count)
        return S2M_ERR_NO_DEVICE;
    if (hipSetDevice(prm.device_id) != hipSuccess) return S2M_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, prm.device_id) != hipSuccess) return S2M_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return S2M_ERR_NO_DEVICE;   // kernels are built for gfx950 only

    s2m_context* h = new (std::nothrow) s2m_context();
    if (!h) return S2M_ERR_HIP;
    h->prm = prm; h->device = prm.device_id;
    if (const char* e = getenv("S2M_NO_GRAPH")) h->use_graph = !(e[0] == '1');
    if (const char* e = getenv("S2M_NO_FUSE")) h->fuse_solve = !(e[0] == '1');
    if (const char* e = getenv("S2M_DENSITY_RAW")) h->density_raw = atoi(e);
    if (const char* e = getenv("S2M_BIG_BLOCKS")) h->big_blocks = !(e[0] == '0');
    if (const char* e = getenv("S2M_SEGMENT")) h->seg_iters = atoi(e);
    if (const char* e = getenv("S2M_SPLIT")) h->split_mode = atoi(e);
    if (const char* e = getenv("S2M_LOCKSTEP")) h->lockstep = !(e[0] == '0');
    if (const char* e = getenv("S2M_SPLIT_FROM")) h->split_from = std::max(1, atoi(e));
    if (const char* e = getenv("S2M_SEARCH_GRID")) h->search_grid = std::max(1, atoi(e));
    if (const char* e = getenv("S2M_CLOSE_IN_SEARCH")) h->close_in_search = (e[0] == '1');
    if (const char* e = getenv("S2M_BATCH_MINW")) h->batch_minw = atoi(e);
    if (const char* e = getenv("S2M_BATCH_ENTRIES")) h->batch_entries = atoi(e);
    if (const char* e = getenv("S2M_LEAN")) h->lean_certify = !(e[0] == '0');
    h->fuse_max_blocks = kFuseMaxBlocks;
    if (const char* e = getenv("S2M_FUSE_MAX")) h->fuse_max_blocks = atoi(e);

    auto bail = [&](int code) { s2m_destroy(h); return code; };
    if (prm.stream) { h->stream = static_cast<hipStream_t>(prm.stream); h->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return bail(S2M_ERR_HIP);
        h->own_stream = true;
    }
    if (hipEventCreate(&h->ev_a) != hipSuccess || hipEventCreate(&h->ev_b) != hipSuccess ||
        hipEventCreate(&h->ev_c) != hipSuccess || hipEventCreate(&h->ev_d) != hipSuccess ||
        hipEventCreate(&h->ev_a2) != hipSuccess || hipEventCreate(&h->ev_b2) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_up, hipEventDisableTiming) != hipSuccess) return bail(S2M_ERR_HIP);
    static_assert(sizeof(DevState) % 8 == 0, "the trace follows the state block");
    if (hipHostMalloc((void**)&h->h_state, sizeof(DevState) * 2 + sizeof(s2m_iter_trace) * kMaxIter) != hipSuccess) return bail(S2M_ERR_HIP);
    h->h_trace = reinterpret_cast<s2m_iter_trace*>(h->h_state + 2);
    if (hipHostMalloc((void**)&h->h_mm, 64) != hipSuccess) return bail(S2M_ERR_HIP);
    if (hipHostMalloc((void**)&h->h_sc, sizeof(double) * 1220) != hipSuccess) return bail(S2M_ERR_HIP);
    if (ensure(h, h->state, sizeof(DevState) + sizeof(s2m_iter_trace) * kMaxIter) ||      // loop state, then the trace
        ensure(h, h->dctx, sizeof(DevCtx)) || ensure(h, h->mm, 64 + sizeof(uint32_t) * 8 * 1024) ||
        ensure(h, h->sc_bins, sizeof(uint32_t) * 1200) || ensure(h, h->sc_out, sizeof(double) * 1220) ||
        ensure(h, h->q_counts, sizeof(int32_t) * kPolarCells))
        return bail(S2M_ERR_HIP);
    if (hipMemsetAsync(h->q_counts.p, 0, sizeof(int32_t) * kPolarCells, h->stream) != hipSuccess) return bail(S2M_ERR_HIP);

    if (!(h->vox = vox_create())) return bail(S2M_ERR_HIP);
    if (!(h->icp = icp_create())) return bail(S2M_ERR_HIP);

    memset(&h->hctx, 0, sizeof(h->hctx));
    h->hctx.nblocks = kBlocksQuantum;
    h->hctx.wpb = kBlock / 64;
    if (ensure_rows(h, kBlocksQuantum)) return bail(S2M_ERR_HIP);
    h->hctx.state = h->state.as<DevState>();
    h->hctx.trace = reinterpret_cast<s2m_iter_trace*>(h->state.as<DevState>() + 1);
    params_to_ctx(h, prm);
    if (const char* e = getenv("S2M_ABLATE")) h->hctx.ablate = atoi(e);
    h->hctx.tune[0] = 0; h->hctx.tune[1] = 0; h->hctx.tune[2] = 0; h->hctx.tune[3] = 2;
    if (const char* e = getenv("S2M_TUNE")) { (void)sscanf(e, "%d,%d,%d,%d", &h->hctx.tune[0], &h->hctx.tune[1], &h->hctx.tune[2], &h->hctx.tune[3]); h->tune_env = true; }
    h->ctx_dirty = true;
    if (upload_ctx(h) != S2M_OK) return bail(S2M_ERR_HIP);
    *out = h;
    return S2M_OK;
}

int s2m_destroy(s2m_handle h)
{
    if (!h) return S2M_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (hipStream_t st : h->branch_streams) (void)hipStreamSynchronize(st);      // (slot work in flight uses the slots' buffers)
    for (s2m_context* k : h->kids) (void)s2m_destroy(k);
    h->kids.clear();
    if (h->kid_states.p) (void)hipFree(h->kid_states.p);
    if (h->h_kid_states) (void)hipHostFree(h->h_kid_states);
    if (h->state_borrowed) { h->state.p = nullptr; h->h_state = nullptr; }
    for (auto& kv : h->batch_graphs) (void)hipGraphExecDestroy(kv.second);
    for (hipStream_t st : h->branch_streams) (void)hipStreamDestroy(st);
    for (hipEvent_t e : h->branch_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->prep_events) (void)hipEventDestroy(e);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_prep) (void)hipEventDestroy(h->ev_prep);
    for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second);
    for (hipEvent_t e : h->iter_events) (void)hipEventDestroy(e);
    DevBuf* bufs[] = { &h->raw_map, &h->map_sorted, &h->m_counts, &h->m_cell_start, &h->m_cell_of, &h->m_rank_of,
                       &h->raw_scan, &h->qx, &h->qy, &h->qz, &h->qperm, &h->npos, &h->front, &h->cert, &h->aux, &h->plane_cache, &h->plane_alt, &h->npos_alt, &h->chunk_parts, &h->chunk_factor, &h->wave_table, &h->n_waves, &h->q_counts, &h->q_cell_start, &h->q_cell_of, &h->q_block_hist,
                       &h->q_rank_of, &h->block_sums, &h->partials, &h->state, &h->dctx, &h->mm,
                       &h->dbg_idx5, &h->dbg_d2, &h->dbg_flag, &h->dbg_coeff, &h->dbg_clk, &h->sc_bins, &h->sc_out,
                       &h->vox_in, &h->vox_out, &h->frames_xf, &h->scan_ds, &h->map_ds,
                       &h->sc_store_desc, &h->sc_store_ring, &h->sc_store_sector, &h->sc_cand, &h->sc_res };
    for (DevBuf* b : bufs) if (b->p) (void)hipFree(b->p);
    vox_destroy(h->vox);
    icp_destroy(h->icp);
    if (h->icp_src.p) (void)hipFree(h->icp_src.p);
    if (h->icp_tgt.p) (void)hipFree(h->icp_tgt.p);
    if (h->h_state) (void)hipHostFree(h->h_state);
    if (h->h_mm) (void)hipHostFree(h->h_mm);
    if (h->h_sc) (void)hipHostFree(h->h_sc);
    if (h->ev_a) (void)hipEventDestroy(h->ev_a);
    if (h->ev_b) (void)hipEventDestroy(h->ev_b);
    if (h->ev_a2) (void)hipEventDestroy(h->ev_a2);
    if (h->ev_b2) (void)hipEventDestroy(h->ev_b2);
    if (h->ev_c) (void)hipEventDestroy(h->ev_c);
    if (h->ev_d) (void)hipEventDestroy(h->ev_d);
    if (h->ev_up) (void)hipEventDestroy(h->ev_up);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return S2M_OK;
}

const char* s2m_last_error(s2m_handle h) { return h ? h->err.c_str() : "null handle"; }

int s2m_get_params(s2m_handle h, s2m_params* out)
{
    if (!h || !out) return S2M_ERR_INVALID_ARG;
    *out = h->prm;
    return S2M_OK;
}

int s2m_set_params(s2m_handle h, const s2m_params* p)
{
    if (!h || !p) return S2M_ERR_INVALID_ARG;
    if (p->struct_size != sizeof(s2m_params)) return fail(h, S2M_ERR_INVALID_ARG, "s2m_params.struct_size does not match this library");
    if (p->device_id != h->prm.device_id || p->stream != h->prm.stream || p->k_neighbors != h->prm.k_neighbors ||
        p->gate_sq != h->prm.gate_sq)
        return fail(h, S2M_ERR_INVALID_ARG, "device_id, stream, k_neighbors and gate_sq are fixed at s2m_create (the search grid is built for the gate)");
    if (p->max_iter < 1 || p->max_iter > kMaxIter) return fail(h, S2M_ERR_INVALID_ARG, "max_iter out of range");
    if (h->opt_pending) return fail(h, S2M_ERR_INVALID_ARG, "an optimize launch is pending: collect it first");
    if (p->max_iter != h->prm.max_iter) {               // the captured loops hold max_iter launches
        S2M_HIP(h, hipSetDevice(h->device));
        S2M_HIP(h, hipStreamSynchronize(h->stream));
        for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second);
        h->graphs.clear();
        for (auto& kv : h->batch_graphs) (void)hipGraphExecDestroy(kv.second);     // (every branch of a batch graph is such a loop)
        h->batch_graphs.clear();
    }
    h->prm = *p;
    params_to_ctx(h, h->prm);                           // uploaded by the next call that launches anything
    return S2M_OK;
}

int s2m_set_map(s2m_handle h, const void* pts, size_t n, size_t stride_bytes)
{ return set_map_impl(h, pts, n, stride_bytes, false); }
int s2m_set_map_device(s2m_handle h, const void* d_pts, size_t n, size_t stride_bytes)
{ return set_map_impl(h, d_pts, n, stride_bytes, true); }
int s2m_set_scan(s2m_handle h, const void* pts, size_t n, size_t stride_bytes)
{ return set_scan_impl(h, pts, n, stride_bytes, false); }
int s2m_set_scan_device(s2m_handle h, const void* d_pts, size_t n, size_t stride_bytes)
{ return set_scan_impl(h, d_pts, n, stride_bytes, true); }

int s2m_optimize_launch(s2m_handle h, const float pose[6])
{
    if (!h || !pose) return S2M_ERR_INVALID_ARG;
    if (!h->have_scan) return fail(h, S2M_ERR_NO_SCAN, "s2m_set_scan has not been called");
    S2M_HIP(h, hipSetDevice(h->device));
    memcpy(h->pending_pose_in, pose, 24);
    h->opt_pending = true;
    h->pending_skipped = 0;
    if (h->n_m == 0) { h->pending_skipped = 1; return S2M_OK; }                    // :1297
    if ((int)h->n_q <= h->prm.min_feats) { h->pending_skipped = 2; return S2M_OK; } // :1300
    int rc;
    if ((rc = push_state(h, pose))) return rc;             // (with the DevCtx block when that changed)
    h->seg_pending = h->prm.early_exit && h->seg_iters > 1 && h->seg_iters < h->prm.max_iter;
    if ((rc = launch_loop(h, h->seg_pending ? 1 : 0))) return rc;        // (state + trace come back with it)
    return S2M_OK;
}

int s2m_optimize_collect(s2m_handle h, float pose[6], const s2m_imu_init* imu, s2m_result* out)
{
    if (!h || !pose) return S2M_ERR_INVALID_ARG;
    if (!h->opt_pending) return fail(h, S2M_ERR_INVALID_ARG, "no optimize launch pending");
    h->opt_pending = false;
    s2m_result r;
    memset(&r, 0, sizeof(r));
    r.skipped = h->pending_skipped;
    h->last_trace_n = 0;
    float t[6];
    memcpy(t, h->pending_pose_in, 24);
    if (!r.skipped) {
        S2M_HIP(h, hipSetDevice(h->device));
        S2M_HIP(h, hipStreamSynchronize(h->stream));
        if (h->parent && !h->slot_mode) h->t_optimize_ms = h->parent->t_optimize_ms;      // a slot of a batch: the batch was timed as a whole
        else S2M_HIP(h, hipEventElapsedTime(&h->t_optimize_ms, h->ev_a, h->ev_b));
        if (h->seg_pending && !h->h_state[1].done) {
            // the first range did not converge: the rest of the loop
            int rc2 = launch_loop(h, 2);
            if (rc2) return rc2;
            S2M_HIP(h, hipStreamSynchronize(h->stream));
            float t2 = 0.0f;
            S2M_HIP(h, hipEventElapsedTime(&t2, h->ev_a2, h->ev_b2));
            h->t_optimize_ms += t2;
        }
        h->seg_pending = false;
        const DevState& s = h->h_state[1];
        memcpy(t, s.pose, 24);
        r.iters_run = s.iters_run; r.converged = s.converged; r.is_degenerate = s.isDegenerate;
        r.n_sel_last = s.n_sel_last;
        h->persist_degenerate = s.isDegenerate;
        memcpy(h->persist_matP, s.matP, sizeof(s.matP));
        // trace: executed iterations; a loop that stalled at iteration k (fewer than min_corr correspondences,
        // pose unchanged, :1178-1180) repeats that no-op record for the iterations the reference would still run
        int n_exec = s.iters_run, stall_at = -1;
        for (int i = 0; i < n_exec && i < kMaxIter; i++) {
            if (stall_at >= 0) { h->last_trace[i] = h->last_trace[stall_at]; continue; }
            h->last_trace[i] = h->h_trace[i];
            if (s.stalled && !h->h_trace[i].stepped) stall_at = i;
        }
        h->last_trace_n = n_exec < kMaxIter ? n_exec : kMaxIter;
        host_transform_update(h->prm, imu, t, r.affine);                           // :1317
    } else {
        host_pose_to_transform(t, r.affine, nullptr);
        r.is_degenerate = h->persist_degenerate;
    }
    memcpy(r.pose, t, 24);
    memcpy(pose, t, 24);
    if (out) *out = r;
    return S2M_OK;
}

int s2m_optimize_resident(s2m_handle h, float pose[6], const s2m_imu_init* imu, s2m_result* out)
{
    int rc = s2m_optimize_launch(h, pose);
    if (rc) return rc;
    return s2m_optimize_collect(h, pose, imu, out);
}

int s2m_optimize(s2m_handle h, const void* scan, size_t n, size_t stride_bytes, float pose[6],
                 const s2m_imu_init* imu, s2m_result* out)
{
    int rc = s2m_set_scan(h, scan, n, stride_bytes);
    if (rc) return rc;
    return s2m_optimize_resident(h, pose, imu, out);
}

// ---- a batch of scans against one resident map -------------------------------------------------------
namespace {

// field by field (the struct has padding): everything a slot has to take over from its parent
bool same_params_but_stream(const s2m_params& a, const s2m_params& b)
{
    return a.struct_size == b.struct_size && a.device_id == b.device_id && a.k_neighbors == b.k_neighbors && a.gate_sq == b.gate_sq &&
           a.plane_tol == b.plane_tol && a.weight_scale == b.weight_scale && a.weight_min == b.weight_min && a.min_corr == b.min_corr &&
           a.min_feats == b.min_feats && a.max_iter == b.max_iter && a.eig_thresh == b.eig_thresh && a.conv_deg == b.conv_deg &&
           a.conv_cm == b.conv_cm && a.z_tol == b.z_tol && a.rot_tol == b.rot_tol && a.imu_type == b.imu_type &&
           a.imu_rpy_weight == b.imu_rpy_weight && a.early_exit == b.early_exit;
}

// child b borrows the parent's map index (no copy) and its per-scan certificates are void when the index changed
int adopt_map(s2m_context* h, s2m_context* k)
{
    if (k->adopted_epoch == h->map_epoch) return S2M_OK;
    k->n_m = h->n_m;
    k->hctx.n_m = h->hctx.n_m;
    k->hctx.g = h->hctx.g;
    k->hctx.map_sorted = h->hctx.map_sorted;
    k->hctx.cell_start = h->hctx.cell_start;
    k->ctx_dirty = true;
    k->adopted_epoch = h->map_epoch;
    if (k->have_scan && k->n_q > 0 && k->cert.p) {
        S2M_HIP(k, hipMemsetAsync(k->cert.p, 0, sizeof(float4) * k->n_q, k->stream));
        S2M_HIP(k, hipMemsetAsync(k->aux.p, 0, sizeof(int4) * k->n_q, k->stream));
    }
    return S2M_OK;
}

int ensure_kids(s2m_context* h, int n)
{
    while ((int)h->kids.size() < n) {
        s2m_params p = h->prm;
        p.stream = h->stream;                              // everything outside the captured graph is ordered on the parent's stream
        s2m_context* k = nullptr;
        const int rc = s2m_create(&p, &k);
        if (rc) return fail(h, rc, "batch: could not create a scan slot");
        k->parent = h;
        k->hctx.ablate = h->hctx.ablate;
        memcpy(k->hctx.tune, h->hctx.tune, sizeof(h->hctx.tune));
        {   // the slot's loop state and trace live in the parent's block
            constexpr size_t kStride = sizeof(DevState) + sizeof(s2m_iter_trace) * kMaxIter;
            if (!h->kid_states.p) {
                if (ensure(h, h->kid_states, kStride * kMaxSlots) != S2M_OK ||
                    hipHostMalloc((void**)&h->h_kid_states, sizeof(DevState) + kStride * kMaxSlots) != hipSuccess) {
                    (void)s2m_destroy(k);
                    return fail(h, S2M_ERR_HIP, "batch: state block allocation failed");
                }
            }
            const size_t slot = h->kids.size();
            (void)hipFree(k->state.p);
            (void)hipHostFree(k->h_state);
            k->state.p = h->kid_states.as<unsigned char>() + kStride * slot; k->state.cap = kStride;
            k->h_state = reinterpret_cast<DevState*>(h->h_kid_states + kStride * slot);      // [1] = the download area of this slot
            k->h_trace = reinterpret_cast<s2m_iter_trace*>(k->h_state + 2);
            k->state_borrowed = true;
            k->hctx.state = k->state.as<DevState>();
            k->hctx.trace = reinterpret_cast<s2m_iter_trace*>(k->state.as<DevState>() + 1);
            k->ctx_dirty = true;
        }
        h->kids.push_back(k);
        hipStream_t st = nullptr;
        hipEvent_t ev = nullptr, ev2 = nullptr;
        // The streams of neighbouring slots get different priorities.  HIP deals a process's streams of one priority onto a small
        // pool of hardware queues in creation order (GPU_MAX_HW_QUEUES, default 4), and two streams that land on one queue run
        // strictly one after the other: with other streams alive in the process (a ROS node has them) the two slots of a stream
        // of scans could end up sharing a queue and lose the overlap of one's preparation with the other's loop.  Streams of
        // different priority never share a hardware queue, whatever else the process has created.
        int prio_least = 0, prio_greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        const int prio = (h->kids.size() & 1) ? prio_greatest : prio_least;
        if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ev2, hipEventDisableTiming) != hipSuccess)
            return fail(h, S2M_ERR_HIP, "batch: stream / event creation failed");
        h->branch_streams.push_back(st);
        h->branch_events.push_back(ev);
        h->prep_events.push_back(ev2);
        h->prep_pending.push_back(0);
    }
    if (!h->ev_fork && hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess) return fail(h, S2M_ERR_HIP, "batch: event creation failed");
    if (!h->ev_prep && hipEventCreateWithFlags(&h->ev_prep, hipEventDisableTiming) != hipSuccess) return fail(h, S2M_ERR_HIP, "batch: event creation failed");
    return S2M_OK;
}

// One graph for the whole batch.  Lockstep (default): the slots' loops advance together, every launch of the loop has one
// grid row per slot (slots of different workgroup shape form groups, one loop per group on a branch of its own).  Otherwise
// (S2M_LOCKSTEP=0, A/B measurements): a fork, one branch per scan with that scan's loop, a join.
// part 0: the whole loop; 1: launches 0 .. seg-1; 2: launches seg .. max_iter-1 (as get_graph)
int get_batch_graph(s2m_context* h, const std::vector<int>& live, int part, hipGraphExec_t* out)
{
    std::vector<int> key;
    key.push_back(part); key.push_back(part ? h->seg_iters : 0);
    for (int b : live) { key.push_back(b); key.push_back(h->kids[(size_t)b]->hctx.table_cap); key.push_back(h->kids[(size_t)b]->hctx.wpb); }
    auto it = h->batch_graphs.find(key);
    if (it != h->batch_graphs.end()) { *out = it->second; return S2M_OK; }
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    // the loops to run: (shape, stream to run it on)
    std::vector<LoopShape> loops;
    if (h->lockstep) {
        for (int wpb : { kBlock / 64, kBigWaves }) {
            LoopShape sh;
            memset(&sh.tbl, 0, sizeof(sh.tbl));
            sh.nslots = 0; sh.wpb = wpb; sh.batch = true;
            for (int b : live) {
                s2m_context* k = h->kids[(size_t)b];
                if (k->hctx.wpb != wpb) continue;
                sh.tbl.ctx[sh.nslots] = k->dctx.as<DevCtx>();
                sh.tbl.st[sh.nslots] = k->state.as<DevState>();
                sh.nslots++;
                sh.nblocks = std::max(sh.nblocks, k->hctx.nblocks);
                sh.table_cap = std::max(sh.table_cap, k->hctx.table_cap);
            }
            sh.split = h->split_mode == 1;         // (default: decided in enqueue_loop - split where the loop is not fused)
            sh.split_auto = h->split_mode == 2 || h->split_mode < 0;   // (default: certify (lean) + search from launch split_from on, where k_finalize closes the iterations)
            if (sh.nslots) loops.push_back(sh);
        }
    } else
        for (int b : live) loops.push_back(shape_of(h->kids[(size_t)b]));
    S2M_HIP(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    bool ok = hipEventRecord(h->ev_fork, h->stream) == hipSuccess;
    hipStream_t keep = h->stream;
    for (size_t j = 0; j < loops.size(); j++) {
        hipStream_t br = h->branch_streams[j];
        ok = ok && hipStreamWaitEvent(br, h->ev_fork, 0) == hipSuccess;
        h->stream = br;                                       // (enqueue_loop launches on the handle's stream; settings are the parent's)
        enqueue_loop(h, loops[j], nullptr, false, part == 2 ? h->seg_iters : 0, part == 1 ? h->seg_iters : -1);
        h->stream = keep;
        ok = ok && hipEventRecord(h->branch_events[j], br) == hipSuccess;
        ok = ok && hipStreamWaitEvent(h->stream, h->branch_events[j], 0) == hipSuccess;
    }
    hipError_t e = hipStreamEndCapture(h->stream, &graph);
    if (!ok || e != hipSuccess || !graph) return fail(h, S2M_ERR_HIP, "batch: stream capture failed", e);
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return fail(h, S2M_ERR_HIP, "batch: hipGraphInstantiate", e);
    h->batch_graphs[key] = exec;
    *out = exec;
    return S2M_OK;
}

}  // namespace

int s2m_batch_set_scan(s2m_handle h, int slot, const void* pts, size_t n, size_t stride_bytes, int on_device)
{
    if (!h || slot < 0 || slot >= 64) return S2M_ERR_INVALID_ARG;
    S2M_HIP(h, hipSetDevice(h->device));
    int rc = ensure_kids(h, slot + 1);
    if (rc) return rc;
    s2m_context* k = h->kids[(size_t)slot];
    if ((rc = adopt_map(h, k))) return fail(h, rc, k->err.c_str());
    // The ordering of one slot's scan does not depend on the others': it runs on the slot's own stream, behind everything
    // issued on the handle's stream so far (the previous batch read the slot's buffers), and the batch launch waits for it.
    hipStream_t br = h->branch_streams[(size_t)slot];
    S2M_HIP(h, hipEventRecord(h->ev_prep, h->stream));
    S2M_HIP(h, hipStreamWaitEvent(br, h->ev_prep, 0));
    k->stream = br;
    rc = set_scan_impl(k, pts, n, stride_bytes, on_device != 0);
    k->stream = h->stream;
    if (rc) { (void)hipStreamSynchronize(br); return fail(h, rc, k->err.c_str()); }
    S2M_HIP(h, hipEventRecord(h->prep_events[(size_t)slot], br));
    h->prep_pending[(size_t)slot] = 1;
    return S2M_OK;
}

int s2m_batch_set_scans(s2m_handle h, int n_scans, const void* const* scans, const size_t* sizes, size_t stride_bytes, int on_device)
{
    if (!h || n_scans < 1 || n_scans > kMaxSlots || !scans || !sizes) return S2M_ERR_INVALID_ARG;
    S2M_HIP(h, hipSetDevice(h->device));
    int rc = ensure_kids(h, n_scans);
    if (rc) return rc;
    // Work still queued on a slot's own stream (a preparation from s2m_batch_set_scan, a loop from s2m_slot_optimize_launch)
    // uses the buffers this call rewrites: the handle's stream waits for it first.
    for (int b = 0; b < n_scans; b++) {
        S2M_HIP(h, hipEventRecord(h->prep_events[(size_t)b], h->branch_streams[(size_t)b]));
        S2M_HIP(h, hipStreamWaitEvent(h->stream, h->prep_events[(size_t)b], 0));
        h->prep_pending[(size_t)b] = 0;
    }
    for (int b0 = 0; b0 < n_scans; b0 += kPrepSlots) {
        PrepTable t;
        const int nb = std::min(kPrepSlots, n_scans - b0);
        for (int j = 0; j < nb; j++) {
            s2m_context* k = h->kids[(size_t)(b0 + j)];
            if ((rc = adopt_map(h, k)) == S2M_OK) rc = scan_slot_prepare(k, scans[b0 + j], sizes[b0 + j], stride_bytes, on_device != 0, &t.s[j]);
            if (rc) {
                // no slot of this call holds a scan after a failure: the groups before this one were ordered, but a batch
                // of which some scans are missing is of no use to the caller
                for (int q = 0; q <= b0 + j; q++) h->kids[(size_t)q]->have_scan = false;
                (void)hipStreamSynchronize(h->stream);
                return fail(h, rc, k->err.c_str());
            }
            k->t_set_scan_ms = 0; k->scan_timing_pending = false;
        }
        launch_scan_prep(h->stream, t, nb);
        S2M_HIP(h, hipGetLastError());
    }
    if (!on_device) S2M_HIP(h, hipStreamSynchronize(h->stream));       // the callers' buffers are free again
    return S2M_OK;
}

// ---- a stream of scans: slot k's preparation and loop run on slot k's own stream, so that the ordering of scan i+1 (slot B)
// overlaps the LM loop of scan i (slot A) - the reference does the two strictly one after the other (:257-265) ----------
int s2m_slot_set_scan(s2m_handle h, int slot, const void* pts, size_t n, size_t stride_bytes, int on_device)
{ return s2m_batch_set_scan(h, slot, pts, n, stride_bytes, on_device); }

int s2m_slot_optimize_launch(s2m_handle h, int slot, const float pose[6])
{
    if (!h || slot < 0 || slot >= (int)h->kids.size() || !pose) return S2M_ERR_INVALID_ARG;
    s2m_context* k = h->kids[(size_t)slot];
    if (!k->have_scan) return fail(h, S2M_ERR_NO_SCAN, "s2m_slot_set_scan has not been called for this slot");
    S2M_HIP(h, hipSetDevice(h->device));
    int rc;
    if (!same_params_but_stream(k->prm, h->prm)) {
        s2m_params p = h->prm;
        p.stream = k->prm.stream;
        if ((rc = s2m_set_params(k, &p))) return fail(h, rc, k->err.c_str());
    }
    h->prep_pending[(size_t)slot] = 0;                       // (the preparation is ahead of the loop on the same stream)
    k->stream = h->branch_streams[(size_t)slot];
    k->slot_mode = true;
    rc = adopt_map(h, k);
    if (!rc) rc = s2m_optimize_launch(k, pose);
    k->stream = h->stream;
    return rc ? fail(h, rc, k->err.c_str()) : S2M_OK;
}

int s2m_slot_optimize_collect(s2m_handle h, int slot, float pose[6], const s2m_imu_init* imu, s2m_result* out)
{
    if (!h || slot < 0 || slot >= (int)h->kids.size() || !pose) return S2M_ERR_INVALID_ARG;
    s2m_context* k = h->kids[(size_t)slot];
    k->stream = h->branch_streams[(size_t)slot];
    const int rc = s2m_optimize_collect(k, pose, imu, out);
    k->stream = h->stream;
    k->slot_mode = false;
    return rc ? fail(h, rc, k->err.c_str()) : S2M_OK;
}

int s2m_optimize_batch_launch(s2m_handle h, int n_scans, const float* poses)
{
    if (!h || n_scans < 1 || n_scans > 64 || !poses) return S2M_ERR_INVALID_ARG;
    if ((int)h->kids.size() < n_scans) return fail(h, S2M_ERR_NO_SCAN, "s2m_batch_set_scan has not been called for every slot");
    S2M_HIP(h, hipSetDevice(h->device));
    std::vector<int> live;
    int rc;
    for (size_t b = 0; b < h->prep_pending.size(); b++)
        if (h->prep_pending[b]) { S2M_HIP(h, hipStreamWaitEvent(h->stream, h->prep_events[b], 0)); h->prep_pending[b] = 0; }
    for (int b = 0; b < n_scans; b++) {
        s2m_context* k = h->kids[(size_t)b];
        if (!k->have_scan) return fail(h, S2M_ERR_NO_SCAN, "s2m_batch_set_scan has not been called for every slot");
        if (!same_params_but_stream(k->prm, h->prm)) {     // any parameter changed on the parent since the slot was made
            s2m_params p = h->prm;
            p.stream = k->prm.stream;
            if ((rc = s2m_set_params(k, &p))) return fail(h, rc, k->err.c_str());
        }
        if ((rc = adopt_map(h, k))) return fail(h, rc, k->err.c_str());
        memcpy(k->pending_pose_in, poses + 6 * (size_t)b, 24);
        k->opt_pending = true;
        k->pending_skipped = 0;
        if (k->n_m == 0) { k->pending_skipped = 1; continue; }                            // :1297
        if ((int)k->n_q <= k->prm.min_feats) { k->pending_skipped = 2; continue; }         // :1300
        live.push_back(b);
    }
    {   // DevCtx blocks that changed and the loop state every live slot starts from: one launch per kCtxSlots / kInitSlots slots
        CtxTable ct; int nc = 0;
        auto flush_ctx = [&]() { if (nc) { for (int j = nc; j < kCtxSlots; j++) ct.dst[j] = nullptr; hipLaunchKernelGGL(k_set_ctxs, dim3(nc), dim3(64), 0, h->stream, ct); nc = 0; } };
        for (int b = 0; b < n_scans; b++) {
            s2m_context* k = h->kids[(size_t)b];
            if (!k->ctx_dirty) continue;
            ct.dst[nc] = k->dctx.as<DevCtx>(); ct.v[nc] = k->hctx; nc++;
            k->ctx_dirty = false;
            if (nc == kCtxSlots) flush_ctx();
        }
        flush_ctx();
        StateInitTable it; int ni = 0;
        auto flush_init = [&]() { if (ni) { for (int j = ni; j < kInitSlots; j++) it.s[j].dst = nullptr; hipLaunchKernelGGL(k_init_states, dim3(ni), dim3(256), 0, h->stream, it); ni = 0; } };
        for (int b : live) {
            s2m_context* k = h->kids[(size_t)b];
            DevState s0;
            fill_state(k, &s0, poses + 6 * (size_t)b);
            StateInit& si = it.s[ni++];
            si.dst = k->state.as<DevState>(); si.n_waves = k->n_waves.as<int32_t>();
            memcpy(si.pose, s0.pose, sizeof(si.pose)); memcpy(si.T, s0.T, sizeof(si.T)); memcpy(si.sc, s0.sc, sizeof(si.sc));
            memcpy(si.matP, s0.matP, sizeof(si.matP)); si.isDegenerate = s0.isDegenerate;
            if (ni == kInitSlots) flush_init();
        }
        flush_init();
        S2M_HIP(h, hipGetLastError());
    }
    // With early exit on the loops are issued as launches 0 .. seg-1 and, only if some slot has not converged by then, the rest
    // (s2m_optimize_batch_collect looks at the slots' `done` flags, which come back with the states anyway): the launches behind
    // the convergence of every slot - idle, but a kernel boundary and a k_finalize each - were a fifth of an early-exit batch.
    h->batch_seg_pending = h->prm.early_exit && h->seg_iters > 1 && h->seg_iters < h->prm.max_iter;
    h->batch_live = live;
    S2M_HIP(h, hipEventRecord(h->ev_a, h->stream));
    if (!live.empty()) {
        hipGraphExec_t exec = nullptr;
        if ((rc = get_batch_graph(h, live, h->batch_seg_pending ? 1 : 0, &exec))) return rc;
        S2M_HIP(h, hipGraphLaunch(exec, h->stream));
    }
    S2M_HIP(h, hipEventRecord(h->ev_b, h->stream));
    for (int b : live) h->kids[(size_t)b]->hctx.density_pending = 0;
    if (!live.empty()) {
        // state + trace of every slot up to the last live one, in one copy (the slots' blocks are contiguous)
        constexpr size_t kStride = sizeof(DevState) + sizeof(s2m_iter_trace) * kMaxIter;
        const size_t upto = (size_t)live.back() + 1;
        S2M_HIP(h, hipMemcpyAsync(h->h_kid_states + sizeof(DevState), h->kid_states.p, kStride * upto, hipMemcpyDeviceToHost, h->stream));
    }
    return S2M_OK;
}

int s2m_optimize_batch_collect(s2m_handle h, int n_scans, float* poses, const s2m_imu_init* imu, s2m_result* out)
{
    if (!h || n_scans < 1 || (int)h->kids.size() < n_scans || !poses) return S2M_ERR_INVALID_ARG;
    S2M_HIP(h, hipSetDevice(h->device));
    S2M_HIP(h, hipStreamSynchronize(h->stream));
    S2M_HIP(h, hipEventElapsedTime(&h->t_optimize_ms, h->ev_a, h->ev_b));
    if (h->batch_seg_pending && !h->batch_live.empty()) {
        h->batch_seg_pending = false;
        bool all_done = true;
        for (int b : h->batch_live) all_done = all_done && h->kids[(size_t)b]->h_state[1].done != 0;
        if (!all_done) {
            // some slot needs more than the first range: the rest of the loop for all of them (a converged slot's launches return at once)
            hipGraphExec_t exec = nullptr;
            int rc2 = get_batch_graph(h, h->batch_live, 2, &exec);
            if (rc2) return rc2;
            S2M_HIP(h, hipEventRecord(h->ev_a2, h->stream));
            S2M_HIP(h, hipGraphLaunch(exec, h->stream));
            S2M_HIP(h, hipEventRecord(h->ev_b2, h->stream));
            constexpr size_t kStride = sizeof(DevState) + sizeof(s2m_iter_trace) * kMaxIter;
            const size_t upto = (size_t)h->batch_live.back() + 1;
            S2M_HIP(h, hipMemcpyAsync(h->h_kid_states + sizeof(DevState), h->kid_states.p, kStride * upto, hipMemcpyDeviceToHost, h->stream));
            S2M_HIP(h, hipStreamSynchronize(h->stream));
            float t2 = 0.0f;
            S2M_HIP(h, hipEventElapsedTime(&t2, h->ev_a2, h->ev_b2));
            h->t_optimize_ms += t2;
        }
    }
    for (int b = 0; b < n_scans; b++) {
        s2m_context* k = h->kids[(size_t)b];
        const int rc = s2m_optimize_collect(k, poses + 6 * (size_t)b, imu ? imu + b : nullptr, out ? out + b : nullptr);
        if (rc) return fail(h, rc, k->err.c_str());
    }
    return S2M_OK;
}

int s2m_optimize_batch(s2m_handle h, int n_scans, const void* const* scans, const size_t* sizes, size_t stride_bytes,
                       float* poses, const s2m_imu_init* imu, s2m_result* out)
{
    if (!h || n_scans < 1 || n_scans > 64 || !scans || !sizes || !poses) return S2M_ERR_INVALID_ARG;
    int rc = s2m_batch_set_scans(h, n_scans, scans, sizes, stride_bytes, 0);
    if (rc) return rc;
    rc = s2m_optimize_batch_launch(h, n_scans, poses);
    if (rc) return rc;
    return s2m_optimize_batch_collect(h, n_scans, poses, imu, out);
}

int s2m_batch_get_trace(s2m_handle h, int slot, s2m_iter_trace* out, int cap)
{
    if (!h || slot < 0 || slot >= (int)h->kids.size()) return S2M_ERR_INVALID_ARG;
    return s2m_get_trace(h->kids[(size_t)slot], out, cap);
}

int s2m_get_trace(s2m_handle h, s2m_iter_trace* out, int cap)
{
    if (!h || (!out && cap > 0)) return S2M_ERR_INVALID_ARG;
    int n = h->last_trace_n < cap ? h->last_trace_n : cap;
    for (int i = 0; i < n; i++) out[i] = h->last_trace[i];
    return n;
}

int s2m_surf_optimization(s2m_handle h, const float pose[6], int32_t* idx5, float* d2_5, uint8_t* flag, float* coeff4)
{
    if (!h || !pose) return S2M_ERR_INVALID_ARG;
    if (!h->have_scan) return fail(h, S2M_ERR_NO_SCAN, "s2m_set_scan has not been called");
    S2M_HIP(h, hipSetDevice(h->device));
    const size_t n = h->n_q;
    if (n == 0) return S2M_OK;
    if (h->n_m == 0) {          // no map: nothing is ever gated
        if (idx5) for (size_t i = 0; i < 5 * n; i++) idx5[i] = -1;
        if (d2_5) for (size_t i = 0; i < 5 * n; i++) d2_5[i] = INFINITY;
        if (flag) memset(flag, 0, n);
        if (coeff4) memset(coeff4, 0, sizeof(float) * 4 * n);
        return S2M_OK;
    }
    int rc;
    if ((rc = ensure(h, h->dbg_idx5, sizeof(int32_t) * 5 * n))) return rc;
    if ((rc = ensure(h, h->dbg_d2, sizeof(float) * 5 * n))) return rc;
    if ((rc = ensure(h, h->dbg_flag, n))) return rc;
    if ((rc = ensure(h, h->dbg_coeff, sizeof(float) * 4 * n))) return rc;
    h->hctx.dbg_idx5 = h->dbg_idx5.as<int32_t>(); h->hctx.dbg_d2 = h->dbg_d2.as<float>();
    h->hctx.dbg_flag = h->dbg_flag.as<uint8_t>(); h->hctx.dbg_coeff = h->dbg_coeff.as<float>();
    h->ctx_dirty = true;
    if ((rc = upload_ctx(h))) return rc;
    if ((rc = push_state(h, pose))) return rc;
    launch_fused(h, shape_of(h), true, 0, 0);
    S2M_HIP(h, hipGetLastError());
    if (idx5) S2M_HIP(h, hipMemcpyAsync(idx5, h->dbg_idx5.p, sizeof(int32_t) * 5 * n, hipMemcpyDeviceToHost, h->stream));
    if (d2_5) S2M_HIP(h, hipMemcpyAsync(d2_5, h->dbg_d2.p, sizeof(float) * 5 * n, hipMemcpyDeviceToHost, h->stream));
    if (flag) S2M_HIP(h, hipMemcpyAsync(flag, h->dbg_flag.p, n, hipMemcpyDeviceToHost, h->stream));
    if (coeff4) S2M_HIP(h, hipMemcpyAsync(coeff4, h->dbg_coeff.p, sizeof(float) * 4 * n, hipMemcpyDeviceToHost, h->stream));
    S2M_HIP(h, hipStreamSynchronize(h->stream));
    h->hctx.dbg_idx5 = nullptr; h->hctx.dbg_d2 = nullptr; h->hctx.dbg_flag = nullptr; h->hctx.dbg_coeff = nullptr;
    h->ctx_dirty = true;
    return upload_ctx(h);
}

int s2m_debug_wave_profile(s2m_handle h, const float pose[6], int launches, uint64_t* out, size_t cap_waves)
{
    if (!h || !pose || !out || launches == 0) return S2M_ERR_INVALID_ARG;
    if (!h->have_scan || h->n_m == 0 || h->n_q == 0) return fail(h, S2M_ERR_NO_SCAN, "needs a resident scan and map");
    S2M_HIP(h, hipSetDevice(h->device));
    const size_t nwaves = (size_t)h->hctx.nblocks * (size_t)h->hctx.wpb;   // one record per wave of the grid
    int rc;
    if ((rc = ensure(h, h->dbg_clk, sizeof(uint64_t) * kProfWords * nwaves))) return rc;
    S2M_HIP(h, hipMemsetAsync(h->dbg_clk.p, 0, sizeof(uint64_t) * kProfWords * nwaves, h->stream));
    h->hctx.dbg_clk = h->dbg_clk.as<unsigned long long>();
    h->ctx_dirty = true;
    if ((rc = upload_ctx(h))) return rc;
    if ((rc = push_state(h, pose))) return rc;
    if (launches < 0) {
        // a real loop as enqueue_loop issues it (density re-split, R0 F0 R1 R2' ...); the recorded launch is number
        // N = -launches - 1 (N = 0: the first launch of a scan), closing the iteration before it when the loop is fused
        const LoopShape sh = shape_of(h);
        const int nblocks = h->hctx.nblocks, N = -launches - 1;
        const bool fuse = h->fuse_solve && nblocks <= h->fuse_max_blocks;
        if (h->density_raw > 0) launch_density(h, sh);
        h->hctx.density_pending = 0;                      // cleared on the device by the kernel: keep the host copy in step
        for (int L = 0; L < N; L++) {
            launch_iteration(h, sh, L, (fuse && L >= 2) ? 1 : 0);
            if (!fuse || L == 0) launch_finalize(h, sh, L, 0);
        }
        launch_fused(h, sh, true, N, (fuse && N >= 2) ? 1 : 0);
    }
    for (int rep = 0; rep < launches; rep++)
        launch_fused(h, shape_of(h), true, 0, 0);
    S2M_HIP(h, hipGetLastError());
    const size_t n = nwaves < cap_waves ? nwaves : cap_waves;
    S2M_HIP(h, hipMemcpyAsync(out, h->dbg_clk.p, sizeof(uint64_t) * kProfWords * n, hipMemcpyDeviceToHost, h->stream));
    S2M_HIP(h, hipStreamSynchronize(h->stream));
    h->hctx.dbg_clk = nullptr;
    h->ctx_dirty = true;
    if ((rc = upload_ctx(h))) return rc;
    return (int)n;
}

int s2m_debug_device_trig(s2m_handle h, const float* x, size_t n, float* s, float* c, float* a)
{
    if (!h || (n > 0 && (!x || !s || !c)) || n > (size_t)0x3fffffff) return S2M_ERR_INVALID_ARG;
    if (n == 0) return S2M_OK;
    S2M_HIP(h, hipSetDevice(h->device));
    int rc = ensure(h, h->vox_in, sizeof(float) * 4 * n);
    if (rc) return rc;
    float* d = h->vox_in.as<float>();
    S2M_HIP(h, hipMemcpyAsync(d, x, sizeof(float) * n, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_debug_sincos, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, (const float*)d, (int)n, d + n, d + 2 * n,
                       a ? d + 3 * n : (float*)nullptr);
    S2M_HIP(h, hipGetLastError());
    S2M_HIP(h, hipMemcpyAsync(s, d + n, sizeof(float) * n, hipMemcpyDeviceToHost, h->stream));
    S2M_HIP(h, hipMemcpyAsync(c, d + 2 * n, sizeof(float) * n, hipMemcpyDeviceToHost, h->stream));
    if (a) S2M_HIP(h, hipMemcpyAsync(a, d + 3 * n, sizeof(float) * n, hipMemcpyDeviceToHost, h->stream));
    S2M_HIP(h, hipStreamSynchronize(h->stream));
    return S2M_OK;
}

int s2m_normal_eq(s2m_handle h, const float pose[6], float AtA[36], float AtB[6], int32_t* n_sel)
{
    if (!h || !pose) return S2M_ERR_INVALID_ARG;
    if (!h->have_scan) return fail(h, S2M_ERR_NO_SCAN, "s2m_set_scan has not been called");
    S2M_HIP(h, hipSetDevice(h->device));
    if (h->n_m == 0 || h->n_q == 0) {
        if (AtA) memset(AtA, 0, sizeof(float) * 36);
        if (AtB) memset(AtB, 0, sizeof(float) * 6);
        if (n_sel) *n_sel = 0;
        return S2M_OK;
    }
    int rc;
    if ((rc = upload_ctx(h))) return rc;
    if ((rc = push_state(h, pose))) return rc;
    {
        const LoopShape sh = shape_of(h);
        launch_fused(h, sh, false, 0, 0);
        launch_finalize(h, sh, 0, 1);
    }
    S2M_HIP(h, hipGetLastError());
    S2M_HIP(h, hipMemcpyAsync(&h->h_state[1], h->state.p, sizeof(DevState), hipMemcpyDeviceToHost, h->stream));
    S2M_HIP(h, hipStreamSynchronize(h->stream));
    if (AtA) memcpy(AtA, h->h_state[1].AtA, sizeof(float) * 36);
    if (AtB) memcpy(AtB, h->h_state[1].AtB, sizeof(float) * 6);
    if (n_sel) *n_sel = h->h_state[1].n_sel_last;
    return S2M_OK;
}

// Diagnostics: workgroups the certify kernels of the last collected loop handed to the search kernel (0 for a fused loop;
// slot >= 0: that scan slot of the last batch).
int s2m_debug_deferred(s2m_handle h, int slot)
{
    if (!h) return S2M_ERR_INVALID_ARG;
    if (slot >= 0) { if (slot >= (int)h->kids.size()) return S2M_ERR_INVALID_ARG; h = h->kids[(size_t)slot]; }
    return h->h_state ? h->h_state[1].deferred_total : 0;
}

int s2m_last_timing(s2m_handle h, float* optimize_ms, float* set_map_ms, float* set_scan_ms)
{
    if (!h) return S2M_ERR_INVALID_ARG;
    if (h->scan_timing_pending) {
        S2M_HIP(h, hipSetDevice(h->device));
        S2M_HIP(h, hipStreamSynchronize(h->stream));
        S2M_HIP(h, hipEventElapsedTime(&h->t_set_scan_ms, h->ev_c, h->ev_d));
        h->scan_timing_pending = false;
    }
    if (optimize_ms) *optimize_ms = h->t_optimize_ms;
    if (set_map_ms) *set_map_ms = h->t_set_map_ms;
    if (set_scan_ms) *set_scan_ms = h->t_set_scan_ms;
    return S2M_OK;
}

static int time_iterations_impl(s2m_handle h, const float pose[6], int reps, float* ms_mean, float* ms_per_iter)
{
    if (!h || !pose || reps < 1) return S2M_ERR_INVALID_ARG;
    if (!h->have_scan || h->n_m == 0 || h->n_q == 0) return fail(h, S2M_ERR_NO_SCAN, "needs a resident scan and map");
    S2M_HIP(h, hipSetDevice(h->device));
    int rc;
    if ((rc = upload_ctx(h))) return rc;
    const int nit = h->prm.max_iter;
    if (h->iter_events.size() < (size_t)(2 * nit)) {
        const size_t old = h->iter_events.size();
        h->iter_events.resize(2 * nit);
        for (size_t k = old; k < h->iter_events.size(); k++) S2M_HIP(h, hipEventCreate(&h->iter_events[k]));
    }
    std::vector<double> per_iter((size_t)nit, 0.0);
    for (int rep = 0; rep < reps; rep++) {
        // a new scan starts without a prior or cached planes (what s2m_set_scan leaves behind)
        S2M_HIP(h, hipMemsetAsync(h->cert.p, 0, sizeof(float4) * h->n_q, h->stream));
        S2M_HIP(h, hipMemsetAsync(h->aux.p, 0, sizeof(int4) * h->n_q, h->stream));
        if ((rc = push_state(h, pose))) return rc;
        enqueue_loop(h, shape_of(h), h->iter_events.data());     // the real loop, launched one by one between event pairs
        h->hctx.density_pending = 0;
        S2M_HIP(h, hipGetLastError());
        S2M_HIP(h, hipStreamSynchronize(h->stream));
        for (int it = 0; it < nit; it++) {
            float ms = 0;
            S2M_HIP(h, hipEventElapsedTime(&ms, h->iter_events[2 * it], h->iter_events[2 * it + 1]));
            per_iter[(size_t)it] += ms;
        }
    }
    double total = 0.0;
    for (int it = 0; it < nit; it++) {
        total += per_iter[(size_t)it];
        if (ms_per_iter) ms_per_iter[it] = (float)(per_iter[(size_t)it] / reps);
    }
    if (ms_mean) *ms_mean = (float)(total / ((double)reps * nit));
    return S2M_OK;
}

// Diagnostics: one full loop from `pose` (max_iter iterations, so early_exit should be off), then `reps` back-to-back
// replays of the loop's last registration launch in the state the loop ended in (nearly every point certified):
// solve_prev = 1 closes the iteration before it in its prologue each time (the steady launch of the fused loop),
// 0 rebuilds the transform only.  Returns the mean time per replayed launch, gaps included.
int s2m_debug_time_steady(s2m_handle h, const float pose[6], int reps, int solve_prev, float* us_per_launch)
{
    if (!h || !pose || reps < 1 || !us_per_launch) return S2M_ERR_INVALID_ARG;
    if (!h->have_scan || h->n_m == 0 || h->n_q == 0) return fail(h, S2M_ERR_NO_SCAN, "needs a resident scan and map");
    float p[6];
    memcpy(p, pose, 24);
    s2m_result r;
    int rc = s2m_optimize_resident(h, p, nullptr, &r);
    if (rc) return rc;
    if (r.skipped || r.iters_run != h->prm.max_iter) return fail(h, S2M_ERR_INVALID_ARG, "the loop ended early: switch early_exit off");
    const LoopShape sh = shape_of(h);
    const int L = h->prm.max_iter;
    S2M_HIP(h, hipEventRecord(h->ev_c, h->stream));
    for (int k = 0; k < reps; k++)
        launch_iteration(h, sh, sh.split ? L + (k & 1) : L, solve_prev ? 1 : 0);   // (split: the worklist slots alternate with the launch parity)
    S2M_HIP(h, hipGetLastError());
    S2M_HIP(h, hipEventRecord(h->ev_d, h->stream));
    S2M_HIP(h, hipStreamSynchronize(h->stream));
    float ms = 0;
    S2M_HIP(h, hipEventElapsedTime(&ms, h->ev_c, h->ev_d));
    *us_per_launch = ms * 1e3f / (float)reps;
    return S2M_OK;
}

// Mean duration of a k_register launch over `reps` whole LM loops (max_iter >= 5, fused loop), gaps between the back-to-back
// launches included: HIP events on the handle's stream around launch 0, launch 1, the run of launches 2 .. n-2 and launch n-1.
int s2m_time_loop_launches(s2m_handle h, const float pose[6], int reps, float* us_per_launch)
{
    if (!h || !pose || reps < 1 || !us_per_launch) return S2M_ERR_INVALID_ARG;
    if (!h->have_scan || h->n_m == 0 || h->n_q == 0) return fail(h, S2M_ERR_NO_SCAN, "needs a resident scan and map");
    const int nit = h->prm.max_iter;
    if (nit < 5) return fail(h, S2M_ERR_INVALID_ARG, "needs max_iter >= 5");
    S2M_HIP(h, hipSetDevice(h->device));
    int rc;
    if ((rc = upload_ctx(h))) return rc;
    while (h->iter_events.size() < 8) { hipEvent_t e; S2M_HIP(h, hipEventCreate(&e)); h->iter_events.push_back(e); }
    double total_ms = 0.0;
    for (int rep = 0; rep < reps; rep++) {
        S2M_HIP(h, hipMemsetAsync(h->cert.p, 0, sizeof(float4) * h->n_q, h->stream));      // a new scan: no certificates, no neighbourhoods
        S2M_HIP(h, hipMemsetAsync(h->aux.p, 0, sizeof(int4) * h->n_q, h->stream));
        if ((rc = push_state(h, pose))) return rc;
        enqueue_loop(h, shape_of(h), h->iter_events.data(), true);
        h->hctx.density_pending = 0;
        S2M_HIP(h, hipGetLastError());
        S2M_HIP(h, hipStreamSynchronize(h->stream));
        for (int k = 0; k < 4; k++) {
            float ms = 0;
            S2M_HIP(h, hipEventElapsedTime(&ms, h->iter_events[2 * k], h->iter_events[2 * k + 1]));
            total_ms += ms;
        }
    }
    *us_per_launch = (float)(total_ms * 1e3 / ((double)reps * nit));
    return S2M_OK;
}

int s2m_time_iteration_kernel(s2m_handle h, const float pose[6], int reps, float* ms_per_launch)
{
    if (!ms_per_launch) return S2M_ERR_INVALID_ARG;
    return time_iterations_impl(h, pose, reps, ms_per_launch, nullptr);
}

int s2m_time_iterations(s2m_handle h, const float pose[6], int reps, float* ms_per_iter, int cap)
{
    if (!h || !ms_per_iter || cap < h->prm.max_iter) return S2M_ERR_INVALID_ARG;
    return time_iterations_impl(h, pose, reps, nullptr, ms_per_iter);
}

// ---- section 8(f) rows F2 / F1: the voxel-grid stages either side of the path ------------------------

static int voxel_downsample_impl(s2m_handle h, const void* pts, size_t n, size_t stride_bytes, float leaf,
                                 void* out, size_t out_stride_bytes, size_t cap, size_t* n_out, bool on_device)
{
    int rc = check_records(h, pts, n, stride_bytes);
    if (rc) return rc;
    if ((rc = check_leaf(h, leaf))) return rc;
    if (!n_out || (cap > 0 && !out) || out_stride_bytes < 12 || (out_stride_bytes & 3))
        return fail(h, S2M_ERR_INVALID_ARG, "bad output buffer");
    *n_out = 0;
    if (n == 0) return S2M_OK;
    S2M_HIP(h, hipSetDevice(h->device));
    VoxResult res;
    if (on_device) {
        hipError_t e = vox_downsample(h->vox, h->stream, static_cast<const unsigned char*>(pts), n, stride_bytes, leaf,
                                      static_cast<unsigned char*>(out), out_stride_bytes, cap, &res);
        if (e != hipSuccess) return fail(h, S2M_ERR_HIP, "voxel grid filter", e);
    } else {
        if ((rc = ensure(h, h->vox_in, n * stride_bytes))) return rc;
        S2M_HIP(h, hipMemcpyAsync(h->vox_in.p, pts, n * stride_bytes, hipMemcpyHostToDevice, h->stream));
        if ((rc = voxel_into(h, h->vox_in.as<unsigned char>(), n, stride_bytes, leaf, h->vox_out, &res))) return rc;
        if ((rc = download_records(h, h->vox_out, res.n_out, out, out_stride_bytes, cap))) return rc;
    }
    *n_out = res.n_out;
    if (res.n_out > cap) return fail(h, S2M_ERR_CAPACITY, "output buffer too small for the filtered cloud");
    return res.leaf_too_small ? S2M_WARN_LEAF_TOO_SMALL : S2M_OK;
}

int s2m_voxel_downsample(s2m_handle h, const void* pts, size_t n, size_t stride_bytes, float leaf,
                         void* out, size_t out_stride_bytes, size_t cap, size_t* n_out)
{ return voxel_downsample_impl(h, pts, n, stride_bytes, leaf, out, out_stride_bytes, cap, n_out, false); }

int s2m_voxel_downsample_device(s2m_handle h, const void* d_pts, size_t n, size_t stride_bytes, float leaf,
                                void* d_out, size_t out_stride_bytes, size_t cap, size_t* n_out)
{ return voxel_downsample_impl(h, d_pts, n, stride_bytes, leaf, d_out, out_stride_bytes, cap, n_out, true); }

int s2m_downsample_scan(s2m_handle h, const void* pts, size_t n, size_t stride_bytes, int on_device, float leaf,
                        void* out, size_t out_stride_bytes, size_t cap, size_t* n_out)
{
    int rc = check_records(h, pts, n, stride_bytes);
    if (rc) return rc;
    if ((rc = check_leaf(h, leaf))) return rc;
    if (!n_out || (cap > 0 && (!out || out_stride_bytes < 12 || (out_stride_bytes & 3))))
        return fail(h, S2M_ERR_INVALID_ARG, "bad output buffer");
    *n_out = 0;
    S2M_HIP(h, hipSetDevice(h->device));
    VoxResult res;
    if (n > 0) {
        const unsigned char* d_in = static_cast<const unsigned char*>(pts);
        if (!on_device) {
            if ((rc = ensure(h, h->vox_in, n * stride_bytes))) return rc;
            S2M_HIP(h, hipMemcpyAsync(h->vox_in.p, pts, n * stride_bytes, hipMemcpyHostToDevice, h->stream));
            d_in = h->vox_in.as<unsigned char>();
        }
        if ((rc = voxel_into(h, d_in, n, stride_bytes, leaf, h->scan_ds, &res))) return rc;
    }
    *n_out = res.n_out;
    // laserCloudSurfLastDS stays on the device as the registration's scan; the host copy is for the key-frame store
    if ((rc = set_scan_impl(h, h->scan_ds.p, res.n_out, kDsStride, true))) return rc;
    if (cap > 0 && (rc = download_records(h, h->scan_ds, res.n_out, out, out_stride_bytes, cap))) return rc;
    if (cap > 0 && res.n_out > cap) return fail(h, S2M_ERR_CAPACITY, "output buffer too small for the filtered scan");
    return res.leaf_too_small ? S2M_WARN_LEAF_TOO_SMALL : S2M_OK;
}

int s2m_extract_cloud(s2m_handle h, int n_frames, const void* const* frames, const size_t* frame_sizes,
                      size_t stride_bytes, int on_device, const float* poses_xyzrpy, float leaf,
                      void* out, size_t out_stride_bytes, size_t cap, size_t* n_out)
{
    if (!h) return S2M_ERR_INVALID_ARG;
    if (n_frames < 0 || (n_frames > 0 && (!frames || !frame_sizes || !poses_xyzrpy)))
        return fail(h, S2M_ERR_INVALID_ARG, "null key-frame table");
    int rc = check_leaf(h, leaf);
    if (rc) return rc;
    if (!n_out || (cap > 0 && (!out || out_stride_bytes < 12 || (out_stride_bytes & 3))))
        return fail(h, S2M_ERR_INVALID_ARG, "bad output buffer");
    *n_out = 0;
    size_t total = 0;
    for (int f = 0; f < n_frames; f++) {
        if ((rc = check_records(h, frames[f], frame_sizes[f], stride_bytes))) return rc;
        total += frame_sizes[f];
        if (total > (size_t)0x3fffffff) return fail(h, S2M_ERR_CAPACITY, "too many points");
    }
    S2M_HIP(h, hipSetDevice(h->device));
    VoxResult res;
    std::vector<int32_t> offsets((size_t)n_frames + 1, 0);
    std::vector<float> T((size_t)n_frames * 12);
    std::vector<const unsigned char*> src((size_t)n_frames, nullptr);
    if (total > 0) {
        if (!on_device && (rc = ensure(h, h->vox_in, total * stride_bytes))) return rc;
        for (int f = 0; f < n_frames; f++) {
            offsets[f + 1] = offsets[f] + (int32_t)frame_sizes[f];
            // transCur = pcl::getTransformation(x, y, z, roll, pitch, yaw) of the key pose (:317)
            const float* p = poses_xyzrpy + 6 * (size_t)f;
            const float rpyxyz[6] = { p[3], p[4], p[5], p[0], p[1], p[2] };
            host_pose_to_transform(rpyxyz, &T[12 * (size_t)f], nullptr);
            if (on_device) src[f] = static_cast<const unsigned char*>(frames[f]);
            else {
                unsigned char* dst = h->vox_in.as<unsigned char>() + (size_t)offsets[f] * stride_bytes;
                if (frame_sizes[f])
                    S2M_HIP(h, hipMemcpyAsync(dst, frames[f], frame_sizes[f] * stride_bytes, hipMemcpyHostToDevice, h->stream));
                src[f] = dst;
            }
        }
        if ((rc = ensure(h, h->frames_xf, kDsStride * total))) return rc;
        hipError_t e = vox_transform_frames(h->vox, h->stream, src.data(), stride_bytes, offsets.data(), T.data(), n_frames,
                                            h->frames_xf.as<unsigned char>(), kDsStride);
        if (e != hipSuccess) return fail(h, S2M_ERR_HIP, "key-frame transform", e);
        if ((rc = voxel_into(h, h->frames_xf.as<unsigned char>(), total, kDsStride, leaf, h->map_ds, &res))) return rc;
    }
    *n_out = res.n_out;
    // laserCloudSurfFromMapDS becomes the search index (the reference's kdtree->setInputCloud, :1302)
    if ((rc = set_map_impl(h, h->map_ds.p, res.n_out, kDsStride, true))) return rc;
    if (cap > 0 && (rc = download_records(h, h->map_ds, res.n_out, out, out_stride_bytes, cap))) return rc;
    if (cap > 0 && res.n_out > cap) return fail(h, S2M_ERR_CAPACITY, "output buffer too small for the local map");
    return res.leaf_too_small ? S2M_WARN_LEAF_TOO_SMALL : S2M_OK;
}

int s2m_transform_cloud(s2m_handle h, const void* pts, size_t n, size_t stride_bytes, const float pose_xyzrpy[6],
                        void* out, size_t out_stride_bytes)
{
    int rc = check_records(h, pts, n, stride_bytes);
    if (rc) return rc;
    if (!pose_xyzrpy || (n > 0 && !out) || out_stride_bytes < 12 || (out_stride_bytes & 3))
        return fail(h, S2M_ERR_INVALID_ARG, "bad pose or output buffer");
    if (n == 0) return S2M_OK;
    S2M_HIP(h, hipSetDevice(h->device));
    if ((rc = ensure(h, h->vox_in, n * stride_bytes))) return rc;
    if ((rc = ensure(h, h->frames_xf, kDsStride * n))) return rc;
    S2M_HIP(h, hipMemcpyAsync(h->vox_in.p, pts, n * stride_bytes, hipMemcpyHostToDevice, h->stream));
    const float rpyxyz[6] = { pose_xyzrpy[3], pose_xyzrpy[4], pose_xyzrpy[5], pose_xyzrpy[0], pose_xyzrpy[1], pose_xyzrpy[2] };
    float T[12];
    host_pose_to_transform(rpyxyz, T, nullptr);
    const int32_t offsets[2] = { 0, (int32_t)n };
    const unsigned char* src[1] = { h->vox_in.as<unsigned char>() };
    hipError_t e = vox_transform_frames(h->vox, h->stream, src, stride_bytes, offsets, T, 1, h->frames_xf.as<unsigned char>(), kDsStride);
    if (e != hipSuccess) return fail(h, S2M_ERR_HIP, "cloud transform", e);
    return download_records(h, h->frames_xf, n, out, out_stride_bytes, n);
}

// ---- section 8(f) row F4: ICP loop-closure alignment -----------------------------------------------

int s2m_icp_default_params(s2m_icp_params* p)
{
    if (!p) return S2M_ERR_INVALID_ARG;
    p->max_correspondence_distance = 20.0;      // historyKeyframeSearchRadius (10, include/utility.h:245) * 2 (:573)
    p->max_iterations = 100;                    // :574
    p->transformation_epsilon = 1e-6;           // :575
    p->euclidean_fitness_epsilon = 1e-6;        // :576
    return S2M_OK;
}

int s2m_icp_align(s2m_handle h, const void* src, size_t n_src, const void* tgt, size_t n_tgt, size_t stride_bytes,
                  const s2m_icp_params* p, s2m_icp_result* out)
{
    int rc = check_records(h, src, n_src, stride_bytes);
    if (rc) return rc;
    if ((rc = check_records(h, tgt, n_tgt, stride_bytes))) return rc;
    if (!out) return S2M_ERR_INVALID_ARG;
    s2m_icp_params prm;
    if (p) prm = *p; else s2m_icp_default_params(&prm);
    if (!(prm.max_correspondence_distance > 0.0) || prm.max_iterations < 1)
        return fail(h, S2M_ERR_INVALID_ARG, "ICP needs a positive correspondence distance and at least one iteration");
    S2M_HIP(h, hipSetDevice(h->device));
    if (n_src) { if ((rc = ensure(h, h->icp_src, n_src * stride_bytes))) return rc;
                 S2M_HIP(h, hipMemcpyAsync(h->icp_src.p, src, n_src * stride_bytes, hipMemcpyHostToDevice, h->stream)); }
    if (n_tgt) { if ((rc = ensure(h, h->icp_tgt, n_tgt * stride_bytes))) return rc;
                 S2M_HIP(h, hipMemcpyAsync(h->icp_tgt.p, tgt, n_tgt * stride_bytes, hipMemcpyHostToDevice, h->stream)); }
    IcpParams ip{ prm.max_correspondence_distance, prm.max_iterations, prm.transformation_epsilon, prm.euclidean_fitness_epsilon };
    IcpResult r;
    hipError_t e = icp_align(h->icp, h->stream, h->icp_src.as<unsigned char>(), n_src, h->icp_tgt.as<unsigned char>(), n_tgt,
                             stride_bytes, ip, &r);
    if (e != hipSuccess) return fail(h, S2M_ERR_HIP, "ICP alignment", e);
    S2M_HIP(h, hipStreamSynchronize(h->stream));
    memcpy(out->T, r.T, sizeof(r.T));
    out->converged = r.converged; out->iterations = r.iterations; out->fitness_score = r.fitness;
    return S2M_OK;
}

int s2m_make_scancontext(s2m_handle h, const void* pts, size_t n, size_t stride_bytes,
                         double desc[S2M_SC_NUM_RING * S2M_SC_NUM_SECTOR], double ringkey[S2M_SC_NUM_RING])
{
    int rc = check_records(h, pts, n, stride_bytes);
    if (rc) return rc;
    if (!desc || !ringkey) return S2M_ERR_INVALID_ARG;
    S2M_HIP(h, hipSetDevice(h->device));
    if ((rc = sc_build_descriptor(h, pts, n, stride_bytes))) return rc;
    S2M_HIP(h, hipMemcpyAsync(h->h_sc, h->sc_out.p, sizeof(double) * 1220, hipMemcpyDeviceToHost, h->stream));
    S2M_HIP(h, hipStreamSynchronize(h->stream));
    memcpy(desc, h->h_sc, sizeof(double) * 1200);
    memcpy(ringkey, h->h_sc + 1200, sizeof(double) * 20);
    return S2M_OK;
}

// ---- section 8(f) row F3: the SCManager loop detector ---------------------------------------------

int s2m_sc_reset(s2m_handle h)
{
    if (!h) return S2M_ERR_INVALID_ARG;
    h->sc_n = 0; h->sc_n_search = 0; h->sc_counter = 0;
    return S2M_OK;
}

int s2m_sc_size(s2m_handle h) { return h ? (int)h->sc_n : S2M_ERR_INVALID_ARG; }

int s2m_sc_add_scan(s2m_handle h, const void* pts, size_t n, size_t stride_bytes)
{
    int rc = check_records(h, pts, n, stride_bytes);
    if (rc) return rc;
    S2M_HIP(h, hipSetDevice(h->device));
    if ((rc = sc_build_descriptor(h, pts, n, stride_bytes))) return rc;
    if ((rc = sc_append_from_out(h))) return rc;
    S2M_HIP(h, hipStreamSynchronize(h->stream));          // the caller's cloud is free again
    return S2M_OK;
}

int s2m_sc_add_descriptor(s2m_handle h, const double desc[S2M_SC_NUM_RING * S2M_SC_NUM_SECTOR])
{
    if (!h || !desc) return S2M_ERR_INVALID_ARG;
    S2M_HIP(h, hipSetDevice(h->device));
    // ring key = row means, as makeRingkeyFromScancontext (:198-211) / k_sc_finish compute them
    memcpy(h->h_sc, desc, sizeof(double) * kScDesc);
    for (int r = 0; r < S2M_SC_NUM_RING; r++) {
        double a = 0.0;
        for (int k = 0; k < S2M_SC_NUM_SECTOR; k++) a += desc[r * S2M_SC_NUM_SECTOR + k];
        h->h_sc[kScDesc + r] = a / 60.0;
    }
    S2M_HIP(h, hipMemcpyAsync(h->sc_out.p, h->h_sc, sizeof(double) * 1220, hipMemcpyHostToDevice, h->stream));
    int rc = sc_append_from_out(h);
    if (rc) return rc;
    S2M_HIP(h, hipStreamSynchronize(h->stream));          // h_sc is reused by the next call
    return S2M_OK;
}

int s2m_sc_detect_loop(s2m_handle h, int32_t* loop_id, float* yaw_diff_rad, s2m_sc_match* detail)
{
    if (!h || !loop_id || !yaw_diff_rad) return S2M_ERR_INVALID_ARG;
    *loop_id = -1; *yaw_diff_rad = 0.0f;
    if (detail) { memset(detail, 0, sizeof(*detail)); detail->min_dist = 10000000; }
    constexpr int NUM_EXCLUDE_RECENT = 30, TREE_MAKING_PERIOD = 10;       // Scancontext.h:89, :99
    if ((int)h->sc_n < NUM_EXCLUDE_RECENT + 1) return S2M_OK;              // :263-267
    // the reference rebuilds its kd-tree every TREE_MAKING_PERIOD_ calls; between rebuilds the search sees the older set
    if (h->sc_counter % TREE_MAKING_PERIOD == 0) h->sc_n_search = h->sc_n - NUM_EXCLUDE_RECENT;
    h->sc_counter = h->sc_counter + 1;
    S2M_HIP(h, hipSetDevice(h->device));
    // the kernel writes its 48-byte result straight into the pinned staging block (host-coherent memory the device can address):
    // a copy behind the kernel would be a second trip through the queue, and one to pageable memory ~50 us more
    static_assert(sizeof(ScDetectOut) <= sizeof(double) * 1220, "the result fits the pinned ScanContext staging block");
    hipLaunchKernelGGL(k_sc_detect, dim3(1), dim3(kScThreads), 0, h->stream, (const double*)h->sc_store_desc.as<double>(),
                       (const float*)h->sc_store_ring.as<float>(), (const double*)h->sc_store_sector.as<double>(),
                       (int)h->sc_n, (int)h->sc_n_search, reinterpret_cast<ScDetectOut*>(h->h_sc));
    S2M_HIP(h, hipGetLastError());
    S2M_HIP(h, hipStreamSynchronize(h->stream));
    ScDetectOut o;
    memcpy(&o, h->h_sc, sizeof(o));
    *loop_id = o.loop_id; *yaw_diff_rad = o.yaw_diff_rad;
    if (detail) {
        detail->min_dist = o.min_dist; detail->nn_idx = o.nn_idx; detail->nn_align = o.nn_align;
        for (int k = 0; k < 3; k++) { detail->cand_idx[k] = o.cand_idx[k]; detail->cand_d2[k] = o.cand_d2[k]; }
    }
    return S2M_OK;
}

int s2m_sc_distance(s2m_handle h, int32_t query_idx, const int32_t* cand_idx, int32_t m, double* dist, int32_t* shift)
{
    if (!h || m < 0 || (m > 0 && (!cand_idx || !dist || !shift))) return S2M_ERR_INVALID_ARG;
    if (query_idx < 0 || (size_t)query_idx >= h->sc_n) return fail(h, S2M_ERR_INVALID_ARG, "query index outside the descriptor store");
    for (int32_t k = 0; k < m; k++)
        if (cand_idx[k] < 0 || (size_t)cand_idx[k] >= h->sc_n) return fail(h, S2M_ERR_INVALID_ARG, "candidate index outside the descriptor store");
    if (m == 0) return S2M_OK;
    S2M_HIP(h, hipSetDevice(h->device));
    const size_t off_d = ((sizeof(int32_t) * (size_t)m + 15) & ~(size_t)15), off_s = off_d + sizeof(double) * (size_t)m;
    int rc = ensure(h, h->sc_cand, off_s + sizeof(int32_t) * (size_t)m);
    if (rc) return rc;
    unsigned char* base = h->sc_cand.as<unsigned char>();
    S2M_HIP(h, hipMemcpyAsync(base, cand_idx, sizeof(int32_t) * (size_t)m, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_sc_distance_batch, dim3((unsigned)m), dim3(kScThreads), 0, h->stream, (const double*)h->sc_store_desc.as<double>(),
                       (const double*)h->sc_store_sector.as<double>(), (int)query_idx, (const int32_t*)base,
                       (double*)(base + off_d), (int32_t*)(base + off_s));
    S2M_HIP(h, hipGetLastError());
    S2M_HIP(h, hipMemcpyAsync(dist, base + off_d, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, h->stream));
    S2M_HIP(h, hipMemcpyAsync(shift, base + off_s, sizeof(int32_t) * (size_t)m, hipMemcpyDeviceToHost, h->stream));
    S2M_HIP(h, hipStreamSynchronize(h->stream));
    return S2M_OK;
}

}  // extern "C"
