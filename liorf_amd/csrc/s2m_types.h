// s2m_types.h — device/host shared structures of the gfx950 scan-to-map path.
// Internal to liorf_amd/csrc; the public boundary is include/liorf_s2m.h.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>
#include "../../include/liorf_s2m.h"
#include "../../include/liorf_s2m_debug.h"

namespace s2m {

constexpr int kBlock = 512;            // threads per workgroup of the registration kernel (8 waves: fewer partial rows for k_finalize)
constexpr int kAcc = 28;               // 21 upper-triangular JtJ + 6 Jtr + 1 correspondence count
constexpr int kFinThreads = 1024;      // finalize kernel workgroup
constexpr int kMaxIter = 64;           // trace capacity
constexpr int kProfWords = 32;         // uint64 words per wave written by the diagnostics variant of k_register
constexpr int kBlocksQuantum = 8;      // graph cache key granularity (workgroups)
constexpr int kNbrCap = 6;            // members of a scan point's front (s2m_register.hpp: kNbr)
constexpr int kMaxBlocks = 512;        // largest k_register grid: two 8-wave workgroups on each of the 256 CUs, all co-resident
constexpr int kBigWaves = 16;          // a scan that needs more than one 8-wave workgroup per CU runs 16-wave workgroups, one per CU: every
                                       // workgroup reads every row of partial sums when it closes an iteration, and rows x workgroups
                                       // is 4x smaller with half as many workgroups

// Uniform search grid over the map: cell edge E >= sqrt(gate_sq)*(1+2^-10), so the 3x3x3
// neighbourhood of a query's cell holds every map point with fp32 d2 < gate_sq.
struct GridDesc {
    float ox, oy, oz;     // origin (min corner minus one cell)
    float inv_e;          // 1 / E
    float e;              // E
    int   nx, ny, nz;     // cells per axis
    int   ncells;
};

// Per-scan loop state, lives in device memory; written by k_finalize and by workgroup 0 of k_register.
struct DevState {
    float pose[6];        // transformTobeMapped                (reference :134): latest value, read by the host
    float pose2[2][6];    // pose launch L runs with, in slot L & 1 (double-buffered: see lm_solve_update)
    float T[12];          // transPointAssociateToMap, row-major (:142)
    float sc[6];          // srx,crx,sry,cry,srz,crz             (:1170-1175)
    float matP[36];       // degeneracy projector                (:140)
    float AtA[36];        // last normal equations (observation hook)
    float AtB[6];
    int32_t isDegenerate; // (:139)
    int32_t done;         // converged (early_exit) or stalled: later launches return at once
    int32_t converged;
    int32_t iters_run;
    int32_t n_sel_last;
    int32_t stalled;      // n_sel < min_corr: every further iteration is the same no-op
    int32_t n_waves;      // wave-table entries of the resident scan (copied in by k_set_state)
    int32_t T_valid;      // 1: T/sc were set by the host (launch 0); 0: k_register rebuilds them from pose2
    int32_t deferred_total; // diagnostics: workgroups the certify kernel handed to the search kernel, summed over the loop
    int32_t pad_;
};

// Everything a kernel needs, in device memory so a captured graph stays valid when
// buffers grow or the grid changes; kernels receive only a pointer to this.
struct DevCtx {
    GridDesc g;
    const float4* map_sorted;     // [n_m] x,y,z, original index (bit pattern), cell-sorted
    const int32_t* cell_start;    // [ncells+1]
    const float* qx; const float* qy; const float* qz;   // [n_q] lidar-frame scan, SoA, locality-sorted
    const int32_t* qperm;         // [n_q] sorted position -> original scan index
    // what a scan point remembers from launch to launch (see s2m_register.hpp); all reset by s2m_set_scan / s2m_set_map
    int32_t* npos;                // [5][n_q] its 5 neighbours, ascending (d2, map index), as positions in map_sorted
    float4*  cert;                // [n_q] {q_ref x,y,z: where the point stood when the tuple was established, slack: how far it may move}
    int4*    aux;                 // [n_q] {r_out (float bits): every map point other than the front members was at least this far from q_ref,
                                  //        state: bits 0-1 plane 0 none / 1 passed the inlier test / 2 failed it, bit 2 the tuple is complete,
                                  //        bit 3 the front is valid, bit 4 plane_alt / npos_alt are valid, number of front members, r_out again}
    float4*  front;               // [kNbr][n_q] the front: the (up to) six nearest map points the point knew when it last searched - EVERY map
                                  //        point within r_out of q_ref - as {x, y, z, position in map_sorted (bits)}: re-measuring them needs no map access
    float4*  plane_cache;         // [n_q] pa,pb,pc,pd of the plane fitted to the tuple; pa = NaN: this point contributes nothing
    // the tuple (and its plane) a point had before its present one - state bit 4 says it is valid: two nearly equidistant
    // neighbours swap places back and forth with the micro-steps of the converged loop, and the plane of either order is kept
    float4*  plane_alt;           // [n_q]
    int32_t* npos_alt;            // [5][n_q]
    int32_t n_q, n_m, nblocks;    // nblocks: workgroups of a k_register launch (<= kMaxBlocks; waves loop over the wave table)
    int32_t wpb;                  // waves per workgroup of k_register for this scan: kBlock / 64 or kBigWaves
    int32_t table_cap;            // capacity of wave_table in entries
    const int2* wave_table;       // [table_cap] {first sorted point, count <= 64} per wave-table entry
    const int32_t* n_waves;       // entries of wave_table in use
    // the same table for the kernels that rebuild it before launch 0 (k_wave_density, k_chunk_table_density)
    int2*    wave_table_rw; int32_t* n_waves_rw;
    const int32_t* chunk_parts;   // [n_chunks] extent-based split wish per 64-point chunk
    int32_t* chunk_factor;        // [n_chunks] density-based extra split, zero between uses
    int32_t  n_chunks;
    int32_t  density_pending;     // 1 from s2m_set_scan until the scan's first optimisation has re-split the table
    double* partials;             // [2][nblocks][kAcc], slot = launch parity
    int32_t* wl_count;            // [2] worklist of the search kernel, slot = launch parity: workgroups of the certify kernel that wrote no row ...
    int32_t* wl_items;            // [2][nblocks] ... and their numbers
    DevState* state;
    s2m_iter_trace* trace;        // [kMaxIter]
    // parameters
    float  gate_f;                // smallest fp32 >= gate_sq: nothing at or beyond it is observable
    float  gate_r;                // sqrtf(gate_sq) rounded down: the gate as a distance
    double gate_sq, plane_tol, weight_scale, weight_min, conv_deg, conv_cm;
    float  eig_thresh;
    int32_t min_corr, max_iter, early_exit;
    int32_t tune[4];              // experiments (env S2M_TUNE=a,b,c,d): [0]-[2] spare,
                                  // [3] a pass of up to this many searching lanes is served lane by lane instead of staging a tile (default 2)
    int32_t ablate;               // diagnostics and tests (env S2M_ABLATE): 1 no certificates (tier A off), 2 no re-measuring (tier B off), 16 ignore the stored tuple in the search, 32 ignore the plane cache, 64 gather path only, 128 tile path whatever the number of lanes
    // observation outputs of the hook variant (original scan order), may be null
    int32_t* dbg_idx5; float* dbg_d2; uint8_t* dbg_flag; float* dbg_coeff;
    unsigned long long* dbg_clk;  // [nwaves][kProfWords] per-wave wall-clock stamps + tile stats (diagnostics)
};

// The registration kernels take their scans by slot: blockIdx.y selects one (a single scan: one slot).  A batch of scans
// advances in lockstep - one launch per iteration for all slots.
constexpr int kMaxSlots = 64;          // scan slots of a batch (include/liorf_s2m.h: n_scans <= 64)
struct SlotTable {
    const DevCtx* ctx[kMaxSlots];
    DevState*     st[kMaxSlots];
};

// Scan preparation (ordering + wave table) of up to kPrepSlots scans per launch, blockIdx.y = slot: everything a slot's six
// small kernels need, by value, so that a batch of scans costs six launches and not six per scan.
struct PrepSlot {
    const unsigned char* pts; size_t stride;
    int32_t n, n_chunks, base_parts, capacity, npb;
    int32_t *cell_of, *rank_of, *block_hist, *counts, *cell_start;
    float *qx, *qy, *qz; int32_t* qperm; float4* cert; int4* aux;
    int32_t* chunk_parts; int2* wave_table; int32_t* n_waves;
};
constexpr int kPrepSlots = 16;
struct PrepTable { PrepSlot s[kPrepSlots]; };

// The loop state a scan starts from, for up to kInitSlots slots per launch (k_init_states): what fill_state sets, the rest is zero.
struct StateInit {
    DevState* dst; const int32_t* n_waves;
    float pose[6], T[12], sc[6], matP[36];
    int32_t isDegenerate;
};
constexpr int kInitSlots = 12;
struct StateInitTable { StateInit s[kInitSlots]; };
constexpr int kCtxSlots = 8;
struct CtxTable { DevCtx* dst[kCtxSlots]; DevCtx v[kCtxSlots]; };

}  // namespace s2m
