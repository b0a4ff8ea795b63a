// s2m_types.h — device/host shared structures of the gfx950 scan-to-map path.
// Internal to liorf_amd/csrc; the public boundary is include/liorf_s2m.h.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>
#include "../../include/liorf_s2m.h"

namespace s2m {

constexpr int kBlock = 512;            // threads per workgroup of the registration kernel (8 waves: fewer partial rows for k_finalize)
constexpr int kAcc = 28;               // 21 upper-triangular JtJ + 6 Jtr + 1 correspondence count
constexpr int kFinThreads = 1024;      // finalize kernel workgroup
constexpr int kMaxIter = 64;           // trace capacity
constexpr int kProfWords = 24;         // uint64 words per wave written by the diagnostics variant of k_register
constexpr int kBlocksQuantum = 8;      // graph cache key granularity (workgroups)

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
};

// Everything a kernel needs, in device memory so a captured graph stays valid when
// buffers grow or the grid changes; kernels receive only a pointer to this.
struct DevCtx {
    GridDesc g;
    const float4* map_sorted;     // [n_m] x,y,z, original index (bit pattern), cell-sorted
    const int32_t* cell_start;    // [ncells+1]
    const float* qx; const float* qy; const float* qz;   // [n_q] lidar-frame scan, SoA, locality-sorted
    const int32_t* qperm;         // [n_q] sorted position -> original scan index
    float4*  prevp;               // [5][n_q] previous launch's neighbours per sorted scan point: x,y,z, map index
    int32_t* prior_valid;         // [n_q] 1 if prevp holds 5 neighbours of the current map for this point
    float4*  plane_cache;         // [n_q] pa,pb,pc,pd of the plane fitted to prev5's tuple
    int32_t* plane_state;         // [n_q] 0 none, 1 plane passed the inlier test, 2 plane failed it
    int32_t n_q, n_m, nblocks;
    const int2* wave_table;       // [nblocks*4] {first sorted point, count <= 64} per wave
    const int32_t* n_waves;       // entries of wave_table in use
    // the same table for the kernels that rebuild it before launch 0 (k_wave_density, k_chunk_table_density)
    int2*    wave_table_rw; int32_t* n_waves_rw;
    const int32_t* chunk_parts;   // [n_chunks] extent-based split wish per 64-point chunk
    int32_t* chunk_factor;        // [n_chunks] density-based extra split, zero between uses
    int32_t  n_chunks;
    int32_t  density_pending;     // 1 from s2m_set_scan until the scan's first optimisation has re-split the table
    double* partials;             // [2][nblocks][kAcc], slot = launch parity
    DevState* state;
    s2m_iter_trace* trace;        // [kMaxIter]
    // parameters
    float  gate_f;                // smallest fp32 >= gate_sq: nothing at or beyond it is observable
    double gate_sq, plane_tol, weight_scale, weight_min, conv_deg, conv_cm;
    float  eig_thresh;
    int32_t min_corr, max_iter, early_exit;
    int32_t ablate;               // diagnostics only (env S2M_ABLATE): 1 skip search, 2 skip plane/Jacobian, 4 skip reduction, 8 skip staging, 16 ignore the prior, 32 ignore the plane cache, 64 gather path only
    // observation outputs of the hook variant (original scan order), may be null
    int32_t* dbg_idx5; float* dbg_d2; uint8_t* dbg_flag; float* dbg_coeff;
    unsigned long long* dbg_clk;  // [nwaves][kProfWords] per-wave wall-clock stamps + tile stats (diagnostics)
};

}  // namespace s2m
