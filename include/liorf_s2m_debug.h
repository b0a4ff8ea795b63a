/*
 * liorf_s2m_debug.h — diagnostic and benchmark entry points of libliorf_s2m.so.
 *
 * NOT part of the drop-in boundary (include/liorf_s2m.h is): nothing a mapOptimization node binds lives here.  These are
 * what the repository's own tools and tests use to look inside the registration path - per-launch timing of the
 * registration kernel, per-wave stage stamps, the device's sinf / cosf / atanf arithmetic, worklist statistics - and the
 * list of environment switches the library reads at s2m_create for A/B experiments.  They may change between rounds.
 *
 * Environment switches (all optional; defaults are what bench.py measures):
 *   S2M_NO_GRAPH=1         plain launches instead of the captured loop graph
 *   S2M_NO_FUSE=1          one k_finalize per iteration instead of closing iterations in the next launch's prologue
 *   S2M_FUSE_MAX=n         largest grid (workgroups x slots) whose iterations are closed in the prologue (512)
 *   S2M_SEGMENT=n          launches in the first range of an early-exit loop (8; 0 = the whole loop in one piece)
 *   S2M_DENSITY_RAW=n      box points above which a wave asks for a finer cut before launch 0 (320; 0 = off)
 *   S2M_BIG_BLOCKS=0       8-wave workgroups whatever the scan size
 *   S2M_SPLIT=0|1|2        fused kernel always / certify + search always / late split in lockstep batches (default)
 *   S2M_SPLIT_FROM=n, S2M_SEARCH_GRID=n, S2M_CLOSE_IN_SEARCH=1, S2M_LEAN=0, S2M_LOCKSTEP=0, S2M_BATCH_MINW=n, S2M_BATCH_ENTRIES=n
 *                          shape of the batch loop (liorf_amd/csrc/s2m_abi.hip, s2m_context)
 *   S2M_ABLATE=bits        switch tiers / search paths off (tests): 1 no certificates, 2 no re-measuring, 16 ignore the prior in
 *                          the search, 32 ignore the plane cache, 64 no tiles (lanes served one by one), 128 tiles for any lane count
 *   S2M_TUNE=a,b,c,d       experiment thresholds (s2m_types.h, DevCtx::tune)
 */
#ifndef LIORF_S2M_DEBUG_H
#define LIORF_S2M_DEBUG_H

#include "liorf_s2m.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Observation hook: sinf / cosf (and atanf when `a` is not NULL) of n host floats as the device computes them -
 * sin/cos when it rebuilds the transform between LM iterations, atan in the ScanContext sector angle. They follow
 * the arithmetic of glibc's sinf / cosf / atanf, so that the device gives what the reference's host libm gives. */
int  s2m_debug_device_trig(s2m_handle h, const float* x, size_t n, float* s, float* c, float* a);
/* Diagnostics: workgroups the certify kernels of the last collected loop handed to the search kernel (slot < 0: the handle's own
 * loop; slot >= 0: that scan slot of the last batch). */
int  s2m_debug_deferred(s2m_handle h, int slot);
/* Benchmark helper: runs `reps` complete LM loops (max_iter iterations, early exit as
 * configured, each loop starting like a fresh scan) on the resident scan + map with plain
 * launches and a HIP-event pair on the handle's stream around every launch of the
 * per-iteration registration kernel (k_register: kNN + plane + Jacobian + block reduction);
 * returns the mean duration of those launches in ms. */
int  s2m_time_iteration_kernel(s2m_handle h, const float pose[6], int reps, float* ms_per_launch);
/* Same measurement, reported per LM iteration: ms_per_iter[it] = mean duration of launch `it` of the
 * loop over `reps` loops (cap >= max_iter entries). The first launches of a scan search without a prior
 * and cost more than the steady state. */
int  s2m_time_iterations(s2m_handle h, const float pose[6], int reps, float* ms_per_iter, int cap);

/* Benchmark helper: mean duration (microseconds) of a k_register launch over `reps` whole LM loops as the fused loop issues them,
 * measured with HIP events on the handle's stream around launch 0, launch 1, the run of back-to-back launches 2 .. max_iter-2 and
 * the last launch - four event pairs per loop instead of max_iter, so the event packets do not break up the back-to-back dispatch
 * (the gaps between consecutive launches are part of the figure). */
int  s2m_time_loop_launches(s2m_handle h, const float pose[6], int reps, float* us_per_launch);

/* Diagnostics: a full loop from `pose` (early_exit must be off), then `reps` back-to-back replays of its last registration
 * launch in the state the loop ended in; solve_prev != 0 closes the iteration before it in the launch's prologue each time
 * (the steady launch of the fused loop), 0 only rebuilds the transform.  Mean microseconds per replayed launch, gaps included. */
int  s2m_debug_time_steady(s2m_handle h, const float pose[6], int reps, int solve_prev, float* us_per_launch);

/* Diagnostics: `launches` > 0: that many k_register passes at `pose`, the last one recorded (1 = the pass
 * that inherits its prior from whatever ran before, 3 = steady state at this pose). `launches` < 0: a real LM
 * loop from `pose` exactly as s2m_optimize issues it, of which launch number N = -launches - 1 is recorded
 * (N = 0: the first launch of a scan, searching without a prior), including the fused close of the
 * iteration before it. Per wave (up to 64 locality-sorted scan points) S2M_PROF_WORDS words:
 * [0..3] wall clock (100 MHz) at start / after the search / after plane+Jacobian / at end; [4] search path
 * (1 LDS tile, 2 gather), [5] box rows, [6] points visited, [7] raw points; [8..12] ticks spent in
 * prior+box / row marking, points in the wave, staging, search; [13..15] path details; [16..22] wall clock
 * of the fused LM close: entry, partial sums reduced, normal equations, QR solved, update done, barrier
 * passed, transform built (0 when the launch closes nothing). Returns the number of waves written. */
#define S2M_PROF_WORDS 32
int  s2m_debug_wave_profile(s2m_handle h, const float pose[6], int launches, uint64_t* out, size_t cap_waves);


#ifdef __cplusplus
}
#endif
#endif /* LIORF_S2M_DEBUG_H */
