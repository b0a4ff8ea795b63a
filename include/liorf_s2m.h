/*
 * liorf_s2m.h — C ABI of the MI355X (gfx950) scan-to-map registration path.
 *
 * This is the drop-in boundary for ONE path of jimmyshe/liorf's mapOptimization
 * node: scan2MapOptimization() -> surfOptimization() + combineOptimizationCoeffs()
 * + LMOptimization() + transformUpdate()   (reference src/mapOptmization.cpp:1295-1363).
 * Everything behind these entry points runs as hand-written HIP kernels; there is
 * no CPU fallback (calls fail with S2M_ERR_NO_DEVICE when no gfx950 device is usable).
 *
 * Conventions
 *   - plain C types only: pointers, sizes, fixed-width ints, float/double.
 *   - every function returns an int status: 0 = S2M_OK, < 0 = error. The reference's
 *     three soft conditions (no map :1297, <= 30 features :1300, < 50 correspondences
 *     :1178) are NOT errors: status 0, pose unchanged, counters set in s2m_result.
 *   - point clouds are handed over as arrays of records of `stride_bytes` bytes whose
 *     first 12 bytes are float x,y,z. pcl::PointXYZI (the reference's PointType,
 *     include/utility.h:61) is stride 32; tightly packed xyz is stride 12.
 *   - pose vectors are float[6] = {roll, pitch, yaw, x, y, z}: the layout of the
 *     reference's transformTobeMapped[6] (src/mapOptmization.cpp:134, :337-351).
 *   - a handle is single-caller (the reference serialises this path under `mtx`,
 *     :252); separate handles are independent (one per GPU / per stream).
 *   - no C++ exception crosses this boundary.
 */
#ifndef LIORF_S2M_H
#define LIORF_S2M_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S2M_ABI_VERSION 1

/* status codes */
#define S2M_OK                 0
#define S2M_ERR_INVALID_ARG   -1
#define S2M_ERR_NO_DEVICE     -2   /* no usable HIP device / kernel image for it   */
#define S2M_ERR_HIP           -3   /* a HIP runtime call failed (see s2m_last_error) */
#define S2M_ERR_NO_SCAN       -4   /* *_resident call without s2m_set_scan          */
#define S2M_ERR_CAPACITY      -5   /* grid / buffer limit exceeded                   */
#define S2M_WARN_LEAF_TOO_SMALL 1   /* voxel filter: PCL's "leaf size is too small" case, output = input */

/* ScanContext descriptor shape (reference include/Scancontext.h:82-84) */
#define S2M_SC_NUM_RING    20
#define S2M_SC_NUM_SECTOR  60

typedef struct s2m_context* s2m_handle;

/*
 * Parameters. s2m_default_params() fills in the reference's constants; the
 * comments give the reference line each one restates (src/mapOptmization.cpp
 * unless another file is named).
 */
typedef struct s2m_params {
    uint32_t struct_size;     /* sizeof(s2m_params), set by s2m_default_params          */
    int32_t  device_id;       /* HIP device ordinal (default 0)                          */
    void*    stream;          /* optional caller-owned hipStream_t; NULL = library makes one */
    int32_t  k_neighbors;     /* 5: nearestKSearch(pointSel, 5, ...)              :1087  */
    double   gate_sq;         /* 1.0: pointSearchSqDis[4] < 1.0                   :1097  */
    double   plane_tol;       /* 0.2: |n.p_j + d| > 0.2 rejects the plane         :1118  */
    double   weight_scale;    /* 0.9: s = 1 - 0.9*|pd2|/sqrt(range)               :1127  */
    double   weight_min;      /* 0.1: keep iff s > 0.1                            :1135  */
    int32_t  min_corr;        /* 50: laserCloudSelNum < 50 -> return false        :1178  */
    int32_t  min_feats;       /* 30: laserCloudSurfLastDSNum > 30                 :1300  */
    int32_t  max_iter;        /* 30: iterCount < 30                               :1304  */
    float    eig_thresh;      /* 100: eignThre[6]                                 :1252  */
    double   conv_deg;        /* 0.05: deltaR < 0.05 (degrees)                    :1289  */
    double   conv_cm;         /* 0.05: deltaT < 0.05 (centimetres)                :1289  */
    float    z_tol;           /* z_tollerance, FLT_MAX default  include/utility.h:230     */
    float    rot_tol;         /* rotation_tollerance, FLT_MAX   include/utility.h:231     */
    int32_t  imu_type;        /* imuType (0: 6-axis, 1: 9-axis) include/utility.h:211     */
    float    imu_rpy_weight;  /* imuRPYWeight 0.01              include/utility.h:218     */
    int32_t  early_exit;      /* 1 = break when LMOptimization() returns true (:1313);
                                 0 = always run max_iter iterations (benchmark mode)      */
} s2m_params;

/*
 * The cloud_info fields the hot path reads (reference msg/cloud_info.msg:10-16;
 * read at src/mapOptmization.cpp:1325-1342 by transformUpdate()). NULL is
 * accepted wherever this struct is taken and means imuAvailable = 0.
 */
typedef struct s2m_imu_init {
    int64_t imuAvailable;     /* cloud_info.imuAvailable; the path tests `imuAvailable == true` (:1325), i.e. == 1 */
    float   imuRollInit;      /* cloud_info.imuRollInit  */
    float   imuPitchInit;     /* cloud_info.imuPitchInit */
    float   imuYawInit;       /* cloud_info.imuYawInit (not read on this path) */
} s2m_imu_init;

/* What scan2MapOptimization() leaves behind in the node's members. */
typedef struct s2m_result {
    int32_t iters_run;        /* LM iterations executed (1..max_iter), 0 if skipped       */
    int32_t converged;        /* 1 if LMOptimization() returned true                      */
    int32_t is_degenerate;    /* isDegenerate (:139) -> pose.covariance[0] (:1724-1727)   */
    int32_t n_sel_last;       /* laserCloudSelNum of the last executed iteration          */
    int32_t skipped;          /* 0 ran; 1 no map (:1297); 2 not enough features (:1300)   */
    float   pose[6];          /* transformTobeMapped after transformUpdate()              */
    float   affine[12];       /* incrementalOdometryAffineBack, row-major 3x4 (:1352)     */
} s2m_result;

/* One record per executed LM iteration (s2m_get_trace). */
typedef struct s2m_iter_trace {
    int32_t n_sel;            /* correspondences kept by surfOptimization()               */
    int32_t stepped;          /* 0 if n_sel < min_corr (pose unchanged), else 1           */
    float   delta[6];         /* matX after the degeneracy projection (:1266-1278)        */
    float   pose[6];          /* transformTobeMapped after this iteration                 */
    float   deltaR, deltaT;   /* :1280-1287                                               */
} s2m_iter_trace;

/* ---- lifecycle ---------------------------------------------------------- */
const char* s2m_version(void);
int  s2m_default_params(s2m_params* p);
int  s2m_create(const s2m_params* p, s2m_handle* out);
int  s2m_destroy(s2m_handle h);
const char* s2m_last_error(s2m_handle h);       /* valid until the next call on h */
/* The reference reads its ParamServer members (z_tollerance, rotation_tollerance, imuType, imuRPYWeight,
 * include/utility.h:211-233) every time the path runs; a caller that changes them after s2m_create applies the new
 * values here. Everything except device_id, stream, k_neighbors and gate_sq (fixed at creation: the search grid is
 * built for the gate) may change between scans; takes effect with the next s2m_optimize* call. */
int  s2m_set_params(s2m_handle h, const s2m_params* p);
int  s2m_get_params(s2m_handle h, s2m_params* out);

/* ---- inputs ------------------------------------------------------------- */
/* Replaces kdtreeSurfFromMap->setInputCloud(laserCloudSurfFromMapDS) (:1302):
 * uploads the local surf map and builds the device neighbour-search index.
 * n == 0 is allowed and models "cloudKeyPoses3D->points.empty()" (:1297). */
int  s2m_set_map(s2m_handle h, const void* pts, size_t n, size_t stride_bytes);
/* Same, for a map that already lives in device memory (hipMalloc'd). */
int  s2m_set_map_device(s2m_handle h, const void* d_pts, size_t n, size_t stride_bytes);
/* Uploads laserCloudSurfLastDS (lidar-frame points) for the *_resident calls. */
int  s2m_set_scan(s2m_handle h, const void* pts, size_t n, size_t stride_bytes);
int  s2m_set_scan_device(s2m_handle h, const void* d_pts, size_t n, size_t stride_bytes);

/* ---- the path ----------------------------------------------------------- */
/* scan2MapOptimization() (:1295-1321) on host buffers: s2m_set_scan() followed
 * by s2m_optimize_resident(). pose is in/out (= transformTobeMapped). */
int  s2m_optimize(s2m_handle h, const void* scan, size_t n, size_t stride_bytes,
                  float pose[6], const s2m_imu_init* imu, s2m_result* out);
/* The <= max_iter x {surfOptimization, combineOptimizationCoeffs, LMOptimization}
 * loop (:1304-1315) plus transformUpdate() (:1317) on the resident scan + map.
 * Runs entirely on the device; one synchronisation at the end. */
int  s2m_optimize_resident(s2m_handle h, float pose[6], const s2m_imu_init* imu,
                           s2m_result* out);
/* Asynchronous form for throughput measurement: enqueue the same device work on
 * the handle's stream and return without synchronising; s2m_optimize_collect()
 * synchronises and fills the outputs of the most recent launch. */
int  s2m_optimize_launch(s2m_handle h, const float pose[6]);
int  s2m_optimize_collect(s2m_handle h, float pose[6], const s2m_imu_init* imu,
                          s2m_result* out);
/* ---- a batch of scans against one resident map (BASELINE config 4 on one GPU; the multi-GPU form shards scans over
 * ranks, liorf_amd/host/s2m_multi_gpu.cpp) --------------------------------------------------------------------------
 * The reference registers one scan at a time (laserCloudInfoHandler holds `mtx`, :252); scans of a batch share nothing but the
 * read-only local map, so n_scans scan2MapOptimization() calls can be in flight at once: every scan
 * slot has its own buffers, loop state and trace, all slots search the map installed with s2m_set_map / s2m_extract_cloud on
 * `h`, and the n_scans LM loops advance in lockstep inside ONE captured graph (one launch, one synchronisation): every kernel
 * launch of the loop carries one grid row per scan, so that while one scan's workgroups wait on a dependent load the others'
 * points are processed.
 * Results are those of n_scans separate s2m_optimize calls, bit for bit.
 *   scans[b], sizes[b]    laserCloudSurfLastDS of scan b (host records, stride_bytes as everywhere)
 *   poses[6*b .. 6*b+5]   in: initial guess of scan b, out: its transformTobeMapped
 *   imu                   NULL, or n_scans entries;  out: NULL, or n_scans entries */
int  s2m_optimize_batch(s2m_handle h, int n_scans, const void* const* scans, const size_t* sizes, size_t stride_bytes,
                        float* poses, const s2m_imu_init* imu, s2m_result* out);
/* The same in steps: install scan b in slot b (host or device records; a device source must stay valid until the collect),
 * enqueue the batch without synchronising, synchronise and fetch. */
int  s2m_batch_set_scan(s2m_handle h, int slot, const void* pts, size_t n, size_t stride_bytes, int on_device);
/* All slots 0 .. n_scans-1 at once: the ordering kernels of the n_scans scans share their launches (six launches for the batch
 * instead of six per scan; the host side of a batch of eight scans is otherwise a quarter of its wall time). */
int  s2m_batch_set_scans(s2m_handle h, int n_scans, const void* const* scans, const size_t* sizes, size_t stride_bytes, int on_device);
int  s2m_optimize_batch_launch(s2m_handle h, int n_scans, const float* poses);
int  s2m_optimize_batch_collect(s2m_handle h, int n_scans, float* poses, const s2m_imu_init* imu, s2m_result* out);
int  s2m_batch_get_trace(s2m_handle h, int slot, s2m_iter_trace* out, int cap);
/* A stream of scans against the resident map, one in flight per slot: slot k's preparation (scan ordering, wave table) and its
 * loop run on slot k's own stream, so the preparation of scan i+1 in one slot overlaps the LM loop of scan i in the other - the
 * reference does downsampleCurrentScan() and scan2MapOptimization() strictly one after the other (:257-265).  Typical use:
 *     s2m_slot_set_scan(h, 0, scan0 ..);
 *     for i:  s2m_slot_optimize_launch(h, i & 1, guess_i);  s2m_slot_set_scan(h, (i + 1) & 1, scan_{i+1} ..);  s2m_slot_optimize_collect(h, i & 1, ..)
 * Every result is bitwise that of s2m_optimize on the same scan.  The map must not be replaced while a slot is in flight. */
int  s2m_slot_set_scan(s2m_handle h, int slot, const void* pts, size_t n, size_t stride_bytes, int on_device);
int  s2m_slot_optimize_launch(s2m_handle h, int slot, const float pose[6]);
int  s2m_slot_optimize_collect(s2m_handle h, int slot, float pose[6], const s2m_imu_init* imu, s2m_result* out);

/* Per-iteration records of the last optimize call; returns the count (<= cap). */
int  s2m_get_trace(s2m_handle h, s2m_iter_trace* out, int cap);

/* ---- observation hooks (used by the parity tests) ----------------------- */
/* One surfOptimization() pass (:1074-1143) at `pose` on the resident scan + map.
 * Outputs are in the scan's ORIGINAL point order; any may be NULL.
 *   idx5   [n*5]  neighbour indices into the map as given to s2m_set_map, ascending d2
 *   d2_5   [n*5]  squared distances (fp32, ((dx^2+dy^2)+dz^2))
 *   flag   [n]    laserCloudOriSurfFlag (:1138)
 *   coeff4 [n*4]  coeffSelSurfVec: s*pa, s*pb, s*pc, s*pd2 (:1130-1133); 0 if !flag
 * idx5/d2_5 are defined for queries whose 5th neighbour passes the gate
 * (d2 < gate_sq); for the others idx5 = -1 (the reference never reads them). */
int  s2m_surf_optimization(s2m_handle h, const float pose[6],
                           int32_t* idx5, float* d2_5, uint8_t* flag, float* coeff4);
/* matAtA / matAtB / laserCloudSelNum of one iteration at `pose` (:1182-1239). */
int  s2m_normal_eq(s2m_handle h, const float pose[6], float AtA[36], float AtB[6],
                   int32_t* n_sel);
/* Raw device time (ms) of the last s2m_optimize* call, measured with HIP events
 * on the handle's stream; and of the last s2m_set_map / s2m_set_scan index build. */
int  s2m_last_timing(s2m_handle h, float* optimize_ms, float* set_map_ms, float* set_scan_ms);
/* Diagnostic and benchmark entry points (per-launch timing, per-wave profiles, the device's trig arithmetic, experiment
 * switches read from the environment) are declared in liorf_s2m_debug.h: they are exported by the same library but are not part
 * of the boundary a node binds. */

/* ---- The voxel-grid stages either side of the path ----------
 * pcl::VoxelGrid<pcl::PointXYZI>::applyFilter with the reference's settings (all fields averaged,
 * no minimum point count): one output record {centroid x, y, z, 1.0f, mean intensity, 0...} per occupied
 * voxel, in ascending voxel-index order like PCL. Input records carry intensity at byte 16 when
 * stride_bytes >= 20 (pcl::PointXYZI). Points with a non-finite coordinate are skipped. Within a voxel the
 * points are summed in ascending input order (PCL's unstable std::sort leaves that order unspecified).
 * `cap` is the capacity of `out` in records; *n_out is always the number of voxels, and a result that does
 * not fit returns S2M_ERR_CAPACITY after writing the first `cap` records. S2M_WARN_LEAF_TOO_SMALL (> 0)
 * reports PCL's index-overflow case, where the output is the unfiltered input. */

/* downSizeFilter*.setInputCloud(cloud); .filter(out) for a host cloud (reference :1064-1065, :1037-1038). */
int  s2m_voxel_downsample(s2m_handle h, const void* pts, size_t n, size_t stride_bytes, float leaf,
                          void* out, size_t out_stride_bytes, size_t cap, size_t* n_out);
/* Same with both clouds in device memory (no host copies; returns after the handle's stream has drained). */
int  s2m_voxel_downsample_device(s2m_handle h, const void* d_pts, size_t n, size_t stride_bytes, float leaf,
                                 void* d_out, size_t out_stride_bytes, size_t cap, size_t* n_out);
/* downsampleCurrentScan() (reference :1061-1067) fused with s2m_set_scan: filters laserCloudSurfLast
 * (host records, or device records when on_device != 0) with leaf mappingSurfLeafSize and installs the result
 * as the scan of the next s2m_optimize_resident / s2m_optimize_launch. If cap > 0 the filtered cloud
 * (laserCloudSurfLastDS, which the node later stores as the key frame's cloud) is also copied to host `out`. */
int  s2m_downsample_scan(s2m_handle h, const void* pts, size_t n, size_t stride_bytes, int on_device, float leaf,
                         void* out, size_t out_stride_bytes, size_t cap, size_t* n_out);
/* extractCloud() (reference :1014-1039) after the key-frame selection, fused with s2m_set_map: for the
 * chosen key frames f = 0..n_frames-1, transformPointCloud(frames[f], pose f) (:310-329; poses_xyzrpy holds
 * {x, y, z, roll, pitch, yaw} of each PointTypePose), concatenated in that order, filtered with leaf
 * surroundingKeyframeMapLeafSize, and installed as the local surf map (the reference's kd-tree build,
 * :1302). frames[] are host buffers, or device buffers when on_device != 0 (a device-resident key-frame
 * store replaces the reference's laserCloudMapContainer cache: re-transforming is cheaper than caching).
 * If cap > 0 the filtered map (laserCloudSurfFromMapDS) is also copied to host `out`. */
int  s2m_extract_cloud(s2m_handle h, int n_frames, const void* const* frames, const size_t* frame_sizes,
                       size_t stride_bytes, int on_device, const float* poses_xyzrpy, float leaf,
                       void* out, size_t out_stride_bytes, size_t cap, size_t* n_out);
/* transformPointCloud(cloudIn, transformIn) (reference :310-329) for one host cloud; out holds n records. */
int  s2m_transform_cloud(s2m_handle h, const void* pts, size_t n, size_t stride_bytes, const float pose_xyzrpy[6],
                         void* out, size_t out_stride_bytes);

/* ---- ScanContext descriptor (BASELINE config 5) ------------------------- */
/* SCManager::makeScancontext + makeRingkeyFromScancontext
 * (reference include/Scancontext.cpp:151-211): desc is 20x60 row-major doubles,
 * ringkey 20 doubles. Points are host records as above. */
int  s2m_make_scancontext(s2m_handle h, const void* pts, size_t n, size_t stride_bytes,
                          double desc[S2M_SC_NUM_RING * S2M_SC_NUM_SECTOR],
                          double ringkey[S2M_SC_NUM_RING]);

/* ---- ScanContext matching ----------------------------------------------
 * The SCManager's containers (reference include/Scancontext.h:102-113) kept on the device, and
 * SCManager::detectLoopClosureID (include/Scancontext.cpp:253-344) with distanceBtnScanContext /
 * fastAlignUsingVkey / distDirectSC (:69-148) as one kernel. Key frames are numbered in the order they are
 * added. The reference rebuilds its ring-key kd-tree every 10th detection (TREE_MAKING_PERIOD_); the same
 * staleness is kept: between rebuilds the search sees the key frames that existed at the last rebuild,
 * minus the 30 most recent (NUM_EXCLUDE_RECENT). The 3 ring-key neighbours (NUM_CANDIDATES_FROM_TREE) are an
 * exact fp32 3-NN in nanoflann's accumulation order; equal distances go to the lower index. */
typedef struct s2m_sc_match {
    double  min_dist;        /* best distanceBtnScanContext over the candidates (10000000 if none)  */
    int32_t nn_idx;          /* its key frame                                                        */
    int32_t nn_align;        /* its column shift (yaw difference in units of 6 degrees)             */
    int32_t cand_idx[3];     /* ring-key neighbours, ascending distance                              */
    float   cand_d2[3];      /* their squared ring-key distances                                     */
} s2m_sc_match;
int  s2m_sc_reset(s2m_handle h);
int  s2m_sc_size(s2m_handle h);                                   /* key frames stored, or a negative status */
/* makeAndSaveScancontextAndKeys(scan) (:236-250): descriptor, ring key and sector key of a host cloud, appended. */
int  s2m_sc_add_scan(s2m_handle h, const void* pts, size_t n, size_t stride_bytes);
/* The same for a descriptor computed elsewhere (20x60 row-major doubles). */
int  s2m_sc_add_descriptor(s2m_handle h, const double desc[S2M_SC_NUM_RING * S2M_SC_NUM_SECTOR]);
/* detectLoopClosureID() for the newest key frame: *loop_id = matching key frame or -1, *yaw_diff_rad as the
 * reference returns it (also when there is no loop); `detail` (optional) receives the intermediate values. */
int  s2m_sc_detect_loop(s2m_handle h, int32_t* loop_id, float* yaw_diff_rad, s2m_sc_match* detail);
/* distanceBtnScanContext(query, candidate k) for m stored candidates at once (one workgroup each): the
 * batched form in which this row is worth running on a GPU. */
int  s2m_sc_distance(s2m_handle h, int32_t query_idx, const int32_t* cand_idx, int32_t m, double* dist, int32_t* shift);

/* ---- ICP loop-closure alignment ------------------------------------------
 * pcl::IterativeClosestPoint<PointType, PointType> as performRSLoopClosure / performSCLoopClosure configure
 * and run it (reference src/mapOptmization.cpp:571-586, :663-678): setMaxCorrespondenceDistance,
 * setMaximumIterations, setTransformationEpsilon, setEuclideanFitnessEpsilon, RANSAC off, identity guess,
 * align(); then hasConverged(), getFitnessScore(), getFinalTransformation(). src = cureKeyframeCloud, tgt =
 * prevKeyframeCloud (host records). T is the row-major 4x4 final transformation (source -> target). As in
 * PCL, reaching max_iterations counts as converged; fewer than 3 correspondences does not. */
typedef struct s2m_icp_params {
    double  max_correspondence_distance;   /* historyKeyframeSearchRadius * 2 (:573)  */
    int32_t max_iterations;                /* 100 (:574)                              */
    double  transformation_epsilon;        /* 1e-6 (:575)                             */
    double  euclidean_fitness_epsilon;     /* 1e-6 (:576)                             */
} s2m_icp_params;
typedef struct s2m_icp_result {
    float   T[16];
    int32_t converged;
    int32_t iterations;
    double  fitness_score;                 /* getFitnessScore(): mean squared nearest-neighbour distance after alignment */
} s2m_icp_result;
int  s2m_icp_default_params(s2m_icp_params* p);
int  s2m_icp_align(s2m_handle h, const void* src, size_t n_src, const void* tgt, size_t n_tgt, size_t stride_bytes,
                   const s2m_icp_params* p /* NULL = defaults */, s2m_icp_result* out);

#ifdef __cplusplus
}
#endif
#endif /* LIORF_S2M_H */
