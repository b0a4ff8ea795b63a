/*
 * s2m_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's scan-to-map registration path
 * (jimmyshe/liorf src/mapOptmization.cpp:302-308, 348-351, 1069-1363) and of the
 * ScanContext descriptor build (include/Scancontext.cpp:23-36, 151-211).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker / the timed CPU baseline. The product
 * (liorf_amd/, include/liorf_s2m.h) never links, imports or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for
 * this path and cannot be built here (needs ROS, PCL/FLANN,
 * Eigen, OpenCV, GTSAM, none present). The arithmetic that the reference delegates
 * to those un-vendored, un-versioned libraries is restated from their published
 * algorithms (named at each function); it is cross-checked by independent means
 * (brute-force kNN, the reference's vendored nanoflann compiled into oracle/_ref,
 * fp64 numpy solves, finite-difference Jacobians) in tests/, not by reference
 * outputs.
 */
#ifndef S2M_ORACLE_H
#define S2M_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_params {
    double  gate_sq;        /* 1.0   :1097 */
    double  plane_tol;      /* 0.2   :1118 */
    double  weight_scale;   /* 0.9   :1127 */
    double  weight_min;     /* 0.1   :1135 */
    int32_t min_corr;       /* 50    :1178 */
    int32_t min_feats;      /* 30    :1300 */
    int32_t max_iter;       /* 30    :1304 */
    float   eig_thresh;     /* 100   :1252 */
    double  conv_deg;       /* 0.05  :1289 */
    double  conv_cm;        /* 0.05  :1289 */
    float   z_tol;          /* include/utility.h:230 */
    float   rot_tol;        /* include/utility.h:231 */
    int32_t imu_type;       /* include/utility.h:211 */
    float   imu_rpy_weight; /* include/utility.h:218 */
    int32_t early_exit;     /* 1 = reference behaviour */
    int32_t num_threads;    /* numberOfCores (:1078; yaml 4, default 2) */
    int32_t knn_backend;    /* 0 = brute force (ground truth), 1 = kd-tree (timed baseline) */
} orc_params;

typedef struct orc_imu_init {
    int64_t imuAvailable;
    float   imuRollInit, imuPitchInit, imuYawInit;
} orc_imu_init;

typedef struct orc_result {
    int32_t iters_run, converged, is_degenerate, n_sel_last, skipped;
    float   pose[6];
    float   affine[12];
} orc_result;

typedef struct orc_iter_trace {
    int32_t n_sel, stepped;
    float   delta[6];
    float   pose[6];
    float   deltaR, deltaT;
} orc_iter_trace;

/* seconds accumulated per stage by orc_scan2MapOptimization */
typedef struct orc_timing {
    double tree_build, knn_plane, compaction, jacobian_solve, total;
} orc_timing;

typedef struct orc_ctx orc_ctx;

void     orc_default_params(orc_params* p);
orc_ctx* orc_create(const orc_params* p);
void     orc_destroy(orc_ctx* c);
/* laserCloudSurfFromMapDS + kdtreeSurfFromMap->setInputCloud (:1302) */
void     orc_set_map(orc_ctx* c, const void* pts, size_t n, size_t stride_bytes);
/* laserCloudSurfLastDS */
void     orc_set_scan(orc_ctx* c, const void* pts, size_t n, size_t stride_bytes);
void     orc_set_pose(orc_ctx* c, const float pose[6]);
void     orc_get_pose(const orc_ctx* c, float pose[6]);

/* pcl::getTransformation(x,y,z,roll,pitch,yaw) via trans2Affine3f (:348-351) */
void     orc_getTransformation(const float pose_rpyxyz[6], float T[12]);
/* one step each, named after the reference functions */
void     orc_surfOptimization(orc_ctx* c);               /* :1074-1143 */
void     orc_combineOptimizationCoeffs(orc_ctx* c);      /* :1145-1156 */
int      orc_LMOptimization(orc_ctx* c, int iterCount);  /* :1158-1293 */
void     orc_transformUpdate(orc_ctx* c, const orc_imu_init* imu);  /* :1323-1353 */
void     orc_scan2MapOptimization(orc_ctx* c, const orc_imu_init* imu, orc_result* out); /* :1295-1321 */

/* observation (original scan order) */
size_t   orc_num_queries(const orc_ctx* c);
void     orc_get_surf_outputs(const orc_ctx* c, int32_t* idx5, float* d2_5,
                              uint8_t* flag, float* coeff4);
int      orc_get_normal_eq(const orc_ctx* c, float AtA[36], float AtB[6]);  /* returns laserCloudSelNum */
int      orc_get_trace(const orc_ctx* c, orc_iter_trace* out, int cap);
void     orc_get_timing(const orc_ctx* c, orc_timing* t);
void     orc_get_matP(const orc_ctx* c, float matP[36], int* isDegenerate);

/* stand-alone pieces exposed for unit tests */
void     orc_knn5_brute(const float* map_xyz, size_t n_m, const float q[3], int32_t idx[5], float d2[5]);
void     orc_knn5_kdtree(orc_ctx* c, const float q[3], int32_t idx[5], float d2[5]);
void     orc_plane_fit_5x3(const float nbr_xyz[15], float x[3]);          /* colPivHouseholderQr().solve(-1) :1104 */
int      orc_solve6_qr(const float A[36], const float b[6], float x[6]);  /* cv::solve(DECOMP_QR) :1240 */
void     orc_eigen6_sym(const float A[36], float evals[6], float evecs[36]); /* cv::eigen :1248 */
int      orc_inv6_lu(const float A[36], float Ainv[36]);                  /* cv::Mat::inv :1263 */
void     orc_jacobian_row(const float pose[6], const float p_ori[3], const float coeff[4],
                          float row[6], float* rhs);                      /* :1216-1234 */

/* ScanContext (include/Scancontext.cpp:23-36, 151-211) */
float    orc_xy2theta(float x, float y);
void     orc_makeScancontext(const void* pts, size_t n, size_t stride_bytes, double desc[20 * 60]);
void     orc_makeRingkeyFromScancontext(const double desc[20 * 60], double key[20]);

/* section 8(f) rows F1/F2: pcl::VoxelGrid<PointXYZI>::applyFilter [ext] (:1037-1039, :1061-1067) and
 * transformPointCloud (:310-329). orc_voxelGrid returns 0 ok, 1 "leaf size too small" (output = input),
 * -1 output capacity too small (*n_out = needed). pose_xyzrpy = PointTypePose {x, y, z, roll, pitch, yaw}. */
int      orc_voxelGrid(const void* pts, size_t n, size_t stride_bytes, float leaf,
                       void* out, size_t out_stride_bytes, size_t cap, size_t* n_out);
void     orc_transformPointCloud(const void* pts, size_t n, size_t stride_bytes, const float pose_xyzrpy[6],
                                 void* out, size_t out_stride_bytes);

/* section 8(f) row F3: ScanContext matching (include/Scancontext.cpp:69-148, 214-344) */
typedef struct orc_sc orc_sc;
void     orc_makeSectorkeyFromScancontext(const double desc[20 * 60], double key[60]);
double   orc_distDirectSC_shifted(const double sc1[20 * 60], const double sc2[20 * 60], int shift);
int      orc_fastAlignUsingVkey(const double vkey1[60], const double vkey2[60]);
void     orc_distanceBtnScanContext(const double sc1[20 * 60], const double sc2[20 * 60], double* dist, int* shift);
orc_sc*  orc_sc_create(void);
void     orc_sc_destroy(orc_sc* m);
size_t   orc_sc_size(const orc_sc* m);
void     orc_sc_add_descriptor(orc_sc* m, const double desc[20 * 60]);
void     orc_sc_add_scan(orc_sc* m, const void* pts, size_t n, size_t stride_bytes);
int      orc_sc_detectLoopClosureID(orc_sc* m, float* yaw_diff_rad, double* min_dist, int* nn_idx, int* nn_align,
                                    int cand_idx[3], float cand_d2[3]);

/* section 8(f) row F4: pcl::IterativeClosestPoint as configured at src/mapOptmization.cpp:571-586 [ext PCL 1.10] */
void     orc_icp_umeyama(const float mean_src[3], const float mean_tgt[3], const float sigma[9], float T[16]);
int      orc_icp_align(const void* src, size_t n_src, const void* tgt, size_t n_tgt, size_t stride_bytes,
                       double max_corr_dist, int max_iter, double trans_eps, double fit_eps, int num_threads,
                       float T_final[16], int* converged, double* fitness, int* iterations);

#ifdef __cplusplus
}
#endif
#endif
