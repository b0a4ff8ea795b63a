/*
 * s2m_oracle.c — CPU ORACLE (test infrastructure, NOT product code). See s2m_oracle.h.
 * PARITY UNPINNED (no reference tests/fixtures exist for this path).
 *
 * Build: gcc -O3 -fopenmp -ffp-contract=off -fPIC -shared (oracle/Makefile).
 * -ffp-contract=off: the reference is built `-O3` for baseline x86-64
 * (CMakeLists.txt:7, no -march), which has no FMA, so every fp32 expression below
 * is evaluated as written, one rounding per operation.
 *
 * All file:line citations are into /root/reference (src/mapOptmization.cpp unless
 * another file is named). "[ext]" marks arithmetic the reference delegates to a
 * library that is not vendored (PCL, FLANN, Eigen, OpenCV, tf); it is restated from
 * the library's published algorithm.
 */
#include "s2m_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* kd-tree (timed baseline back-end): single exact tree, leaf 15, L2          */
/* [ext] stands in for pcl::KdTreeFLANN -> flann::KDTreeSingleIndex (:1302,   */
/* :1087). Ties in d2 are broken by the smaller map index in BOTH back-ends   */
/* (FLANN's tie order is traversal dependent and unobservable here).          */
/* ------------------------------------------------------------------------- */
#define ORC_LEAF_MAX 15

typedef struct kd_node {
    int32_t left, right;     /* children, -1 for a leaf        */
    int32_t lo, hi;          /* leaf: range in vind [lo,hi)    */
    int32_t dim;
    float   divlow, divhigh; /* gap between the two children   */
} kd_node;

typedef struct kd_tree {
    kd_node* nodes;
    int32_t  n_nodes, cap_nodes;
    int32_t* vind;
    float    bbox_lo[3], bbox_hi[3];
} kd_tree;

struct orc_ctx {
    orc_params p;
    /* laserCloudSurfFromMapDS */
    float*  map; size_t n_m;
    kd_tree tree; int have_tree;
    /* laserCloudSurfLastDS */
    float*  scan; size_t n_q;
    /* laserCloudOriSurfVec / coeffSelSurfVec / laserCloudOriSurfFlag (:110-115, :218-222) */
    float*   oriSurfVec; float* coeffSurfVec; uint8_t* surfFlag;
    int32_t* idx5; float* d2_5;
    /* laserCloudOri / coeffSel (:110-111) */
    float*  cloudOri; float* coeffSel; int32_t selNum;
    /* node state */
    float   transformTobeMapped[6];           /* :134 */
    float   transPointAssociateToMap[12];     /* :142 */
    float   incrementalOdometryAffineBack[12];/* :157 */
    int     isDegenerate;                     /* :139 */
    float   matP[36];                         /* :140 */
    float   matAtA[36], matAtB[6];
    orc_iter_trace trace[64]; int n_trace;
    orc_timing tm;
};

static double now_s(void)
{
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void orc_default_params(orc_params* p)
{
    memset(p, 0, sizeof(*p));
    p->gate_sq = 1.0; p->plane_tol = 0.2; p->weight_scale = 0.9; p->weight_min = 0.1;
    p->min_corr = 50; p->min_feats = 30; p->max_iter = 30; p->eig_thresh = 100.0f;
    p->conv_deg = 0.05; p->conv_cm = 0.05;
    p->z_tol = FLT_MAX; p->rot_tol = FLT_MAX; p->imu_type = 0; p->imu_rpy_weight = 0.01f;
    p->early_exit = 1; p->num_threads = 4; p->knn_backend = 1;
}

orc_ctx* orc_create(const orc_params* p)
{
    orc_ctx* c = (orc_ctx*)calloc(1, sizeof(orc_ctx));
    if (p) c->p = *p; else orc_default_params(&c->p);
    /* matP starts all-zero (:233): calloc */
    return c;
}

static void kd_free(kd_tree* t) { free(t->nodes); free(t->vind); memset(t, 0, sizeof(*t)); }

void orc_destroy(orc_ctx* c)
{
    if (!c) return;
    kd_free(&c->tree);
    free(c->map); free(c->scan); free(c->oriSurfVec); free(c->coeffSurfVec); free(c->surfFlag);
    free(c->idx5); free(c->d2_5); free(c->cloudOri); free(c->coeffSel);
    free(c);
}

/* ---- top-5 set ordered by (d2, idx) ------------------------------------- */
typedef struct top5 { float d2[5]; int32_t idx[5]; } top5;

static inline void top5_init(top5* t)
{
    for (int i = 0; i < 5; i++) { t->d2[i] = INFINITY; t->idx[i] = INT32_MAX; }
}
static inline int pair_lt(float da, int32_t ia, float db, int32_t ib)
{
    return da < db || (da == db && ia < ib);
}
static inline void top5_push(top5* t, float d2, int32_t idx)
{
    if (!pair_lt(d2, idx, t->d2[4], t->idx[4])) return;
    int j = 4;
    while (j > 0 && pair_lt(d2, idx, t->d2[j - 1], t->idx[j - 1])) {
        t->d2[j] = t->d2[j - 1]; t->idx[j] = t->idx[j - 1]; j--;
    }
    t->d2[j] = d2; t->idx[j] = idx;
}

/* squared L2 distance, accumulation order of flann::L2_Simple [ext]:
 * result = 0; result += dx*dx; += dy*dy; += dz*dz  (cf. include/nanoflann.hpp:432-440) */
static inline float dist2(const float* a, const float* b)
{
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return (dx * dx + dy * dy) + dz * dz;
}

void orc_knn5_brute(const float* map_xyz, size_t n_m, const float q[3], int32_t idx[5], float d2[5])
{
    top5 t; top5_init(&t);
    for (size_t j = 0; j < n_m; j++) top5_push(&t, dist2(q, map_xyz + 3 * j), (int32_t)j);
    for (int i = 0; i < 5; i++) { idx[i] = t.idx[i]; d2[i] = t.d2[i]; }
}

/* ---- kd-tree build ------------------------------------------------------- */
static int32_t kd_new_node(kd_tree* t)
{
    if (t->n_nodes == t->cap_nodes) {
        t->cap_nodes = t->cap_nodes ? 2 * t->cap_nodes : 1024;
        t->nodes = (kd_node*)realloc(t->nodes, sizeof(kd_node) * (size_t)t->cap_nodes);
    }
    kd_node* n = &t->nodes[t->n_nodes];
    n->left = n->right = -1; n->lo = n->hi = 0; n->dim = 0; n->divlow = n->divhigh = 0.0f;
    return t->n_nodes++;
}

static int32_t kd_divide(kd_tree* t, const float* pts, int32_t lo, int32_t hi)
{
    int32_t id = kd_new_node(t);
    if (hi - lo <= ORC_LEAF_MAX) { t->nodes[id].lo = lo; t->nodes[id].hi = hi; return id; }
    float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int32_t i = lo; i < hi; i++) {
        const float* p = pts + 3 * (size_t)t->vind[i];
        for (int d = 0; d < 3; d++) { if (p[d] < mn[d]) mn[d] = p[d]; if (p[d] > mx[d]) mx[d] = p[d]; }
    }
    int dim = 0; float span = mx[0] - mn[0];
    for (int d = 1; d < 3; d++) if (mx[d] - mn[d] > span) { span = mx[d] - mn[d]; dim = d; }
    if (!(span > 0.0f)) { t->nodes[id].lo = lo; t->nodes[id].hi = hi; return id; } /* all coincident */
    float split = 0.5f * (mn[dim] + mx[dim]);
    /* partition: < split to the left */
    int32_t i = lo, j = hi - 1;
    while (i <= j) {
        while (i <= j && pts[3 * (size_t)t->vind[i] + dim] < split) i++;
        while (i <= j && pts[3 * (size_t)t->vind[j] + dim] >= split) j--;
        if (i < j) { int32_t tmp = t->vind[i]; t->vind[i] = t->vind[j]; t->vind[j] = tmp; i++; j--; }
    }
    int32_t mid = i;
    if (mid == lo || mid == hi) mid = lo + (hi - lo) / 2;   /* cannot happen with span > 0; guard */
    float lmax = -INFINITY, rmin = INFINITY;
    for (int32_t k = lo; k < mid; k++) { float v = pts[3 * (size_t)t->vind[k] + dim]; if (v > lmax) lmax = v; }
    for (int32_t k = mid; k < hi; k++) { float v = pts[3 * (size_t)t->vind[k] + dim]; if (v < rmin) rmin = v; }
    int32_t l = kd_divide(t, pts, lo, mid);
    int32_t r = kd_divide(t, pts, mid, hi);
    kd_node* n = &t->nodes[id];
    n->left = l; n->right = r; n->dim = dim; n->divlow = lmax; n->divhigh = rmin;
    return id;
}

static void kd_build(kd_tree* t, const float* pts, size_t n)
{
    kd_free(t);
    t->vind = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
    for (size_t i = 0; i < n; i++) t->vind[i] = (int32_t)i;
    for (int d = 0; d < 3; d++) { t->bbox_lo[d] = INFINITY; t->bbox_hi[d] = -INFINITY; }
    for (size_t i = 0; i < n; i++)
        for (int d = 0; d < 3; d++) {
            float v = pts[3 * i + d];
            if (v < t->bbox_lo[d]) t->bbox_lo[d] = v;
            if (v > t->bbox_hi[d]) t->bbox_hi[d] = v;
        }
    if (n) kd_divide(t, pts, 0, (int32_t)n);
}

/* exact search: a sub-tree is skipped only when its lower bound (kept in fp64,
 * shrunk by a relative slack that covers fp32 rounding of dist2) is strictly
 * worse than the current 5th candidate, so results equal brute force bit for bit. */
static void kd_search(const kd_tree* t, const float* pts, int32_t id, const float q[3],
                      double mind, double dists[3], top5* res)
{
    const kd_node* n = &t->nodes[id];
    if (n->left < 0) {
        for (int32_t i = n->lo; i < n->hi; i++) {
            int32_t j = t->vind[i];
            top5_push(res, dist2(q, pts + 3 * (size_t)j), j);
        }
        return;
    }
    int d = n->dim;
    double val = q[d], diff1 = val - (double)n->divlow, diff2 = val - (double)n->divhigh;
    int32_t best, other; double cut;
    if (diff1 + diff2 < 0) { best = n->left; other = n->right; cut = diff2 * diff2; }
    else                   { best = n->right; other = n->left; cut = diff1 * diff1; }
    kd_search(t, pts, best, q, mind, dists, res);
    double dst = dists[d];
    double mind2 = mind + cut - dst;
    dists[d] = cut;
    if (mind2 * (1.0 - 1e-5) <= (double)res->d2[4]) kd_search(t, pts, other, q, mind2, dists, res);
    dists[d] = dst;
}

void orc_knn5_kdtree(orc_ctx* c, const float q[3], int32_t idx[5], float d2[5])
{
    top5 t; top5_init(&t);
    if (c->n_m) {
        double dists[3] = { 0, 0, 0 }, mind = 0;
        for (int d = 0; d < 3; d++) {
            if (q[d] < c->tree.bbox_lo[d]) { double e = (double)q[d] - c->tree.bbox_lo[d]; dists[d] = e * e; }
            if (q[d] > c->tree.bbox_hi[d]) { double e = (double)q[d] - c->tree.bbox_hi[d]; dists[d] = e * e; }
            mind += dists[d];
        }
        kd_search(&c->tree, c->map, 0, q, mind, dists, &t);
    }
    for (int i = 0; i < 5; i++) { idx[i] = t.idx[i]; d2[i] = t.d2[i]; }
}

/* ---- inputs --------------------------------------------------------------- */
static float* copy_xyz(const void* pts, size_t n, size_t stride)
{
    float* out = (float*)malloc(sizeof(float) * 3 * (n ? n : 1));
    const unsigned char* b = (const unsigned char*)pts;
    for (size_t i = 0; i < n; i++) memcpy(out + 3 * i, b + i * stride, 12);
    return out;
}

void orc_set_map(orc_ctx* c, const void* pts, size_t n, size_t stride_bytes)
{
    free(c->map); c->map = copy_xyz(pts, n, stride_bytes); c->n_m = n;
    double t0 = now_s();
    kd_build(&c->tree, c->map, n);          /* kdtreeSurfFromMap->setInputCloud :1302 */
    c->have_tree = 1;
    c->tm.tree_build = now_s() - t0;
}

void orc_set_scan(orc_ctx* c, const void* pts, size_t n, size_t stride_bytes)
{
    free(c->scan); c->scan = copy_xyz(pts, n, stride_bytes); c->n_q = n;
    size_t m = n ? n : 1;
    c->oriSurfVec   = (float*)realloc(c->oriSurfVec, sizeof(float) * 3 * m);
    c->coeffSurfVec = (float*)realloc(c->coeffSurfVec, sizeof(float) * 4 * m);
    c->surfFlag     = (uint8_t*)realloc(c->surfFlag, m);
    c->idx5         = (int32_t*)realloc(c->idx5, sizeof(int32_t) * 5 * m);
    c->d2_5         = (float*)realloc(c->d2_5, sizeof(float) * 5 * m);
    c->cloudOri     = (float*)realloc(c->cloudOri, sizeof(float) * 3 * m);
    c->coeffSel     = (float*)realloc(c->coeffSel, sizeof(float) * 4 * m);
    memset(c->surfFlag, 0, m);              /* std::fill(..., false) :222 */
    c->selNum = 0;
}

void orc_set_pose(orc_ctx* c, const float pose[6]) { memcpy(c->transformTobeMapped, pose, 24); }
void orc_get_pose(const orc_ctx* c, float pose[6]) { memcpy(pose, c->transformTobeMapped, 24); }
size_t orc_num_queries(const orc_ctx* c) { return c->n_q; }

/* ---- pcl::getTransformation [ext] ----------------------------------------- */
/* trans2Affine3f(t) = pcl::getTransformation(t[3],t[4],t[5], t[0],t[1],t[2]) (:348-351).
 * PCL's published formula (common/eigen.hpp, Scalar = float):
 *   A=cos(yaw) B=sin(yaw) C=cos(pitch) D=sin(pitch) E=cos(roll) F=sin(roll), DE=D*E, DF=D*F
 *   row0 = A*C, A*DF - B*E, B*F + A*DE, x
 *   row1 = B*C, A*E + B*DF, B*DE - A*F, y
 *   row2 = -D,  C*F,        C*E,        z                                               */
void orc_getTransformation(const float t[6], float T[12])
{
    float roll = t[0], pitch = t[1], yaw = t[2];
    float A = cosf(yaw), B = sinf(yaw), C = cosf(pitch), D = sinf(pitch), E = cosf(roll), F = sinf(roll);
    float DE = D * E, DF = D * F;
    T[0] = A * C; T[1] = A * DF - B * E; T[2]  = B * F + A * DE; T[3]  = t[3];
    T[4] = B * C; T[5] = A * E + B * DF; T[6]  = B * DE - A * F; T[7]  = t[4];
    T[8] = -D;    T[9] = C * F;          T[10] = C * E;          T[11] = t[5];
}

/* ---- Eigen::ColPivHouseholderQR<Matrix<float,5,3>>::solve [ext] ------------ */
/* Restates Eigen 3.3's published algorithm (ColPivHouseholderQR.h computeInPlace /
 * _solve_impl, Householder.h makeHouseholder / applyHouseholderOnTheLeft): column
 * pivoting on down-dated column norms, in-place Householder vectors, rank from
 * nonzeroPivots(), back substitution, permutation. Inner products are summed in
 * index order (Eigen's packet reductions may associate differently: unverifiable). */
#define QR_ROWS 5
#define QR_COLS 3
static void colpiv_qr_solve_5x3(float qr[QR_ROWS][QR_COLS], const float rhs[QR_ROWS], float x[QR_COLS])
{
    float hCoeffs[QR_COLS], normsUpdated[QR_COLS], normsDirect[QR_COLS];
    int   perm[QR_COLS];
    for (int k = 0; k < QR_COLS; k++) {
        float s = 0.0f;
        for (int i = 0; i < QR_ROWS; i++) s += qr[i][k] * qr[i][k];
        normsDirect[k] = sqrtf(s); normsUpdated[k] = normsDirect[k]; perm[k] = k;
    }
    float maxn = normsUpdated[0];
    for (int k = 1; k < QR_COLS; k++) if (normsUpdated[k] > maxn) maxn = normsUpdated[k];
    float th = maxn * FLT_EPSILON;
    const float threshold_helper = (th * th) / (float)QR_ROWS;
    const float norm_downdate_threshold = sqrtf(FLT_EPSILON);
    int nonzero_pivots = QR_COLS;

    for (int k = 0; k < QR_COLS; k++) {
        int big = k; float bign = normsUpdated[k];
        for (int j = k + 1; j < QR_COLS; j++) if (normsUpdated[j] > bign) { bign = normsUpdated[j]; big = j; }
        float big_sq = bign * bign;
        if (nonzero_pivots == QR_COLS && big_sq < threshold_helper * (float)(QR_ROWS - k)) nonzero_pivots = k;
        if (k != big) {
            for (int i = 0; i < QR_ROWS; i++) { float t = qr[i][k]; qr[i][k] = qr[i][big]; qr[i][big] = t; }
            float t = normsUpdated[k]; normsUpdated[k] = normsUpdated[big]; normsUpdated[big] = t;
            t = normsDirect[k]; normsDirect[k] = normsDirect[big]; normsDirect[big] = t;
            int ti = perm[k]; perm[k] = perm[big]; perm[big] = ti;
        }
        /* makeHouseholderInPlace on qr[k..,k] */
        float tailSq = 0.0f;
        for (int i = k + 1; i < QR_ROWS; i++) tailSq += qr[i][k] * qr[i][k];
        float c0 = qr[k][k], beta, tau;
        if (tailSq <= FLT_MIN) {
            tau = 0.0f; beta = c0;
            for (int i = k + 1; i < QR_ROWS; i++) qr[i][k] = 0.0f;
        } else {
            beta = sqrtf(c0 * c0 + tailSq);
            if (c0 >= 0.0f) beta = -beta;
            float den = c0 - beta;
            for (int i = k + 1; i < QR_ROWS; i++) qr[i][k] = qr[i][k] / den;
            tau = (beta - c0) / beta;
        }
        qr[k][k] = beta; hCoeffs[k] = tau;
        /* apply H_k to the trailing columns */
        if (tau != 0.0f) {
            for (int j = k + 1; j < QR_COLS; j++) {
                float tmp = 0.0f;
                for (int i = k + 1; i < QR_ROWS; i++) tmp += qr[i][k] * qr[i][j];
                tmp += qr[k][j];
                qr[k][j] -= tau * tmp;
                for (int i = k + 1; i < QR_ROWS; i++) qr[i][j] -= (tau * qr[i][k]) * tmp;
            }
        }
        /* down-date the remaining column norms */
        for (int j = k + 1; j < QR_COLS; j++) {
            if (normsUpdated[j] != 0.0f) {
                float temp = fabsf(qr[k][j]) / normsUpdated[j];
                temp = (1.0f + temp) * (1.0f - temp);
                temp = temp < 0.0f ? 0.0f : temp;
                float r = normsUpdated[j] / normsDirect[j];
                float temp2 = temp * (r * r);
                if (temp2 <= norm_downdate_threshold) {
                    float s = 0.0f;
                    for (int i = k + 1; i < QR_ROWS; i++) s += qr[i][j] * qr[i][j];
                    normsDirect[j] = sqrtf(s); normsUpdated[j] = normsDirect[j];
                } else {
                    normsUpdated[j] *= sqrtf(temp);
                }
            }
        }
    }

    /* solve */
    x[0] = x[1] = x[2] = 0.0f;
    if (nonzero_pivots == 0) return;
    float c[QR_ROWS];
    for (int i = 0; i < QR_ROWS; i++) c[i] = rhs[i];
    for (int k = 0; k < nonzero_pivots; k++) {          /* c <- H_k c, k ascending (Q^T c) */
        float tau = hCoeffs[k];
        if (tau != 0.0f) {
            float tmp = 0.0f;
            for (int i = k + 1; i < QR_ROWS; i++) tmp += qr[i][k] * c[i];
            tmp += c[k];
            c[k] -= tau * tmp;
            for (int i = k + 1; i < QR_ROWS; i++) c[i] -= (tau * qr[i][k]) * tmp;
        }
    }
    for (int i = nonzero_pivots - 1; i >= 0; i--) {     /* upper-triangular solve, column oriented */
        if (c[i] != 0.0f) {
            c[i] /= qr[i][i];
            for (int r = 0; r < i; r++) c[r] -= c[i] * qr[r][i];
        }
    }
    for (int i = 0; i < nonzero_pivots; i++) x[perm[i]] = c[i];
}

void orc_plane_fit_5x3(const float nbr_xyz[15], float x[3])
{
    float A[QR_ROWS][QR_COLS], b[QR_ROWS];
    for (int j = 0; j < 5; j++) { A[j][0] = nbr_xyz[3 * j]; A[j][1] = nbr_xyz[3 * j + 1]; A[j][2] = nbr_xyz[3 * j + 2]; b[j] = -1.0f; }
    colpiv_qr_solve_5x3(A, b, x);
}

/* ---- surfOptimization (:1074-1143) ---------------------------------------- */
void orc_surfOptimization(orc_ctx* c)
{
    orc_getTransformation(c->transformTobeMapped, c->transPointAssociateToMap);   /* :1076 */
    const float* T = c->transPointAssociateToMap;
    const int nq = (int)c->n_q;
    int nthreads = c->p.num_threads > 0 ? c->p.num_threads : 1;
    (void)nthreads;
    #pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int i = 0; i < nq; i++) {
        const float* po = c->scan + 3 * (size_t)i;                                /* pointOri :1085 */
        float sel[3];                                                             /* pointAssociateToMap :302-308 */
        sel[0] = T[0] * po[0] + T[1] * po[1] + T[2]  * po[2] + T[3];
        sel[1] = T[4] * po[0] + T[5] * po[1] + T[6]  * po[2] + T[7];
        sel[2] = T[8] * po[0] + T[9] * po[1] + T[10] * po[2] + T[11];
        int32_t idx[5]; float d2[5];
        if (c->p.knn_backend == 1) orc_knn5_kdtree(c, sel, idx, d2);               /* :1087 */
        else orc_knn5_brute(c->map, c->n_m, sel, idx, d2);
        for (int j = 0; j < 5; j++) { c->idx5[5 * (size_t)i + j] = idx[j]; c->d2_5[5 * (size_t)i + j] = d2[j]; }
        c->surfFlag[i] = 0;
        if ((double)d2[4] < c->p.gate_sq) {                                       /* :1097 */
            float A[QR_ROWS][QR_COLS], b[QR_ROWS], X[3];
            for (int j = 0; j < 5; j++) {
                const float* m = c->map + 3 * (size_t)idx[j];
                A[j][0] = m[0]; A[j][1] = m[1]; A[j][2] = m[2]; b[j] = -1.0f;     /* :1094, :1099-1101 */
            }
            colpiv_qr_solve_5x3(A, b, X);                                         /* :1104 */
            float pa = X[0], pb = X[1], pc = X[2], pd = 1.0f;
            float ps = sqrtf(pa * pa + pb * pb + pc * pc);                        /* :1111 */
            pa /= ps; pb /= ps; pc /= ps; pd /= ps;
            int planeValid = 1;
            for (int j = 0; j < 5; j++) {                                         /* :1115-1122 */
                const float* m = c->map + 3 * (size_t)idx[j];
                float r = pa * m[0] + pb * m[1] + pc * m[2] + pd;
                if ((double)fabsf(r) > c->p.plane_tol) { planeValid = 0; break; }
            }
            if (planeValid) {
                float pd2 = pa * sel[0] + pb * sel[1] + pc * sel[2] + pd;         /* :1125 */
                float rr = sqrtf(sqrtf(po[0] * po[0] + po[1] * po[1] + po[2] * po[2]));
                float s = (float)(1.0 - c->p.weight_scale * (double)fabsf(pd2) / (double)rr); /* :1127 */
                if ((double)s > c->p.weight_min) {                                /* :1135 */
                    float* o = c->oriSurfVec + 3 * (size_t)i; o[0] = po[0]; o[1] = po[1]; o[2] = po[2];
                    float* cf = c->coeffSurfVec + 4 * (size_t)i;
                    cf[0] = s * pa; cf[1] = s * pb; cf[2] = s * pc; cf[3] = s * pd2;   /* :1130-1133 */
                    c->surfFlag[i] = 1;
                }
            }
        }
    }
}

/* ---- combineOptimizationCoeffs (:1145-1156) -------------------------------- */
void orc_combineOptimizationCoeffs(orc_ctx* c)
{
    int32_t n = 0;
    for (size_t i = 0; i < c->n_q; i++) {
        if (c->surfFlag[i]) {
            memcpy(c->cloudOri + 3 * (size_t)n, c->oriSurfVec + 3 * i, 12);
            memcpy(c->coeffSel + 4 * (size_t)n, c->coeffSurfVec + 4 * i, 16);
            n++;
        }
    }
    c->selNum = n;
    /* the reference resets the flags here (:1155); they are kept readable for
     * orc_get_surf_outputs and reset at the top of the next orc_surfOptimization. */
}

void orc_get_surf_outputs(const orc_ctx* c, int32_t* idx5, float* d2_5, uint8_t* flag, float* coeff4)
{
    for (size_t i = 0; i < c->n_q; i++) {
        int gated = (double)c->d2_5[5 * i + 4] < c->p.gate_sq;
        for (int j = 0; j < 5; j++) {
            if (idx5) idx5[5 * i + j] = gated ? c->idx5[5 * i + j] : -1;
            if (d2_5) d2_5[5 * i + j] = c->d2_5[5 * i + j];
        }
        if (flag) flag[i] = c->surfFlag[i];
        if (coeff4) for (int j = 0; j < 4; j++) coeff4[4 * i + j] = c->surfFlag[i] ? c->coeffSurfVec[4 * i + j] : 0.0f;
    }
}

/* ---- one Jacobian row (:1216-1234) ----------------------------------------- */
/* srx,crx = sin/cos(yaw t[2]); sry,cry = (pitch t[1]); srz,crz = (roll t[0]) (:1170-1175).
 * row = (d/droll, d/dpitch, d/dyaw, cx, cy, cz), rhs = -coeff.intensity.          */
static inline void jac_row(const float sc[6], const float p[3], const float cf[4], float row[6], float* rhs)
{
    const float srx = sc[0], crx = sc[1], sry = sc[2], cry = sc[3], srz = sc[4], crz = sc[5];
    const float px = p[0], py = p[1], pz = p[2];
    float arx = (-srx * cry * px - (srx * sry * srz + crx * crz) * py + (crx * srz - srx * sry * crz) * pz) * cf[0]
              + (crx * cry * px - (srx * crz - crx * sry * srz) * py + (crx * sry * crz + srx * srz) * pz) * cf[1];
    float ary = (-crx * sry * px + crx * cry * srz * py + crx * cry * crz * pz) * cf[0]
              + (-srx * sry * px + srx * sry * srz * py + srx * cry * crz * pz) * cf[1]
              + (-cry * px - sry * srz * py - sry * crz * pz) * cf[2];
    float arz = ((crx * sry * crz + srx * srz) * py + (srx * crz - crx * sry * srz) * pz) * cf[0]
              + ((-crx * srz + srx * sry * crz) * py + (-srx * sry * srz - crx * crz) * pz) * cf[1]
              + (cry * crz * py - cry * srz * pz) * cf[2];
    row[0] = arz; row[1] = ary; row[2] = arx; row[3] = cf[0]; row[4] = cf[1]; row[5] = cf[2];
    *rhs = -cf[3];
}

static void pose_trig(const float t[6], float sc[6])
{
    sc[0] = sinf(t[2]); sc[1] = cosf(t[2]);
    sc[2] = sinf(t[1]); sc[3] = cosf(t[1]);
    sc[4] = sinf(t[0]); sc[5] = cosf(t[0]);
}

void orc_jacobian_row(const float pose[6], const float p_ori[3], const float coeff[4], float row[6], float* rhs)
{
    float sc[6]; pose_trig(pose, sc);
    jac_row(sc, p_ori, coeff, row, rhs);
}

/* ---- cv::solve(A, b, x, DECOMP_QR), CV_32F [ext] ---------------------------- */
/* Restates OpenCV (>= 3.3) hal::QR32f / QRImpl: un-pivoted Householder QR in fp32,
 * reflectors normalised to unit length, rhs transformed, back substitution.
 * (OpenCV < 3.3 routed DECOMP_QR to SVD; versions are unpinned: unverifiable.)   */
int orc_solve6_qr(const float Ain[36], const float bin[6], float x[6])
{
    enum { N = 6 };
    float A[N][N], b[N], vl[N], hf[N];
    for (int i = 0; i < N; i++) { b[i] = bin[i]; for (int j = 0; j < N; j++) A[i][j] = Ain[i * N + j]; }
    for (int l = 0; l < N; l++) {
        int vs = N - l; float nrm = 0.0f;
        for (int i = 0; i < vs; i++) { vl[i] = A[l + i][l]; nrm += vl[i] * vl[i]; }
        float tmpV = vl[0];
        vl[0] = vl[0] + (vl[0] >= 0.0f ? 1.0f : -1.0f) * sqrtf(nrm);
        nrm = sqrtf(nrm + vl[0] * vl[0] - tmpV * tmpV);
        for (int i = 0; i < vs; i++) vl[i] /= nrm;
        for (int j = l; j < N; j++) {
            float v = 0.0f;
            for (int i = l; i < N; i++) v += vl[i - l] * A[i][j];
            for (int i = l; i < N; i++) A[i][j] -= 2.0f * vl[i - l] * v;
        }
        hf[l] = vl[0] * vl[0];
        for (int i = 1; i < vs; i++) A[l + i][l] = vl[i] / vl[0];
    }
    for (int l = 0; l < N; l++) {
        vl[0] = 1.0f;
        for (int j = 1; j < N - l; j++) vl[j] = A[j + l][l];
        float v = 0.0f;
        for (int i = l; i < N; i++) v += vl[i - l] * b[i];
        for (int i = l; i < N; i++) b[i] -= 2.0f * vl[i - l] * v * hf[l];
    }
    for (int i = N - 1; i >= 0; i--) {
        for (int j = N - 1; j > i; j--) b[i] -= b[j] * A[i][j];
        if (fabsf(A[i][i]) < FLT_EPSILON * 10.0f) {   /* singular: cv::solve reports failure */ for (int k = 0; k < N; k++) x[k] = 0.0f; return 0; }
        b[i] /= A[i][i];
    }
    for (int i = 0; i < N; i++) x[i] = b[i];
    return 1;
}

/* ---- cv::eigen(symmetric 6x6, CV_32F) [ext] ---------------------------------- */
/* Restates OpenCV's JacobiImpl_: max-pivot cyclic Jacobi in fp32, eigenvalues
 * sorted descending, eigenvectors returned as ROWS of V.                          */
void orc_eigen6_sym(const float Ain[36], float W[6], float Vout[36])
{
    enum { N = 6 };
    float A[N][N], V[N][N]; int indR[N], indC[N];
    const float eps = FLT_EPSILON;
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) { A[i][j] = Ain[i * N + j]; V[i][j] = (i == j) ? 1.0f : 0.0f; }
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
    const int maxIters = N * N * 30;
    for (int it = 0; it < maxIters; it++) {
        int k = 0, l; float mv = fabsf(A[0][indR[0]]);
        for (int i = 1; i < N - 1; i++) { float v = fabsf(A[i][indR[i]]); if (mv < v) { mv = v; k = i; } }
        l = indR[k];
        for (int i = 1; i < N; i++) { float v = fabsf(A[indC[i]][i]); if (mv < v) { mv = v; k = indC[i]; l = i; } }
        float p = A[k][l];
        if (fabsf(p) <= eps) break;
        float y = (W[l] - W[k]) * 0.5f;
        float t = fabsf(y) + hypotf(p, y);
        float s = hypotf(p, t);
        float c = t / s;
        s = p / s; t = (p / t) * p;
        if (y < 0.0f) { s = -s; t = -t; }
        A[k][l] = 0.0f;
        W[k] -= t; W[l] += t;
#define ORC_ROT(v0, v1) do { float a0 = (v0), b0 = (v1); (v0) = a0 * c - b0 * s; (v1) = a0 * s + b0 * c; } while (0)
        for (int i = 0; i < k; i++)     ORC_ROT(A[i][k], A[i][l]);
        for (int i = k + 1; i < l; i++) ORC_ROT(A[k][i], A[i][l]);
        for (int i = l + 1; i < N; i++) ORC_ROT(A[k][i], A[l][i]);
        for (int i = 0; i < N; i++)     ORC_ROT(V[k][i], V[l][i]);
#undef ORC_ROT
        for (int j = 0; j < 2; j++) {
            int idx = j == 0 ? k : l;
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
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) Vout[i * N + j] = V[i][j];
}

/* ---- cv::Mat::inv() (DECOMP_LU), CV_32F [ext] --------------------------------- */
/* Restates OpenCV's LUImpl applied to [A | I]: partial pivoting, fp32.            */
int orc_inv6_lu(const float Ain[36], float Ainv[36])
{
    enum { N = 6 };
    float A[N][N], B[N][N];
    const float eps = FLT_EPSILON * 10.0f;
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) { A[i][j] = Ain[i * N + j]; B[i][j] = (i == j) ? 1.0f : 0.0f; }
    for (int i = 0; i < N; i++) {
        int k = i;
        for (int j = i + 1; j < N; j++) if (fabsf(A[j][i]) > fabsf(A[k][i])) k = j;
        if (fabsf(A[k][i]) < eps) { memset(Ainv, 0, sizeof(float) * 36); return 0; }
        if (k != i) {
            for (int j = i; j < N; j++) { float t = A[i][j]; A[i][j] = A[k][j]; A[k][j] = t; }
            for (int j = 0; j < N; j++) { float t = B[i][j]; B[i][j] = B[k][j]; B[k][j] = t; }
        }
        float d = -1.0f / A[i][i];
        for (int j = i + 1; j < N; j++) {
            float alpha = A[j][i] * d;
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
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) Ainv[i * N + j] = B[i][j];
    return 1;
}

/* ---- LMOptimization (:1158-1293) ----------------------------------------------- */
int orc_LMOptimization(orc_ctx* c, int iterCount)
{
    orc_iter_trace* tr = (c->n_trace < 64) ? &c->trace[c->n_trace++] : &c->trace[63];
    memset(tr, 0, sizeof(*tr));
    float sc[6]; pose_trig(c->transformTobeMapped, sc);                 /* :1170-1175 */
    int n = c->selNum;
    tr->n_sel = n;
    memcpy(tr->pose, c->transformTobeMapped, 24);
    if (n < c->p.min_corr) return 0;                                    /* :1178-1180 */

    /* matA (n x 6), matB (n x 1) fp32; matAtA = At*A, matAtB = At*B accumulate in
     * double and store fp32 (OpenCV gemm for CV_32F uses a double work type) [ext]. */
    double acc[27]; for (int k = 0; k < 27; k++) acc[k] = 0.0;
    for (int i = 0; i < n; i++) {
        float row[6], rhs;
        jac_row(sc, c->cloudOri + 3 * (size_t)i, c->coeffSel + 4 * (size_t)i, row, &rhs);
        int k = 0;
        for (int a = 0; a < 6; a++) for (int b = a; b < 6; b++) acc[k++] += (double)row[a] * (double)row[b];
        for (int a = 0; a < 6; a++) acc[21 + a] += (double)row[a] * (double)rhs;
    }
    float* AtA = c->matAtA; float* AtB = c->matAtB;
    { int k = 0; for (int a = 0; a < 6; a++) for (int b = a; b < 6; b++) { float v = (float)acc[k++]; AtA[a * 6 + b] = v; AtA[b * 6 + a] = v; } }
    for (int a = 0; a < 6; a++) AtB[a] = (float)acc[21 + a];

    float X[6];
    orc_solve6_qr(AtA, AtB, X);                                         /* :1240 */

    if (iterCount == 0) {                                               /* :1242-1264 */
        float E[6], V[36], V2[36], Vinv[36];
        orc_eigen6_sym(AtA, E, V);
        memcpy(V2, V, sizeof(V));
        c->isDegenerate = 0;
        for (int i = 5; i >= 0; i--) {
            if (E[i] < c->p.eig_thresh) { for (int j = 0; j < 6; j++) V2[i * 6 + j] = 0.0f; c->isDegenerate = 1; }
            else break;
        }
        orc_inv6_lu(V, Vinv);
        for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) {       /* matP = V.inv() * V2 */
            double s = 0.0; for (int k = 0; k < 6; k++) s += (double)Vinv[i * 6 + k] * (double)V2[k * 6 + j];
            c->matP[i * 6 + j] = (float)s;
        }
    }
    if (c->isDegenerate) {                                              /* :1266-1271 */
        float X2[6]; memcpy(X2, X, 24);
        for (int i = 0; i < 6; i++) { double s = 0.0; for (int k = 0; k < 6; k++) s += (double)c->matP[i * 6 + k] * (double)X2[k]; X[i] = (float)s; }
    }
    for (int k = 0; k < 6; k++) c->transformTobeMapped[k] += X[k];      /* :1273-1278 */

    /* pcl::rad2deg(float) = alpha * 57.29578f [ext]; pow(float, int) promotes to double */
    double r0 = (double)(X[0] * 57.29578f), r1 = (double)(X[1] * 57.29578f), r2 = (double)(X[2] * 57.29578f);
    float deltaR = (float)sqrt(r0 * r0 + r1 * r1 + r2 * r2);
    double t0 = (double)(X[3] * 100), t1 = (double)(X[4] * 100), t2 = (double)(X[5] * 100);
    float deltaT = (float)sqrt(t0 * t0 + t1 * t1 + t2 * t2);

    tr->stepped = 1; memcpy(tr->delta, X, 24); memcpy(tr->pose, c->transformTobeMapped, 24);
    tr->deltaR = deltaR; tr->deltaT = deltaT;
    if ((double)deltaR < c->p.conv_deg && (double)deltaT < c->p.conv_cm) return 1;   /* :1289 */
    return 0;
}

int orc_get_normal_eq(const orc_ctx* c, float AtA[36], float AtB[6])
{
    /* recompute from the compacted correspondences at the current pose */
    float sc[6]; pose_trig(c->transformTobeMapped, sc);
    double acc[27]; for (int k = 0; k < 27; k++) acc[k] = 0.0;
    for (int i = 0; i < c->selNum; i++) {
        float row[6], rhs;
        jac_row(sc, c->cloudOri + 3 * (size_t)i, c->coeffSel + 4 * (size_t)i, row, &rhs);
        int k = 0;
        for (int a = 0; a < 6; a++) for (int b = a; b < 6; b++) acc[k++] += (double)row[a] * (double)row[b];
        for (int a = 0; a < 6; a++) acc[21 + a] += (double)row[a] * (double)rhs;
    }
    int k = 0; for (int a = 0; a < 6; a++) for (int b = a; b < 6; b++) { float v = (float)acc[k++]; AtA[a * 6 + b] = v; AtA[b * 6 + a] = v; }
    for (int a = 0; a < 6; a++) AtB[a] = (float)acc[21 + a];
    return c->selNum;
}

/* ---- transformUpdate (:1323-1353) ------------------------------------------------ */
/* tf::Quaternion::setRPY / slerp and tf::Matrix3x3::getRPY restated in double [ext]. */
typedef struct { double x, y, z, w; } quat;
static quat quat_rpy(double roll, double pitch, double yaw)
{
    double hy = yaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
    double cy = cos(hy), sy = sin(hy), cp = cos(hp), sp = sin(hp), cr = cos(hr), sr = sin(hr);
    quat q = { sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy };
    return q;
}
static quat quat_slerp(quat a, quat b, double t)
{
    double dot = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    double la = sqrt(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w), lb = sqrt(b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w);
    double cs = dot / (la * lb); if (cs > 1) cs = 1; if (cs < -1) cs = -1;
    double theta = acos(cs < 0 ? -cs : cs);      /* angleShortestPath */
    if (theta != 0.0) {
        double d = 1.0 / sin(theta), s0 = sin((1.0 - t) * theta), s1 = sin(t * theta);
        if (dot < 0) s1 = -s1;
        quat r = { (a.x * s0 + b.x * s1) * d, (a.y * s0 + b.y * s1) * d, (a.z * s0 + b.z * s1) * d, (a.w * s0 + b.w * s1) * d };
        return r;
    }
    return a;
}
static void quat_get_rpy(quat q, double* roll, double* pitch, double* yaw)
{
    double d = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w, s = 2.0 / d;
    double xs = q.x * s, ys = q.y * s, zs = q.z * s;
    double wx = q.w * xs, wy = q.w * ys, wz = q.w * zs, xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
    double yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
    double m00 = 1.0 - (yy + zz), m10 = xy + wz, m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy);
    double m01 = xy - wz, m02 = xz + wy; (void)m02;
    if (fabs(m20) >= 1.0) {   /* gimbal lock branch of getEulerYPR */
        *yaw = 0.0;
        if (m20 < 0) { *pitch = M_PI / 2.0; *roll = atan2(m01, m02); }
        else { *pitch = -M_PI / 2.0; *roll = atan2(-m01, -m02); }
        return;
    }
    *pitch = -asin(m20);
    double cp = cos(*pitch);
    *roll = atan2(m21 / cp, m22 / cp);
    *yaw = atan2(m10 / cp, m00 / cp);
}
static float constraintTransformation(float value, float limit)      /* :1355-1363 */
{
    if (value < -limit) value = -limit;
    if (value > limit) value = limit;
    return value;
}

void orc_transformUpdate(orc_ctx* c, const orc_imu_init* imu)
{
    float* t = c->transformTobeMapped;
    if (imu && imu->imuAvailable == 1 && c->p.imu_type) {            /* `cloudInfo.imuAvailable == true` :1325: the int64 field against 1 */
        if ((double)fabsf(imu->imuPitchInit) < 1.4) {                /* std::abs(float) < 1.4 :1327: float promoted, double compare */
            double w = (double)c->p.imu_rpy_weight, r, p, y;
            quat_get_rpy(quat_slerp(quat_rpy(t[0], 0, 0), quat_rpy(imu->imuRollInit, 0, 0), w), &r, &p, &y);
            t[0] = (float)r;                                         /* :1338 */
            quat_get_rpy(quat_slerp(quat_rpy(0, t[1], 0), quat_rpy(0, imu->imuPitchInit, 0), w), &r, &p, &y);
            t[1] = (float)p;                                         /* :1344 */
        }
    }
    t[0] = constraintTransformation(t[0], c->p.rot_tol);             /* :1348-1350 */
    t[1] = constraintTransformation(t[1], c->p.rot_tol);
    t[5] = constraintTransformation(t[5], c->p.z_tol);
    orc_getTransformation(t, c->incrementalOdometryAffineBack);      /* :1352 */
}

/* ---- scan2MapOptimization (:1295-1321) --------------------------------------------- */
void orc_scan2MapOptimization(orc_ctx* c, const orc_imu_init* imu, orc_result* out)
{
    orc_result r; memset(&r, 0, sizeof(r));
    c->n_trace = 0;
    double tb = c->tm.tree_build; memset(&c->tm, 0, sizeof(c->tm)); c->tm.tree_build = tb;
    double t_all = now_s();
    if (c->n_m == 0) { r.skipped = 1; }                              /* cloudKeyPoses3D empty :1297 */
    else if ((int)c->n_q > c->p.min_feats) {                         /* :1300 */
        for (int iter = 0; iter < c->p.max_iter; iter++) {           /* :1304 */
            double t0 = now_s();
            orc_surfOptimization(c);                                 /* :1309 */
            double t1 = now_s();
            orc_combineOptimizationCoeffs(c);                        /* :1311 */
            double t2 = now_s();
            int conv = orc_LMOptimization(c, iter);                  /* :1313 */
            double t3 = now_s();
            c->tm.knn_plane += t1 - t0; c->tm.compaction += t2 - t1; c->tm.jacobian_solve += t3 - t2;
            r.iters_run = iter + 1; r.n_sel_last = c->selNum;
            if (conv) { r.converged = 1; if (c->p.early_exit) break; }
        }
        orc_transformUpdate(c, imu);                                 /* :1317 */
    } else { r.skipped = 2; }                                        /* ROS_WARN :1319 */
    c->tm.total = now_s() - t_all;
    r.is_degenerate = c->isDegenerate;
    memcpy(r.pose, c->transformTobeMapped, 24);
    memcpy(r.affine, c->incrementalOdometryAffineBack, 48);
    if (out) *out = r;
}

int orc_get_trace(const orc_ctx* c, orc_iter_trace* out, int cap)
{
    int n = c->n_trace < cap ? c->n_trace : cap;
    for (int i = 0; i < n; i++) out[i] = c->trace[i];
    return n;
}
void orc_get_timing(const orc_ctx* c, orc_timing* t) { *t = c->tm; }
void orc_get_matP(const orc_ctx* c, float matP[36], int* isDegenerate)
{
    memcpy(matP, c->matP, sizeof(float) * 36); if (isDegenerate) *isDegenerate = c->isDegenerate;
}

/* ---- ScanContext (include/Scancontext.cpp) ------------------------------------------ */
/* xy2theta (:23-36): quadrant-wise atan in degrees. The float overloads of atan/sqrt
 * are assumed visible unqualified (libstdc++ <math.h> via the OpenCV C headers) [ext]. */
float orc_xy2theta(float x, float y)
{
    const double k = 180.0 / M_PI;
    if ((x >= 0) & (y >= 0)) return (float)(k * (double)atanf(y / x));
    if ((x < 0) & (y >= 0))  return (float)(180.0 - (k * (double)atanf(y / (-x))));
    if ((x < 0) & (y < 0))   return (float)(180.0 + (k * (double)atanf(y / x)));
    if ((x >= 0) & (y < 0))  return (float)(360.0 - (k * (double)atanf((-y) / x)));
    return NAN;   /* NaN inputs fall through every branch in the reference (UB there) */
}

static int clamp_ceil_bin(double v, int nbins)
{
    /* std::max(std::min(N, int(ceil(v))), 1); int(NaN) is INT_MIN on x86-64 */
    int iv;
    if (v != v) iv = INT32_MIN;
    else { double cv = ceil(v); iv = cv >= 2147483647.0 ? INT32_MAX : (cv <= -2147483648.0 ? INT32_MIN : (int)cv); }
    int m = nbins < iv ? nbins : iv;
    return m > 1 ? m : 1;
}

void orc_makeScancontext(const void* pts, size_t n, size_t stride_bytes, double desc[20 * 60])
{
    const double LIDAR_HEIGHT = 2.0, PC_MAX_RADIUS = 80.0;           /* Scancontext.h:80-84 */
    const int NR = 20, NS = 60; const double NO_POINT = -1000.0;
    for (int i = 0; i < NR * NS; i++) desc[i] = NO_POINT;             /* :158-159 */
    const unsigned char* b = (const unsigned char*)pts;
    for (size_t i = 0; i < n; i++) {
        float p[3]; memcpy(p, b + i * stride_bytes, 12);
        float px = p[0], py = p[1];
        float pz = (float)((double)p[2] + LIDAR_HEIGHT);              /* :168 (float member) */
        float azim_range = sqrtf(px * px + py * py);                  /* :171 */
        float azim_angle = orc_xy2theta(px, py);                      /* :172 */
        if ((double)azim_range > PC_MAX_RADIUS) continue;             /* :175 */
        int ring = clamp_ceil_bin(((double)azim_range / PC_MAX_RADIUS) * NR, NR);   /* :178 */
        int sect = clamp_ceil_bin(((double)azim_angle / 360.0) * NS, NS);           /* :179 */
        double* d = &desc[(ring - 1) * NS + (sect - 1)];
        if (*d < (double)pz) *d = (double)pz;                         /* :182-183 */
    }
    for (int i = 0; i < NR * NS; i++) if (desc[i] == NO_POINT) desc[i] = 0.0;       /* :187-190 */
}

void orc_makeRingkeyFromScancontext(const double desc[20 * 60], double key[20])
{
    for (int r = 0; r < 20; r++) {                                    /* Eigen row mean = sum / 60 [ext] */
        double s = 0.0; for (int k = 0; k < 60; k++) s += desc[r * 60 + k];
        key[r] = s / 60.0;
    }
}

/* ==== section 8(f) rows F1 / F2: the voxel-grid filters either side of the path ================
 * downsampleCurrentScan (:1061-1067, leaf mappingSurfLeafSize) and the VoxelGrid at the end of
 * extractCloud (:1037-1039, leaf surroundingKeyframeMapLeafSize) both run
 * pcl::VoxelGrid<pcl::PointXYZI>::applyFilter with default settings (downsample_all_data_ = true,
 * min_points_per_voxel_ = 0, no filter field). PCL is not vendored: restated from PCL 1.10
 * (filters/include/pcl/filters/impl/voxel_grid.hpp, common/include/pcl/common/impl/centroid.hpp) [ext].
 *
 * One thing PCL leaves to the standard library is restated as a definition: std::sort of the
 * (voxel idx, point index) pairs compares idx only and is not stable, so the order in which the
 * points of one voxel are summed is unspecified there; here it is ascending point index. The
 * centroid is therefore reproducible, and differs from any one PCL build by fp32 summation order only.
 * Points with a non-finite coordinate are skipped (PCL does so when !is_dense).
 */
typedef struct { uint32_t idx; uint32_t pt; } vox_pair;
static int vox_pair_cmp(const void* a, const void* b)
{
    const vox_pair* x = (const vox_pair*)a; const vox_pair* y = (const vox_pair*)b;
    if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
    return x->pt < y->pt ? -1 : (x->pt > y->pt ? 1 : 0);
}

int orc_voxelGrid(const void* pts, size_t n, size_t stride_bytes, float leaf,
                  void* out, size_t out_stride_bytes, size_t cap, size_t* n_out)
{
    const unsigned char* b = (const unsigned char*)pts;
    unsigned char* ob = (unsigned char*)out;
    *n_out = 0;
    /* getMinMax3D over the finite points */
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    size_t n_valid = 0;
    for (size_t i = 0; i < n; i++) {
        float p[3]; memcpy(p, b + i * stride_bytes, 12);
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
        for (int d = 0; d < 3; d++) { if (p[d] < mn[d]) mn[d] = p[d]; if (p[d] > mx[d]) mx[d] = p[d]; }
        n_valid++;
    }
    if (n_valid == 0) return 0;
    const float inv = 1.0f / leaf;                                     /* inverse_leaf_size_ = Ones / leaf_size_ */
    /* "Leaf size is too small for the input dataset": output = input */
    int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1, dy = (int64_t)((mx[1] - mn[1]) * inv) + 1,
            dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
    /* PCL multiplies in int64 (undefined past 2^63); the product is taken in double here, equal wherever PCL's is defined */
    if ((double)dx * (double)dy * (double)dz > (double)INT32_MAX) {
        if (n > cap) { *n_out = n; return -1; }
        for (size_t i = 0; i < n; i++) {
            memset(ob + i * out_stride_bytes, 0, out_stride_bytes);
            memcpy(ob + i * out_stride_bytes, b + i * stride_bytes,
                   stride_bytes < out_stride_bytes ? stride_bytes : out_stride_bytes);
        }
        *n_out = n;
        return 1;
    }
    int min_b[3], max_b[3], div_b[3];
    for (int d = 0; d < 3; d++) {
        min_b[d] = (int)floorf(mn[d] * inv);
        max_b[d] = (int)floorf(mx[d] * inv);
        div_b[d] = max_b[d] - min_b[d] + 1;
    }
    const int mul1 = div_b[0], mul2 = div_b[0] * div_b[1];             /* divb_mul_ = (1, div0, div0*div1) */
    vox_pair* iv = (vox_pair*)malloc(sizeof(vox_pair) * n_valid);
    size_t k = 0;
    for (size_t i = 0; i < n; i++) {
        float p[3]; memcpy(p, b + i * stride_bytes, 12);
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
        int ijk0 = (int)(floorf(p[0] * inv) - (float)min_b[0]);
        int ijk1 = (int)(floorf(p[1] * inv) - (float)min_b[1]);
        int ijk2 = (int)(floorf(p[2] * inv) - (float)min_b[2]);
        iv[k].idx = (uint32_t)(ijk0 + ijk1 * mul1 + ijk2 * mul2);
        iv[k].pt = (uint32_t)i;
        k++;
    }
    qsort(iv, n_valid, sizeof(vox_pair), vox_pair_cmp);
    size_t total = 0, first = 0;
    int rc = 0;
    while (first < n_valid) {
        size_t last = first + 1;
        while (last < n_valid && iv[last].idx == iv[first].idx) last++;
        if (total < cap) {
            /* CentroidPoint<PointXYZI>: AccumulatorXYZ (Vector3f sum), AccumulatorIntensity (float sum); get() = sum / n */
            float sx = 0.0f, sy = 0.0f, sz = 0.0f, si = 0.0f;
            for (size_t j = first; j < last; j++) {
                const unsigned char* r = b + (size_t)iv[j].pt * stride_bytes;
                float p[3]; memcpy(p, r, 12);
                float in = 0.0f; if (stride_bytes >= 20) memcpy(&in, r + 16, 4);
                sx += p[0]; sy += p[1]; sz += p[2]; si += in;
            }
            const float cnt = (float)(last - first);
            float rec[5] = { sx / cnt, sy / cnt, sz / cnt, 1.0f, si / cnt };
            unsigned char* o = ob + total * out_stride_bytes;
            memset(o, 0, out_stride_bytes);
            memcpy(o, rec, out_stride_bytes >= 20 ? 20 : (out_stride_bytes >= 16 ? 16 : 12));
        } else rc = -1;
        total++;
        first = last;
    }
    free(iv);
    *n_out = total;
    return rc;
}

/* transformPointCloud (:310-329): T = pcl::getTransformation(x, y, z, roll, pitch, yaw) of a
 * PointTypePose, applied as written (fp32, three products and three adds per row); intensity copied. */
void orc_transformPointCloud(const void* pts, size_t n, size_t stride_bytes, const float pose_xyzrpy[6],
                             void* out, size_t out_stride_bytes)
{
    float t[6] = { pose_xyzrpy[3], pose_xyzrpy[4], pose_xyzrpy[5], pose_xyzrpy[0], pose_xyzrpy[1], pose_xyzrpy[2] };
    float T[12];
    orc_getTransformation(t, T);
    const unsigned char* b = (const unsigned char*)pts;
    unsigned char* ob = (unsigned char*)out;
    for (size_t i = 0; i < n; i++) {
        const unsigned char* r = b + i * stride_bytes;
        float p[3]; memcpy(p, r, 12);
        float in = 0.0f; if (stride_bytes >= 20) memcpy(&in, r + 16, 4);
        float rec[5];
        rec[0] = T[0] * p[0] + T[1] * p[1] + T[2] * p[2] + T[3];
        rec[1] = T[4] * p[0] + T[5] * p[1] + T[6] * p[2] + T[7];
        rec[2] = T[8] * p[0] + T[9] * p[1] + T[10] * p[2] + T[11];
        rec[3] = 1.0f; rec[4] = in;
        unsigned char* o = ob + i * out_stride_bytes;
        memset(o, 0, out_stride_bytes);
        memcpy(o, rec, out_stride_bytes >= 20 ? 20 : (out_stride_bytes >= 16 ? 16 : 12));
    }
}

/* ==== section 8(f) row F3: ScanContext matching (include/Scancontext.cpp:69-148, 214-344) =========
 * Descriptors are 20 x 60 row-major doubles as orc_makeScancontext writes them (ring r, sector s at
 * desc[r*60 + s]); the reference's Eigen::MatrixXd is column-major, which changes no arithmetic.
 * [ext] Eigen's mean()/norm()/dot() on 20- and 60-vectors reduce in a vectorised order that depends on
 * the Eigen version and instruction set; restated as plain left-to-right sums (differences ~1e-16
 * relative).  nanoflann's kNN over the float ring keys is restated as brute force with the vendored
 * L2_Adaptor's accumulation order (include/nanoflann.hpp:383-408: groups of four); equal distances,
 * which the kd-tree would order by traversal, go to the lower index.
 */
#define SC_NR 20
#define SC_NS 60

void orc_makeSectorkeyFromScancontext(const double desc[SC_NR * SC_NS], double key[SC_NS])
{
    for (int s = 0; s < SC_NS; s++) {                                  /* :214-227 column mean */
        double a = 0.0; for (int r = 0; r < SC_NR; r++) a += desc[r * SC_NS + s];
        key[s] = a / (double)SC_NR;
    }
}

/* distDirectSC(_sc1, circshift(_sc2, shift)) (:69-91 with :39-59): column j of the shifted matrix is
 * column (j - shift) mod 60 of _sc2 */
double orc_distDirectSC_shifted(const double sc1[SC_NR * SC_NS], const double sc2[SC_NR * SC_NS], int shift)
{
    int num_eff_cols = 0;
    double sum_sector_similarity = 0;
    for (int j = 0; j < SC_NS; j++) {
        const int j2 = ((j - shift) % SC_NS + SC_NS) % SC_NS;
        double n1 = 0.0, n2 = 0.0, dot = 0.0;
        for (int r = 0; r < SC_NR; r++) {
            const double a = sc1[r * SC_NS + j], b = sc2[r * SC_NS + j2];
            n1 += a * a; n2 += b * b; dot += a * b;
        }
        n1 = sqrt(n1); n2 = sqrt(n2);
        if ((n1 == 0) | (n2 == 0)) continue;                           /* :78-79 */
        sum_sector_similarity = sum_sector_similarity + dot / (n1 * n2);   /* :81-83 */
        num_eff_cols = num_eff_cols + 1;
    }
    const double sc_sim = sum_sector_similarity / num_eff_cols;       /* 0/0 = NaN when no sector counts */
    return 1.0 - sc_sim;
}

int orc_fastAlignUsingVkey(const double vkey1[SC_NS], const double vkey2[SC_NS])
{
    int argmin_vkey_shift = 0;                                         /* :94-113 */
    double min_veky_diff_norm = 10000000;
    for (int shift = 0; shift < SC_NS; shift++) {
        double a = 0.0;
        for (int j = 0; j < SC_NS; j++) {
            const double d = vkey1[j] - vkey2[((j - shift) % SC_NS + SC_NS) % SC_NS];
            a += d * d;
        }
        const double cur = sqrt(a);
        if (cur < min_veky_diff_norm) { argmin_vkey_shift = shift; min_veky_diff_norm = cur; }
    }
    return argmin_vkey_shift;
}

static int cmp_int(const void* a, const void* b) { return *(const int*)a - *(const int*)b; }

/* distanceBtnScanContext (:116-148): SEARCH_RATIO 0.1 (Scancontext.h:93) */
void orc_distanceBtnScanContext(const double sc1[SC_NR * SC_NS], const double sc2[SC_NR * SC_NS], double* dist, int* shift)
{
    double v1[SC_NS], v2[SC_NS];
    orc_makeSectorkeyFromScancontext(sc1, v1);
    orc_makeSectorkeyFromScancontext(sc2, v2);
    const int a0 = orc_fastAlignUsingVkey(v1, v2);
    const int SEARCH_RADIUS = (int)round(0.5 * 0.1 * SC_NS);
    int space[1 + 2 * SC_NS], n = 0;
    space[n++] = a0;
    for (int ii = 1; ii < SEARCH_RADIUS + 1; ii++) {
        space[n++] = (a0 + ii + SC_NS) % SC_NS;
        space[n++] = (a0 - ii + SC_NS) % SC_NS;
    }
    qsort(space, (size_t)n, sizeof(int), cmp_int);
    int argmin_shift = 0;
    double min_sc_dist = 10000000;
    for (int k = 0; k < n; k++) {
        const double cur = orc_distDirectSC_shifted(sc1, sc2, space[k]);
        if (cur < min_sc_dist) { argmin_shift = space[k]; min_sc_dist = cur; }
    }
    *dist = min_sc_dist; *shift = argmin_shift;
}

/* SCManager state (Scancontext.h:102-113) and detectLoopClosureID (:253-344) */
struct orc_sc {
    double* desc; float* invkey; size_t n, cap;
    size_t n_search;            /* keys in the tree at its last rebuild */
    int    counter;             /* tree_making_period_conter */
};

orc_sc* orc_sc_create(void) { return (orc_sc*)calloc(1, sizeof(orc_sc)); }
void orc_sc_destroy(orc_sc* m) { if (m) { free(m->desc); free(m->invkey); free(m); } }
size_t orc_sc_size(const orc_sc* m) { return m->n; }

void orc_sc_add_descriptor(orc_sc* m, const double desc[SC_NR * SC_NS])   /* makeAndSaveScancontextAndKeys :236-250 */
{
    if (m->n == m->cap) {
        m->cap = m->cap ? 2 * m->cap : 64;
        m->desc = (double*)realloc(m->desc, m->cap * SC_NR * SC_NS * sizeof(double));
        m->invkey = (float*)realloc(m->invkey, m->cap * SC_NR * sizeof(float));
    }
    memcpy(m->desc + m->n * SC_NR * SC_NS, desc, SC_NR * SC_NS * sizeof(double));
    double key[SC_NR];
    orc_makeRingkeyFromScancontext(desc, key);
    for (int r = 0; r < SC_NR; r++) m->invkey[m->n * SC_NR + r] = (float)key[r];   /* eig2stdvec :62-66 */
    m->n++;
}

void orc_sc_add_scan(orc_sc* m, const void* pts, size_t n, size_t stride_bytes)
{
    double desc[SC_NR * SC_NS];
    orc_makeScancontext(pts, n, stride_bytes, desc);
    orc_sc_add_descriptor(m, desc);
}

/* Returns loop_id (-1: none); also the values the reference computes on the way. cand_* hold the
 * NUM_CANDIDATES_FROM_TREE (3) ring-key neighbours in ascending distance. */
int orc_sc_detectLoopClosureID(orc_sc* m, float* yaw_diff_rad, double* min_dist_out, int* nn_idx_out, int* nn_align_out,
                               int cand_idx[3], float cand_d2[3])
{
    const int NUM_EXCLUDE_RECENT = 30, NUM_CANDIDATES = 3, TREE_MAKING_PERIOD = 10;
    const double SC_DIST_THRES = 0.3, PC_UNIT_SECTORANGLE = 360.0 / 60.0;
    *yaw_diff_rad = 0.0f; *min_dist_out = 10000000; *nn_idx_out = 0; *nn_align_out = 0;
    for (int k = 0; k < 3; k++) { cand_idx[k] = 0; cand_d2[k] = 0.0f; }
    if ((int)m->n < NUM_EXCLUDE_RECENT + 1) return -1;                 /* :263-267 */
    if (m->counter % TREE_MAKING_PERIOD == 0) m->n_search = m->n - NUM_EXCLUDE_RECENT;   /* :270-281 */
    m->counter = m->counter + 1;

    const float* q = m->invkey + (m->n - 1) * SC_NR;
    const double* curr = m->desc + (m->n - 1) * SC_NR * SC_NS;
    /* knn search (:288-296): exact 3-NN, ascending */
    int nfound = 0;
    for (size_t i = 0; i < m->n_search; i++) {
        const float* b = m->invkey + i * SC_NR;
        float result = 0.0f;
        for (int d = 0; d < SC_NR; d += 4) {
            const float d0 = q[d] - b[d], d1 = q[d + 1] - b[d + 1], d2 = q[d + 2] - b[d + 2], d3 = q[d + 3] - b[d + 3];
            result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        }
        int pos = nfound < NUM_CANDIDATES ? nfound : NUM_CANDIDATES;
        while (pos > 0 && cand_d2[pos - 1] > result) pos--;
        if (pos < NUM_CANDIDATES) {
            const int last = nfound < NUM_CANDIDATES ? nfound : NUM_CANDIDATES - 1;
            for (int k = last; k > pos; k--) { cand_d2[k] = cand_d2[k - 1]; cand_idx[k] = cand_idx[k - 1]; }
            cand_d2[pos] = result; cand_idx[pos] = (int)i;
            if (nfound < NUM_CANDIDATES) nfound++;
        }
    }
    /* fewer keys than candidates: the reference's index vector keeps its zero initialisation (:289) */
    double min_dist = 10000000;
    int nn_align = 0, nn_idx = 0;
    for (int c = 0; c < NUM_CANDIDATES; c++) {                         /* :302-316 */
        double dist; int align;
        orc_distanceBtnScanContext(curr, m->desc + (size_t)cand_idx[c] * SC_NR * SC_NS, &dist, &align);
        if (dist < min_dist) { min_dist = dist; nn_align = align; nn_idx = cand_idx[c]; }
    }
    int loop_id = -1;
    if (min_dist < SC_DIST_THRES) loop_id = nn_idx;                    /* :322-324 */
    *yaw_diff_rad = (float)((double)(float)(nn_align * PC_UNIT_SECTORANGLE) * M_PI / 180.0);   /* deg2rad(float) :17-20, :338 */
    *min_dist_out = min_dist; *nn_idx_out = nn_idx; *nn_align_out = nn_align;
    return loop_id;
}

/* ==== section 8(f) row F4: ICP loop-closure alignment (src/mapOptmization.cpp:571-586, :663-678) ========
 * pcl::IterativeClosestPoint<PointXYZI, PointXYZI> (Scalar = float) with the settings of the call site:
 * max correspondence distance 2 * historyKeyframeSearchRadius, 100 iterations, transformation epsilon 1e-6,
 * Euclidean fitness epsilon 1e-6, no RANSAC, identity guess.  PCL is not vendored: restated from PCL 1.10
 * [ext] - registration/impl/icp.hpp (computeTransformation), correspondence_estimation.hpp
 * (determineCorrespondences: nearest target point, kept if d2 <= max_dist^2),
 * transformation_estimation_svd.hpp -> Eigen::umeyama without scaling (mean, demeaned cross-covariance / n,
 * SVD, R = U S V^T with S(2) = -1 if det(U) det(V) < 0, t = mean_tgt - R mean_src),
 * default_convergence_criteria.hpp (hasConverged, state machine as published), and Registration::
 * getFitnessScore.  Restatement choices where PCL delegates: the nearest neighbour is brute force with
 * ties to the lower index (PCL: FLANN kd-tree); the 3x3 SVD is a fp64 one-sided Jacobi (Eigen:
 * JacobiSVD<Matrix3f>); means and covariance are summed left to right in fp32 as Eigen would in a scalar
 * build.  Differences against a PCL build are of the order of fp32 rounding of those sums.
 */
static void icp_svd3(const double A[9], double U[9], double S[3], double V[9])
{
    /* one-sided Jacobi on the columns of W = A (row-major 3x3): W V = U S */
    double W[9]; memcpy(W, A, sizeof(W));
    for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double a = 0, b = 0, c = 0;
                for (int k = 0; k < 3; k++) { a += W[k * 3 + p] * W[k * 3 + p]; b += W[k * 3 + q] * W[k * 3 + q]; c += W[k * 3 + p] * W[k * 3 + q]; }
                off += c * c;
                if (fabs(c) <= 1e-300 || fabs(c) <= 1e-17 * sqrt(a * b)) continue;
                const double zeta = (b - a) / (2.0 * c);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (int k = 0; k < 3; k++) {
                    const double wp = W[k * 3 + p], wq = W[k * 3 + q];
                    W[k * 3 + p] = cs * wp - sn * wq; W[k * 3 + q] = sn * wp + cs * wq;
                    const double vp = V[k * 3 + p], vq = V[k * 3 + q];
                    V[k * 3 + p] = cs * vp - sn * vq; V[k * 3 + q] = sn * vp + cs * vq;
                }
            }
        if (off < 1e-300) break;
    }
    for (int j = 0; j < 3; j++) {
        double nrm = 0; for (int k = 0; k < 3; k++) nrm += W[k * 3 + j] * W[k * 3 + j];
        S[j] = sqrt(nrm);
    }
    /* sort descending (like Eigen), permuting the columns of W and V */
    for (int i = 0; i < 2; i++) for (int j = i + 1; j < 3; j++) if (S[j] > S[i]) {
        double ts = S[i]; S[i] = S[j]; S[j] = ts;
        for (int k = 0; k < 3; k++) { double t1 = W[k * 3 + i]; W[k * 3 + i] = W[k * 3 + j]; W[k * 3 + j] = t1; double t2 = V[k * 3 + i]; V[k * 3 + i] = V[k * 3 + j]; V[k * 3 + j] = t2; }
    }
    for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) U[k * 3 + j] = S[j] > 1e-300 ? W[k * 3 + j] / S[j] : 0.0;
    /* complete a rank-deficient U to an orthonormal basis (cross products) */
    if (S[2] <= 1e-12 * S[0]) {
        U[2] = U[3] * U[7] - U[6] * U[4]; U[5] = U[6] * U[1] - U[0] * U[7]; U[8] = U[0] * U[4] - U[3] * U[1];
    }
}

static double det3(const double M[9])
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

/* rigid transform (row-major 4x4, float) from the sums of one iteration; shared with nothing on the GPU side */
void orc_icp_umeyama(const float mean_src[3], const float mean_tgt[3], const float sigma[9] /* (1/n) sum tgt_d src_d^T */, float T[16])
{
    double A[9], U[9], S[3], V[9];
    for (int i = 0; i < 9; i++) A[i] = (double)sigma[i];
    icp_svd3(A, U, S, V);
    double sgn[3] = { 1.0, 1.0, (det3(U) * det3(V) < 0) ? -1.0 : 1.0 };
    float R[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double a = 0; for (int k = 0; k < 3; k++) a += U[i * 3 + k] * sgn[k] * V[j * 3 + k];
        R[i * 3 + j] = (float)a;
    }
    for (int i = 0; i < 16; i++) T[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[i * 4 + j] = R[i * 3 + j];
        T[i * 4 + 3] = mean_tgt[i] - (R[i * 3 + 0] * mean_src[0] + R[i * 3 + 1] * mean_src[1] + R[i * 3 + 2] * mean_src[2]);
    }
}

int orc_icp_align(const void* src, size_t n_src, const void* tgt, size_t n_tgt, size_t stride_bytes,
                  double max_corr_dist, int max_iter, double trans_eps, double fit_eps, int num_threads,
                  float T_final[16], int* converged, double* fitness, int* iterations)
{
    const unsigned char* sb = (const unsigned char*)src; const unsigned char* tb = (const unsigned char*)tgt;
    float* cur = (float*)malloc(sizeof(float) * 3 * (n_src ? n_src : 1));       /* input_transformed */
    float* tg = (float*)malloc(sizeof(float) * 3 * (n_tgt ? n_tgt : 1));
    int32_t* ci = (int32_t*)malloc(sizeof(int32_t) * (n_src ? n_src : 1));
    float* cd = (float*)malloc(sizeof(float) * (n_src ? n_src : 1));
    for (size_t i = 0; i < n_src; i++) memcpy(cur + 3 * i, sb + i * stride_bytes, 12);
    for (size_t i = 0; i < n_tgt; i++) memcpy(tg + 3 * i, tb + i * stride_bytes, 12);
    for (int i = 0; i < 16; i++) T_final[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    const double max_d2 = max_corr_dist * max_corr_dist;
    const double rot_thr = 1.0 - trans_eps, trl_thr = trans_eps, mse_abs = 1e-12, mse_rel = fit_eps;
    double prev_mse = DBL_MAX;
    int it = 0, conv = 0, similar = 0;
    const int max_similar = 0;                                          /* max_iterations_similar_transforms_ */
    if (num_threads < 1) num_threads = 1;
    for (;;) {
        /* determineCorrespondences */
        #pragma omp parallel for num_threads(num_threads) schedule(static)
        for (long i = 0; i < (long)n_src; i++) {
            const float* p = cur + 3 * i;
            float best = INFINITY; int32_t bi = -1;
            if (isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))
                for (size_t j = 0; j < n_tgt; j++) {
                    const float dx = p[0] - tg[3 * j], dy = p[1] - tg[3 * j + 1], dz = p[2] - tg[3 * j + 2];
                    const float d2 = (dx * dx + dy * dy) + dz * dz;
                    if (d2 < best) { best = d2; bi = (int32_t)j; }
                }
            if (bi >= 0 && (double)best <= max_d2) { ci[i] = bi; cd[i] = best; } else ci[i] = -1;
        }
        size_t cnt = 0;
        for (size_t i = 0; i < n_src; i++) cnt += ci[i] >= 0;
        if (cnt < 3) { conv = 0; break; }                               /* min_number_correspondences_ */
        /* estimateRigidTransformation: umeyama */
        float ms[3] = { 0, 0, 0 }, mt[3] = { 0, 0, 0 };
        for (size_t i = 0; i < n_src; i++) if (ci[i] >= 0)
            for (int d = 0; d < 3; d++) { ms[d] += cur[3 * i + d]; mt[d] += tg[3 * (size_t)ci[i] + d]; }
        for (int d = 0; d < 3; d++) { ms[d] /= (float)cnt; mt[d] /= (float)cnt; }
        float sg[9] = { 0 };
        double mse = 0.0;
        for (size_t i = 0; i < n_src; i++) if (ci[i] >= 0) {
            float a[3], b[3];
            for (int d = 0; d < 3; d++) { a[d] = cur[3 * i + d] - ms[d]; b[d] = tg[3 * (size_t)ci[i] + d] - mt[d]; }
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) sg[r * 3 + c] += b[r] * a[c];
            mse += (double)cd[i];
        }
        for (int k = 0; k < 9; k++) sg[k] /= (float)cnt;
        mse /= (double)cnt;
        float T[16];
        orc_icp_umeyama(ms, mt, sg, T);
        /* transformCloud (input_transformed in place) and final = T * final */
        for (size_t i = 0; i < n_src; i++) {
            const float x = cur[3 * i], y = cur[3 * i + 1], z = cur[3 * i + 2];
            cur[3 * i]     = T[0] * x + T[1] * y + T[2]  * z + T[3];
            cur[3 * i + 1] = T[4] * x + T[5] * y + T[6]  * z + T[7];
            cur[3 * i + 2] = T[8] * x + T[9] * y + T[10] * z + T[11];
        }
        float F[16];
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) {
            float a = 0; for (int k = 0; k < 4; k++) a += T[r * 4 + k] * T_final[k * 4 + c];
            F[r * 4 + c] = a;
        }
        memcpy(T_final, F, sizeof(F));
        ++it;
        /* DefaultConvergenceCriteria::hasConverged */
        int is_similar = 0;
        if (it >= max_iter) { conv = 1; break; }                        /* CONVERGENCE_CRITERIA_ITERATIONS */
        const double cos_angle = 0.5 * ((double)T[0] + (double)T[5] + (double)T[10] - 1.0);
        const double tsq = (double)T[3] * T[3] + (double)T[7] * T[7] + (double)T[11] * T[11];
        if (cos_angle >= rot_thr && tsq <= trl_thr) { if (similar >= max_similar) { conv = 1; break; } is_similar = 1; }   /* TRANSFORM */
        if (fabs(mse - prev_mse) < mse_abs) { if (similar >= max_similar) { conv = 1; break; } is_similar = 1; }            /* ABS_MSE */
        if (fabs(mse - prev_mse) / prev_mse < mse_rel) { if (similar >= max_similar) { conv = 1; break; } is_similar = 1; }   /* REL_MSE */
        similar = is_similar ? similar + 1 : 0;
        prev_mse = mse;
    }
    /* getFitnessScore(): mean squared distance of the aligned source to its nearest target points */
    double fs = 0.0; size_t nr = 0;
    {
        float* a = (float*)malloc(sizeof(float) * 3 * (n_src ? n_src : 1));
        for (size_t i = 0; i < n_src; i++) {
            float p[3]; memcpy(p, sb + i * stride_bytes, 12);
            a[3 * i]     = T_final[0] * p[0] + T_final[1] * p[1] + T_final[2]  * p[2] + T_final[3];
            a[3 * i + 1] = T_final[4] * p[0] + T_final[5] * p[1] + T_final[6]  * p[2] + T_final[7];
            a[3 * i + 2] = T_final[8] * p[0] + T_final[9] * p[1] + T_final[10] * p[2] + T_final[11];
        }
        #pragma omp parallel for num_threads(num_threads) schedule(static)
        for (long i = 0; i < (long)n_src; i++) {
            const float* p = a + 3 * i;
            float best = INFINITY;
            if (isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))
                for (size_t j = 0; j < n_tgt; j++) {
                    const float dx = p[0] - tg[3 * j], dy = p[1] - tg[3 * j + 1], dz = p[2] - tg[3 * j + 2];
                    const float d2 = (dx * dx + dy * dy) + dz * dz;
                    if (d2 < best) best = d2;
                }
            cd[i] = best;
        }
        for (size_t i = 0; i < n_src; i++) if (cd[i] < INFINITY) { fs += (double)cd[i]; nr++; }
        free(a);
    }
    *converged = conv; *iterations = it; *fitness = nr ? fs / (double)nr : DBL_MAX;
    free(cur); free(tg); free(ci); free(cd);
    return 0;
}
