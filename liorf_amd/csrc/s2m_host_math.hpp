// s2m_host_math.hpp — host-side scalar pieces of the path (per scan, not per point):
// trans2Affine3f / pcl::getTransformation (reference src/mapOptmization.cpp:348-351),
// the LMOptimization trig (:1170-1175) and transformUpdate() (:1323-1363).
#pragma once
#include <cmath>
#include <utility>
#include "s2m_types.h"

namespace s2m {

// pose = {roll, pitch, yaw, x, y, z}; T row-major 3x4. fp32 throughout, libm sinf/cosf,
// term order of pcl::getTransformation (A=cos yaw, B=sin yaw, C=cos pitch, D=sin pitch,
// E=cos roll, F=sin roll).
inline void host_pose_to_transform(const float t[6], float T[12], float sc[6])
{
    const float A = cosf(t[2]), B = sinf(t[2]), C = cosf(t[1]), D = sinf(t[1]), E = cosf(t[0]), F = sinf(t[0]);
    const float DE = D * E, DF = D * F;
    T[0] = A * C; T[1] = A * DF - B * E; T[2]  = B * F + A * DE; T[3]  = t[3];
    T[4] = B * C; T[5] = A * E + B * DF; T[6]  = B * DE - A * F; T[7]  = t[4];
    T[8] = -D;    T[9] = C * F;          T[10] = C * E;          T[11] = t[5];
    if (sc) { sc[0] = B; sc[1] = A; sc[2] = D; sc[3] = C; sc[4] = F; sc[5] = E; }
}

struct Quat { double x, y, z, w; };

inline Quat quat_from_rpy(double roll, double pitch, double yaw)     // tf::Quaternion::setRPY
{
    const double hy = yaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
    const double cy = std::cos(hy), sy = std::sin(hy), cp = std::cos(hp), sp = std::sin(hp);
    const double cr = std::cos(hr), sr = std::sin(hr);
    return Quat{ sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy,
                 cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy };
}

inline Quat quat_slerp(const Quat& a, const Quat& b, double t)       // tf::Quaternion::slerp
{
    const double dot = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    const double la = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w);
    const double lb = std::sqrt(b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w);
    double cs = dot / (la * lb);
    cs = cs > 1.0 ? 1.0 : (cs < -1.0 ? -1.0 : cs);
    const double theta = std::acos(cs < 0 ? -cs : cs);
    if (theta == 0.0) return a;
    const double d = 1.0 / std::sin(theta), s0 = std::sin((1.0 - t) * theta);
    double s1 = std::sin(t * theta);
    if (dot < 0) s1 = -s1;
    return Quat{ (a.x * s0 + b.x * s1) * d, (a.y * s0 + b.y * s1) * d,
                 (a.z * s0 + b.z * s1) * d, (a.w * s0 + b.w * s1) * d };
}

inline void quat_to_rpy(const Quat& q, double& roll, double& pitch, double& yaw)   // Matrix3x3::getRPY
{
    const double d = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w, s = 2.0 / d;
    const double xs = q.x * s, ys = q.y * s, zs = q.z * s;
    const double wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
    const double xx = q.x * xs, xy = q.x * ys, xz = q.x * zs, yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
    const double m00 = 1.0 - (yy + zz), m01 = xy - wz, m02 = xz + wy;
    const double m10 = xy + wz, m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy);
    if (std::fabs(m20) >= 1.0) {
        yaw = 0.0;
        if (m20 < 0) { pitch = M_PI / 2.0; roll = std::atan2(m01, m02); }
        else         { pitch = -M_PI / 2.0; roll = std::atan2(-m01, -m02); }
        return;
    }
    pitch = -std::asin(m20);
    const double cp = std::cos(pitch);
    roll = std::atan2(m21 / cp, m22 / cp);
    yaw = std::atan2(m10 / cp, m00 / cp);
}

inline float constraint_transformation(float value, float limit)      // :1355-1363
{
    if (value < -limit) value = -limit;
    if (value > limit) value = limit;
    return value;
}

// transformUpdate (:1323-1353): optional IMU roll/pitch slerp, clamps, affine of the result.
inline void host_transform_update(const s2m_params& p, const s2m_imu_init* imu, float t[6], float affine[12])
{
    if (imu && imu->imuAvailable == 1 && p.imu_type) {                 // `cloudInfo.imuAvailable == true` (:1325): int64 == 1
        if ((double)std::fabs(imu->imuPitchInit) < 1.4) {               // std::abs(float) < 1.4 (:1327): compared as doubles
            const double w = (double)p.imu_rpy_weight;
            double r, pi, y;
            quat_to_rpy(quat_slerp(quat_from_rpy(t[0], 0, 0), quat_from_rpy(imu->imuRollInit, 0, 0), w), r, pi, y);
            t[0] = (float)r;
            quat_to_rpy(quat_slerp(quat_from_rpy(0, t[1], 0), quat_from_rpy(0, imu->imuPitchInit, 0), w), r, pi, y);
            t[1] = (float)pi;
        }
    }
    t[0] = constraint_transformation(t[0], p.rot_tol);
    t[1] = constraint_transformation(t[1], p.rot_tol);
    t[5] = constraint_transformation(t[5], p.z_tol);
    host_pose_to_transform(t, affine, nullptr);
}

// Eigen::umeyama without scaling, as pcl::registration::TransformationEstimationSVD uses it (reference
// src/mapOptmization.cpp:583 -> icp.align, PCL 1.10 [ext]): R = U S V^T of the cross-covariance sigma =
// (1/n) sum (tgt - mean_tgt)(src - mean_src)^T with S(2) = -1 if det(U) det(V) < 0, t = mean_tgt - R mean_src.
// The 3x3 SVD is a one-sided Jacobi in fp64 (Eigen: JacobiSVD<Matrix3f>); T is row-major 4x4.
inline void host_svd3(const double A[9], double U[9], double S[3], double V[9])
{
    double W[9];
    for (int i = 0; i < 9; i++) { W[i] = A[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double a = 0, b = 0, c = 0;
                for (int k = 0; k < 3; k++) { a += W[k * 3 + p] * W[k * 3 + p]; b += W[k * 3 + q] * W[k * 3 + q]; c += W[k * 3 + p] * W[k * 3 + q]; }
                off += c * c;
                if (std::fabs(c) <= 1e-300 || std::fabs(c) <= 1e-17 * std::sqrt(a * b)) continue;
                const double zeta = (b - a) / (2.0 * c);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
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
        double nrm = 0;
        for (int k = 0; k < 3; k++) nrm += W[k * 3 + j] * W[k * 3 + j];
        S[j] = std::sqrt(nrm);
    }
    for (int i = 0; i < 2; i++)
        for (int j = i + 1; j < 3; j++)
            if (S[j] > S[i]) {
                std::swap(S[i], S[j]);
                for (int k = 0; k < 3; k++) { std::swap(W[k * 3 + i], W[k * 3 + j]); std::swap(V[k * 3 + i], V[k * 3 + j]); }
            }
    for (int j = 0; j < 3; j++)
        for (int k = 0; k < 3; k++) U[k * 3 + j] = S[j] > 1e-300 ? W[k * 3 + j] / S[j] : 0.0;
    if (S[2] <= 1e-12 * S[0]) {          // rank-deficient: complete U to an orthonormal basis
        U[2] = U[3] * U[7] - U[6] * U[4]; U[5] = U[6] * U[1] - U[0] * U[7]; U[8] = U[0] * U[4] - U[3] * U[1];
    }
}

inline double host_det3(const double M[9])
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

inline void host_umeyama(const float mean_src[3], const float mean_tgt[3], const float sigma[9], float T[16])
{
    double A[9], U[9], S[3], V[9];
    for (int i = 0; i < 9; i++) A[i] = (double)sigma[i];
    host_svd3(A, U, S, V);
    const double sgn[3] = { 1.0, 1.0, (host_det3(U) * host_det3(V) < 0) ? -1.0 : 1.0 };
    float R[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double a = 0;
            for (int k = 0; k < 3; k++) a += U[i * 3 + k] * sgn[k] * V[j * 3 + k];
            R[i * 3 + j] = (float)a;
        }
    for (int i = 0; i < 16; i++) T[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[i * 4 + j] = R[i * 3 + j];
        T[i * 4 + 3] = mean_tgt[i] - (R[i * 3 + 0] * mean_src[0] + R[i * 3 + 1] * mean_src[1] + R[i * 3 + 2] * mean_src[2]);
    }
}

}  // namespace s2m
