// s2m_host_math.hpp — host-side scalar pieces of the path (per scan, not per point):
// trans2Affine3f / pcl::getTransformation (reference src/mapOptmization.cpp:348-351),
// the LMOptimization trig (:1170-1175) and transformUpdate() (:1323-1363).
#pragma once
#include <cmath>
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
    if (imu && imu->imuAvailable && p.imu_type) {
        if (std::fabs(imu->imuPitchInit) < 1.4f) {
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

}  // namespace s2m
