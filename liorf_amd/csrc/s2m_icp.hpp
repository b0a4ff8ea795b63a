// s2m_icp.hpp — ICP loop-closure alignment on the device:
// pcl::IterativeClosestPoint<PointXYZI, PointXYZI> as the reference configures it at
// src/mapOptmization.cpp:571-586 and :663-678.  Implemented in s2m_icp.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace s2m {

struct IcpWorkspace;
IcpWorkspace* icp_create();
void          icp_destroy(IcpWorkspace* w);

struct IcpParams { double max_corr_dist; int max_iter; double trans_eps; double fit_eps; };
struct IcpResult { float T[16]; int converged; int iterations; double fitness; };

// src / tgt: device records (x, y, z at byte 0/4/8). Synchronises `stream` once per iteration (the
// convergence test of the reference runs on the host between iterations).
hipError_t icp_align(IcpWorkspace* w, hipStream_t stream, const unsigned char* d_src, size_t n_src,
                     const unsigned char* d_tgt, size_t n_tgt, size_t stride, const IcpParams& prm, IcpResult* res);

}  // namespace s2m
