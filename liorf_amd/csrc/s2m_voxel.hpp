// s2m_voxel.hpp — device voxel-grid filter and cloud transform: the two stages that produce the
// inputs of the scan-to-map path.
//   downsampleCurrentScan()  reference src/mapOptmization.cpp:1061-1067   (VoxelGrid, leaf 0.4)
//   extractCloud()           reference src/mapOptmization.cpp:1014-1039   (transformPointCloud of
//                            the chosen key frames :310-329, concatenation, VoxelGrid, leaf 0.4-0.5)
// Implemented in s2m_voxel.hip (own translation unit; the stable radix sort of the (voxel, point) pairs is hand-written there).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace s2m {

struct VoxWorkspace;                       // grow-only device scratch of the filter

VoxWorkspace* vox_create();
void          vox_destroy(VoxWorkspace* w);

struct VoxResult {
    size_t n_out = 0;                      // voxels (records needed in `out`)
    int    leaf_too_small = 0;             // PCL's "Leaf size is too small" case: output = input
};

// pcl::VoxelGrid<PointXYZI>::applyFilter on device records (x, y, z at byte 0/4/8, intensity at
// byte 16 when stride >= 20). Writes min(n_out, cap) records {centroid xyz, 1.0f, mean intensity, 0...}
// in ascending voxel-index order and synchronises `stream` to return the count.
hipError_t vox_downsample(VoxWorkspace* w, hipStream_t stream, const unsigned char* d_in, size_t n, size_t stride,
                          float leaf, unsigned char* d_out, size_t out_stride, size_t cap, VoxResult* res);

// transformPointCloud (:310-329) of `n_frames` key-frame clouds: frame f is the device buffer h_src[f] holding
// h_offsets[f+1] - h_offsets[f] records and uses the row-major 3x4 transform h_T[12*f ..]; the transformed
// records are written back to back (the reference's `*laserCloudSurfFromMap += ...`) into d_out.
hipError_t vox_transform_frames(VoxWorkspace* w, hipStream_t stream, const unsigned char* const* h_src, size_t stride,
                                const int32_t* h_offsets, const float* h_T, int n_frames,
                                unsigned char* d_out, size_t out_stride);

}  // namespace s2m
