// nanoflann_ref.cpp — ORACLE-SIDE cross-check driver (test infrastructure, NOT product code).
//
// Compiles the reference's VENDORED third-party kd-tree (include/nanoflann.hpp, v0x132,
// used in the reference by ScanContext, include/Scancontext.cpp:278,295) from where it
// lies under /root/reference/include, into oracle/_ref/libnanoflann_ref.so. The reference's
// hot path uses FLANN through PCL (src/mapOptmization.cpp:1087), which is not vendored;
// nanoflann is the same algorithm family (single exact kd-tree, L2_Simple, sorted result
// set) and serves as an independent check of the oracle's 5-NN indices and distances.
// No reference source is copied: this file only #includes the header at build time.
#include <cstddef>
#include <cstdint>
#include <vector>
#include "nanoflann.hpp"

namespace {
struct Cloud {
    const float* xyz; size_t n;
    inline size_t kdtree_get_point_count() const { return n; }
    inline float kdtree_get_pt(const size_t idx, const size_t dim) const { return xyz[3 * idx + dim]; }
    template <class BBOX> bool kdtree_get_bbox(BBOX&) const { return false; }
};
typedef nanoflann::KDTreeSingleIndexAdaptor<nanoflann::L2_Simple_Adaptor<float, Cloud>, Cloud, 3, int32_t> Tree;
}

extern "C" int nfref_knn5(const float* map_xyz, size_t n_m, const float* q_xyz, size_t n_q,
                          int32_t* idx5, float* d2_5, int leaf_max)
{
    if (n_m < 5) return -1;
    Cloud cloud{ map_xyz, n_m };
    Tree tree(3, cloud, nanoflann::KDTreeSingleIndexAdaptorParams(leaf_max > 0 ? leaf_max : 15));
    tree.buildIndex();
    for (size_t i = 0; i < n_q; i++) {
        nanoflann::KNNResultSet<float, int32_t> rs(5);
        rs.init(idx5 + 5 * i, d2_5 + 5 * i);
        tree.findNeighbors(rs, q_xyz + 3 * i, nanoflann::SearchParams());
    }
    return 0;
}
