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

// ---- ScanContext ring-key search (section 8(f) row F3) through the reference's OWN adaptor ------------------
// include/KDTreeVectorOfVectorsAdaptor.h needs only nanoflann.hpp + <vector>, so SCManager's search can be run
// here exactly as the reference runs it: the types of include/Scancontext.h:42-43 (KeyMat = vector<vector<float>>,
// InvKeyTree = KDTreeVectorOfVectorsAdaptor<KeyMat, float>: metric_L2, run-time dimension), the tree built as
// at include/Scancontext.cpp:278 (dim PC_NUM_RING = 20, max leaf 10) over the keys to search, and the query of
// :289-295 (KNNResultSet<float> of NUM_CANDIDATES_FROM_TREE entries over zero-initialised vectors,
// SearchParams(10)).  keys: n_search rows of `dim` floats; out_idx / out_d2: k entries.
#include "KDTreeVectorOfVectorsAdaptor.h"

extern "C" int nfref_ringkey_knn(const float* keys, size_t n_search, int dim, const float* query, int k,
                                 int64_t* out_idx, float* out_d2)
{
    if (n_search == 0 || dim <= 0 || k <= 0) return -1;
    typedef std::vector<std::vector<float> > KeyMat;
    typedef KDTreeVectorOfVectorsAdaptor<KeyMat, float> InvKeyTree;
    KeyMat to_search(n_search);
    for (size_t i = 0; i < n_search; i++) to_search[i].assign(keys + i * (size_t)dim, keys + (i + 1) * (size_t)dim);
    std::vector<float> curr_key(query, query + dim);
    InvKeyTree tree((size_t)dim, to_search, 10);
    std::vector<size_t> candidate_indexes((size_t)k);
    std::vector<float> out_dists_sqr((size_t)k);
    nanoflann::KNNResultSet<float> knnsearch_result((size_t)k);
    knnsearch_result.init(&candidate_indexes[0], &out_dists_sqr[0]);
    tree.index->findNeighbors(knnsearch_result, &curr_key[0], nanoflann::SearchParams(10));
    for (int j = 0; j < k; j++) { out_idx[j] = (int64_t)candidate_indexes[(size_t)j]; out_d2[j] = out_dists_sqr[(size_t)j]; }
    return (int)knnsearch_result.size();
}
