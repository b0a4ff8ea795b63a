"""Writes tests/golden/sc_ringkey_ref_golden.npz: the ring-key candidates of SCManager::detectLoopClosureID
(reference include/Scancontext.cpp:288-296) for the 340-key-frame test sequence, computed by the REFERENCE's own
search code - include/KDTreeVectorOfVectorsAdaptor.h + include/nanoflann.hpp compiled from /root/reference into
oracle/_ref by oracle/Makefile (nfref_ringkey_knn in oracle/nanoflann_ref.cpp).  Run in the authoring container
(needs /root/reference):  python tests/golden/make_golden_ringkey.py

Stored: the fp32 ring keys of every key frame (so a test can tell a changed input from a changed result), and per
detection the number of keys searched (tree rebuilt every 10th call over all but the 30 newest keys, :270-283), the
3 candidate indices and their squared distances.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import oracle as O                      # noqa: E402
from test_scancontext_cpu import sequence_340       # noqa: E402


def main():
    descs = sequence_340()
    keys = np.stack([O.make_ringkey(d).astype(np.float32) for d in descs])       # eig2stdvec: double -> float (:62-66)
    n_search = np.zeros(len(descs), np.int32)
    cand_idx = np.zeros((len(descs), 3), np.int32)
    cand_d2 = np.zeros((len(descs), 3), np.float32)
    counter, ns = 0, 0
    for i in range(len(descs)):
        n = i + 1
        if n < 31:
            continue
        if counter % 10 == 0:
            ns = n - 30
        counter += 1
        idx, d2, found = O.nanoflann_ringkey_knn(keys[:ns], keys[i], 3)
        n_search[i], cand_idx[i], cand_d2[i] = ns, idx, d2
        if found < 3:                                # the reference's result vectors keep their zero initialisation (:289-290)
            cand_idx[i, found:] = 0
            cand_d2[i, found:] = 0
    out = os.path.join(HERE, "sc_ringkey_ref_golden.npz")
    np.savez_compressed(out, ringkeys=keys, n_search=n_search, cand_idx=cand_idx, cand_d2=cand_d2)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
