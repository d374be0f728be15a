"""CPU: the union set of a step, its global pieces, the classes of equal inputs per piece pair and the unit list
(graal_amd/csrc/strict_sets.h -- shared by k_gprep / k_strict2 and the host) against brute force: the candidate move applied to
every fragment ITSELF (frag_ops.h: apply_move, which tests/test_layout_algebra.py holds to the oracle's kernels,
kernels3.cu:239-2070), for every pair of fragments of the K sets and every one of the K x 13 candidates."""
import ctypes

import numpy as np
import pytest

from tests import util

_i32p = ctypes.POINTER(ctypes.c_int32)


def union_check(s, fA, fBs, max_id, quirk, seg, tile=64):
    hc = util.hostcheck()
    hc.hc_union_check.restype = ctypes.c_int
    fb = np.asarray(fBs, np.int32)
    info = np.zeros(10, np.int64)
    bad = hc.hc_union_check(int(fA), fb.ctypes.data_as(_i32p), len(fb), int(max_id), util._ptrs(s), len(s["pos"]), int(quirk), int(seg),
                            info.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), int(tile))
    return bad, info


@pytest.mark.parametrize("seed,n,n_contigs,K,p_circ", [(1, 40, 3, 3, 0.0), (2, 60, 4, 5, 0.4), (3, 90, 2, 10, 0.3), (4, 30, 12, 4, 0.3),
                                                       (5, 150, 3, 5, 0.0), (6, 200, 2, 10, 0.5), (7, 25, 1, 6, 1.0), (8, 80, 6, 10, 0.2)])
def test_union_set_classes_and_units_against_brute_force(seed, n, n_contigs, K, p_circ):
    rng = np.random.RandomState(seed)
    tot_classes = tot_cands = 0
    for trial in range(6):
        s = util.random_layout(rng, n, n_contigs=n_contigs, p_circ=p_circ, max_len_bp=3000)
        if trial % 3 == 2:   # an inactive copy of a repeated bin somewhere (activity swaps, kernels3.cu:283)
            f = int(rng.randint(n)); s["rep"][f] = 1
        max_id = int(s["id_c"].max())
        for _ in range(3):
            fA = int(rng.randint(n))
            if trial % 2 == 0:   # neighbours as the sampler draws them: mostly close to fA along its contig
                same = np.nonzero(s["id_c"] == s["id_c"][fA])[0]
                near = same[np.argsort(np.abs(s["pos"][same] - s["pos"][fA]))][1:K + 3]
                pool = np.concatenate([near, rng.randint(0, n, size=K)])
                fBs = []
                for f in pool:
                    if int(f) != fA and int(f) not in fBs:
                        fBs.append(int(f))
                fBs = sorted(fBs[:K])
            else:
                fBs = sorted(int(v) for v in rng.choice(n, K, replace=False))   # (may contain fA: that neighbour is not live)
            for quirk in (0, 1):
                # (tiles of 64 fragments, or 32: what several sub-fragments per bin are tiled by)
                bad, info = union_check(s, fA, fBs, max_id, quirk, seg=int(rng.choice([1, 2, 4, 16])), tile=int(rng.choice([64, 32])))
                assert bad == 0, (seed, trial, fA, fBs, quirk, info)
                assert info[0] <= 3 * K + 3 and info[2] <= K + 1
                tot_classes += info[4]; tot_cands += info[7]
    assert tot_classes > 0 and tot_cands > 0
