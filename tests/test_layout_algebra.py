"""CPU: the engine's layout algebra (graal_amd/csrc/frag_ops.h, host build) against the oracle's restatement of
the reference mutation kernels, on randomised layouts incl. circular / singleton / two-fragment contigs."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import util


def _cases(seed, n_layouts, n_frags, pairs_per_layout):
    rng = np.random.RandomState(seed)
    for _ in range(n_layouts):
        n = int(rng.randint(2, n_frags + 1))
        s = util.random_layout(rng, n)
        util.check_invariants(s)
        max_id = int(s["id_c"].max())
        for _ in range(pairs_per_layout):
            fA, fB = rng.choice(n, 2, replace=False)
            yield s, int(fA), int(fB), max_id


def test_random_layouts_are_valid():
    rng = np.random.RandomState(0)
    for _ in range(50):
        util.check_invariants(util.random_layout(rng, int(rng.randint(1, 40))))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_apply_move_matches_oracle_bit_exact(seed):
    n_checked = 0
    for s, fA, fB, max_id in _cases(seed, 60, 14, 6):
        for op in range(13):
            want, stale = util.oracle_candidate(s, fA, fB, op, max_id)
            got, n_stale = util.hc_apply_move(s, fA, fB, op, max_id)
            assert not stale and n_stale == 0, "the paste stale-slot branch must be unreachable from a valid layout"
            for k in O.FIELDS:
                assert np.array_equal(got[k], want[k]), (seed, fA, fB, op, k, got[k], want[k], {f: s[f] for f in O.FIELDS})
            n_checked += 1
    assert n_checked > 4000


@pytest.mark.parametrize("seed", [11, 12])
def test_candidates_keep_the_layout_valid(seed):
    for s, fA, fB, max_id in _cases(seed, 40, 12, 4):
        for op in range(13):
            got, _ = util.hc_apply_move(s, fA, fB, op, max_id)
            util.check_invariants(got)


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_piece_transforms_reproduce_every_candidate(seed):
    """For each candidate, every fragment's (label, start_bp, ori, circ, l_cont_bp) follows from its piece's
    affine transform -- the only thing the device scan knows about a candidate."""
    for s, fA, fB, max_id in _cases(seed, 50, 14, 5):
        piece, xf, changed, rep = util.hc_piece_tables(s, fA, fB, max_id)
        cA, cB = s["id_c"][fA], s["id_c"][fB]
        inside = (s["id_c"] == cA) | (s["id_c"] == cB)
        assert np.array_equal(piece > 0, inside)
        for p in range(1, 7):  # representative exists iff the piece is non-empty, and lies in it
            assert (rep[p] >= 0) == bool(np.any(piece == p))
            if rep[p] >= 0:
                assert piece[rep[p]] == p
        for op in range(13):
            want, _ = util.oracle_candidate(s, fA, fB, op, max_id)
            for f in range(len(piece)):
                p = piece[f]
                if p == 0:
                    for k in ("id_c", "start_bp", "ori", "circ", "l_cont_bp", "pos"):
                        assert want[k][f] == s[k][f]
                    continue
                label, sigma, off, circ, lbp = xf[op, p]
                start = s["start_bp"][f] + off if sigma > 0 else off - (s["start_bp"][f] + s["len_bp"][f])
                assert (label, start, s["ori"][f] * sigma, circ, lbp) == (
                    want["id_c"][f], want["start_bp"][f], want["ori"][f], want["circ"][f], want["l_cont_bp"][f]), (fA, fB, op, f, p)


def _centre2(state, f):  # twice the centre coordinate (exact integer)
    return 2 * int(state["start_bp"][f]) + int(state["len_bp"][f])


@pytest.mark.parametrize("seed", [31, 32])
def test_relation_flags_cover_every_geometry_change(seed):
    """A pair of fragments whose cis/trans status, centre distance or circular model differs between the current
    layout and a candidate must sit in pieces flagged `changed` (so the scan revisits its contacts)."""
    for s, fA, fB, max_id in _cases(seed, 40, 12, 4):
        piece, xf, changed, rep = util.hc_piece_tables(s, fA, fB, max_id)
        n = len(piece)
        for op in range(13):
            want, _ = util.oracle_candidate(s, fA, fB, op, max_id)
            for x in range(n):
                for y in range(x + 1, n):
                    cis0, cis1 = s["id_c"][x] == s["id_c"][y], want["id_c"][x] == want["id_c"][y]
                    differs = cis0 != cis1
                    if cis0 and cis1:
                        differs = abs(_centre2(s, x) - _centre2(s, y)) != abs(_centre2(want, x) - _centre2(want, y))
                        differs |= s["circ"][x] != want["circ"][x]
                        differs |= bool(want["circ"][x] == 1 and s["l_cont_bp"][x] != want["l_cont_bp"][x])
                    if differs:
                        p, q = int(piece[x]), int(piece[y])
                        assert p > 0 and q > 0, (fA, fB, op, x, y)
                        assert (int(changed[op]) >> (p * 8 + q)) & 1, (fA, fB, op, x, y, p, q)
