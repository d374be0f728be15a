// strict_sets.h -- reference arithmetic (GRAAL_MODE_STRICT), the step's UNION SET and its classes of equal inputs, written once
// for host and device (the CPU suite checks it against brute force through host_check.cpp; the kernels are k_gprep / k_strict2).
//
// The reference prices the 13 candidates of every neighbour fB_k independently: every pixel of contig(fA) u contig(fB_k) again,
// from the float32 kb coordinates of the candidate layout (sub_compute_likelihood, kernels3.cu:3259-3718, driven by
// cuda_lib_gl.py:2508-2545).  What a pixel's value depends on are the model's INPUTS -- cis or trans, the two fragments' new bp
// offsets and orientations, the circular model -- and over the K x 13 candidates of a step most of them coincide: ejecting or
// flipping fA does not depend on fB_k at all, the current layout's value is the same for every neighbour, and the K sets are
// mostly the same two or three contigs.  So the step is priced over the UNION of the K sets, cut at fA and at every fB_k into
// "global pieces" (maximal position ranges that every one of the K x 13 candidates maps with one affine transform), and per
// pair of global pieces the candidates fall into classes of equal inputs (frag_ops.h: inputs_key): a class is priced once, its
// value added to every candidate in it; the class of the current layout's own inputs is never priced.  Every term that is
// dropped is exactly equal to one that is kept, or exactly zero: the sums are those of the O(m^2) kernel, bit for bit.
#pragma once
#include "frag_ops.h"

namespace graal {

constexpr int US_MAXK = 10;                                   // = GRAAL_MAX_NEIGHBOURS
constexpr int US_MAXC = US_MAXK + 1;                          // contigs of a step: contig(fA) + one per neighbour
constexpr int US_MAXP = 3 * US_MAXK + 3;                      // global pieces: 2 per cut fragment (itself, the run behind it) + 1 per contig
constexpr int US_MAXPAIRS = US_MAXP * (US_MAXP + 1) / 2;      // unordered pairs (g <= h) of global pieces
constexpr int US_NCAND = US_MAXK * N_OPS;                     // candidates of a step, index k * 13 + op
constexpr int US_TILE = 64;                                   // fragments per tile at most (one wave); a set is tiled by USet::tile of them: 64, or 32 with
                                                              // several sub-fragments per bin (the two halves of a wave then take two fragments of the segment)

struct UContig { int label, base, len, lbp, circ, pad; };     // perm[base .. base + len): its fragments in position order
struct UPiece {
    int contig, lo, n, tile0;                                 // contig index, first position, fragments, first tile
    unsigned char pk[US_MAXK];                                // its piece id under neighbour k (frag_ops.h: piece_of), 0 = not in set k
    unsigned char pad[2];
};
struct USet {
    int n_contigs, n_pieces, n_tiles;
    unsigned live;                                            // bit k: neighbour k is a real pair (fB_k != fA): its set is part of the union
    unsigned mass;                                            // bit k: ... and its fragment pairs are priced here (else the table kernel did: small sets)
    int tile;                                                 // fragments per tile (US_TILE or 32)
    UContig c[US_MAXC];
    UPiece p[US_MAXP];
};
// what the layout says about one end of a proposal (fA, or a neighbour fB_k): its contig
struct UEnd { int label, pos, base, len, lbp, circ; };

GR_HD int upiece_tiles(int n, int tile) { return (n + tile - 1) / tile; }
GR_HD int upair_index(int g, int h) { return g * US_MAXP - g * (g - 1) / 2 + (h - g); }   // g <= h

// Build the union set, in two parts: the contigs, cuts, pieces and tiles (one thread), then every piece's id under every neighbour
// (independent entries: one thread each on the device).  cuts: workspace of US_MAXC * (US_MAXK + 1) ints, ncut: US_MAXC ints.
GR_HD void uset_add_piece(USet& U, int& np, int& tile, int ci, int lo, int n)
{
    if (n <= 0) return;
    UPiece& P = U.p[np];
    np += 1;
    P.contig = ci; P.lo = lo; P.n = n; P.tile0 = tile;
    P.pad[0] = 0; P.pad[1] = 0;
    tile += upiece_tiles(n, U.tile);
}
GR_HD void uset_build_geometry(USet& U, const UEnd& A, const UEnd* B, int K, unsigned live, unsigned mass, int* cuts, int* ncut, int tile_frags = US_TILE)
{
    constexpr int CW = US_MAXK + 1;
    U.live = live; U.mass = mass & live; U.tile = tile_frags;
    int nc = 1;
    U.c[0].label = A.label; U.c[0].base = A.base; U.c[0].len = A.len; U.c[0].lbp = A.lbp; U.c[0].circ = A.circ; U.c[0].pad = 0;
    cuts[0] = A.pos; ncut[0] = 1;
    for (int k = 0; k < K; k++) {
        if (!((live >> k) & 1u)) continue;
        int ci = -1;
        for (int i = 0; i < nc; i++) if (U.c[i].label == B[k].label) ci = i;
        if (ci < 0) {
            ci = nc++;
            U.c[ci].label = B[k].label; U.c[ci].base = B[k].base; U.c[ci].len = B[k].len; U.c[ci].lbp = B[k].lbp; U.c[ci].circ = B[k].circ;
            U.c[ci].pad = 0;
            ncut[ci] = 0;
        }
        // sorted insert, no duplicates
        int* cc = cuts + ci * CW;
        int n = ncut[ci], at = 0;
        while (at < n && cc[at] < B[k].pos) at++;
        if (at < n && cc[at] == B[k].pos) continue;
        for (int i = n; i > at; i--) cc[i] = cc[i - 1];
        cc[at] = B[k].pos;
        ncut[ci] = n + 1;
    }
    U.n_contigs = nc;
    int np = 0, tile = 0;
    for (int ci = 0; ci < nc; ci++) {
        const int* cc = cuts + ci * CW;
        int prev = 0;
        for (int i = 0; i < ncut[ci]; i++) {
            uset_add_piece(U, np, tile, ci, prev, cc[i] - prev);
            uset_add_piece(U, np, tile, ci, cc[i], 1);
            prev = cc[i] + 1;
        }
        uset_add_piece(U, np, tile, ci, prev, U.c[ci].len - prev);
    }
    U.n_pieces = np;
    U.n_tiles = tile;
}
GR_HD void uset_piece_pk(USet& U, const PieceKey* keys, int K, int g, int k)
{
    UPiece& P = U.p[g];
    P.pk[k] = (unsigned char)((k < K && ((U.live >> k) & 1u)) ? piece_of(keys[k], U.c[P.contig].label, P.lo) : 0);
}
GR_HD void uset_build(USet& U, const UEnd& A, const UEnd* B, const PieceKey* keys, int K, unsigned live, unsigned mass, int* cuts, int* ncut, int tile_frags = US_TILE)
{
    uset_build_geometry(U, A, B, K, live, mass, cuts, ncut, tile_frags);
    for (int g = 0; g < U.n_pieces; g++)
        for (int k = 0; k < US_MAXK; k++) uset_piece_pk(U, keys, K, g, k);
}

// the global piece of tile t, and the tile's offset (in tiles) inside it
GR_HD int utile_piece(const USet& U, int t, int& off)
{
    int g = 0;
    for (int i = 1; i < U.n_pieces; i++) g += (t >= U.p[i].tile0) ? 1 : 0;
    off = t - U.p[g].tile0;
    return g;
}

// the global piece of a fragment of the layout (label, position); -1: not in the union set
GR_HD int ufrag_piece(const USet& U, int label, int pos)
{
    int g = -1;
    for (int i = 0; i < U.n_pieces; i++) {
        const UPiece& P = U.p[i];
        if (U.c[P.contig].label == label && pos >= P.lo && pos < P.lo + P.n) g = i;
    }
    return g;
}

// the current layout's transform of a global piece (xf_identity of any of its fragments)
GR_HD Xf uxf_old(const USet& U, int g)
{
    const UContig& C = U.c[U.p[g].contig];
    Xf x; x.label = C.label; x.sigma = 1; x.off = 0; x.circ = C.circ; x.lbp = C.lbp;
    return x;
}

// One class of equal inputs of a piece pair (g <= h; "x" = piece g, "y" = piece h): the representative's transforms, reduced to what the
// contact model reads, and the candidates (k * 13 + op) of the class.  64 bytes.
struct GClass {
    int offx, offy, lbp;
    unsigned flags;                 // 1: sigma_x > 0   2: sigma_y > 0   4: cis (same label afterwards)   8: the contig is circular afterwards
    unsigned long long m0, m1;      // candidates 0..63, 64..127
    unsigned m2, rep;               // candidates 128, 129; the representative candidate
    unsigned long long w0, w1;      // the same, restricted to the neighbours whose fragment pairs are priced here (USet::mass): the queued
    unsigned w2, pad;               // contacts are priced for every candidate of the class, the fragment pairs for these
};
GR_HD GClass gclass_make(const Xf& a, const Xf& b, int rep)
{
    GClass c;
    const bool cis = a.label == b.label;
    c.offx = a.off; c.offy = b.off; c.lbp = (cis && a.circ == 1) ? a.lbp : 0;
    c.flags = (a.sigma > 0 ? 1u : 0u) | (b.sigma > 0 ? 2u : 0u) | (cis ? 4u : 0u) | ((cis && a.circ == 1) ? 8u : 0u);
    c.m0 = 0; c.m1 = 0; c.m2 = 0; c.rep = (unsigned)rep; c.w0 = 0; c.w1 = 0; c.w2 = 0; c.pad = 0;
    return c;
}
GR_HD bool ikey_eq(const InputsKey& a, const InputsKey& b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; }

// new start / orientation of a fragment under one side of a class
GR_HD int gclass_start(int sigma_pos, int off, int start_bp, int len_bp) { return sigma_pos ? start_bp + off : off - (start_bp + len_bp); }

// ---- the unit list: (tile ti <= tile tj, a segment of <= seg fragments of the "segment side" tile)
// The lanes of a wave take the fragments of the tile that holds MORE of them (a cut fragment is a piece, hence a tile, of its own:
// against a full tile it is one segment of one fragment, not 64 lanes of which one works); the other tile is walked in segments.
//   bits  0..15 ti   16..31 tj   32..37 first fragment of the segment (inside its tile)   38..43 fragments of the segment
//   bit 44: the lanes are tile ti's fragments (else tile tj's)
//   bits 45..47 r, 48..51 R: the unit is dealt to R waves, wave r of them prices the classes r, r + R, r + 2R, ... of its piece pair (each with
//   the current layout's values of its own: when a step has few units, their depth -- all classes of a pair one after the other --
//   is what the step waits for)
//   bits 52..61: the index of the tiles' pair of global pieces (upair_index): what k_strict2 needs to find the unit's classes -- and their
//   number, by which it deals a step's few units to its many waves -- without walking the set's pieces
GR_HD unsigned long long uunit_pack(int ti, int tj, int j0, int cnt, int lanes_first, int r = 0, int R = 1, int pair = 0)
{
    return (unsigned long long)ti | ((unsigned long long)tj << 16) | ((unsigned long long)j0 << 32) | ((unsigned long long)cnt << 38) |
           ((unsigned long long)(lanes_first ? 1 : 0) << 44) | ((unsigned long long)r << 45) | ((unsigned long long)R << 48) |
           ((unsigned long long)pair << 52);
}
GR_HD int uunit_pair(unsigned long long u) { return (int)((u >> 52) & 1023ull); }
GR_HD int utile_count(const USet& U, int t)
{
    int off;
    const int g = utile_piece(U, t, off);
    const int left = U.p[g].n - off * U.tile;
    return left < U.tile ? left : U.tile;
}

} // namespace graal
