// frag_ops.h -- fragment-layout algebra of the GRAAL sampler, written once for host and device.
//
// The reference materialises every candidate layout as a full 14-array SoA with one elementwise
// kernel per mutation (kernels3.cu:239-2070, driven by cuda_lib_gl.py:841-954).  Here a mutation is a
// pure function on ONE fragment record given the records of the two fragments that define the move,
// so that the same code (a) commits the accepted move (one elementwise launch) and (b) is evaluated on
// a handful of representative fragments to obtain, per candidate, a per-piece affine transform of the
// genomic coordinates -- the only thing the likelihood scan needs.
//
// Semantics follow the reference kernels branch for branch; tests/ compare every function with the
// oracle on randomised layouts (linear, circular, degenerate contigs).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GR_HD __host__ __device__ __forceinline__
#else
#define GR_HD inline
#endif

namespace graal {

enum { N_OPS = 13, MAX_PIECES = 6 };

struct Rec {
    int pos, id_c, start_bp, len_bp, circ, prev, next, l_cont, l_cont_bp, ori, rep, activ, id_d;
};

// ---------------------------------------------------------------- single mutations
// pop_out_frag (kernels3.cu:329): P = record of the ejected fragment in the input layout.
GR_HD Rec m_pop_out(Rec r, int f, const Rec& P, int id_f_pop, int max_id)
{
    (void)id_f_pop;
    if (P.l_cont < 2 || r.id_c != P.id_c) return r;
    const bool big = P.l_cont > 2;
    if (r.pos == P.pos) { // the ejected fragment becomes a singleton contig
        r.pos = 0; r.id_c = max_id + 1; r.start_bp = 0; r.circ = 0; r.ori = 1;
        r.prev = -1; r.next = -1; r.l_cont = 1; r.l_cont_bp = r.len_bp;
        return r;
    }
    if (r.pos < P.pos) {
        if (big) {
            if (f == P.next && P.circ == 1) r.prev = P.prev;
            if (r.pos == P.pos - 1) r.next = P.next;
        } else { r.circ = 0; r.prev = -1; r.next = -1; }
    } else {
        if (big) {
            const int prev_fi = r.prev, next_fi = r.next;
            r.prev = (r.pos == P.pos + 1) ? P.prev : prev_fi;
            r.next = (f == P.prev && P.circ == 1) ? P.next : next_fi;
        } else { r.circ = 0; r.prev = -1; r.next = -1; }
        r.pos -= 1;
        r.start_bp -= P.len_bp;
    }
    r.l_cont -= 1;
    r.l_cont_bp -= P.len_bp;
    return r;
}

// flip_frag (kernels3.cu:239)
GR_HD Rec m_flip(Rec r, int f, int id_f_flip)
{
    if (f == id_f_flip) r.ori = -r.ori;
    return r;
}

// swap_activity_frag (kernels3.cu:283)
GR_HD Rec m_swap_activity(Rec r, int f, int id_f, int max_id)
{
    if (f == id_f && r.rep == 1) {
        const int a = r.activ;
        r.activ = (a == 0) ? 1 : 0;
        if (a == 0) r.id_c = max_id + 1; else if (a != 1) r.id_c = 0;
    }
    return r;
}

// pop_in_frag_1..4 (kernels3.cu:565/814/1081/1267).  P, I = records of the inserted fragment and of
// the insertion target in the INPUT layout (i.e. after pop_out).
GR_HD Rec m_pop_in(int which, Rec r, int f, const Rec& P, const Rec& I, int id_f_pop, int id_f_ins, int max_id,
                   int ori_f_pop)
{
    if (!(I.activ == 1 && P.activ == 1)) return r;
    const int pos = r.pos, start = r.start_bp, prev = r.prev, next = r.next;
    const bool is_pop = (f == id_f_pop);
    if (!is_pop && r.id_c != I.id_c) return r;
    const int tailpos = I.pos + 1, tailbp = I.start_bp + I.len_bp; // first position / bp after f_ins
    switch (which) {
    case 1: // split the target contig before f_ins; f_pop becomes the head of [f_pop, f_ins, ...]
        if (I.circ == 0) {
            const int n_new = I.l_cont - I.pos + 1, bp_new = I.l_cont_bp - I.start_bp + P.len_bp;
            if (is_pop) {
                r.pos = 0; r.start_bp = 0; r.circ = 0; r.ori = ori_f_pop; r.prev = -1; r.next = id_f_ins;
                r.id_c = max_id + 1; r.l_cont = n_new; r.l_cont_bp = bp_new;
            } else if (pos < I.pos) {
                r.circ = 0; if (pos == I.pos - 1) r.next = -1;
                r.l_cont = I.pos; r.l_cont_bp = I.start_bp;
            } else {
                r.id_c = max_id + 1; r.circ = 0; r.l_cont = n_new; r.l_cont_bp = bp_new;
                if (pos == I.pos) { r.pos = 1; r.start_bp = P.len_bp; r.ori = I.ori; r.prev = id_f_pop; r.next = I.next; }
                else { r.pos = pos - I.pos + 1; r.start_bp = start - I.start_bp + P.len_bp; }
            }
        } else { // circular target: linearised at f_ins, keeps its label
            if (is_pop) {
                r.pos = 0; r.start_bp = 0; r.ori = ori_f_pop; r.prev = -1; r.next = id_f_ins;
            } else if (pos < I.pos) {
                r.pos = I.l_cont - I.pos + pos + 1;
                r.start_bp = I.l_cont_bp - I.start_bp + start + P.len_bp;
                if (pos == I.pos - 1) r.next = -1;
            } else if (pos == I.pos) {
                r.pos = 1; r.start_bp = P.len_bp; r.len_bp = I.len_bp; r.ori = I.ori; r.prev = id_f_pop; r.next = I.next;
            } else {
                r.pos = pos - I.pos + 1; r.start_bp = start - I.start_bp + P.len_bp;
                if (f == I.prev) r.next = -1;
            }
            r.id_c = I.id_c; r.circ = 0; r.l_cont = I.l_cont + 1; r.l_cont_bp = I.l_cont_bp + P.len_bp;
        }
        return r;
    case 2: // split the target contig after f_ins; f_pop becomes the tail of [..., f_ins, f_pop]
        if (I.circ == 0) {
            const int n_new = I.pos + 2, bp_new = tailbp + P.len_bp;
            if (is_pop) {
                r.pos = tailpos; r.id_c = I.id_c; r.start_bp = tailbp; r.circ = 0; r.ori = ori_f_pop;
                r.prev = id_f_ins; r.next = -1; r.l_cont = n_new; r.l_cont_bp = bp_new;
            } else if (pos <= I.pos) {
                r.circ = 0; r.l_cont = n_new; r.l_cont_bp = bp_new;
                if (pos == I.pos) { r.ori = I.ori; r.prev = I.prev; r.next = id_f_pop; }
            } else {
                r.pos = pos - tailpos; r.id_c = max_id + 1; r.start_bp = start - tailbp; r.circ = 0;
                if (pos == tailpos) r.prev = -1;
                r.l_cont = I.l_cont - tailpos; r.l_cont_bp = I.l_cont_bp - tailbp;
            }
        } else {
            const int wrap = I.l_cont - tailpos, wrapbp = I.l_cont_bp - tailbp; // fragments after f_ins go first
            if (is_pop) {
                r.pos = wrap + tailpos; r.start_bp = wrapbp + tailbp; r.ori = ori_f_pop; r.prev = id_f_ins; r.next = -1;
            } else if (pos < I.pos) {
                r.pos = wrap + pos; r.start_bp = wrapbp + start;
                if (f == I.next) r.prev = -1;
            } else if (pos == I.pos) {
                r.pos = wrap + I.pos; r.start_bp = wrapbp + I.start_bp; r.len_bp = I.len_bp;
                r.prev = I.prev; r.next = id_f_pop;
            } else {
                r.pos = pos - tailpos; r.start_bp = start - tailbp;
                if (pos == tailpos) r.prev = -1;
            }
            r.id_c = I.id_c; r.circ = 0; r.l_cont = I.l_cont + 1; r.l_cont_bp = I.l_cont_bp + P.len_bp;
        }
        return r;
    case 3: // plain insertion right of f_ins
        if (is_pop) {
            r.pos = tailpos; r.start_bp = tailbp; r.ori = ori_f_pop; r.prev = id_f_ins; r.next = I.next;
        } else if (pos < I.pos) {
            if (f == I.next && I.circ == 1) r.prev = id_f_pop;
        } else if (pos == I.pos) {
            r.ori = I.ori; r.next = id_f_pop;
        } else {
            r.pos = pos + 1; r.start_bp = start + P.len_bp;
            if (pos == tailpos) r.prev = id_f_pop;
        }
        r.id_c = I.id_c; r.circ = I.circ; r.l_cont = I.l_cont + 1; r.l_cont_bp = I.l_cont_bp + P.len_bp;
        return r;
    default: // 4: plain insertion left of f_ins
        if (is_pop) {
            r.pos = I.pos; r.start_bp = I.start_bp; r.ori = ori_f_pop; r.prev = I.prev; r.next = id_f_ins;
        } else if (pos < I.pos) {
            if (pos == I.pos - 1) r.next = id_f_pop;
        } else if (pos == I.pos) {
            r.pos = I.pos + 1; r.start_bp = I.start_bp + P.len_bp; r.ori = I.ori; r.prev = id_f_pop; r.next = I.next;
        } else {
            r.pos = pos + 1; r.start_bp = start + P.len_bp;
        }
        r.id_c = I.id_c; r.circ = I.circ; r.l_cont = I.l_cont + 1; r.l_cont_bp = I.l_cont_bp + P.len_bp;
        return r;
    }
    (void)prev; (void)next;
}

// does split_contig hand out the label max_id + 1 to at least one fragment?
GR_HD bool split_makes_label(const Rec& C, int upstream)
{
    if (!(C.activ == 1 && C.l_cont > 1) || C.circ != 0) return false;
    return upstream == 1 ? true : (C.pos < C.l_cont - 1);
}

// split_contig (kernels3.cu:1451).  C = record of the cut fragment in the input layout.
GR_HD Rec m_split(Rec r, int f, const Rec& C, int id_f_cut, int upstream, int max_id)
{
    (void)id_f_cut;
    if (!(C.activ == 1 && C.l_cont > 1) || r.id_c != C.id_c) return r;
    const int pos = r.pos, start = r.start_bp;
    // first position / bp of the downstream part
    const int cutpos = upstream == 1 ? C.pos : C.pos + 1;
    const int cutbp = upstream == 1 ? C.start_bp : C.start_bp + C.len_bp;
    if (C.circ == 0) {
        if (pos < cutpos) {
            r.circ = 0; r.l_cont = cutpos; r.l_cont_bp = cutbp;
            if (pos == cutpos - 1) r.next = -1; // (for upstream == 0 this is the cut fragment itself)
            if (upstream == 0 && pos == C.pos) { r.prev = C.prev; r.len_bp = C.len_bp; }
        } else {
            r.pos = pos - cutpos; r.id_c = max_id + 1; r.start_bp = start - cutbp; r.circ = 0;
            if (pos == cutpos) r.prev = -1;
            if (upstream == 1 && pos == C.pos) { r.next = C.next; r.len_bp = C.len_bp; }
            r.l_cont = C.l_cont - cutpos; r.l_cont_bp = C.l_cont_bp - cutbp;
        }
    } else { // circular: rotated so that the downstream part comes first; same label and length
        const int wrap = C.l_cont - cutpos, wrapbp = C.l_cont_bp - cutbp;
        if (pos < cutpos) {
            r.pos = wrap + pos; r.start_bp = wrapbp + start;
            if (upstream == 1) { if (pos == C.pos - 1) r.next = -1; }
            else {
                if (pos == C.pos) { r.start_bp = wrapbp + C.start_bp; r.len_bp = C.len_bp; r.prev = C.prev; r.next = -1; }
                else if (f == C.next) r.prev = -1;
            }
        } else {
            r.pos = pos - cutpos; r.start_bp = start - cutbp;
            if (upstream == 1) {
                if (pos == C.pos) { r.start_bp = 0; r.len_bp = C.len_bp; r.prev = -1; r.next = C.next; }
                else if (f == C.prev) r.next = -1;
            } else { if (pos == cutpos) r.prev = -1; }
        }
        r.id_c = C.id_c; r.circ = 0; r.l_cont = C.l_cont; r.l_cont_bp = C.l_cont_bp;
    }
    return r;
}

// paste_contigs (kernels3.cu:1786).  A, B = records of fA / fB in the input layout.
// *written = false reproduces the reference's unwritten "stale slot" case (same contig, fA/fB not its
// two ends, kernels3.cu:1977-2033); callers treat it as "layout unchanged" and count it.
GR_HD Rec m_paste(Rec r, int f, const Rec& A, const Rec& B, int id_fA, int id_fB, bool* written)
{
    (void)f;
    *written = true;
    if (!(A.activ == 1 && B.activ == 1)) return r;
    const int pos = r.pos, start = r.start_bp, prev = r.prev, next = r.next;
    if (A.id_c != B.id_c) {
        const int n = A.l_cont + B.l_cont, bp = A.l_cont_bp + B.l_cont_bp;
        if (r.id_c == A.id_c) {
            if (A.pos == 0) { // reverse contig A so that fA is its tail
                r.pos = A.l_cont - (pos + 1); r.start_bp = A.l_cont_bp - (start + r.len_bp); r.ori = -r.ori;
                r.prev = (pos == A.l_cont - 1) ? -1 : next;
                r.next = (pos == A.pos) ? id_fB : prev;
            } else if (pos == A.pos) r.next = id_fB;
            r.circ = 0; r.l_cont = n; r.l_cont_bp = bp;
        } else if (r.id_c == B.id_c) {
            if (B.pos == 0) {
                r.pos = A.l_cont + pos; r.start_bp = A.l_cont_bp + start;
                if (pos == B.pos) r.prev = id_fA;
            } else { // reverse contig B so that fB is its head
                r.pos = A.l_cont + (B.l_cont - (pos + 1));
                r.start_bp = A.l_cont_bp + (B.l_cont_bp - (start + r.len_bp)); r.ori = -r.ori;
                r.prev = (pos == B.pos) ? id_fA : next;
                r.next = (pos == 0) ? -1 : prev;
            }
            r.id_c = A.id_c; r.circ = 0; r.l_cont = n; r.l_cont_bp = bp;
        }
        return r;
    }
    if (r.id_c != A.id_c) return r;
    const int last = A.l_cont - 1;
    if (A.pos == 0 && B.pos == last) {
        r.circ = 1;
        if (pos == A.pos) r.prev = id_fB;
        if (pos == last) r.next = id_fA;
    } else if (A.pos == last && B.pos == 0) {
        r.circ = 1;
        if (pos == B.pos) r.prev = id_fA;
        if (pos == last) r.next = id_fB;
    } else { *written = false; return r; }
    r.l_cont = A.l_cont; r.l_cont_bp = A.l_cont_bp;
    return r;
}

// ---------------------------------------------------------------- a whole candidate move
// The 13 candidates of cuda_lib_gl.py:864-909 (ops 0-8) and :926-954 (ops 9-12) as one composed map.
struct Move {
    int op, fA, fB, max_id;
    Rec A0, B0;      // fA / fB in the current layout
    Rec A1, B1;      // after pop_out (ops 0, 2-8) or after the first split (ops 9-12)
    Rec A2, B2;      // after the second split (ops 9-12)
    int max_id1, max_id2;
    bool identity;   // fA == fB and an op that involves fB (2-7, 9-12): the reference's behaviour is undefined; we score and
                     // apply a no-op.  Ops 0, 1 and 8 (eject, flip, swap activity) never look at fB and are applied as they
                     // are -- explode_genome commits (i, 0, op 0) for every fragment, fragment 0 included (cuda_lib_gl.py:1539-1544)
};

GR_HD Move make_move(int op, int fA, int fB, int max_id, const Rec& A0, const Rec& B0)
{
    Move m;
    m.op = op; m.fA = fA; m.fB = fB; m.max_id = max_id; m.A0 = A0; m.B0 = B0;
    m.A1 = A0; m.B1 = B0; m.A2 = A0; m.B2 = B0; m.max_id1 = max_id; m.max_id2 = max_id;
    m.identity = (fA == fB) && !(op == 0 || op == 1 || op == 8);
    if (m.identity) return m;
    if (op <= 8) {
        m.A1 = m_pop_out(A0, fA, A0, fA, max_id);
        m.B1 = m_pop_out(B0, fB, A0, fA, max_id);
        m.max_id1 = max_id + (A0.l_cont >= 2 ? 1 : 0); // max(pop_id_contigs), cuda_lib_gl.py:857
    } else {
        const int upA = (op - 9) >> 1, upB = (op - 9) & 1;
        m.A1 = m_split(A0, fA, A0, fA, upA, max_id);
        m.B1 = m_split(B0, fB, A0, fA, upA, max_id);
        m.max_id1 = max_id + (split_makes_label(A0, upA) ? 1 : 0);
        m.A2 = m_split(m.A1, fA, m.B1, fB, upB, m.max_id1);
        m.B2 = m_split(m.B1, fB, m.B1, fB, upB, m.max_id1);
        m.max_id2 = m.max_id1 + (split_makes_label(m.B1, upB) ? 1 : 0);
    }
    return m;
}

// new record of fragment f under the move; *stale counts the paste stale-slot case
GR_HD Rec apply_move(const Move& m, int f, Rec r, bool* stale)
{
    *stale = false;
    if (m.identity) return r;
    const int op = m.op;
    if (op == 1) return m_flip(r, f, m.fA);
    if (op <= 8) {
        r = m_pop_out(r, f, m.A0, m.fA, m.max_id);
        if (op == 0) return r;
        if (op == 8) return m_swap_activity(r, f, m.fA, m.max_id1);
        const int which = (op - 2) / 2 + 1;          // 2,3 -> 1 ; 4,5 -> 2 ; 6,7 -> 3
        const int ori = (op & 1) ? -1 : 1;           // even op: +1, odd op: -1
        return m_pop_in(which, r, f, m.A1, m.B1, m.fA, m.fB, m.max_id1, ori);
    }
    const int upA = (op - 9) >> 1, upB = (op - 9) & 1;
    r = m_split(r, f, m.A0, m.fA, upA, m.max_id);
    r = m_split(r, f, m.B1, m.fB, upB, m.max_id1);
    bool written;
    r = m_paste(r, f, m.A2, m.B2, m.fA, m.fB, &written);
    *stale = !written;
    return r;
}

// ---------------------------------------------------------------- pieces
// For one neighbour fB, the fragments of contig(fA) u contig(fB) fall into <= 6 "pieces": maximal
// position ranges that every one of the 13 candidates maps with ONE affine coordinate transform.
// Break points are the positions of fA and fB (all mutation kernels branch only on pos </==/> those).
struct PieceKey { int cA, a, cB, b; };   // contig / position of fA and fB in the current layout

GR_HD int piece_of(const PieceKey& k, int id_c, int pos)
{
    if (k.cA != k.cB) {
        if (id_c == k.cA) return pos < k.a ? 1 : (pos == k.a ? 2 : 3);
        if (id_c == k.cB) return pos < k.b ? 4 : (pos == k.b ? 5 : 6);
        return 0;
    }
    if (id_c != k.cA) return 0;
    const int lo = k.a < k.b ? k.a : k.b, hi = k.a < k.b ? k.b : k.a;
    if (pos < lo) return 1;
    if (pos == lo) return 2;
    if (pos < hi) return 3;
    if (pos == hi) return 4;
    return 5;
}

// Representative fragment of each piece (a neighbour of fA / fB along the contig), -1 if the piece is empty.
GR_HD void piece_representatives(const PieceKey& k, int fA, int fB, const Rec& A0, const Rec& B0, int rep[MAX_PIECES + 1])
{
    for (int p = 0; p <= MAX_PIECES; p++) rep[p] = -1;
    if (fA == fB) { rep[2] = fA; return; }
    if (k.cA != k.cB) {
        rep[1] = (A0.pos > 0) ? A0.prev : -1;
        rep[2] = fA;
        rep[3] = (A0.pos < A0.l_cont - 1) ? A0.next : -1;
        rep[4] = (B0.pos > 0) ? B0.prev : -1;
        rep[5] = fB;
        rep[6] = (B0.pos < B0.l_cont - 1) ? B0.next : -1;
        return;
    }
    const bool a_first = A0.pos < B0.pos;
    const Rec& L = a_first ? A0 : B0;
    const Rec& H = a_first ? B0 : A0;
    rep[1] = (L.pos > 0) ? L.prev : -1;
    rep[2] = a_first ? fA : fB;
    rep[3] = (H.pos - L.pos > 1) ? L.next : -1;
    rep[4] = a_first ? fB : fA;
    rep[5] = (H.pos < H.l_cont - 1) ? H.next : -1;
}

// Affine map of one piece under one candidate: label', and for the bp coordinate x of any point of
// the piece  x' = sigma * x + off  (sigma = -1: the piece is mirrored, every ori flips).
struct Xf { int label, sigma, off, circ, lbp; };

GR_HD Xf xf_identity(const Rec& r)
{
    Xf x; x.label = r.id_c; x.sigma = 1; x.off = 0; x.circ = r.circ; x.lbp = r.l_cont_bp; return x;
}

GR_HD Xf xf_from(const Rec& r_old, const Rec& r_new)
{
    Xf x;
    x.label = r_new.id_c;
    x.sigma = (r_new.ori == r_old.ori) ? 1 : -1;
    x.off = x.sigma > 0 ? r_new.start_bp - r_old.start_bp : r_new.start_bp + r_old.start_bp + r_old.len_bp;
    x.circ = r_new.circ;
    x.lbp = r_new.l_cont_bp;
    return x;
}

// new start_bp / ori of a fragment of the piece
GR_HD int xf_start(const Xf& x, int start_bp, int len_bp) { return x.sigma > 0 ? start_bp + x.off : x.off - (start_bp + len_bp); }

// Pair relation: does the (contact-model relevant) geometry between piece p and piece q differ
// between two layouts?  Trans pairs have no geometry; cis pairs are compared through the relative
// affine map.  Circular contigs add the contig length to the geometry (kernels3.cu:135-166).
GR_HD bool rel_changed(const Xf& p_old, const Xf& q_old, const Xf& p_new, const Xf& q_new)
{
    const bool cis_old = p_old.label == q_old.label, cis_new = p_new.label == q_new.label;
    if (cis_old != cis_new) return true;
    if (!cis_new) return false;
    if (p_new.circ != p_old.circ) return true;
    if (p_new.circ == 1 && p_new.lbp != p_old.lbp) return true;
    // old layouts are identities: sigma = 1, off = 0 (callers pass xf_identity for *_old)
    const int rho_old = p_old.sigma * q_old.sigma, rho_new = p_new.sigma * q_new.sigma;
    const long long d_old = (long long)p_old.sigma * ((long long)q_old.off - p_old.off);
    const long long d_new = (long long)p_new.sigma * ((long long)q_new.off - p_new.off);
    return rho_old != rho_new || d_old != d_new;
}

// Reference arithmetic (GRAAL_MODE_STRICT): do two layouts hand the contact model the SAME INPUTS for every fragment pair of
// pieces (p, q)?  The dense reference prices a pixel from the float32 kb coordinates of its two fragments, so -- unlike
// rel_changed -- the absolute offsets matter, not only the relative map (kernels3.cu:2997-3078).  Equal inputs = bit-equal
// values; anything else is priced again.  A trans pair has no geometry; with the reference's RF-count indexing in the trans
// branch (`quirk`, kernels3.cu:3155) its value depends on the orientation of the lower-id bin, i.e. on the two mirrors.
GR_HD bool same_inputs(const Xf& p1, const Xf& q1, const Xf& p2, const Xf& q2, bool quirk)
{
    const bool cis1 = p1.label == q1.label, cis2 = p2.label == q2.label;
    if (cis1 != cis2) return false;
    if (!cis1) return !quirk || (p1.sigma == p2.sigma && q1.sigma == q2.sigma);
    if (p1.sigma != p2.sigma || q1.sigma != q2.sigma || p1.off != p2.off || q1.off != q2.off) return false;
    if (p1.circ != p2.circ) return false;
    return p1.circ != 1 || p1.lbp == p2.lbp;
}

// what same_inputs compares, packed into four words: inputs_key(p1, q1) == inputs_key(p2, q2) (all four words) <=> same_inputs(p1, q1,
// p2, q2).  The table kernel compares these keys -- one 16-byte LDS read per comparison instead of four 5-word transforms
// (tests/test_host_logic.py checks the equivalence on the host).
struct InputsKey { int x, y, z, w; };
GR_HD InputsKey inputs_key(const Xf& a, const Xf& b, bool quirk)
{
    const bool cis = a.label == b.label;
    InputsKey k;
    k.x = cis ? a.off : 0; k.y = cis ? b.off : 0; k.z = (cis && a.circ == 1) ? a.lbp : 0;
    const bool sig = cis || quirk;
    k.w = (cis ? 1 : 0) | ((sig && a.sigma == 1) ? 2 : 0) | ((sig && b.sigma == 1) ? 4 : 0) | (cis ? (int)((unsigned)a.circ << 3) : 0);
    return k;
}

// geometry inside ONE piece changes only through the circular model (or if the piece is torn, which
// cannot happen: pieces are the tear units)
GR_HD bool intra_changed(const Xf& p_old, const Xf& p_new)
{
    if (p_new.circ != p_old.circ) return true;
    return p_new.circ == 1 && p_new.lbp != p_old.lbp;
}

} // namespace graal
