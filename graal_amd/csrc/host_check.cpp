// host_check.cpp -- TEST ONLY.  Host build of frag_ops.h so that the CPU test-suite can compare the engine's
// layout algebra (mutations, pieces, per-piece transforms, relation flags) with the oracle without a GPU.
// No likelihood code here and nothing in graal_amd/ loads this library.
#include <stdint.h>

#include "frag_ops.h"

using namespace graal;

namespace {
enum { F_POS, F_IDC, F_START, F_LEN, F_CIRC, F_ID, F_PREV, F_NEXT, F_LCONT, F_LCONTBP, F_ORI, F_REP, F_ACTIV, F_IDD };

Rec ld(int32_t* const* s, int f)
{
    Rec r;
    r.pos = s[F_POS][f]; r.id_c = s[F_IDC][f]; r.start_bp = s[F_START][f]; r.len_bp = s[F_LEN][f]; r.circ = s[F_CIRC][f];
    r.prev = s[F_PREV][f]; r.next = s[F_NEXT][f]; r.l_cont = s[F_LCONT][f]; r.l_cont_bp = s[F_LCONTBP][f];
    r.ori = s[F_ORI][f]; r.rep = s[F_REP][f]; r.activ = s[F_ACTIV][f]; r.id_d = s[F_IDD][f];
    return r;
}

void st(int32_t* const* s, int f, const Rec& r)
{
    s[F_POS][f] = r.pos; s[F_IDC][f] = r.id_c; s[F_START][f] = r.start_bp; s[F_LEN][f] = r.len_bp; s[F_CIRC][f] = r.circ;
    s[F_ID][f] = f; s[F_PREV][f] = r.prev; s[F_NEXT][f] = r.next; s[F_LCONT][f] = r.l_cont; s[F_LCONTBP][f] = r.l_cont_bp;
    s[F_ORI][f] = r.ori; s[F_REP][f] = r.rep; s[F_ACTIV][f] = r.activ; s[F_IDD][f] = r.id_d;
}
} // namespace

extern "C" {

// out = apply_move(in) for every fragment; returns the number of stale (unwritten paste) fragments
int hc_apply_move(int op, int fA, int fB, int max_id, int32_t* const* in, int32_t* const* out, int n)
{
    const Move m = make_move(op, fA, fB, max_id, ld(in, fA), ld(in, fB));
    int n_stale = 0;
    for (int f = 0; f < n; f++) {
        bool stale;
        const Rec r = apply_move(m, f, ld(in, f), &stale);
        st(out, f, r);
        n_stale += stale;
    }
    return n_stale;
}

// piece id of every fragment, the 13 x 7 transforms (label, sigma, off, circ, lbp) and the relation flags
// changed[op] (bit p*8+q) for one neighbour -- the same glue k_tables runs on the device.
void hc_piece_tables(int fA, int fB, int max_id, int32_t* const* s, int n, int32_t* piece, int32_t* xf_out /*[13][7][5]*/,
                     uint64_t* changed /*[13]*/, int32_t* rep_out /*[7]*/)
{
    const Rec A0 = ld(s, fA), B0 = ld(s, fB);
    PieceKey key; key.cA = A0.id_c; key.a = A0.pos; key.cB = B0.id_c; key.b = B0.pos;
    for (int f = 0; f < n; f++) piece[f] = (fA == fB) ? 0 : piece_of(key, s[F_IDC][f], s[F_POS][f]);
    int rep[MAX_PIECES + 1];
    piece_representatives(key, fA, fB, A0, B0, rep);
    Xf old[MAX_PIECES + 1], cur[MAX_PIECES + 1];
    Rec rold[MAX_PIECES + 1];
    for (int p = 0; p <= MAX_PIECES; p++) {
        rep_out[p] = rep[p];
        if (rep[p] >= 0) { rold[p] = ld(s, rep[p]); old[p] = xf_identity(rold[p]); }
        else { old[p].label = -1 - p; old[p].sigma = 1; old[p].off = 0; old[p].circ = 0; old[p].lbp = 0; }
    }
    for (int op = 0; op < N_OPS; op++) {
        const Move m = make_move(op, fA, fB, max_id, A0, B0);
        for (int p = 0; p <= MAX_PIECES; p++) {
            cur[p] = old[p];
            if (p >= 1 && rep[p] >= 0) {
                bool stale;
                cur[p] = xf_from(rold[p], apply_move(m, rep[p], rold[p], &stale));
            }
            int32_t* o = xf_out + (op * (MAX_PIECES + 1) + p) * 5;
            o[0] = cur[p].label; o[1] = cur[p].sigma; o[2] = cur[p].off; o[3] = cur[p].circ; o[4] = cur[p].lbp;
        }
        uint64_t c = 0;
        for (int p = 1; p <= MAX_PIECES; p++)
            for (int q = p; q <= MAX_PIECES; q++) {
                if (rep[p] < 0 || rep[q] < 0) continue;
                const bool chg = (p == q) ? intra_changed(old[p], cur[p]) : rel_changed(old[p], old[q], cur[p], cur[q]);
                if (chg) c |= (1ull << (p * 8 + q)) | (1ull << (q * 8 + p));
            }
        changed[op] = c;
    }
}

// inputs_key against same_inputs on n random quadruples of transforms drawn from small ranges (so that equal inputs occur):
// returns the number of disagreements
int hc_inputs_key_disagreements(int n, unsigned seed, int quirk)
{
    unsigned st_ = seed * 2654435761u + 12345u;
    auto rnd = [&](int m) { st_ = st_ * 1664525u + 1013904223u; return (int)((st_ >> 16) % (unsigned)m); };
    auto rxf = [&]() { Xf x; x.label = rnd(3); x.sigma = rnd(2) ? 1 : -1; x.off = rnd(4) * 1000 - 1000; x.circ = rnd(2); x.lbp = 5000 + 1000 * rnd(3); return x; };
    int bad = 0;
    for (int i = 0; i < n; i++) {
        const Xf p1 = rxf(), q1 = rxf(), p2 = rxf(), q2 = rxf();
        const InputsKey a = inputs_key(p1, q1, quirk != 0), b = inputs_key(p2, q2, quirk != 0);
        const bool eq = a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w;
        bad += eq != same_inputs(p1, q1, p2, q2, quirk != 0);
    }
    return bad;
}

} // extern "C"
