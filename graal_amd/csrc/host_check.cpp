// host_check.cpp -- TEST ONLY.  Host build of frag_ops.h / strict_sets.h so that the CPU test-suite can compare the engine's
// layout algebra (mutations, pieces, per-piece transforms, relation flags; the union set of a step, its classes of equal inputs
// and its work units) with the oracle and with brute force without a GPU.
// No likelihood code here and nothing in graal_amd/ loads this library.
#include <stddef.h>
#include <stdint.h>

#include <vector>
#include <map>

#include "frag_ops.h"
#include "strict_sets.h"

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

// ---- strict_sets.h: the union set of a step, its global pieces, the classes of equal inputs per piece pair and the unit list,
// checked against brute force on one proposal (fA, fB[0..K)) of a layout:
//   (1) every fragment of the K sets lies in exactly one global piece, whose per-neighbour piece ids are piece_of's;
//   (2) for every unordered pair of union fragments and every candidate (k, op) whose set holds both: the model's inputs obtained by
//       applying the candidate move to the two fragments THEMSELVES (apply_move) equal those of the pair's class representative
//       (or the current layout's, when the candidate is in no class);
//   (3) two candidates of one class have equal inputs_key, two classes different ones, no class has the current layout's;
//   (4) the unit list with every tile pair alive covers every unordered pair of union fragments exactly once.
// Returns the number of violations (0 = fine); info[0..9]: pieces, tiles, contigs, piece pairs with classes, classes, units, pairs, candidates checked, model evaluations
// (no window) over the union's classes, the same one neighbour at a time.
int hc_union_check(int fA, const int32_t* fB, int K, int max_id, int32_t* const* s, int n, int quirk, int seg, int64_t* info, int tile_frags)
{
    int bad = 0;
    // position index
    std::map<int, int> base_of;
    std::vector<int> perm(n), cbase(n);
    {
        std::map<int, int> len;
        for (int f = 0; f < n; f++) len[s[F_IDC][f]] = s[F_LCONT][f];
        int b = 0;
        for (auto& kv : len) { base_of[kv.first] = b; b += kv.second; }
        for (int f = 0; f < n; f++) { cbase[f] = base_of[s[F_IDC][f]]; perm[cbase[f] + s[F_POS][f]] = f; }
    }
    auto end_of = [&](int f) { UEnd e; e.label = s[F_IDC][f]; e.pos = s[F_POS][f]; e.base = cbase[f]; e.len = s[F_LCONT][f]; e.lbp = s[F_LCONTBP][f]; e.circ = s[F_CIRC][f]; return e; };
    // per-neighbour tables (as k_tm builds them)
    std::vector<Xf> xf((size_t)K * N_OPS * (MAX_PIECES + 1));
    std::vector<PieceKey> keys(K);
    std::vector<UEnd> B(K);
    unsigned live = 0;
    const Rec A0 = ld(s, fA);
    for (int k = 0; k < K; k++) {
        const Rec B0 = ld(s, fB[k]);
        PieceKey key; key.cA = A0.id_c; key.a = A0.pos; key.cB = B0.id_c; key.b = B0.pos;
        keys[k] = key; B[k] = end_of(fB[k]);
        if (fB[k] != fA) live |= 1u << k;
        int rep[MAX_PIECES + 1];
        piece_representatives(key, fA, fB[k], A0, B0, rep);
        for (int op = 0; op < N_OPS; op++) {
            const Move m = make_move(op, fA, fB[k], max_id, A0, B0);
            for (int p = 0; p <= MAX_PIECES; p++) {
                Xf x; x.label = -1 - p; x.sigma = 1; x.off = 0; x.circ = 0; x.lbp = 0;
                if (rep[p] >= 0) {
                    const Rec ro = ld(s, rep[p]);
                    x = xf_identity(ro);
                    if (p >= 1) { bool stale; x = xf_from(ro, apply_move(m, rep[p], ro, &stale)); }
                }
                xf[((size_t)k * N_OPS + op) * (MAX_PIECES + 1) + p] = x;
            }
        }
    }
    auto XF = [&](int k, int op, int p) -> const Xf& { return xf[((size_t)k * N_OPS + op) * (MAX_PIECES + 1) + p]; };
    USet U;
    int cuts[US_MAXC * (US_MAXK + 1)], ncut[US_MAXC];
    uset_build(U, end_of(fA), B.data(), keys.data(), K, live, live, cuts, ncut, tile_frags == 32 ? 32 : US_TILE);
    info[0] = U.n_pieces; info[1] = U.n_tiles; info[2] = U.n_contigs;
    if (U.n_pieces > US_MAXP || U.n_contigs > US_MAXC) return 1000000;
    // (1) membership
    std::vector<int> piece(n, -1);
    for (int f = 0; f < n; f++) {
        bool in_union = s[F_IDC][f] == A0.id_c;
        for (int k = 0; k < K; k++) if (((live >> k) & 1u) && s[F_IDC][f] == s[F_IDC][fB[k]]) in_union = true;
        const int g = ufrag_piece(U, s[F_IDC][f], s[F_POS][f]);
        if (in_union != (g >= 0)) { bad++; continue; }
        piece[f] = g;
        if (g < 0) continue;
        int hits = 0;
        for (int i = 0; i < U.n_pieces; i++) hits += (U.c[U.p[i].contig].label == s[F_IDC][f] && s[F_POS][f] >= U.p[i].lo && s[F_POS][f] < U.p[i].lo + U.p[i].n);
        if (hits != 1) bad++;
        for (int k = 0; k < K; k++) {
            const int want = ((live >> k) & 1u) ? piece_of(keys[k], s[F_IDC][f], s[F_POS][f]) : 0;
            if (want != U.p[g].pk[k]) bad++;
        }
        if (perm[U.c[U.p[g].contig].base + s[F_POS][f]] != f) bad++;
    }
    // classes (the algorithm of k_gprep: candidates in index order, a class is represented by its first candidate)
    struct Cls { GClass c; InputsKey key; };
    std::vector<std::vector<Cls>> classes(US_MAXPAIRS);
    std::vector<InputsKey> old_key(US_MAXPAIRS);
    long long n_cls = 0, n_pairs_cls = 0;
    double work_union = 0.0, work_single = 0.0;
    for (int g = 0; g < U.n_pieces; g++)
        for (int h = g; h < U.n_pieces; h++) {
            const int pr = upair_index(g, h);
            old_key[pr] = inputs_key(uxf_old(U, g), uxf_old(U, h), quirk != 0);
            for (int k = 0; k < K; k++) {
                if (!U.p[g].pk[k] || !U.p[h].pk[k]) continue;
                for (int op = 0; op < N_OPS; op++) {
                    const Xf &a = XF(k, op, U.p[g].pk[k]), &b = XF(k, op, U.p[h].pk[k]);
                    const InputsKey key = inputs_key(a, b, quirk != 0);
                    if (ikey_eq(key, old_key[pr])) continue;
                    const int cand = k * N_OPS + op;
                    Cls* hit = nullptr;
                    for (auto& c : classes[pr]) if (ikey_eq(c.key, key)) { hit = &c; break; }
                    if (!hit) { Cls c; c.c = gclass_make(a, b, cand); c.key = key; classes[pr].push_back(c); hit = &classes[pr].back(); }
                    if (cand < 64) hit->c.m0 |= 1ull << cand; else if (cand < 128) hit->c.m1 |= 1ull << (cand - 64); else hit->c.m2 |= 1u << (cand - 128);
                }
            }
            n_cls += (long long)classes[pr].size();
            n_pairs_cls += classes[pr].empty() ? 0 : 1;
            {   // evaluations without a window: the union's classes (+ the current layout once) against one neighbour at a time
                const double np_ = g == h ? 0.5 * U.p[g].n * (U.p[g].n - 1.0) : (double)U.p[g].n * U.p[h].n;
                if (!classes[pr].empty()) work_union += np_ * (1.0 + (double)classes[pr].size());
                for (int k = 0; k < K; k++) {
                    if (!U.p[g].pk[k] || !U.p[h].pk[k]) continue;
                    std::vector<InputsKey> seen_k;
                    for (int op = 0; op < N_OPS; op++) {
                        const InputsKey key = inputs_key(XF(k, op, U.p[g].pk[k]), XF(k, op, U.p[h].pk[k]), quirk != 0);
                        if (ikey_eq(key, old_key[pr])) continue;
                        bool dup = false;
                        for (auto& o : seen_k) dup = dup || ikey_eq(o, key);
                        if (!dup) seen_k.push_back(key);
                    }
                    if (!seen_k.empty()) work_single += np_ * (1.0 + (double)seen_k.size());
                }
            }
            // (3)
            for (size_t i = 0; i < classes[pr].size(); i++) {
                if (ikey_eq(classes[pr][i].key, old_key[pr])) bad++;
                for (size_t j = i + 1; j < classes[pr].size(); j++) if (ikey_eq(classes[pr][i].key, classes[pr][j].key)) bad++;
            }
        }
    info[3] = n_pairs_cls; info[4] = n_cls;
    info[8] = (int64_t)work_union; info[9] = (int64_t)work_single;
    // (2) brute force
    long long n_checked = 0, n_pairs = 0;
    std::vector<int> uf;
    for (int f = 0; f < n; f++) if (piece[f] >= 0) uf.push_back(f);
    // union order: contig index, then position
    std::vector<long long> ukey(n, 0);
    for (int f : uf) ukey[f] = (long long)U.p[piece[f]].contig * (1ll << 32) + s[F_POS][f];
    for (size_t i = 0; i < uf.size(); i++)
        for (size_t j = 0; j < uf.size(); j++) {
            const int fx = uf[i], fy = uf[j];
            if (!(ukey[fx] < ukey[fy])) continue;
            n_pairs++;
            const int g = piece[fx], h = piece[fy];
            if (g > h) { bad++; continue; }
            const int pr = upair_index(g, h);
            const Rec rx = ld(s, fx), ry = ld(s, fy);
            for (int k = 0; k < K; k++) {
                if (!U.p[g].pk[k] || !U.p[h].pk[k]) continue;
                const Rec B0 = ld(s, fB[k]);
                for (int op = 0; op < N_OPS; op++) {
                    const Move m = make_move(op, fA, fB[k], max_id, A0, B0);
                    bool st1, st2;
                    const Rec nx = apply_move(m, fx, rx, &st1), ny = apply_move(m, fy, ry, &st2);
                    const int cand = k * N_OPS + op;
                    const GClass* c = nullptr;
                    for (auto& cc : classes[pr]) {
                        const bool in = cand < 64 ? ((cc.c.m0 >> cand) & 1ull) : (cand < 128 ? ((cc.c.m1 >> (cand - 64)) & 1ull) : ((cc.c.m2 >> (cand - 128)) & 1u));
                        if (in) { if (c) bad++; c = &cc.c; }
                    }
                    // predicted inputs
                    bool cis; int sx, sy, ox, oy, circ, lbp;
                    if (c) {
                        cis = (c->flags & 4u) != 0;
                        sx = gclass_start(c->flags & 1u, c->offx, rx.start_bp, rx.len_bp); sy = gclass_start(c->flags & 2u, c->offy, ry.start_bp, ry.len_bp);
                        ox = (c->flags & 1u) ? rx.ori : -rx.ori; oy = (c->flags & 2u) ? ry.ori : -ry.ori;
                        circ = (c->flags & 8u) ? 1 : 0; lbp = c->lbp;
                    } else {
                        cis = rx.id_c == ry.id_c; sx = rx.start_bp; sy = ry.start_bp; ox = rx.ori; oy = ry.ori; circ = rx.circ; lbp = rx.l_cont_bp;
                    }
                    const bool cis_t = nx.id_c == ny.id_c;
                    n_checked++;
                    if (cis != cis_t) { bad++; continue; }
                    if (cis_t) {
                        if (sx != nx.start_bp || sy != ny.start_bp || ox != nx.ori || oy != ny.ori) bad++;
                        if ((circ == 1) != (nx.circ == 1) || nx.circ != ny.circ) bad++;
                        else if (nx.circ == 1 && (lbp != nx.l_cont_bp || nx.l_cont_bp != ny.l_cont_bp)) bad++;
                    } else if (quirk) { if (ox != nx.ori || oy != ny.ori) bad++; }
                }
            }
        }
    info[6] = n_pairs; info[7] = n_checked;
    // (4) the unit list, every tile pair alive
    std::map<std::pair<int, int>, int> seen;
    long long n_units = 0;
    auto frag_at = [&](int t, int l) { int off; const int g = utile_piece(U, t, off); return perm[U.c[U.p[g].contig].base + U.p[g].lo + off * U.tile + l]; };
    for (int ti = 0; ti < U.n_tiles; ti++)
        for (int tj = ti; tj < U.n_tiles; tj++) {
            const int ci = utile_count(U, ti), cj = utile_count(U, tj);
            const int lanes_first = ci >= cj ? 1 : 0;
            const int cs = lanes_first ? cj : ci;            // fragments of the segment side
            for (int j0 = 0; j0 < cs; j0 += seg) {
                const int cnt = cs - j0 < seg ? cs - j0 : seg;
                int offi_, offj_;
                const int gi = utile_piece(U, ti, offi_), gj = utile_piece(U, tj, offj_);
                const int pair = upair_index(gi, gj);                          // (tiles are numbered piece by piece: gi <= gj)
                const unsigned long long u = uunit_pack(ti, tj, j0, cnt, lanes_first, 0, 1, pair);
                if (gi > gj || pair < 0 || pair >= US_MAXPAIRS || uunit_pair(u) != pair) bad++;   // the entry carries its piece pair (k_strict2's plan reads it)
                n_units++;
                // decode as k_strict2 does
                const int dti = (int)(u & 0xffffull), dtj = (int)((u >> 16) & 0xffffull), dj0 = (int)((u >> 32) & 63ull), dcnt = (int)((u >> 38) & 63ull);
                const int lf = (int)((u >> 44) & 1ull);
                const int tl = lf ? dti : dtj, ts = lf ? dtj : dti, cl = utile_count(U, tl);
                for (int lane = 0; lane < U.tile; lane++) {
                    if (lane >= cl) continue;
                    for (int j = 0; j < dcnt; j++) {
                        if (dti == dtj && !(lane < dj0 + j)) continue;     // the diagonal tile pair: every unordered pair once
                        const int fl = frag_at(tl, lane), fs = frag_at(ts, dj0 + j);
                        const int first = lf ? fl : fs, second = lf ? fs : fl;
                        if (!(ukey[first] < ukey[second])) bad++;
                        seen[{first, second}]++;
                    }
                }
            }
        }
    info[5] = n_units;
    if ((long long)seen.size() != n_pairs) bad++;
    for (auto& kv : seen) if (kv.second != 1) bad++;
    return bad;
}

} // extern "C"
