// graal_hip.hip -- MI355X (gfx950) engine behind include/graal_hip.h.
//
// Reformulation of the reference's dense pixel loops (kernels3.cu:2802-3718) for a sparse contact list:
//
//   logL = sum_{contacts} [ ob * log(ex) - lf(ob) ]  -  sum_{all pixels} ex
//   sum_{all pixels} ex = T_all + sum_{cis pixels with s < d_max} (ex - ex_trans)
//
// T_all (every pixel priced as trans) does not depend on the layout; the cis correction is a windowed sum.
// A candidate move changes the geometry only BETWEEN the <= 6 "pieces" (frag_ops.h) of the two contigs it
// touches, so its delta needs (a) the contacts whose two ends lie in different pieces -- found by ONE
// streaming pass over the COO triples that serves all 13*K candidates of the step -- and (b) the windowed
// cis correction between pieces ("mass tasks").  All sums are int64 fixed point (2^-30), added with
// integer atomics: independent of summation order, launch geometry and number of GPUs.
//
// Wave64 code for gfx950 only.  No CPU fallback: without a device graal_create fails.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/graal_hip.h"
#include "frag_ops.h"

using namespace graal;

namespace {

constexpr double Q_SCALE = 1073741824.0; // 2^GRAAL_Q_BITS
constexpr int MAXK = GRAAL_MAX_NEIGHBOURS;
constexpr int NP = MAX_PIECES + 1;       // piece ids 0..6
constexpr int MAX_TASKS = 21 * (N_OPS + 1); // per neighbour: 21 piece pairs x (old + 13 candidate layouts), before dedupe
constexpr int LABEL_BITS = 20;           // relabel sort key = l_cont << 20 | label

struct Par { float kuhn, lm, c1, slope, d, d_max, fact, v_inter; };

struct SoaPtr { int* p[GRAAL_N_FIELDS]; }; // order of struct frag
enum { F_POS, F_IDC, F_START, F_LEN, F_CIRC, F_ID, F_PREV, F_NEXT, F_LCONT, F_LCONTBP, F_ORI, F_REP, F_ACTIV, F_IDD };

__device__ __forceinline__ Rec ld_rec(const SoaPtr& s, int f)
{
    Rec r;
    r.pos = s.p[F_POS][f]; r.id_c = s.p[F_IDC][f]; r.start_bp = s.p[F_START][f]; r.len_bp = s.p[F_LEN][f];
    r.circ = s.p[F_CIRC][f]; r.prev = s.p[F_PREV][f]; r.next = s.p[F_NEXT][f]; r.l_cont = s.p[F_LCONT][f];
    r.l_cont_bp = s.p[F_LCONTBP][f]; r.ori = s.p[F_ORI][f]; r.rep = s.p[F_REP][f]; r.activ = s.p[F_ACTIV][f];
    r.id_d = s.p[F_IDD][f];
    return r;
}

__device__ __forceinline__ void st_rec(const SoaPtr& s, int f, const Rec& r)
{
    s.p[F_POS][f] = r.pos; s.p[F_IDC][f] = r.id_c; s.p[F_START][f] = r.start_bp; s.p[F_LEN][f] = r.len_bp;
    s.p[F_CIRC][f] = r.circ; s.p[F_ID][f] = f; s.p[F_PREV][f] = r.prev; s.p[F_NEXT][f] = r.next;
    s.p[F_LCONT][f] = r.l_cont; s.p[F_LCONTBP][f] = r.l_cont_bp; s.p[F_ORI][f] = r.ori; s.p[F_REP][f] = r.rep;
    s.p[F_ACTIV][f] = r.activ; s.p[F_IDD][f] = r.id_d;
}

// ------------------------------------------------------------------ contact model (float32, as the reference)
// rippe_contacts kernels3.cu:120
__device__ __forceinline__ float rippe(float s, const Par& p)
{
    float result = 0.0f;
    if ((s > 0.0f) && (s < p.d_max))
        result = (p.c1 * powf(s, p.slope) * expf((p.d - 2) / (powf(s * p.lm / p.kuhn, 2.0f) + p.d))) * p.fact;
    return fmaxf(result, p.v_inter);
}

// rippe_contacts_circ kernels3.cu:135
__device__ __forceinline__ float rippe_circ(float s, float s_tot, const Par& p)
{
    float result = 0.0f;
    if ((s > 0.0f) && (s < p.d_max)) {
        const float K = p.lm / p.kuhn;
        const float nmax = K * 1;
        const float n = K * s * (s_tot - s) / s_tot;
        const float norm_lin = rippe(s, p);
        const float norm_circ =
            (powf(p.kuhn, -3.0f) * powf(nmax, p.slope) * expf((p.d - 2.0f) / (powf(nmax, 2.0f) + p.d))) * p.fact;
        const float val = (powf(p.kuhn, -3.0f) * powf(n, p.slope) * expf((p.d - 2.0f) / (powf(n, 2.0f) + p.d))) * p.fact;
        result = val * norm_lin / norm_circ;
    }
    return fmaxf(result, p.v_inter);
}

// dynamic geometry of one fragment (16-byte record, rebuilt after every layout change)
struct Geo { int id_c, start_bp, len_bp, flags; }; // flags: bit0 ori==+1, bit1 circ
// static data of one bin: sub-fragment lengths (kb) and RF counts (simulation_loader.py:673-704)
struct Stat { float len[3]; int n; int accu[3]; int pad; };

// centre (kb) of the sub-fragment stored in data slot `slot`, walking the bin in its orientation with
// the reference's float32 operation order (kernels3.cu:2997-3060)
__device__ __forceinline__ float centre_kb(int start_bp, bool fwd, const Stat& st, int slot)
{
    const int limit = st.n - 1;
    const int w = fwd ? slot : limit - slot; // walk index of that slot
    const float s0 = (float)start_bp / 1000.0f;
    const float l0 = fwd ? st.len[0] : st.len[limit];
    if (w == 0) return s0 + l0 / 2.0f;
    float run = s0 + l0;
    const float l1 = fwd ? st.len[1] : st.len[limit - 1];
    if (w == 1) return run + l1 / 2.0f;
    run = run + l1;
    const float l2 = fwd ? st.len[2] : st.len[limit - 2];
    return run + l2 / 2.0f;
}

struct End { // one end of a fragment pair in some layout
    int label, start_bp; bool fwd; int circ, lbp;
};

__device__ __forceinline__ End end_cur(const Geo& g, const int* __restrict__ lcontbp, int f)
{
    End e; e.label = g.id_c; e.start_bp = g.start_bp; e.fwd = g.flags & 1; e.circ = (g.flags >> 1) & 1;
    e.lbp = e.circ ? lcontbp[f] : 0;
    return e;
}

__device__ __forceinline__ End end_xf(const Geo& g, const Xf& x)
{
    End e; e.label = x.label; e.start_bp = xf_start(x, g.start_bp, g.len_bp);
    e.fwd = ((g.flags & 1) != 0) == (x.sigma > 0); e.circ = x.circ; e.lbp = x.lbp;
    return e;
}

// trans value of a slot pair: p.v_inter * norm_accu (kernels3.cu:3187-3191), accu quirk NOT applied
__device__ __forceinline__ float ex_trans(int ax, int ay, float nfpb, const Par& p)
{
    return p.v_inter * ((float)(ax * ay) / nfpb);
}

// expected contacts between slot sx of fragment X and slot sy of fragment Y (kernels3.cu:3062-3078 / 3184-3195)
__device__ __forceinline__ float ex_pair(const End& X, const Stat& sx, int slx, const End& Y, const Stat& sy, int sly,
                                         float nfpb, const Par& p)
{
    const float norm = (float)(sx.accu[slx] * sy.accu[sly]) / nfpb;
    if (X.label != Y.label) return p.v_inter * norm;
    const float s = fabsf(centre_kb(Y.start_bp, Y.fwd, sy, sly) - centre_kb(X.start_bp, X.fwd, sx, slx));
    if (X.circ == 1) return rippe_circ(s, (float)X.lbp / 1000.0f, p) * norm;
    return rippe(s, p) * norm;
}

__device__ __forceinline__ long long to_q(double v) { return __double2ll_rn(v * Q_SCALE); }

__device__ __forceinline__ long long wave_sum_ll(long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// ------------------------------------------------------------------ small maintenance kernels
__global__ void k_refresh_geo(SoaPtr s, Geo* __restrict__ geo, int n)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    Geo g; g.id_c = s.p[F_IDC][f]; g.start_bp = s.p[F_START][f]; g.len_bp = s.p[F_LEN][f];
    g.flags = (s.p[F_ORI][f] == 1 ? 1 : 0) | (s.p[F_CIRC][f] == 1 ? 2 : 0);
    geo[f] = g;
}

// out: [0] #heads (pos==0) [1] sum l_cont [2] #(start_bp==0) [3] sum l_cont_bp over start_bp==0 [4] max l_cont
//      [5] min l_cont [6] #(rep != 0 or activ != 1 or id_d != f)  [7] max label  [14] #(circ == 1)
__global__ void k_stats(SoaPtr s, int n, long long* __restrict__ out)
{
    long long heads = 0, sl = 0, nst = 0, sbp = 0, mx = 0, mn = 0x7fffffff, bad = 0, mlab = -1, ncirc = 0;
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < n; f += gridDim.x * blockDim.x) {
        const int lc = s.p[F_LCONT][f];
        heads += s.p[F_POS][f] == 0;
        sl += lc;
        if (s.p[F_START][f] == 0) { nst += 1; sbp += s.p[F_LCONTBP][f]; }
        mx = lc > mx ? lc : mx;
        mn = lc < mn ? lc : mn;
        bad += (s.p[F_REP][f] != 0) || (s.p[F_ACTIV][f] != 1) || (s.p[F_IDD][f] != f);
        ncirc += s.p[F_CIRC][f] == 1;
        const int c = s.p[F_IDC][f];
        mlab = c > mlab ? c : mlab;
    }
    for (int o = 32; o > 0; o >>= 1) {
        heads += __shfl_down(heads, o, 64); sl += __shfl_down(sl, o, 64); nst += __shfl_down(nst, o, 64);
        sbp += __shfl_down(sbp, o, 64); bad += __shfl_down(bad, o, 64); ncirc += __shfl_down(ncirc, o, 64);
        const long long a = __shfl_down(mx, o, 64), b = __shfl_down(mn, o, 64), c = __shfl_down(mlab, o, 64);
        mx = a > mx ? a : mx; mn = b < mn ? b : mn; mlab = c > mlab ? c : mlab;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd((unsigned long long*)&out[0], (unsigned long long)heads);
        atomicAdd((unsigned long long*)&out[1], (unsigned long long)sl);
        atomicAdd((unsigned long long*)&out[2], (unsigned long long)nst);
        atomicAdd((unsigned long long*)&out[3], (unsigned long long)sbp);
        atomicMax((long long*)&out[4], mx);
        atomicMin((long long*)&out[5], mn);
        atomicAdd((unsigned long long*)&out[6], (unsigned long long)bad);
        atomicMax((long long*)&out[7], mlab);
        atomicAdd((unsigned long long*)&out[14], (unsigned long long)ncirc);
    }
}

// relabel: sort key of every contig head
__global__ void k_relabel_keys(SoaPtr s, int n, unsigned long long* __restrict__ keys)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    keys[f] = (s.p[F_POS][f] == 0)
                  ? (((unsigned long long)(unsigned)s.p[F_LCONT][f] << LABEL_BITS) | (unsigned)s.p[F_IDC][f])
                  : ((1ull << (2 * LABEL_BITS)) - 1ull);
}

__global__ void k_relabel_o2n(const unsigned long long* __restrict__ sorted, int n_contigs, int* __restrict__ o2n,
                              int* __restrict__ len_of)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_contigs) return;
    const unsigned long long k = sorted[i];
    o2n[(int)(k & ((1u << LABEL_BITS) - 1u))] = i;
    len_of[i] = (int)(k >> LABEL_BITS);
}

__global__ void k_relabel_apply(SoaPtr s, int n, const int* __restrict__ o2n)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    s.p[F_IDC][f] = o2n[s.p[F_IDC][f]];
}

__global__ void k_build_perm(SoaPtr s, int n, const int* __restrict__ contig_off, int* __restrict__ perm)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    perm[contig_off[s.p[F_IDC][f]] + s.p[F_POS][f]] = f;
}

// commit one candidate (test_copy_struct, cuda_lib_gl.py:1156): out = apply_move(in)
__global__ void k_apply(SoaPtr in, SoaPtr out, int n, int op, int fA, int fB, int max_id, int* __restrict__ n_stale)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    const Rec A0 = ld_rec(in, fA), B0 = ld_rec(in, fB);
    const Move m = make_move(op, fA, fB, max_id, A0, B0);
    bool stale;
    const Rec r = apply_move(m, f, ld_rec(in, f), &stale);
    st_rec(out, f, r);
    if (stale) atomicAdd(n_stale, 1);
}

// ------------------------------------------------------------------ full likelihood
// contacts part: sum ob * log(ex) in Q (the log-factorial constant is added on the host)
__global__ __launch_bounds__(256) void k_full_nnz(const int* __restrict__ row, const int* __restrict__ col,
                                                   const int* __restrict__ cnt, long long nnz,
                                                   const int* __restrict__ sub2bin, const Geo* __restrict__ geo,
                                                   const Stat* __restrict__ stat, const int* __restrict__ lcontbp,
                                                   float nfpb, Par par, long long* __restrict__ out)
{
    long long acc = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (long long)gridDim.x * blockDim.x) {
        const int a = sub2bin[row[i]], b = sub2bin[col[i]];
        const int fx = a >> 2, fy = b >> 2;
        const End X = end_cur(geo[fx], lcontbp, fx), Y = end_cur(geo[fy], lcontbp, fy);
        const float ex = ex_pair(X, stat[fx], a & 3, Y, stat[fy], b & 3, nfpb, par);
        acc += to_q((double)cnt[i] * log((double)ex));
    }
    acc = wave_sum_ll(acc);
    if ((threadIdx.x & 63) == 0 && acc != 0) atomicAdd((unsigned long long*)out, (unsigned long long)acc);
}

// cis correction of the expected mass for the current layout: one thread per fragment x (in contig order),
// pairs (x, y later in the same contig) while the gap is below d_max, plus x's own sub-fragment pairs.
__global__ __launch_bounds__(64) void k_full_mass(int n, const int* __restrict__ perm, const int* __restrict__ contig_off,
                                                   const Geo* __restrict__ geo, const Stat* __restrict__ stat,
                                                   const int* __restrict__ lcont, const int* __restrict__ lcontbp,
                                                   const int* __restrict__ pos, float nfpb, Par par, int reach_bp,
                                                   long long* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    if (i < n) {
        const int fx = perm[i];
        const Geo gx = geo[fx];
        const Stat sx = stat[fx];
        const End X = end_cur(gx, lcontbp, fx);
        for (int a = 0; a < sx.n; a++)
            for (int b = a + 1; b < sx.n; b++) acc += (double)ex_pair(X, sx, a, X, sx, b, nfpb, par);
        const int remaining = lcont[fx] - 1 - pos[fx];
        for (int k = 1; k <= remaining; k++) {
            const int fy = perm[i + k];
            const Geo gy = geo[fy];
            if (gy.start_bp - (gx.start_bp + gx.len_bp) > reach_bp) break;
            const Stat sy = stat[fy];
            const End Y = end_cur(gy, lcontbp, fy);
            for (int a = 0; a < sx.n; a++)
                for (int b = 0; b < sy.n; b++)
                    acc += (double)ex_pair(X, sx, a, Y, sy, b, nfpb, par) - (double)ex_trans(sx.accu[a], sy.accu[b], nfpb, par);
        }
    }
    (void)contig_off;
    const long long q = wave_sum_ll(to_q(acc)); // one Q rounding per fragment: fixed partition
    if ((threadIdx.x & 63) == 0 && q != 0) atomicAdd((unsigned long long*)out, (unsigned long long)q);
}

// ------------------------------------------------------------------ candidate tables
struct Task {            // windowed cis sum between piece p and piece q in one layout
    int p, q;            // piece ids (p <= q); p == q: pairs inside the piece + bin diagonals
    Xf xp, xq;           // transforms of the two pieces in that layout
    unsigned plus, minus; // 13-bit op masks: layouts (ops) in which this sum is the NEW (+) / OLD (-) value
};

struct NbTables {        // everything the scan / mass kernels need about one neighbour
    PieceKey key;
    int fB;
    int lo[NP], hi[NP], contig[NP];  // old pos range / contig label of every piece (hi < lo: empty)
    Xf xf[N_OPS][NP];
    unsigned long long changed[N_OPS]; // bit p*8+q (both orders): relation (p,q) changed; p==q: intra
    unsigned intra_any;                // bit p: some op changes piece p internally
    int n_tasks;
    int item_start[MAX_TASKS + 1];     // prefix sum of 64-fragment chunks per task: the mass work list, in a fixed order
    Task task[MAX_TASKS];
};

struct Neigh { int fB[MAXK]; };

__global__ __launch_bounds__(256) void k_tables(SoaPtr s, int fA, Neigh nb, int max_id, NbTables* __restrict__ tabs,
                                                 int* __restrict__ overflow)
{
    const int k = blockIdx.x;
    const int fB = nb.fB[k];
    NbTables& T = tabs[k];
    __shared__ Rec A0, B0;
    __shared__ int rep[NP];
    __shared__ Rec rep_old[NP];
    __shared__ Xf xf_old[NP];
    __shared__ Xf xf[N_OPS][NP];
    __shared__ unsigned long long changed[N_OPS];
    __shared__ unsigned intra_any;
    // dedupe table: entries 0..20 = old relation of (p<=q); then 13*21 new relations
    constexpr int NPAIR = 21, NENT = NPAIR * (N_OPS + 1);
    __shared__ int e_valid[NENT], e_owner[NENT], e_slot[NENT];
    __shared__ unsigned e_plus[NENT], e_minus[NENT];
    __shared__ int n_tasks;
    const int t = threadIdx.x;
    if (t == 0) {
        A0 = ld_rec(s, fA); B0 = ld_rec(s, fB);
        PieceKey key; key.cA = A0.id_c; key.a = A0.pos; key.cB = B0.id_c; key.b = B0.pos;
        T.key = key; T.fB = fB;
        piece_representatives(key, fA, fB, A0, B0, rep);
        intra_any = 0; n_tasks = 0;
        for (int p = 0; p < NP; p++) { T.lo[p] = 0; T.hi[p] = -1; T.contig[p] = -1; }
        if (fA != fB) {
            if (key.cA != key.cB) {
                T.lo[1] = 0; T.hi[1] = key.a - 1; T.lo[2] = T.hi[2] = key.a; T.lo[3] = key.a + 1; T.hi[3] = A0.l_cont - 1;
                T.lo[4] = 0; T.hi[4] = key.b - 1; T.lo[5] = T.hi[5] = key.b; T.lo[6] = key.b + 1; T.hi[6] = B0.l_cont - 1;
                for (int p = 1; p <= 3; p++) { T.contig[p] = key.cA; T.contig[p + 3] = key.cB; }
            } else {
                const int lo = key.a < key.b ? key.a : key.b, hi = key.a < key.b ? key.b : key.a;
                T.lo[1] = 0; T.hi[1] = lo - 1; T.lo[2] = T.hi[2] = lo; T.lo[3] = lo + 1; T.hi[3] = hi - 1;
                T.lo[4] = T.hi[4] = hi; T.lo[5] = hi + 1; T.hi[5] = A0.l_cont - 1;
                for (int p = 1; p <= 5; p++) T.contig[p] = key.cA;
            }
        }
    }
    if (t < N_OPS) changed[t] = 0;
    for (int e = t; e < NENT; e += blockDim.x) { e_valid[e] = 0; e_plus[e] = 0; e_minus[e] = 0; e_owner[e] = e; e_slot[e] = -1; }
    __syncthreads();
    if (t < NP) {
        if (rep[t] >= 0) { rep_old[t] = ld_rec(s, rep[t]); xf_old[t] = xf_identity(rep_old[t]); }
        else { Xf x; x.label = -1 - t; x.sigma = 1; x.off = 0; x.circ = 0; x.lbp = 0; xf_old[t] = x; }
    }
    __syncthreads();
    // transforms: one thread per (op, piece)
    if (t < N_OPS * NP) {
        const int op = t / NP, p = t % NP;
        Xf x = xf_old[p];
        if (p >= 1 && rep[p] >= 0) {
            const Move m = make_move(op, fA, fB, max_id, A0, B0);
            bool stale;
            const Rec rn = apply_move(m, rep[p], rep_old[p], &stale);
            x = xf_from(rep_old[p], rn);
        }
        xf[op][p] = x;
        T.xf[op][p] = x;
    }
    __syncthreads();
    // relations: one thread per (op, p <= q)
    for (int e = t; e < N_OPS * NPAIR; e += blockDim.x) {
        const int op = e / NPAIR;
        int idx = e % NPAIR, p = 1, q = 1;
        for (p = 1; p <= MAX_PIECES; p++) { const int row = MAX_PIECES - p + 1; if (idx < row) { q = p + idx; break; } idx -= row; }
        if (rep[p] < 0 || rep[q] < 0) continue;
        const bool chg = (p == q) ? intra_changed(xf_old[p], xf[op][p]) : rel_changed(xf_old[p], xf_old[q], xf[op][p], xf[op][q]);
        if (!chg) continue;
        atomicOr(&changed[op], (1ull << (p * 8 + q)) | (1ull << (q * 8 + p)));
        if (p == q) atomicOr(&intra_any, 1u << p);
        const int pair = e % NPAIR;
        // old relation entry (shared by all ops) and new relation entry
        if (xf_old[p].label == xf_old[q].label) { e_valid[pair] = 1; atomicOr(&e_minus[pair], 1u << op); }
        if (xf[op][p].label == xf[op][q].label) { e_valid[NPAIR + e] = 1; atomicOr(&e_plus[NPAIR + e], 1u << op); }
    }
    __syncthreads();
    // dedupe new-relation entries against every earlier entry with the same relative geometry
    for (int e = NPAIR + t; e < NENT; e += blockDim.x) {
        if (!e_valid[e]) continue;
        const int ee = e - NPAIR, op = ee / NPAIR, pair = ee % NPAIR;
        int idx = pair, p = 1, q = 1;
        for (p = 1; p <= MAX_PIECES; p++) { const int row = MAX_PIECES - p + 1; if (idx < row) { q = p + idx; break; } idx -= row; }
        const Xf xp = xf[op][p], xq = xf[op][q];
        int owner = e;
        // same pair in the old layout?
        if (e_valid[pair]) {
            const bool same = (p == q) ? !intra_changed(xf_old[p], xp) : !rel_changed(xf_old[p], xf_old[q], xp, xq);
            if (same) owner = pair; // cannot happen for changed relations, kept for safety
        }
        for (int o = 0; o < op && owner == e; o++) {
            const int e2 = NPAIR + o * NPAIR + pair;
            if (!e_valid[e2]) continue;
            const bool same = (p == q) ? !intra_changed(xf[o][p], xp) : !rel_changed(xf[o][p], xf[o][q], xp, xq);
            if (same) owner = e2;
        }
        e_owner[e] = owner;
    }
    __syncthreads();
    for (int e = NPAIR + t; e < NENT; e += blockDim.x)
        if (e_valid[e] && e_owner[e] != e) { atomicOr(&e_plus[e_owner[e]], e_plus[e]); }
    __syncthreads();
    if (t == 0) {
        int n = 0;
        for (int e = 0; e < NENT; e++)
            if (e_valid[e] && e_owner[e] == e) e_slot[e] = n++;   // n <= NENT == MAX_TASKS by construction
        n_tasks = n;
        T.n_tasks = n;
        T.intra_any = intra_any;
        for (int op = 0; op < N_OPS; op++) T.changed[op] = changed[op];
    }
    __syncthreads();
    for (int e = t; e < NENT; e += blockDim.x) {
        if (e_slot[e] < 0) continue;
        const int pair = (e < NPAIR) ? e : (e - NPAIR) % NPAIR;
        int idx = pair, p = 1, q = 1;
        for (p = 1; p <= MAX_PIECES; p++) { const int row = MAX_PIECES - p + 1; if (idx < row) { q = p + idx; break; } idx -= row; }
        Task tk; tk.p = p; tk.q = q;
        if (e < NPAIR) { tk.xp = xf_old[p]; tk.xq = xf_old[q]; }
        else { const int op = (e - NPAIR) / NPAIR; tk.xp = xf[op][p]; tk.xq = xf[op][q]; }
        if (T.hi[tk.p] - T.lo[tk.p] < T.hi[tk.q] - T.lo[tk.q]) { // the larger piece goes on the parallel (lane) axis
            const int tp = tk.p; tk.p = tk.q; tk.q = tp;
            const Xf tx = tk.xp; tk.xp = tk.xq; tk.xq = tx;
        }
        tk.plus = e_plus[e]; tk.minus = e_minus[e];
        T.task[e_slot[e]] = tk;
    }
    __syncthreads();
    if (t == 0) { // work list: chunks of 64 fragments of the task's first piece, tasks in slot order
        int acc = 0;
        for (int i = 0; i < n_tasks; i++) {
            T.item_start[i] = acc;
            const int p = T.task[i].p;
            acc += (T.hi[p] - T.lo[p] + 1 + 63) / 64;
        }
        T.item_start[n_tasks] = acc;
    }
    (void)overflow;
}

// per-fragment relevance code: 4 bits per neighbour = piece id (3) | "piece changes internally" (1)
__global__ void k_codes(SoaPtr s, int n, int K, const NbTables* __restrict__ tabs, unsigned* __restrict__ codes)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    const int c = s.p[F_IDC][f], pos = s.p[F_POS][f];
    unsigned code = 0;
    for (int k = 0; k < K; k++) {
        const int p = (tabs[k].fB == -1) ? 0 : piece_of(tabs[k].key, c, pos);
        const unsigned intra = (tabs[k].intra_any >> p) & 1u;
        code |= ((unsigned)p | (intra << 3)) << (4 * k);
    }
    codes[f] = code;
}

// ------------------------------------------------------------------ the fused candidate scan
// One pass over this rank's COO triples for all 13*K candidates.  Fast path per contact: two 4-byte gathers
// (relevance codes, L2 resident) and ~15 integer ops.  Slow path (contact joins two different pieces of some
// neighbour, or lies in a piece whose circular model changes): evaluate ex under the current layout and under
// each candidate whose relation changed, add ob * (log ex_new - log ex_old) in Q to the LDS accumulators.
template <bool SINGLE_SUB>
__global__ __launch_bounds__(256) void k_scan(const int* __restrict__ row, const int* __restrict__ col,
                                               const int* __restrict__ cnt, long long nnz,
                                               const int* __restrict__ sub2bin, const unsigned* __restrict__ codes,
                                               const Geo* __restrict__ geo, const Stat* __restrict__ stat,
                                               const int* __restrict__ lcontbp, const NbTables* __restrict__ tabs, int K,
                                               float nfpb, Par par, long long* __restrict__ out,
                                               unsigned long long* __restrict__ n_relevant)
{
    __shared__ long long acc[MAXK * N_OPS];
    __shared__ Xf s_xf[MAXK][N_OPS][NP];
    __shared__ unsigned long long s_changed[MAXK][N_OPS];
    for (int i = threadIdx.x; i < MAXK * N_OPS; i += blockDim.x) acc[i] = 0;
    for (int i = threadIdx.x; i < K * N_OPS * NP; i += blockDim.x) {
        const int k = i / (N_OPS * NP), r = i % (N_OPS * NP);
        s_xf[k][r / NP][r % NP] = tabs[k].xf[r / NP][r % NP];
    }
    for (int i = threadIdx.x; i < K * N_OPS; i += blockDim.x) s_changed[i / N_OPS][i % N_OPS] = tabs[i / N_OPS].changed[i % N_OPS];
    __syncthreads();
    unsigned long long n_rel = 0;
    const long long n4 = nnz >> 2;
    const int4* row4 = reinterpret_cast<const int4*>(row);
    const int4* col4 = reinterpret_cast<const int4*>(col);
    const int4* cnt4 = reinterpret_cast<const int4*>(cnt);
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n4 + 1; g += stride) {
        int r[4], c[4], v[4], m = 4;
        if (g < n4) {
            const int4 r4 = row4[g], c4 = col4[g], v4 = cnt4[g];
            r[0] = r4.x; r[1] = r4.y; r[2] = r4.z; r[3] = r4.w;
            c[0] = c4.x; c[1] = c4.y; c[2] = c4.z; c[3] = c4.w;
            v[0] = v4.x; v[1] = v4.y; v[2] = v4.z; v[3] = v4.w;
        } else { // tail (nnz % 4 contacts), handled by exactly one thread
            m = (int)(nnz - (n4 << 2));
            for (int j = 0; j < m; j++) { r[j] = row[(n4 << 2) + j]; c[j] = col[(n4 << 2) + j]; v[j] = cnt[(n4 << 2) + j]; }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (j >= m) break;
            int fx, fy, slx = 0, sly = 0;
            if (SINGLE_SUB) { fx = r[j]; fy = c[j]; }
            else { const int a = sub2bin[r[j]], b = sub2bin[c[j]]; fx = a >> 2; fy = b >> 2; slx = a & 3; sly = b & 3; }
            const unsigned ci = codes[fx], cj = codes[fy];
            const unsigned ti = ci & 0x77777777u, tj = cj & 0x77777777u;
            const unsigned nzi = (ti | (ti >> 1) | (ti >> 2)) & 0x11111111u;
            const unsigned nzj = (tj | (tj >> 1) | (tj >> 2)) & 0x11111111u;
            const unsigned df = ti ^ tj;
            const unsigned dnz = (df | (df >> 1) | (df >> 2)) & 0x11111111u;
            unsigned rel = nzi & nzj & (dnz | ((ci >> 3) & 0x11111111u));
            if (rel == 0) continue;
            if (!SINGLE_SUB && fx == fy) continue; // a bin's own pixel is never revisited (kernels3.cu:3356-3380)
            // ---- slow path ----
            const Geo gx = geo[fx], gy = geo[fy];
            const Stat sx = stat[fx], sy = stat[fy];
            const End X0 = end_cur(gx, lcontbp, fx), Y0 = end_cur(gy, lcontbp, fy);
            const double ln_old = log((double)ex_pair(X0, sx, slx, Y0, sy, sly, nfpb, par));
            const double ob = (double)v[j];
            while (rel) {
                const int k = (__ffs((int)rel) - 1) >> 2;
                rel &= rel - 1;
                n_rel++;
                const int p = (ci >> (4 * k)) & 7, q = (cj >> (4 * k)) & 7;
                const unsigned long long bit = 1ull << (p * 8 + q);
                for (int op = 0; op < N_OPS; op++) {
                    if (!(s_changed[k][op] & bit)) continue;
                    const End X = end_xf(gx, s_xf[k][op][p]), Y = end_xf(gy, s_xf[k][op][q]);
                    const double ln_new = log((double)ex_pair(X, sx, slx, Y, sy, sly, nfpb, par));
                    const long long qv = to_q(ob * (ln_new - ln_old));
                    if (qv != 0) atomicAdd((unsigned long long*)&acc[k * N_OPS + op], (unsigned long long)qv);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K * N_OPS; i += blockDim.x)
        if (acc[i] != 0) atomicAdd((unsigned long long*)&out[i], (unsigned long long)acc[i]);
    n_rel = (unsigned long long)wave_sum_ll((long long)n_rel);
    if ((threadIdx.x & 63) == 0 && n_rel) atomicAdd(n_relevant, n_rel);
}

// ------------------------------------------------------------------ mass tasks
// work item = (neighbour, task, chunk of 64 fragments of piece p); one wave per item, lane = fragment x.
__global__ __launch_bounds__(64) void k_mass(const NbTables* __restrict__ tabs, int K,
                                              const int* __restrict__ perm, const int* __restrict__ contig_off,
                                              const Geo* __restrict__ geo, const Stat* __restrict__ stat, float nfpb,
                                              Par par, int reach_bp, int rank, int world, long long* __restrict__ out,
                                              unsigned long long* __restrict__ n_items)
{
    const int lane = threadIdx.x;
    unsigned long long items = 0;
    for (int k = 0; k < K; k++) {
    const NbTables& T = tabs[k];
    const int n_tasks = T.n_tasks, total = T.item_start[n_tasks];
    // static sharding of the fixed, ordered item list over ranks; every wave reaches the loop exit
    for (int w = blockIdx.x * world + rank; w < total; w += gridDim.x * world) {
        int lo_t = 0, hi_t = n_tasks - 1; // task of item w: last task with item_start <= w
        while (lo_t < hi_t) { const int mid = (lo_t + hi_t + 1) >> 1; if (T.item_start[mid] <= w) lo_t = mid; else hi_t = mid - 1; }
        const int ti = lo_t, chunk = w - T.item_start[ti];
        const Task& tk = T.task[ti];
        const int np = T.hi[tk.p] - T.lo[tk.p] + 1, nq = T.hi[tk.q] - T.lo[tk.q] + 1;
        items++;
        const int base_p = contig_off[T.contig[tk.p]] + T.lo[tk.p];
        const int base_q = contig_off[T.contig[tk.q]] + T.lo[tk.q];
        const int ix = chunk * 64 + lane;
        double acc = 0.0;
        if (ix < np) {
            const int fx = perm[base_p + ix];
            const Geo gx = geo[fx];
            const Stat sx = stat[fx];
            const End X = end_xf(gx, tk.xp);
            if (tk.p == tk.q) {
                // (x's own sub-fragment pairs are left out: candidates never revisit a bin's own pixel)
                // later fragments of the same piece, walking away from x in the new layout
                for (int iy = ix + 1; iy < np; iy++) {
                    const int fy = perm[base_p + iy];
                    const Geo gy = geo[fy];
                    const End Y = end_xf(gy, tk.xp);
                    const int gap = X.start_bp < Y.start_bp ? Y.start_bp - (X.start_bp + gx.len_bp) : X.start_bp - (Y.start_bp + gy.len_bp);
                    if (gap > reach_bp) break;
                    const Stat sy = stat[fy];
                    for (int a = 0; a < sx.n; a++)
                        for (int b = 0; b < sy.n; b++)
                            acc += (double)ex_pair(X, sx, a, Y, sy, b, nfpb, par) - (double)ex_trans(sx.accu[a], sy.accu[b], nfpb, par);
                }
            } else {
                // q's fragment nearest to x in this layout: pieces map to disjoint intervals, so the side is
                // fixed by comparing x with q's first fragment
                const int fq0 = perm[base_q];
                const End Q0 = end_xf(geo[fq0], tk.xq);
                const bool x_below = X.start_bp < Q0.start_bp;
                const bool asc = (tk.xq.sigma > 0) == x_below; // walk q by increasing old position?
                for (int s = 0; s < nq; s++) {
                    const int iy = asc ? s : nq - 1 - s;
                    const int fy = perm[base_q + iy];
                    const Geo gy = geo[fy];
                    const End Y = end_xf(gy, tk.xq);
                    const int gap = x_below ? Y.start_bp - (X.start_bp + gx.len_bp) : X.start_bp - (Y.start_bp + gy.len_bp);
                    if (gap > reach_bp) break;
                    const Stat sy = stat[fy];
                    for (int a = 0; a < sx.n; a++)
                        for (int b = 0; b < sy.n; b++)
                            acc += (double)ex_pair(X, sx, a, Y, sy, b, nfpb, par) - (double)ex_trans(sx.accu[a], sy.accu[b], nfpb, par);
                }
            }
        }
        // one Q rounding per fragment x: the partition is fixed by the layout, not by the launch
        const long long qv = wave_sum_ll(to_q(acc));
        if (lane == 0 && qv != 0) {
            // logL = contacts - mass: the NEW layout's mass counts negative, the OLD one positive
            for (int op = 0; op < N_OPS; op++) {
                const long long sgn = (long long)((tk.minus >> op) & 1u) - (long long)((tk.plus >> op) & 1u);
                if (sgn != 0) atomicAdd((unsigned long long*)&out[k * N_OPS + op], (unsigned long long)(sgn * qv));
            }
        }
    }
    }
    if (lane == 0 && items) atomicAdd(n_items, items);
}

// ------------------------------------------------------------------ host side
struct Ctx {
    int device = 0;
    std::string err;
    hipStream_t stream = nullptr;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    bool timing_valid = false;
    // problem
    int n = 0, n_bins = 0, n_sub_total = 0;
    long long nnz = 0;
    bool single_sub = true;
    float nfpb = 1.0f;
    Par par{};
    bool have_par = false, have_sub = false, have_frags = false, have_contacts = false, order_valid = false;
    int n_contigs = 0;
    double t_all = 0.0;         // layout independent all-trans expected mass
    double c_lf = 0.0;          // sum of log-factorial terms of this shard's contacts
    std::vector<int> h_accu;    // [n_bins][3]
    std::vector<int> h_nsub;
    // device
    int* soa_mem[2] = {nullptr, nullptr};
    SoaPtr soa[2];
    int cur = 0;
    Geo* geo = nullptr;
    Stat* stat = nullptr;
    int* sub2bin = nullptr;
    int *row = nullptr, *col = nullptr, *cnt = nullptr;
    unsigned* codes = nullptr;
    unsigned long long *keys = nullptr, *keys_sorted = nullptr;
    int *o2n = nullptr, *len_of = nullptr, *contig_off = nullptr, *perm = nullptr;
    void* cub_tmp = nullptr;
    size_t cub_tmp_bytes = 0;
    NbTables* tabs = nullptr;
    long long* d_scalars = nullptr; // [0..7] stats, [8..9] full q, [10] n_relevant, [11] n_items, [12] overflow/stale (int)
    long long* d_qout = nullptr;    // K*13
    long long counters[4] = {0, 0, 0, 0};
    float timing[4] = {0, 0, 0, 0};
};

#define CK(call)                                                                                     \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            char b_[256];                                                                            \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            h->err = b_;                                                                             \
            return GRAAL_E_HIP;                                                                      \
        }                                                                                            \
    } while (0)

int fail(Ctx* h, int code, const char* msg) { h->err = msg; return code; }

inline int blocks_for(long long n, int bs) { return (int)((n + bs - 1) / bs); }

// float32 log-factorial term of evaluate_likelihood_double (kernels3.cu:191-210, factorial :80-93)
double lf_term(double ob)
{
    if (ob >= 15.0) return ob * log(ob) - ob + log(sqrt(ob * 2.0 * M_PI));
    if (ob > 0.0) {
        float n = floorf((float)ob), result = 1.0f;
        if (n < 10) { for (int c = 1; c <= n; c++) result = result * c; }
        else result = powf(n, n) * expf(-n) * sqrtf((float)(2 * M_PI * n));
        return log((double)result);
    }
    return 0.0;
}

void compute_t_all(Ctx* h)
{
    // T_all = sum over pairs of DIFFERENT bins, all slot pairs, of float32(v * float32(float32(ax*ay)/nfpb))
    if (!h->have_par || !h->have_sub) return;
    std::vector<long long> hist;
    for (int b = 0; b < h->n_bins; b++)
        for (int s = 0; s < h->h_nsub[b]; s++) {
            const int a = h->h_accu[3 * b + s];
            if ((size_t)a >= hist.size()) hist.resize(a + 1, 0);
            hist[a]++;
        }
    auto c = [&](long long m) { return (double)(h->par.v_inter * ((float)(int)m / h->nfpb)); };
    double all = 0.0;
    for (size_t u = 0; u < hist.size(); u++)
        if (hist[u])
            for (size_t w = 0; w < hist.size(); w++)
                if (hist[w]) all += (double)hist[u] * (double)hist[w] * c((long long)u * (long long)w);
    double self = 0.0;
    for (int b = 0; b < h->n_bins; b++)
        for (int s = 0; s < h->h_nsub[b]; s++)
            for (int t = 0; t < h->h_nsub[b]; t++) self += c((long long)h->h_accu[3 * b + s] * h->h_accu[3 * b + t]);
    h->t_all = 0.5 * (all - self);
}

int reach_bp(const Ctx* h) { return (int)ceil((double)h->par.d_max * 1000.0) + 1000; }

int refresh(Ctx* h)
{
    k_refresh_geo<<<blocks_for(h->n, 256), 256, 0, h->stream>>>(h->soa[h->cur], h->geo, h->n);
    CK(hipGetLastError());
    return GRAAL_OK;
}

} // namespace

struct graal_ctx : Ctx {};

extern "C" {

int graal_abi_version(void) { return GRAAL_ABI_VERSION; }

int graal_create(int device, graal_ctx** out)
{
    if (!out) return GRAAL_E_ARG;
    *out = nullptr;
    graal_ctx* h = new graal_ctx();
    *out = h; // returned even on failure so that graal_last_error works; caller destroys it
    h->device = device;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        h->err = "no HIP device available (this library has no CPU fallback)";
        return GRAAL_E_HIP;
    }
    if (device < 0 || device >= count) { h->err = "device index out of range"; return GRAAL_E_ARG; }
    CK(hipSetDevice(device));
    CK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    for (auto& ev : h->ev) CK(hipEventCreate(&ev));
    CK(hipMalloc(&h->d_scalars, 16 * sizeof(long long)));
    CK(hipMalloc(&h->d_qout, MAXK * N_OPS * sizeof(long long)));
    CK(hipMalloc(&h->tabs, MAXK * sizeof(NbTables)));
    return GRAAL_OK;
}

void graal_destroy(graal_ctx* h)
{
    if (!h) return;
    if (h->stream) {
        hipSetDevice(h->device);
        hipStreamSynchronize(h->stream);
        void* ptrs[] = {h->soa_mem[0], h->soa_mem[1], h->geo, h->stat, h->sub2bin, h->row, h->col, h->cnt, h->codes,
                        h->keys, h->keys_sorted, h->o2n, h->len_of, h->contig_off, h->perm, h->cub_tmp, h->tabs,
                        h->d_scalars, h->d_qout};
        for (void* p : ptrs) if (p) hipFree(p);
        for (auto& ev : h->ev) if (ev) hipEventDestroy(ev);
        hipStreamDestroy(h->stream);
    }
    delete h;
}

const char* graal_last_error(const graal_ctx* h) { return h ? h->err.c_str() : "null handle"; }

int graal_set_params(graal_ctx* h, const float* p)
{
    if (!h || !p) return GRAAL_E_ARG;
    memcpy(&h->par, p, sizeof(Par));
    if (!(h->par.v_inter > 0.0f)) return fail(h, GRAAL_E_ARG, "v_inter must be > 0 (the sparse form prices every pixel at >= v_inter)");
    if (!(h->par.d_max > 0.0f) || !(h->par.d_max < 2.0e6f)) return fail(h, GRAAL_E_ARG, "d_max out of range");
    h->have_par = true;
    compute_t_all(h);
    return GRAAL_OK;
}

int graal_upload_subfrags(graal_ctx* h, const int32_t* sub_id, const float* sub_len, const int32_t* sub_accu, int32_t n_bins,
                          int32_t n_sub_total, float nfpb)
{
    if (!h || !sub_id || !sub_len || !sub_accu || n_bins <= 0 || n_sub_total < n_bins || !(nfpb > 0)) return GRAAL_E_ARG;
    CK(hipSetDevice(h->device));
    std::vector<Stat> st(n_bins);
    std::vector<int> s2b(n_sub_total, -1);
    h->h_accu.assign(sub_accu, sub_accu + 3 * (size_t)n_bins);
    h->h_nsub.resize(n_bins);
    bool single = true;
    for (int b = 0; b < n_bins; b++) {
        const int ns = sub_id[4 * b + 3];
        if (ns < 1 || ns > 3) return fail(h, GRAAL_E_ARG, "n_sub must be 1..3 (sub-sampling factor 3 is baked in, simulation_loader.py:682-698)");
        single &= (ns == 1);
        h->h_nsub[b] = ns;
        Stat s{};
        s.n = ns;
        for (int k = 0; k < 3; k++) { s.len[k] = k < ns ? sub_len[3 * b + k] : 0.0f; s.accu[k] = k < ns ? sub_accu[3 * b + k] : 0; }
        for (int k = 0; k < ns; k++) {
            const int sid = sub_id[4 * b + k];
            if (sid < 0 || sid >= n_sub_total || s2b[sid] != -1) return fail(h, GRAAL_E_ARG, "sub_id must map sub-fragments to bins one-to-one");
            if (sub_accu[3 * b + k] <= 0 || sub_accu[3 * b + k] > 30000) return fail(h, GRAAL_E_ARG, "sub_accu out of range");
            s2b[sid] = b * 4 + k;
        }
        st[b] = s;
    }
    for (int v : s2b) if (v < 0) return fail(h, GRAAL_E_ARG, "every sub-fragment must belong to a bin");
    // single_sub additionally needs sub id == bin id so that the scan can skip the sub2bin gather
    for (int b = 0; single && b < n_bins; b++) single = (sub_id[4 * b] == b);
    if (h->stat) { hipFree(h->stat); hipFree(h->sub2bin); }
    CK(hipMalloc(&h->stat, sizeof(Stat) * (size_t)n_bins));
    CK(hipMalloc(&h->sub2bin, sizeof(int) * (size_t)n_sub_total));
    CK(hipMemcpy(h->stat, st.data(), sizeof(Stat) * (size_t)n_bins, hipMemcpyHostToDevice));
    CK(hipMemcpy(h->sub2bin, s2b.data(), sizeof(int) * (size_t)n_sub_total, hipMemcpyHostToDevice));
    h->n_bins = n_bins; h->n_sub_total = n_sub_total; h->nfpb = nfpb; h->single_sub = single; h->have_sub = true;
    compute_t_all(h);
    return GRAAL_OK;
}

int graal_upload_contacts(graal_ctx* h, const int32_t* row, const int32_t* col, const int32_t* count, int64_t nnz)
{
    if (!h || nnz < 0 || (nnz > 0 && (!row || !col || !count))) return GRAAL_E_ARG;
    if (!h->have_sub) return fail(h, GRAAL_E_STATE, "upload_subfrags first");
    CK(hipSetDevice(h->device));
    double c_lf = 0.0;
    double lf_small[16];
    for (int i = 0; i < 16; i++) lf_small[i] = lf_term((double)i);
    for (int64_t i = 0; i < nnz; i++) {
        if (row[i] < 0 || col[i] >= h->n_sub_total || row[i] >= col[i]) return fail(h, GRAAL_E_ARG, "contacts need 0 <= row < col < n_sub_total");
        if (count[i] <= 0) return fail(h, GRAAL_E_ARG, "contact counts must be > 0");
        c_lf += count[i] < 16 ? lf_small[count[i]] : lf_term((double)count[i]);
    }
    if (h->row) { hipFree(h->row); hipFree(h->col); hipFree(h->cnt); h->row = h->col = h->cnt = nullptr; }
    const size_t bytes = sizeof(int) * (size_t)(nnz + 4); // +4: int4 tail reads stay in bounds
    CK(hipMalloc(&h->row, bytes)); CK(hipMalloc(&h->col, bytes)); CK(hipMalloc(&h->cnt, bytes));
    if (nnz) {
        CK(hipMemcpy(h->row, row, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice));
        CK(hipMemcpy(h->col, col, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice));
        CK(hipMemcpy(h->cnt, count, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice));
    }
    h->nnz = nnz; h->c_lf = c_lf; h->have_contacts = true;
    return GRAAL_OK;
}

int graal_upload_frags(graal_ctx* h, const int32_t* const soa[GRAAL_N_FIELDS], int32_t n)
{
    if (!h || !soa || n <= 0) return GRAAL_E_ARG;
    if (!h->have_sub) return fail(h, GRAAL_E_STATE, "upload_subfrags first");
    if (n != h->n_bins) return fail(h, GRAAL_E_UNSUPPORTED, "repeated fragments (n_new_frags != n_frags) are not supported yet");
    if ((long long)n >= (1ll << LABEL_BITS) / 2 - 4) return fail(h, GRAAL_E_ARG, "too many fragments for the relabel key");
    CK(hipSetDevice(h->device));
    for (int i = 0; i < n; i++) {
        if (soa[F_IDC][i] < 0 || soa[F_IDC][i] >= 2 * n + 4) return fail(h, GRAAL_E_ARG, "id_c must be in [0, 2n+4)");
        if (soa[F_REP][i] != 0 || soa[F_ACTIV][i] != 1 || soa[F_IDD][i] != i)
            return fail(h, GRAAL_E_UNSUPPORTED, "repeats / inactive fragments are not supported yet");
    }
    if (h->n != n) {
        void* old[] = {h->soa_mem[0], h->soa_mem[1], h->geo, h->codes, h->keys, h->keys_sorted, h->o2n, h->len_of, h->contig_off, h->perm, h->cub_tmp};
        for (void* p : old) if (p) hipFree(p);
        for (int b = 0; b < 2; b++) {
            CK(hipMalloc(&h->soa_mem[b], sizeof(int) * (size_t)n * GRAAL_N_FIELDS));
            for (int k = 0; k < GRAAL_N_FIELDS; k++) h->soa[b].p[k] = h->soa_mem[b] + (size_t)k * n;
        }
        CK(hipMalloc(&h->geo, sizeof(Geo) * (size_t)n));
        CK(hipMalloc(&h->codes, sizeof(unsigned) * (size_t)n));
        CK(hipMalloc(&h->keys, sizeof(unsigned long long) * (size_t)n));
        CK(hipMalloc(&h->keys_sorted, sizeof(unsigned long long) * (size_t)n));
        CK(hipMalloc(&h->o2n, sizeof(int) * (size_t)(2 * n + 8)));
        CK(hipMalloc(&h->len_of, sizeof(int) * (size_t)(n + 1)));
        CK(hipMalloc(&h->contig_off, sizeof(int) * (size_t)(n + 1)));
        CK(hipMalloc(&h->perm, sizeof(int) * (size_t)n));
        size_t b1 = 0, b2 = 0;
        hipcub::DeviceRadixSort::SortKeys(nullptr, b1, h->keys, h->keys_sorted, n, 0, 2 * LABEL_BITS, h->stream);
        hipcub::DeviceScan::ExclusiveSum(nullptr, b2, h->len_of, h->contig_off, n, h->stream);
        h->cub_tmp_bytes = b1 > b2 ? b1 : b2;
        CK(hipMalloc(&h->cub_tmp, h->cub_tmp_bytes));
        h->n = n;
    }
    h->cur = 0;
    for (int k = 0; k < GRAAL_N_FIELDS; k++)
        CK(hipMemcpy(h->soa[0].p[k], soa[k], sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
    h->have_frags = true; h->order_valid = false;
    int rc = refresh(h);
    if (rc) return rc;
    CK(hipStreamSynchronize(h->stream));
    return GRAAL_OK;
}

int graal_download_frags(graal_ctx* h, int32_t* const soa[GRAAL_N_FIELDS])
{
    if (!h || !soa) return GRAAL_E_ARG;
    if (!h->have_frags) return fail(h, GRAAL_E_STATE, "no fragments uploaded");
    CK(hipSetDevice(h->device));
    CK(hipStreamSynchronize(h->stream));
    for (int k = 0; k < GRAAL_N_FIELDS; k++)
        CK(hipMemcpy(soa[k], h->soa[h->cur].p[k], sizeof(int) * (size_t)h->n, hipMemcpyDeviceToHost));
    return GRAAL_OK;
}

int graal_layout_stats(graal_ctx* h, int64_t out[8])
{
    if (!h || !out) return GRAAL_E_ARG;
    if (!h->have_frags) return fail(h, GRAAL_E_STATE, "no fragments uploaded");
    CK(hipSetDevice(h->device));
    long long init[8] = {0, 0, 0, 0, 0, 0x7fffffff, 0, -1};
    CK(hipMemcpyAsync(h->d_scalars, init, sizeof init, hipMemcpyHostToDevice, h->stream));
    CK(hipMemsetAsync(h->d_scalars + 14, 0, sizeof(long long), h->stream));
    k_stats<<<std::min(blocks_for(h->n, 256), 512), 256, 0, h->stream>>>(h->soa[h->cur], h->n, h->d_scalars);
    CK(hipGetLastError());
    long long res[16];
    CK(hipMemcpyAsync(res, h->d_scalars, sizeof res, hipMemcpyDeviceToHost, h->stream));
    CK(hipStreamSynchronize(h->stream));
    out[0] = res[0]; out[1] = res[1]; out[2] = res[2]; out[3] = res[3]; out[4] = res[4]; out[5] = res[5];
    out[6] = res[14]; out[7] = 0;
    h->n_contigs = (int)res[0];
    return GRAAL_OK;
}

int graal_relabel_contigs(graal_ctx* h, int32_t* max_id)
{
    if (!h) return GRAAL_E_ARG;
    if (!h->have_frags) return fail(h, GRAAL_E_STATE, "no fragments uploaded");
    CK(hipSetDevice(h->device));
    const int n = h->n, bs = 256, nb = blocks_for(n, bs);
    SoaPtr s = h->soa[h->cur];
    long long init[8] = {0, 0, 0, 0, 0, 0x7fffffff, 0, -1};
    CK(hipMemcpyAsync(h->d_scalars, init, sizeof init, hipMemcpyHostToDevice, h->stream));
    k_stats<<<std::min(nb, 512), bs, 0, h->stream>>>(s, n, h->d_scalars);
    k_relabel_keys<<<nb, bs, 0, h->stream>>>(s, n, h->keys);
    CK(hipGetLastError());
    size_t tb = h->cub_tmp_bytes;
    CK(hipcub::DeviceRadixSort::SortKeys(h->cub_tmp, tb, h->keys, h->keys_sorted, n, 0, 2 * LABEL_BITS, h->stream));
    long long res[8];
    CK(hipMemcpyAsync(res, h->d_scalars, sizeof res, hipMemcpyDeviceToHost, h->stream));
    CK(hipStreamSynchronize(h->stream));
    const int nc = (int)res[0];
    if (nc <= 0 || nc > n || res[7] >= 2 * n + 8) return fail(h, GRAAL_E_STATE, "corrupt layout: contig heads / labels out of range");
    if (res[6] != 0) return fail(h, GRAAL_E_UNSUPPORTED, "repeats / inactive fragments are not supported yet");
    k_relabel_o2n<<<blocks_for(nc, bs), bs, 0, h->stream>>>(h->keys_sorted, nc, h->o2n, h->len_of);
    k_relabel_apply<<<nb, bs, 0, h->stream>>>(s, n, h->o2n);
    CK(hipGetLastError());
    tb = h->cub_tmp_bytes;
    CK(hipcub::DeviceScan::ExclusiveSum(h->cub_tmp, tb, h->len_of, h->contig_off, nc, h->stream));
    k_build_perm<<<nb, bs, 0, h->stream>>>(s, n, h->contig_off, h->perm);
    CK(hipGetLastError());
    int rc = refresh(h);
    if (rc) return rc;
    CK(hipStreamSynchronize(h->stream));
    h->n_contigs = nc; h->order_valid = true;
    if (max_id) *max_id = nc - 1;
    return GRAAL_OK;
}

int graal_eval_full_q(graal_ctx* h, int64_t q_out[2])
{
    if (!h || !q_out) return GRAAL_E_ARG;
    if (!(h->have_frags && h->have_contacts && h->have_par)) return fail(h, GRAAL_E_STATE, "upload fragments, contacts and parameters first");
    if (!h->order_valid) return fail(h, GRAAL_E_STATE, "call graal_relabel_contigs after changing the layout");
    CK(hipSetDevice(h->device));
    CK(hipMemsetAsync(h->d_scalars + 8, 0, 2 * sizeof(long long), h->stream));
    SoaPtr s = h->soa[h->cur];
    if (h->nnz) {
        const int nb = (int)std::min<long long>((h->nnz + 255) / 256, 256 * 16);
        k_full_nnz<<<nb, 256, 0, h->stream>>>(h->row, h->col, h->cnt, h->nnz, h->sub2bin, h->geo, h->stat, s.p[F_LCONTBP],
                                               h->nfpb, h->par, h->d_scalars + 8);
    }
    k_full_mass<<<blocks_for(h->n, 64), 64, 0, h->stream>>>(h->n, h->perm, h->contig_off, h->geo, h->stat, s.p[F_LCONT],
                                                             s.p[F_LCONTBP], s.p[F_POS], h->nfpb, h->par, reach_bp(h),
                                                             h->d_scalars + 9);
    CK(hipGetLastError());
    long long res[2];
    CK(hipMemcpyAsync(res, h->d_scalars + 8, sizeof res, hipMemcpyDeviceToHost, h->stream));
    CK(hipStreamSynchronize(h->stream));
    q_out[0] = res[0] - (int64_t)llrint(h->c_lf * Q_SCALE);
    q_out[1] = -(res[1] + (int64_t)llrint(h->t_all * Q_SCALE));
    return GRAAL_OK;
}

int graal_eval_candidates_q(graal_ctx* h, int32_t fA, const int32_t* fB, int32_t K, int32_t max_id, int32_t rank, int32_t world,
                            int64_t* d_q_out, void* stream_v)
{
    if (!h || !fB || !d_q_out || K < 1 || K > MAXK || world < 1 || rank < 0 || rank >= world) return GRAAL_E_ARG;
    if (!(h->have_frags && h->have_contacts && h->have_par)) return fail(h, GRAAL_E_STATE, "upload fragments, contacts and parameters first");
    if (!h->order_valid) return fail(h, GRAAL_E_STATE, "call graal_relabel_contigs after changing the layout");
    if (fA < 0 || fA >= h->n) return fail(h, GRAAL_E_ARG, "fA out of range");
    Neigh nb;
    for (int k = 0; k < MAXK; k++) {
        nb.fB[k] = k < K ? fB[k] : -1;
        if (k < K && (fB[k] < 0 || fB[k] >= h->n)) return fail(h, GRAAL_E_ARG, "fB out of range");
    }
    CK(hipSetDevice(h->device));
    hipStream_t st = stream_v ? (hipStream_t)stream_v : h->stream;
    SoaPtr s = h->soa[h->cur];
    const int n = h->n;
    CK(hipEventRecord(h->ev[0], st));
    CK(hipMemsetAsync(d_q_out, 0, sizeof(long long) * (size_t)K * N_OPS, st));
    CK(hipMemsetAsync(h->d_scalars + 10, 0, 3 * sizeof(long long), st));
    k_tables<<<K, 256, 0, st>>>(s, fA, nb, max_id, h->tabs, (int*)(h->d_scalars + 12));
    k_codes<<<blocks_for(n, 256), 256, 0, st>>>(s, n, K, h->tabs, h->codes);
    CK(hipEventRecord(h->ev[1], st));
    {
        const long long groups = (h->nnz >> 2) + 1;
        const int nbk = (int)std::min<long long>((groups + 255) / 256, 256 * 8);
        if (h->single_sub)
            k_scan<true><<<nbk, 256, 0, st>>>(h->row, h->col, h->cnt, h->nnz, h->sub2bin, h->codes, h->geo, h->stat,
                                              s.p[F_LCONTBP], h->tabs, K, h->nfpb, h->par, (long long*)d_q_out,
                                              (unsigned long long*)(h->d_scalars + 10));
        else
            k_scan<false><<<nbk, 256, 0, st>>>(h->row, h->col, h->cnt, h->nnz, h->sub2bin, h->codes, h->geo, h->stat,
                                               s.p[F_LCONTBP], h->tabs, K, h->nfpb, h->par, (long long*)d_q_out,
                                               (unsigned long long*)(h->d_scalars + 10));
    }
    CK(hipEventRecord(h->ev[2], st));
    {
        const int nbm = std::min(std::max((n + 63) / 64 * 4, 256), 256 * 16);
        k_mass<<<nbm, 64, 0, st>>>(h->tabs, K, h->perm, h->contig_off, h->geo, h->stat, h->nfpb, h->par,
                                    reach_bp(h), rank, world, (long long*)d_q_out, (unsigned long long*)(h->d_scalars + 11));
    }
    CK(hipEventRecord(h->ev[3], st));
    CK(hipGetLastError());
    h->timing_valid = true;
    return GRAAL_OK;
}

int graal_eval_candidates(graal_ctx* h, int32_t fA, const int32_t* fB, int32_t K, int32_t max_id, double* delta)
{
    if (!h || !delta) return GRAAL_E_ARG;
    int rc = graal_eval_candidates_q(h, fA, fB, K, max_id, 0, 1, (int64_t*)h->d_qout, nullptr);
    if (rc) return rc;
    long long q[MAXK * N_OPS];
    CK(hipMemcpyAsync(q, h->d_qout, sizeof(long long) * (size_t)K * N_OPS, hipMemcpyDeviceToHost, h->stream));
    CK(hipStreamSynchronize(h->stream));
    for (int i = 0; i < K * N_OPS; i++) delta[i] = (double)q[i] / Q_SCALE;
    return GRAAL_OK;
}

int graal_apply_move(graal_ctx* h, int32_t fA, int32_t fB, int32_t op, int32_t max_id, int32_t* n_stale)
{
    if (!h || op < 0 || op >= N_OPS) return GRAAL_E_ARG;
    if (!h->have_frags) return fail(h, GRAAL_E_STATE, "no fragments uploaded");
    if (fA < 0 || fA >= h->n || fB < 0 || fB >= h->n) return fail(h, GRAAL_E_ARG, "fragment index out of range");
    CK(hipSetDevice(h->device));
    int* d_stale = (int*)(h->d_scalars + 13);
    CK(hipMemsetAsync(d_stale, 0, sizeof(int), h->stream));
    k_apply<<<blocks_for(h->n, 256), 256, 0, h->stream>>>(h->soa[h->cur], h->soa[1 - h->cur], h->n, op, fA, fB, max_id, d_stale);
    CK(hipGetLastError());
    h->cur = 1 - h->cur;
    h->order_valid = false;
    int rc = refresh(h);
    if (rc) return rc;
    int stale = 0;
    CK(hipMemcpyAsync(&stale, d_stale, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    CK(hipStreamSynchronize(h->stream));
    if (n_stale) *n_stale = stale;
    return GRAAL_OK;
}

int graal_last_timing(graal_ctx* h, float out[4])
{
    if (!h || !out) return GRAAL_E_ARG;
    if (!h->timing_valid) return fail(h, GRAAL_E_STATE, "no candidate evaluation yet");
    CK(hipSetDevice(h->device));
    CK(hipEventSynchronize(h->ev[3]));
    for (int i = 0; i < 3; i++) CK(hipEventElapsedTime(&out[i], h->ev[i], h->ev[i + 1]));
    out[3] = 0.0f;
    return GRAAL_OK;
}

int graal_last_counters(graal_ctx* h, int64_t out[4])
{
    if (!h || !out) return GRAAL_E_ARG;
    CK(hipSetDevice(h->device));
    long long res[3];
    CK(hipMemcpy(res, h->d_scalars + 10, sizeof res, hipMemcpyDeviceToHost));
    out[0] = h->nnz; out[1] = res[0]; out[2] = 0; out[3] = res[1];
    return GRAAL_OK;
}

} // extern "C"
