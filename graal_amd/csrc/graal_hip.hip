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
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <dlfcn.h>
#include <unistd.h>
#include <map>
#include <string>
#include <vector>

#include "../../include/graal_hip.h"
#include "frag_ops.h"
#include "strict_sets.h"
#include "model_math.h"

using namespace graal;

namespace {

constexpr double Q_SCALE = 1073741824.0; // 2^GRAAL_Q_BITS
constexpr int MAXK = GRAAL_MAX_NEIGHBOURS;
constexpr int NP = MAX_PIECES + 1;       // piece ids 0..6
constexpr int CODE_BITS = 3;             // a fragment's piece id (0..6) per neighbour, packed into one word
constexpr unsigned CODE_LSB = 0x09249249u; // bit 0 of each of the 10 fields
static_assert(MAXK * CODE_BITS <= 32, "piece codes of all neighbours must fit one word");
constexpr int N_PAIRS = 21;                 // unordered pairs (p <= q) of the 6 pieces
constexpr int MAX_TASKS = N_PAIRS * (N_OPS + 1); // per neighbour: 21 piece pairs x (old + 13 candidate layouts), before dedupe
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
// pow(x, 2.0f) of the reference: the correctly rounded square IS the single float32 multiplication (what a correctly
// rounded powf returns, e.g. glibc's in the oracle); the general powf costs ~100 instructions more
__device__ __forceinline__ float sq(float x) { return x * x; }

// powf / expf / log of the model: correctly rounded, self-contained (model_math.h).  The contact model is ~110 instructions per
// evaluation with them (it was ~200 with the device library's float-float powf).
// rippe_contacts kernels3.cu:120
__device__ __forceinline__ float rippe(float s, const Par& p)
{
    float result = 0.0f;
    if ((s > 0.0f) && (s < p.d_max))
    {
        // (x / 1.0f is x: a Kuhn length of one -- the fitted models' -- needs no division; wave-uniform, the parameters are kernel arguments)
        float n = s * p.lm;
        if (p.kuhn != 1.0f) { asm volatile("" : "+v"(n)); n = n / p.kuhn; }   // (the barrier keeps this a branch: as a select, both sides were computed)
        result = (p.c1 * mm_powf_pos(s, p.slope) * mm_expf((p.d - 2) / (sq(n) + p.d))) * p.fact;
    }
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
            (mm_powf(p.kuhn, -3.0f) * mm_powf(nmax, p.slope) * mm_expf((p.d - 2.0f) / (sq(nmax) + p.d))) * p.fact;
        const float val = (mm_powf(p.kuhn, -3.0f) * mm_powf(n, p.slope) * mm_expf((p.d - 2.0f) / (sq(n) + p.d))) * p.fact;
        result = val * norm_lin / norm_circ;
    }
    return fmaxf(result, p.v_inter);
}

// dynamic geometry of one fragment (16-byte record, rebuilt after every layout change)
struct Geo { int id_c, flags, start_bp, len_bp; }; // flags: bit0 ori==+1, bit1 circ, bit2 inactive, bit3 rep, bits 4..31 pos
__device__ __forceinline__ int geo_flags(int ori, int circ, int pos, int activ, int rep)
{
    return (ori == 1 ? 1 : 0) | (circ == 1 ? 2 : 0) | (activ == 1 ? 0 : 4) | (rep == 1 ? 8 : 0) | (pos << 4);
}
__device__ __forceinline__ int geo_pos(int flags) { return (int)((unsigned)flags >> 4); }
__device__ __forceinline__ bool geo_active(int flags) { return (flags & 4) == 0; }
// the rest of a fragment's layout record, 16 bytes: with Geo it replaces 13 scattered SoA words by two 16-byte loads
struct Link { int l_cont, l_cont_bp, prev, next; };
constexpr int N_MATES = 8; // mates[f][0..8): the fragments of f's contig in position order (-1 padded) when it holds <= 8 of them,
                           // else a row that starts with MATES_LONG
constexpr int MATES_LONG = -2;

// static data of one bin: sub-fragment lengths (kb) and RF counts (simulation_loader.py:673-704)
struct Stat { float l0, l1, l2; int n; int a0, a1, a2; int pad; };
// 3-way selects instead of indexed arrays: indexed private arrays would live in scratch memory
__device__ __forceinline__ float sel3(float x0, float x1, float x2, int i) { return i == 0 ? x0 : (i == 1 ? x1 : x2); }
__device__ __forceinline__ int sel3(int x0, int x1, int x2, int i) { return i == 0 ? x0 : (i == 1 ? x1 : x2); }
__device__ __forceinline__ float stat_len(const Stat& s, int i) { return sel3(s.l0, s.l1, s.l2, i); }
__device__ __forceinline__ int stat_accu(const Stat& s, int i) { return sel3(s.a0, s.a1, s.a2, i); }
__device__ __forceinline__ bool stat_uniform(const Stat& s) { return (s.n < 2 || s.a1 == s.a0) && (s.n < 3 || s.a2 == s.a0); } // equal RF counts

// centre (kb) of the sub-fragment stored in data slot `slot`, walking the bin in its orientation with
// the reference's float32 operation order (kernels3.cu:2997-3060)
__device__ __forceinline__ float centre_kb(int start_bp, bool fwd, const Stat& st, int slot)
{
    const int limit = st.n - 1;
    const int w = fwd ? slot : limit - slot; // walk index of that slot
    const float s0 = (float)start_bp / 1000.0f;
    const float l0 = stat_len(st, fwd ? 0 : limit);
    if (w == 0) return s0 + l0 / 2.0f;
    float run = s0 + l0;
    const float l1 = stat_len(st, fwd ? 1 : limit - 1);
    if (w == 1) return run + l1 / 2.0f;
    run = run + l1;
    const float l2 = stat_len(st, fwd ? 2 : limit - 2);
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

// the same with the contig length handed in (0 unless the contig is circular)
__device__ __forceinline__ End end_old(const Geo& g, int lbp)
{
    End e; e.label = g.id_c; e.start_bp = g.start_bp; e.fwd = g.flags & 1; e.circ = (g.flags >> 1) & 1; e.lbp = lbp;
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
    const float norm = (float)(stat_accu(sx, slx) * stat_accu(sy, sly)) / nfpb;
    if (X.label != Y.label) return p.v_inter * norm;
    const float s = fabsf(centre_kb(Y.start_bp, Y.fwd, sy, sly) - centre_kb(X.start_bp, X.fwd, sx, slx));
    if (X.circ == 1) return rippe_circ(s, (float)X.lbp / 1000.0f, p) * norm;
    return rippe(s, p) * norm;
}

// One term in Q.  A term that is not finite (ln of an overflowed / zero expected value) or does not fit (|v| >= 2^31; the
// running int64 sums hold +-8.6e9 log-likelihood units) gives Q_BAD: the caller flags the candidate (nf_flag) and the host
// reports NaN for it -- the reference's evaluate_likelihood_double would have produced -inf / NaN there (kernels3.cu:191-210).
constexpr long long Q_BAD = (long long)0x8000000000000000ull;
constexpr long long Q_NAN = Q_BAD;       // handed out instead of a flagged candidate sum: the EXACT value INT64_MIN (like the full likelihood's flag).  A
                                         // legitimate sum may be anything else -- a hopeless candidate's -8.6e8 log-likelihood units are -9.2e17 in Q30, beyond
                                         // the 2^58 that used to stand for NaN; ranks' values are summed with the marker sticky (eval_sync), and the device
                                         // buffer of the RCCL path carries the flags as counts of their own (hand_out), which an all-reduce can sum
// a candidate's value from its two sums (the host's conversion everywhere): NaN if flagged, else coarse + fine / 2^30
static inline double q_value(long long q, long long c)
{
    if (q == Q_NAN) return (double)NAN;
    return c == 0 ? (double)q / Q_SCALE : (double)c + (double)q / Q_SCALE;
}
constexpr int NF_OFF = 14;               // counters[NF_OFF .. NF_OFF + 2]: bit (k * 13 + op) = that candidate met a bad term
__device__ __forceinline__ long long to_q(double v) { return fabs(v) < 2147483648.0 ? __double2ll_rn(v * Q_SCALE) : Q_BAD; }
// the same value in fewer instructions for the pair loops (|v| < 2^30: v = hi + lo with hi = rint(v); hi 2^30 is an even integer, so rounding
// lo 2^30 to the nearest-even integer rounds the sum the same way; both parts fit int32)
__device__ __forceinline__ long long to_q_fast(double v)
{
    if (!(fabs(v) < 1073741824.0)) return to_q(v);
    const double hi = rint(v);
    const int lo = __double2int_rn((v - hi) * Q_SCALE);
    return (long long)__double2int_rn(hi) * (1ll << 30) + (long long)lo;
}
__device__ __forceinline__ void nf_flag(unsigned long long* nf, int k, int op)
{
    const int i = k * N_OPS + op;
    atomicOr(&nf[i >> 6], 1ull << (i & 63));
}
__device__ __forceinline__ void nf_flag_ops(unsigned long long* nf, int k, unsigned ops)
{
    while (ops) { nf_flag(nf, k, __ffs((int)ops) - 1); ops &= ops - 1; }
}

// A term that does not fit Q30 (|v| >= 2^31 log-likelihood units) but is finite: the reference adds it into its float64 sum like any
// other (kernels3.cu:191-210, 3703-3717) and reports a finite, hopeless score -- a circular contig closing over bp-sized fragments prices a
// pair at > 2e9 under C5's parameters.  Such terms are rounded to whole units (relative error < 2.4e-10) and summed in a SECOND int64
// per candidate (acc[MAXK * N_OPS + candidate], "coarse"), order independent like the first; the candidate's value is coarse + fine / 2^30.
// They are rare: added with one device-scope atomic each, where they turn up.  NaN / inf / |v| >= 2^62 still flag the candidate.
__device__ __forceinline__ bool to_coarse(double v, long long& c)
{
    if (!(fabs(v) < 4611686018427387904.0)) return false;
    c = __double2ll_rn(v);
    return true;
}
// candidates `ops` (13-bit mask) of neighbour k += v, coarse; or flag them
__device__ __forceinline__ void coarse_add_ops(long long* __restrict__ acc, unsigned long long* __restrict__ nf, int k, unsigned ops, double v)
{
    long long c;
    if (!to_coarse(v, c)) { nf_flag_ops(nf, k, ops); return; }
    while (ops) { const int b = __ffs((int)ops) - 1; ops &= ops - 1; atomicAdd((unsigned long long*)&acc[MAXK * N_OPS + k * N_OPS + b], (unsigned long long)c); }
}


// The same with the reference's own accu indexing in the TRANS branch (kernels3.cu:3155 / 3638): when the bin with the LOWER
// id of the pixel ("fi" = min, kernels3.cu:2884-2888 / 3364-3365) is reversed, every one of its slots is priced with the RF count
// of its LAST sub-fragment (list_accu_data_i[i] = accu_sub_fi[limit_fi] instead of [limit_fi - i]).  Only bins whose
// sub-fragments carry different RF counts see a difference.  `quirk` = GRAAL_MODE_REF_TRANS_ACCU.
__device__ __forceinline__ float ex_pair_ref(const End& X, const Stat& sx, int slx, int bin_x, const End& Y, const Stat& sy, int sly,
                                             int bin_y, float nfpb, const Par& p, bool quirk)
{
    if (X.label != Y.label) {
        int ax = stat_accu(sx, slx), ay = stat_accu(sy, sly);
        if (quirk) {
            if (bin_x < bin_y) { if (!X.fwd) ax = stat_accu(sx, sx.n - 1); }
            else if (!Y.fwd) ay = stat_accu(sy, sy.n - 1);
        }
        return p.v_inter * ((float)(ax * ay) / nfpb);
    }
    return ex_pair(X, sx, slx, Y, sy, sly, nfpb, p);
}

// Reference arithmetic, one fragment pair under one candidate class: sum over its slot pairs of (old value - new value), both
// priced from the float32 coordinates of their layout (kernels3.cu:3383-3697 against the stored evaluate_likelihood value),
// rounded to Q once per fragment pair.  X0 / Y0: the two fragments in the current layout, X / Y: under the candidate.
__device__ __forceinline__ long long strict_pair_q(const End& X0, const End& Y0, const End& X, const End& Y, const Stat& sx, int fx,
                                                   const Stat& sy, int fy, float nfpb, const Par& par, bool quirk, double& acc)
{
    acc = 0.0;
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++)
            if (a < sx.n && b < sy.n)
                acc += (double)ex_pair_ref(X0, sx, a, fx, Y0, sy, b, fy, nfpb, par, quirk) - (double)ex_pair_ref(X, sx, a, fx, Y, sy, b, fy, nfpb, par, quirk);
    return to_q(acc);
}

// "This thread's device-scope atomics have been performed."  They execute at the memory side, and on gfx9 the vector-memory
// counter also counts stores and atomics without return and is decremented when they are acknowledged: waiting for it is all
// a block needs before it takes its completion ticket.  __threadfence() -- a release fence at agent scope -- adds a write-back
// of the XCD's L2 (buffer_wbl2), which the atomics do not need and which takes a microsecond or more per block: the 768 blocks
// of k_fin doing it one after the other were a 15 us tail of that kernel (in-kernel stamps, C2 stand-in: last block at the
// ticket -> last block past it).
#define ATOMICS_DONE() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

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

// layout record of fragment f from its Geo + Link
__device__ __forceinline__ Rec rec_gl(const Geo& g, const Link& l, int f)
{
    Rec r;
    r.pos = geo_pos(g.flags); r.id_c = g.id_c; r.start_bp = g.start_bp; r.len_bp = g.len_bp; r.circ = (g.flags >> 1) & 1;
    r.prev = l.prev; r.next = l.next; r.l_cont = l.l_cont; r.l_cont_bp = l.l_cont_bp; r.ori = (g.flags & 1) ? 1 : -1;
    r.rep = (g.flags >> 3) & 1; r.activ = geo_active(g.flags) ? 1 : 0; r.id_d = f; // (id_d is not used by any mutation)
    return r;
}

// 16-byte streaming (nontemporal) load
__device__ __forceinline__ int4 ld_stream(const int4* p)
{
    typedef int v4i __attribute__((ext_vector_type(4)));
    const v4i v = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(p));
    return make_int4(v.x, v.y, v.z, v.w);
}

// Debug build (-DGRAAL_STAMPS): selected threads write the 100 MHz wall clock at a few points of the per-step kernels
#ifdef GRAAL_STAMPS
__device__ unsigned long long g_stamps[32];
__device__ unsigned long long g_s2eq[8];      // (GRAAL_S2_COUNTS: k_strict2's evaluations whose float32 inputs equal the current layout's, strict2.h)
__device__ unsigned long long g_hitstat[8];   // [0] launches' max third-stage duration (ticks) [1] latest third-stage end [2] count [3] sum of durations [4] max passes
#define STAMP(i, cond) do { if (cond) g_stamps[i] = wall_clock64(); } while (0)
#define STAMP_MAX(i, cond) do { if (cond) atomicMax(&g_stamps[i], wall_clock64()); } while (0)   // (the clock only grows: the latest wave of the latest step)
#define HITSTAT_BEGIN() const unsigned long long hs_t0 = wall_clock64()
#define HITSTAT_END() do { if (lane == 0 && !dry) { const unsigned long long hs_t1 = wall_clock64(); atomicMax(&g_hitstat[0], hs_t1 - hs_t0); atomicMax(&g_hitstat[1], hs_t1); atomicAdd(&g_hitstat[2], 1ull); atomicAdd(&g_hitstat[3], hs_t1 - hs_t0); } } while (0)
__device__ unsigned long long g_blk[4096 * 4]; // per block of k_scan: start, past prologue, loop done
#define STAMP_BLK(j, cond) do { if (cond) g_blk[4 * blockIdx.x + (j)] = wall_clock64(); } while (0)
#define STAMP_FBLK(j, cond) do { if ((cond) && blockIdx.x < 2048) g_blk[4 * (2048 + blockIdx.x) + (j)] = wall_clock64(); } while (0) // k_fin's blocks
#else
#define STAMP(i, cond) do { } while (0)
#define STAMP_MAX(i, cond) do { } while (0)
#define HITSTAT_BEGIN() do { } while (0)
#define HITSTAT_END() do { } while (0)
#define STAMP_BLK(j, cond) do { } while (0)
#define STAMP_FBLK(j, cond) do { } while (0)
#endif

// ------------------------------------------------------------------ small maintenance kernels
__global__ void k_refresh_geo(SoaPtr s, Geo* __restrict__ geo, Link* __restrict__ link, int n)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    Geo g; g.id_c = s.p[F_IDC][f]; g.start_bp = s.p[F_START][f]; g.len_bp = s.p[F_LEN][f];
    g.flags = geo_flags(s.p[F_ORI][f], s.p[F_CIRC][f], s.p[F_POS][f], s.p[F_ACTIV][f], s.p[F_REP][f]);
    geo[f] = g;
    Link l; l.l_cont = s.p[F_LCONT][f]; l.l_cont_bp = s.p[F_LCONTBP][f]; l.prev = s.p[F_PREV][f]; l.next = s.p[F_NEXT][f];
    link[f] = l;
}

// mates rows (see N_MATES); runs after the position index is complete
__global__ void k_mates(int n, const int* __restrict__ perm, const int* __restrict__ cbase, const Link* __restrict__ link,
                        int* __restrict__ mates, int* __restrict__ chg_words, int chg_n)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    // last kernel of every relabel: the record of the commit it consumed is cleared for the next commit (no memset launch)
    if (f < chg_n) chg_words[f] = 0;
    if (f >= n) return;
    const int base = cbase[f], lc = link[f].l_cont;
    int m[N_MATES];
#pragma unroll
    for (int i = 0; i < N_MATES; i++) m[i] = lc > N_MATES ? MATES_LONG : (i < lc ? perm[base + i] : -1);
    int4* out = reinterpret_cast<int4*>(mates + (size_t)f * N_MATES);
    out[0] = make_int4(m[0], m[1], m[2], m[3]);
    out[1] = make_int4(m[4], m[5], m[6], m[7]);
}

// out: [0] #heads (pos==0) [1] sum l_cont [2] #(start_bp==0) [3] sum l_cont_bp over start_bp==0 [4] max l_cont
//      [5] min l_cont [6] #(rep != 0 or activ != 1 or id_d != f)  [7] max label  [14] #(circ == 1)
constexpr int NC_WORD = 29;   // d_scalars[NC_WORD]: number of contigs of the ranked layout (= max_id + 1), kept on the device so that the
                              // scoring kernels can be launched before the host has read the statistics
constexpr int N_STAT = 9;
struct StatAcc { long long v[9]; }; // heads, sum l_cont, #start0, sum lbp, max, min, bad, max label, #circ
__device__ __forceinline__ StatAcc stat_zero() { StatAcc a = {{0, 0, 0, 0, 0, 0x7fffffff, 0, -1, 0}}; return a; }
__device__ __forceinline__ void stat_add(StatAcc& a, int f, int pos, int lc, int start_bp, int lbp, int rep, int activ, int id_d, int c, int circ)
{
    a.v[0] += pos == 0;
    a.v[1] += lc;
    if (start_bp == 0) { a.v[2] += 1; a.v[3] += lbp; }
    a.v[4] = lc > a.v[4] ? lc : a.v[4];
    a.v[5] = lc < a.v[5] ? lc : a.v[5];
    a.v[6] += (rep != 0) || (activ != 1) || (id_d != f);
    a.v[7] = c > a.v[7] ? c : a.v[7];
    a.v[8] += circ == 1;
}

// block reduction of the per-thread statistics (a multiple of 64 threads, <= 1024) + one atomic per statistic per block.  With `host` the LAST
// block (ticket in out[21]) also does k_stats_fin's job -- publish the 16 words to pinned host memory followed by the
// sequence number, re-arm the accumulators -- so that the host has the statistics while the kernels queued behind this one
// still run.  Every thread of the block must call it.
__device__ __forceinline__ void stat_reduce_publish(const StatAcc& a, long long* __restrict__ out, volatile long long* host, long long seq,
                                                    int reset_stale)
{
    __shared__ long long sh[16][9]; // up to 1024 threads
#pragma unroll
    for (int i = 0; i < 9; i++) {
        long long x = a.v[i];
        for (int o = 32; o > 0; o >>= 1) {
            const long long y = __shfl_down(x, o, 64);
            x = (i == 4 || i == 7) ? (y > x ? y : x) : (i == 5 ? (y < x ? y : x) : x + y);
        }
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][i] = x;
    }
    __syncthreads();
    if (threadIdx.x < 9) { // one atomic per statistic per block
        const int i = threadIdx.x;
        long long x = sh[0][i];
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) {
            const long long y = sh[w][i];
            x = (i == 4 || i == 7) ? (y > x ? y : x) : (i == 5 ? (y < x ? y : x) : x + y);
        }
        const int slot = i < 8 ? i : 14;
        if (i == 4 || i == 7) atomicMax((long long*)&out[slot], x);
        else if (i == 5) atomicMin((long long*)&out[slot], x);
        else atomicAdd((unsigned long long*)&out[slot], (unsigned long long)x);
    }
    if (host == nullptr) return;
    __shared__ int s_last;
    ATOMICS_DONE();
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd((unsigned long long*)&out[21], 1ull) == (unsigned long long)gridDim.x - 1ull;
    __syncthreads();
    if (!s_last) return;
    const int t = threadIdx.x;
    // (device-scope atomic reads: the other blocks' atomics were performed at the memory side, a plain load could hit a
    // stale line of this XCD's L2)
    if (t < 16) host[1 + t] = (long long)atomicAdd((unsigned long long*)&out[t], 0ull);
    if (t == 16) out[NC_WORD] = (long long)atomicAdd((unsigned long long*)&out[0], 0ull);
    __syncthreads();
    if (t < 8) out[t] = (t == 5) ? 0x7fffffffll : (t == 7 ? -1ll : 0ll);
    if (t == 13 && reset_stale) out[13] = 0;
    if (t == 14) out[14] = 0;
    if (t == 21) out[21] = 0;
    __threadfence_system();
    __syncthreads();
    if (t == 0) { host[0] = seq; __threadfence_system(); }
}

// the block's statistics as one row of partials (no atomics, no ticket): the commit kernel leaves them for the relabel kernel
// that follows it on the stream, whose first block combines the rows and publishes (k_incr)
__device__ __forceinline__ void stat_block_partials(const StatAcc& a, long long* __restrict__ part)
{
    __shared__ long long sh[16][N_STAT];
#pragma unroll
    for (int i = 0; i < N_STAT; i++) {
        long long x = a.v[i];
        for (int o = 32; o > 0; o >>= 1) {
            const long long y = __shfl_down(x, o, 64);
            x = (i == 4 || i == 7) ? (y > x ? y : x) : (i == 5 ? (y < x ? y : x) : x + y);
        }
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][i] = x;
    }
    __syncthreads();
    if (threadIdx.x < N_STAT) {
        const int i = threadIdx.x;
        long long x = sh[0][i];
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) {
            const long long y = sh[w][i];
            x = (i == 4 || i == 7) ? (y > x ? y : x) : (i == 5 ? (y < x ? y : x) : x + y);
        }
        part[(size_t)blockIdx.x * N_STAT + i] = x;
    }
}

__global__ __launch_bounds__(256) void k_stats(SoaPtr s, int n, long long* __restrict__ out, volatile long long* host = nullptr,
                                               long long seq = 0, int reset_stale = 0)
{
    StatAcc a = stat_zero();
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < n; f += gridDim.x * blockDim.x)
        stat_add(a, f, s.p[F_POS][f], s.p[F_LCONT][f], s.p[F_START][f], s.p[F_LCONTBP][f], s.p[F_REP][f], s.p[F_ACTIV][f],
                 s.p[F_IDD][f], s.p[F_IDC][f], s.p[F_CIRC][f]);
    stat_reduce_publish(a, out, host, seq, reset_stale);
}

// last kernel of graal_begin_step / graal_layout_stats: publishes the 16 statistics words to pinned host memory (followed by
// a sequence number the host spins on: no device->host copy, no stream synchronise) and re-arms the accumulators for the
// next k_stats (no host->device copy of initial values either)
__global__ void k_stats_fin(long long* __restrict__ sc, volatile long long* host, long long seq, int reset_stale)
{
    const int t = threadIdx.x;
    if (t < 16) host[1 + t] = sc[t];
    if (t == 16) sc[NC_WORD] = sc[0];
    __syncthreads();
    if (t < 8) sc[t] = (t == 5) ? 0x7fffffffll : (t == 7 ? -1ll : 0ll);
    if (t == 13 && reset_stale) sc[13] = 0;
    if (t == 14) sc[14] = 0;
    __threadfence_system();
    __syncthreads();
    if (t == 0) { host[0] = seq; __threadfence_system(); }
}

// exchange self-test: this rank's GPU writes a tag into the spare last word of its two slots (graal_exchange_selftest)
__global__ void k_exchange_tag(volatile long long* slot0, volatile long long* slot1, int word, long long tag)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) { slot0[word] = tag; slot1[word] = tag; __threadfence_system(); }
}

// Genome distance of dist_inter_genome (cuda_lib_gl.py:475-541) in HALF units: the reference subtracts, per fragment, terms
// that are multiples of 0.5 from 3 * (fragments counted); this kernel returns twice the sum of those terms (an integer, so
// the host's float64 result is exactly the reference loop's).  ref[f] = (initial prev, initial next, initial ori,
// bit 0 orientable | bit 1 counted).  Like the reference, bin ids (id_d) index the per-fragment arrays.
__global__ __launch_bounds__(256) void k_dist(SoaPtr s, int n, const int4* __restrict__ ref, unsigned long long* __restrict__ acc,
                                              volatile long long* host, long long seq)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    int h = 0;
    if (f < n) {
        const int4 r = ref[f];
        if (r.w & 2) {
            const int* __restrict__ id_d = s.p[F_IDD];
            const int* __restrict__ ori = s.p[F_ORI];
            const int pv = s.p[F_PREV][f], nx = s.p[F_NEXT][f];
            const int prev_t1 = pv != -1 ? id_d[pv] : -1, next_t1 = nx != -1 ? id_d[nx] : -1;
            const int p0 = r.x, n0 = r.y;
            if ((prev_t1 == p0 && next_t1 == n0) || (prev_t1 == n0 && next_t1 == p0)) h += 2;
            if (r.w & 1) {
                const bool flipped = r.z != ori[f];
                const int p1 = flipped ? next_t1 : prev_t1, n1 = flipped ? prev_t1 : next_t1, swap = flipped ? -1 : 1;
                for (int side = 0; side < 2; side++) {
                    const int t0 = side ? n0 : p0, t1 = side ? n1 : p1;
                    if (t0 != t1) continue;
                    if (t0 == -1) { h += 2; continue; }
                    const int idx = min(max(t1, 0), n - 1);
                    if ((ref[idx].w & 1) == 0) { h += 2; continue; }
                    h += 1 + (ref[min(max(t0, 0), n - 1)].z == swap * ori[idx] ? 1 : 0);
                }
            } else {
                h += (prev_t1 == p0 || prev_t1 == n0) ? 2 : 0;
                h += (next_t1 == n0 || next_t1 == p0) ? 2 : 0;
            }
        }
    }
    __shared__ int s_h[4];
    const long long w = wave_sum_ll((long long)h);
    if ((threadIdx.x & 63) == 0) s_h[threadIdx.x >> 6] = (int)w;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&acc[0], (unsigned long long)(s_h[0] + s_h[1] + s_h[2] + s_h[3]));
        ATOMICS_DONE();
        if (atomicAdd(&acc[1], 1ull) == (unsigned long long)gridDim.x - 1ull) { // last block: publish to pinned host memory
            __threadfence();
            host[1] = (long long)atomicExch(&acc[0], 0ull);
            acc[1] = 0;
            __threadfence_system();
            host[0] = seq;
            __threadfence_system();
        }
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

// the number of contigs is read from the statistics block on the device (k_stats ran earlier on the stream): no host
// round trip in the middle of the relabel
__global__ void k_relabel_o2n(const unsigned long long* __restrict__ sorted, int n, const long long* __restrict__ stats,
                              int* __restrict__ o2n, int* __restrict__ len_of)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int n_contigs = (int)stats[0];
    if (i >= n_contigs) { len_of[i] = 0; if (i == n - 1) len_of[n] = 0; return; }
    if (i == n - 1) len_of[n] = 0;
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

__global__ void k_build_perm(SoaPtr s, int n, const int* __restrict__ contig_off, int* __restrict__ perm, int* __restrict__ cbase,
                             int* __restrict__ pstart)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    const int base = contig_off[s.p[F_IDC][f]];
    cbase[f] = base; // first slot of the fragment's contig in the position index
    perm[base + s.p[F_POS][f]] = f;
    pstart[base + s.p[F_POS][f]] = s.p[F_START][f];   // start_bp in position order: a tile's extent in one round trip (k_gprep)
}

// What one commit did to the contig set: only contig(fA), contig(fB) and up to two fresh labels can change.
struct Changed {
    int cA, cB;        // labels (= ranks) of the two touched contigs before the commit
    int lab[4];        // candidate labels after the commit: cA, cB, max_id + 1, max_id + 2
    int len[4];        // l_cont of the contig carrying that label after the commit
    int exists[4];
};

// A bin's OWN pixel -- its sub-fragment pairs (a < b) -- as the full evaluation prices it (evaluate_likelihood's on_diag pixels,
// kernels3.cu:3213): the contacts between two of its sub-fragments, each rounded to Q like k_full_nnz's terms, minus the expected mass of
// its pairs, rounded once like k_full_mass's.  No candidate delta of the reference contains it (sub_compute_likelihood's pixel set has no
// diagonal pixels, kernels3.cu:3356-3380), yet its float32 value moves with the bin's coordinates and with its contig's circular model: it is
// what separates a candidate's delta from full(after) - full(before) when bins hold several sub-fragments (RF counts uniform or the
// trans-branch indexing off: tests/test_carried_total_gpu.py).  own[3] = the observed counts of the pairs (0,1), (0,2), (1,2), 0 = none.
// One lane's part of it: sub-fragment pair `pair` ((0,1), (0,2), (1,2)) of the bin in the layout record r -- its expected value and, rounded to Q, its
// observed contacts' term (0 without contacts); `valid` = the bin has that pair.
__device__ __forceinline__ void own_pixel_pair(const Rec& r, const Stat& st, const float* __restrict__ own, int pair, float nfpb, const Par& par,
                                               bool valid, float& ex, long long& t, bool& bad)
{
    ex = 0.0f; t = 0;
    if (!valid) return;
    End X; X.label = 0; X.start_bp = r.start_bp; X.fwd = r.ori == 1; X.circ = r.circ; X.lbp = r.l_cont_bp;
    const int a = pair == 2 ? 1 : 0, b = pair == 0 ? 1 : 2;
    ex = ex_pair(X, st, a, X, st, b, nfpb, par);
    const float ob = own[pair];
    if (ob != 0.0f) { t = to_q((double)ob * mm_ln(ex)); if (t == Q_BAD) { bad = true; t = 0; } }
}

// commit one candidate (test_copy_struct, cuda_lib_gl.py:1156): out = apply_move(in); also records which contigs exist
// afterwards among the <= 4 labels the move can touch (for the incremental relabel of the next graal_begin_step), and -- it
// has every new record in registers anyway -- the statistics of the NEW layout, which its last block publishes to pinned host
// memory: the next graal_begin_step finds them there and needs no statistics kernel.
__global__ __launch_bounds__(256) void k_apply(SoaPtr in, SoaPtr out, int n, int op, int fA, int fB, int max_id, int* __restrict__ n_stale,
                                               Changed* __restrict__ chg, long long* __restrict__ part)
{
    STAMP(7, blockIdx.x == 0 && threadIdx.x == 0);
    StatAcc a = stat_zero();
    const Rec A0 = ld_rec(in, fA), B0 = ld_rec(in, fB);
    const Move m = make_move(op, fA, fB, max_id, A0, B0);
    // (grid-stride with a small grid: the statistics cost 10 same-address atomics per BLOCK.  Measured on 50k fragments:
    // 64 x 256 threads 15 us; 49 blocks x 1024 threads ~28 us; 196 x 256 ~25 us)
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < n; f += gridDim.x * blockDim.x) {
        bool stale;
        const Rec r0 = ld_rec(in, f);
        const Rec r = apply_move(m, f, r0, &stale);
        st_rec(out, f, r);
        if (stale) atomicAdd(n_stale, 1);
        if (f == 0) { chg->cA = A0.id_c; chg->cB = B0.id_c; chg->lab[0] = A0.id_c; chg->lab[1] = B0.id_c; chg->lab[2] = max_id + 1; chg->lab[3] = max_id + 2; }
        if (r.pos == 0) {
            const int c = r.id_c;
            const int j = c == A0.id_c ? 0 : (c == B0.id_c ? 1 : (c == max_id + 1 ? 2 : (c == max_id + 2 ? 3 : -1)));
            if (j >= 0) { chg->len[j] = r.l_cont; chg->exists[j] = 1; }
        }
        stat_add(a, f, r.pos, r.l_cont, r.start_bp, r.l_cont_bp, r.rep, r.activ, r.id_d, r.id_c, r.circ);
    }
    // (round 2 reduced and published here: 10 of this kernel's 16 us were that atomics -> ticket -> publish chain, in front of
    // the relabel and the whole next step; now the rows go to the relabel kernel, whose block 0 publishes them next to its own work)
    stat_block_partials(a, part);
    STAMP(15, blockIdx.x == gridDim.x - 1 && threadIdx.x == 0);
}

// The commit's OWN-PIXEL CORRECTION (several sub-fragments per bin: graal_take_carry_correction), next to the commit on a stream of its own:
// nobody needs it before the NEXT step forms its scores, and inside the commit kernel it was 7 us of every step's critical path (the
// sub-fragment centres, the contact model, a float64 logarithm: 13 us per commit at the C2 shape against 6 for the plain commit).  It reads the
// layout the commit reads and derives every fragment's new record itself (apply_move: registers only).  Eight lanes per fragment: lanes 0-2 price
// the bin's three sub-fragment pairs in the NEW layout, lanes 3-5 in the OLD one; the mass of a side is the float64 sum of its pairs' expected
// values in the full evaluation's order (0,1), (0,2), (1,2), rounded to Q once (k_full_mass's own-pair term), gathered by shuffles.  A bin's pixel
// = its pairs' observed terms - that mass; the correction = new - old.  Block sums by atomics, the last block (a ticket) publishes
// {sum, unknown terms} and the commit's number to pinned host memory and re-arms the words.
__global__ __launch_bounds__(256) void k_own_corr(SoaPtr in, int n, int op, int fA, int fB, int max_id, const Stat* __restrict__ stat,
                                                  const float* __restrict__ own_obs, float nfpb, Par par, int quirk /* GRAAL_MODE_REF_TRANS_ACCU, mixed RF counts exist */,
                                                  long long* __restrict__ acc /* [0] sum, [1] unknown, [2] ticket */, volatile long long* host, long long seq)
{
    const Rec A0 = ld_rec(in, fA), B0 = ld_rec(in, fB);
    const Move m = make_move(op, fA, fB, max_id, A0, B0);
    const int lane = threadIdx.x & 63, sub = lane & 7, oct = lane & ~7;
    const int per_pass = (int)((gridDim.x * blockDim.x) >> 3);
    long long my_q = 0, my_bad = 0;
    // (whole octets run the loop together: the shuffles below need all eight lanes of a fragment, and the bound is on the octet's fragment alone)
    for (int f = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 3); f < n; f += per_pass) {
        bool stale;
        const Rec r0 = ld_rec(in, f);
        const Rec r = apply_move(m, f, r0, &stale);
        const bool changed = r.start_bp != r0.start_bp || r.ori != r0.ori || r.circ != r0.circ || (r.circ == 1 && r.l_cont_bp != r0.l_cont_bp);
        if (!changed) continue;
        const Stat st = stat[f];
        if (st.n < 2) continue;                                       // (octet-uniform, like `changed`)
        // (the trans-branch indexing prices a reversed bin with its LAST RF count, kernels3.cu:3155: mirroring a bin whose sub-fragments
        // carry different counts also changes its trans pixels with every bin OUTSIDE the two contigs -- in no delta, not an own pixel:
        // this commit's correction is unknown and the step that follows evaluates the layout in full)
        bool bad = quirk && r.ori != r0.ori && !stat_uniform(st);
        const int pair = sub % 3, side = sub / 3;                     // side 0: the new record, 1: the old one; lanes 6, 7: nothing
        const bool valid = sub < 6 && (pair == 0 || st.n > 2);
        float ex; long long t;
        own_pixel_pair(side == 0 ? r : r0, st, own_obs + 3 * (size_t)r.id_d, pair, nfpb, par, valid, ex, t, bad);
        const int base = oct + 3 * (sub < 3 ? 0 : 1);
        const float e0 = __shfl(ex, base, 64), e1 = __shfl(ex, base + 1, 64), e2 = __shfl(ex, base + 2, 64);
        long long c = sub >= 6 ? 0ll : (side == 0 ? t : -t);
        if (sub == 0 || sub == 3) {                                   // the side's mass, by its first lane
            double mass = 0.0;
            mass += (double)e0;
            if (st.n > 2) { mass += (double)e1; mass += (double)e2; }
            const long long mq = to_q(mass);
            if (mq == Q_BAD) bad = true; else c += sub == 0 ? -mq : mq;
        }
        for (int o = 1; o < 8; o <<= 1) c += __shfl_xor(c, o, 64);
        const unsigned any_bad = (unsigned)((__ballot(bad) >> oct) & 0xffull);
        if (sub == 0) { if (any_bad) my_bad += 1; else my_q += c; }
    }
    const long long wq = wave_sum_ll(my_q), wb = wave_sum_ll(my_bad);
    if (lane == 0) {
        if (wq != 0) atomicAdd((unsigned long long*)&acc[0], (unsigned long long)wq);
        if (wb != 0) atomicAdd((unsigned long long*)&acc[1], (unsigned long long)wb);
    }
    ATOMICS_DONE();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long ticket = atomicAdd((unsigned long long*)&acc[2], 1ull);
        if (ticket == (unsigned long long)gridDim.x - 1ull) {
            // (device-scope atomic reads: the other blocks' atomics were performed at the memory side)
            host[1] = (long long)atomicExch((unsigned long long*)&acc[0], 0ull);
            host[2] = (long long)atomicExch((unsigned long long*)&acc[1], 0ull);
            atomicExch((unsigned long long*)&acc[2], 0ull);
            __threadfence_system();
            host[0] = seq;
            __threadfence_system();
        }
    }
}

// ---- incremental relabel.  Invariant after every graal_begin_step: labels ARE ranks (contigs sorted by (l_cont, label)),
// len_of[rank] is sorted and contig_off is its exclusive prefix sum.  One commit changes <= 4 contigs, so the new stable
// ranking follows from the old one by counting: rank' = rank - #removed before + #inserted before.  Same result as the
// full sort (cuda_lib_gl.py:1697-1722 with a stable argsort), without sorting 50k keys per step.
// The statistics of the committed layout: the commit kernel's rows of partials combined and published to pinned host memory,
// followed by the sequence number; one wave (lane = rows u, u + 64, ...: nine loads in flight per row, then wave reductions).
__device__ __forceinline__ void publish_partials(const long long* __restrict__ part, int n_part, long long* __restrict__ stats,
                                                 volatile long long* host, long long seq, int u)
{
    if (host != nullptr) {
        long long v[N_STAT];
#pragma unroll
        for (int i = 0; i < N_STAT; i++) v[i] = (i == 5 ? 0x7fffffffll : (i == 7 ? -1ll : 0ll));
        for (int r = u; r < n_part; r += 64) {
#pragma unroll
            for (int i = 0; i < N_STAT; i++) {
                const long long y = part[(size_t)r * N_STAT + i];
                v[i] = (i == 4 || i == 7) ? (y > v[i] ? y : v[i]) : (i == 5 ? (y < v[i] ? y : v[i]) : v[i] + y);
            }
        }
        const long long stale = u == 0 ? stats[13] : 0;               // fragments that hit the unwritten paste branch
#pragma unroll
        for (int i = 0; i < N_STAT; i++) {
            long long x = v[i];
            for (int o = 32; o > 0; o >>= 1) {
                const long long y = __shfl_down(x, o, 64);
                x = (i == 4 || i == 7) ? (y > x ? y : x) : (i == 5 ? (y < x ? y : x) : x + y);
            }
            if (u == 0) host[1 + (i < 8 ? i : 14)] = x;
        }
        if (u == 0) host[1 + 13] = stale;
        // (words 9..12 of the host block are not statistics: nobody reads them)
        __threadfence_system();
    }
    if (u == 0) stats[13] = 0;   // (re-armed for the next commit)
    if (host != nullptr && u == 0) { host[0] = seq; __threadfence_system(); }
}

struct IncrPlan {
    int n_removed, removed[2], removed_len[2];   // old ranks of contig(fA), contig(fB) and their old lengths
    int n_new, new_lab[4], new_len[4], new_rank[4], new_off[4];
    int nc_new;
};

__device__ __forceinline__ bool key_less(int l1, int c1, int l2, int c2) { return l1 < l2 || (l1 == l2 && c1 < c2); }

// One kernel: every block derives the plan (a dozen loads; the searches of the <= 4 new contigs in the sorted length array
// run side by side, 32 probes per round), then the elementwise pass: new label of every fragment, sorted-length / offset
// arrays (contig heads write them), position index, geometry records and the mates rows of the touched contigs.  Also
// re-arms the stale-paste counter the commit's statistics reported and clears the commit record the NEXT commit fills.
__global__ __launch_bounds__(256) void k_incr(SoaPtr s, int n, const Changed* __restrict__ chg, const int* __restrict__ len_old,
                                              const int* __restrict__ off_old, int nc_old, int* __restrict__ len_new,
                                              int* __restrict__ off_new, int* __restrict__ perm, int* __restrict__ pstart, int* __restrict__ cbase,
                                              Geo* __restrict__ geo, Link* __restrict__ link, long long* __restrict__ stats,
                                              int* __restrict__ mates, int* __restrict__ chg_clear, int chg_n,
                                              const long long* __restrict__ part, int n_part, volatile long long* host, long long seq,
                                              volatile long long* guard /* pinned host, 8 words: an index out of range (never yet seen to be this kernel's) */)
{
    __shared__ IncrPlan sp;
    __shared__ int s_lb[4], s_ub[4];
    const int t = threadIdx.x;
    // the LAST block (the grid is one block larger than the fragments need) has one job: the statistics of this layout
    // (publish_partials) -- next to the other blocks, not in front of them: the two system-scope fences are microseconds
    STAMP(22, blockIdx.x == 0 && t == 0);
    if (blockIdx.x == gridDim.x - 1) {
        if (t < 64 && part != nullptr) publish_partials(part, n_part, stats, host, seq, t);   // (nullptr: k_tm's extra block publishes)
        STAMP(31, t == 0);
        return;
    }
#if defined(GRAAL_STAMPS) && defined(GRAAL_EXP_INCR_CHECK)
    // (diagnostics, tools/incr_check.py: is the committed layout complete when this kernel STARTS?  The fragment's label and position
    // are read here and again behind the plan; g_stamps[27] counts the fragments for which the two reads differ)
    const int f_early = blockIdx.x * blockDim.x + t;
    const int c_early = f_early < n ? __hip_atomic_load(&s.p[F_IDC][f_early], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    const int pos_early = f_early < n ? __hip_atomic_load(&s.p[F_POS][f_early], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
#endif
    if (t == 0) {
        // (built in LDS: as a private struct its dynamically indexed arrays lived in scratch memory -- two dozen dependent scratch accesses in
        // front of every block of this kernel)
        IncrPlan& p = sp;
        const int cA = chg->cA, cB = chg->cB;
        const int lA = len_old[cA], lB = len_old[cB];
        int ex[4], lab[4], ln[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { ex[j] = chg->exists[j]; lab[j] = chg->lab[j]; ln[j] = chg->len[j]; }
        p.removed[0] = cA; p.removed_len[0] = lA;
        const int n_removed = cB == cA ? 1 : 2;
        p.removed[1] = cB == cA ? -1 : cB; p.removed_len[1] = cB == cA ? 0 : lB;
        p.n_removed = n_removed;
        int n_new = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (!ex[j]) continue;
            if (j == 1 && lab[1] == lab[0]) continue;
            p.new_lab[n_new] = lab[j]; p.new_len[n_new] = ln[j]; n_new++;
        }
        for (int i = n_new; i < 4; i++) { p.new_lab[i] = -1; p.new_len[i] = 0; }
#pragma unroll
        for (int i = 0; i < 4; i++) { p.new_rank[i] = -1; p.new_off[i] = 0; }
        p.n_new = n_new;
        p.nc_new = nc_old - n_removed + n_new;
        if (blockIdx.x == 0) stats[NC_WORD] = nc_old - n_removed + n_new;
    }
    __syncthreads();
    {   // old contigs with key < (l, c): lower / upper bound of l in the sorted length array, for each of the <= 4 new contigs.
        // Eight searches side by side, 32 lanes each (the two halves of a wave share the contig): 32 probes per round, so
        // 50k contigs take 4 rounds of one load instead of 16 dependent ones.
        const int g = t >> 5, j = t & 31, i = g >> 1;
        const bool upper = g & 1, act = i < sp.n_new;
        const int l = act ? sp.new_len[i] : 0;
        int lo = 0, hi = nc_old;
        for (;;) {
            const bool go = act && lo < hi;
            if (__ballot(go) == 0) break;
            const int span = hi - lo;
            const int m = lo + (int)(((long long)(j + 1) * span) / 33);
            const int v = go ? len_old[m] : 0;
            const bool pred = go && (upper ? v <= l : v < l);
            const unsigned long long bal = __ballot(pred);
            const int cnt = __popc((unsigned)(bal >> (32 * (g & 1))));   // probes of this search that lie below the bound
            if (go) {
                const int nlo = cnt > 0 ? lo + (int)(((long long)cnt * span) / 33) + 1 : lo;
                const int nhi = cnt < 32 ? lo + (int)(((long long)(cnt + 1) * span) / 33) : hi;
                lo = nlo; hi = nhi;
            }
        }
        if (act && j == 0) { if (upper) s_ub[i] = lo; else s_lb[i] = lo; }
    }
    __syncthreads();
    if (t < sp.n_new) {
        const int i = t, l = sp.new_len[i], c = sp.new_lab[i], lb = s_lb[i], ub = s_ub[i];
        int within = c - lb; within = within < 0 ? 0 : (within > ub - lb ? ub - lb : within);
        const int pcount = lb + within;                    // number of old contigs with a smaller key
        int rank = pcount, off = pcount < nc_old ? off_old[pcount] : n;
        for (int r = 0; r < sp.n_removed; r++)
            if (key_less(sp.removed_len[r], sp.removed[r], l, c)) { rank -= 1; off -= sp.removed_len[r]; }
        for (int j = 0; j < sp.n_new; j++)
            if (j != i && key_less(sp.new_len[j], sp.new_lab[j], l, c)) { rank += 1; off += sp.new_len[j]; }
        sp.new_rank[i] = rank; sp.new_off[i] = off;
    }
    __syncthreads();
    const int f = blockIdx.x * blockDim.x + t;
    const IncrPlan p = sp;
    if (f < chg_n) chg_clear[f] = 0; // the OTHER commit record (consumed by the previous relabel): clear for the next commit
    if (f == 0) off_new[p.nc_new] = n;
    if (f >= p.nc_new && f < n) len_new[f] = 0; // keep the tail of the length array zero
    if (f >= n) return;
    const int c = s.p[F_IDC][f];
#if defined(GRAAL_STAMPS) && defined(GRAAL_EXP_INCR_CHECK)
    if (c != c_early || s.p[F_POS][f] != pos_early) atomicAdd(&g_stamps[27], 1ull);
#endif
    int rank = -1, off = 0, lenc = 0;
    for (int i = 0; i < 4; i++)
        if (i < p.n_new && c == p.new_lab[i]) { rank = p.new_rank[i]; off = p.new_off[i]; lenc = p.new_len[i]; }
    const bool touched = rank >= 0;
    if (rank < 0) { // untouched contig: c is its old rank
        lenc = len_old[c]; rank = c; off = off_old[c];
        for (int r = 0; r < 2; r++)
            if (r < p.n_removed && p.removed[r] < c) { rank -= 1; off -= p.removed_len[r]; }
        for (int i = 0; i < 4; i++)
            if (i < p.n_new && key_less(p.new_len[i], p.new_lab[i], lenc, c)) { rank += 1; off += p.new_len[i]; }
    }
    const int pos = s.p[F_POS][f];
    if ((unsigned)(off + pos) >= (unsigned)n || (unsigned)rank >= (unsigned)n) {
        // GUARD (the open fault of the two-rank rehearsal: an access 48+ words behind the position index's twin): an
        // index this relabel would write out of range is recorded for the host -- graal_begin_step fails with it -- and not written
        // (the first offender of the handle's life with its details; every later one still trips the word the host looks at)
        if (atomicAdd((unsigned long long*)&stats[31], 1ull) == 0ull) {
            guard[1] = f; guard[2] = c; guard[3] = rank; guard[4] = off; guard[5] = pos; guard[6] = lenc; guard[7] = ((long long)p.n_new << 32) | (unsigned)p.nc_new;
        }
        __threadfence_system();
        guard[0] = seq != 0 ? seq : 1;
        return;
    }
    s.p[F_IDC][f] = rank;
    if (pos == 0) { len_new[rank] = lenc; off_new[rank] = off; }
    perm[off + pos] = f;
    cbase[f] = off;
    Geo g; g.id_c = rank; g.start_bp = s.p[F_START][f]; g.len_bp = s.p[F_LEN][f];
    pstart[off + pos] = g.start_bp;
    g.flags = geo_flags(s.p[F_ORI][f], s.p[F_CIRC][f], pos, s.p[F_ACTIV][f], s.p[F_REP][f]);
    geo[f] = g;
    Link l; l.l_cont = s.p[F_LCONT][f]; l.l_cont_bp = s.p[F_LCONTBP][f]; l.prev = s.p[F_PREV][f]; l.next = s.p[F_NEXT][f];
    link[f] = l;
    // mates row (first N_MATES fragments of the fragment's contig, read only for contigs that short): members and order of
    // an untouched contig did not change, so only the fragments of the <= 4 touched contigs rewrite theirs -- by walking
    // the links of the committed layout (complete before this kernel started; the position index is being written by it)
    if (touched && lenc <= N_MATES) {
        int head = f;
        for (int i = 0; i < pos && i < N_MATES; i++) { const int h2 = s.p[F_PREV][head]; if ((unsigned)h2 >= (unsigned)n) break; head = h2; }
        int m[N_MATES], cur = head;
#pragma unroll
        for (int i = 0; i < N_MATES; i++) {
            m[i] = (i < lenc && cur >= 0) ? cur : -1;
            if (i + 1 < lenc && cur >= 0) { const int nx = s.p[F_NEXT][cur]; cur = (unsigned)nx < (unsigned)n ? nx : -1; }
        }
        int4* out = reinterpret_cast<int4*>(mates + (size_t)f * N_MATES);
        out[0] = make_int4(m[0], m[1], m[2], m[3]);
        out[1] = make_int4(m[4], m[5], m[6], m[7]);
    } else if (touched) { // a longer contig: marker row (k_scan then walks the position index instead)
        int4* out = reinterpret_cast<int4*>(mates + (size_t)f * N_MATES);
        out[0] = make_int4(MATES_LONG, MATES_LONG, MATES_LONG, MATES_LONG);
        out[1] = out[0];
    }
    STAMP(23, blockIdx.x == gridDim.x - 2 && t == 0);
}

// ------------------------------------------------------------------ full likelihood
// contacts part: sum ob * log(ex) in Q (the log-factorial constant is added on the host)  [evaluate_likelihood,
// kernels3.cu:2802-3222, as called by cuda_lib_gl.py:1986-2017 -- restricted to the pixels that hold a contact]
//
// Everything a contact needs of one of its two sub-fragments fits 16 bytes: contig label, centre coordinate (kb, float32,
// computed by centre_kb exactly as the pricing kernels do), RF count, and the contig length when the contig is circular.
// k_subrec rebuilds that table from the geometry records before every full evaluation (50k fragments: a few microseconds),
// and the contact kernel below is a pure stream: three 16-byte nontemporal loads per group of 4 contacts (row, col, count),
// two groups in flight per lane, two 16-byte gathers per contact from a table that stays in L2.  Round 1's kernel loaded
// the three words of ONE contact per lane and iteration and gathered a 16-byte + a 32-byte record per end: 234 us for
// 20 M contacts (13 % of the 240 MB / 8 TB/s bound).
struct SubRec8 { int label; float centre; };   // compact form for uniform RF counts (k_full_nnz_u): label | circular << 31
struct SubRec {
    int label; float centre;
    int accu;   // RF count | RF count under the reference's trans-branch indexing (ex_pair_ref) << 16
    int bin;    // fragment (= bin) id | contig is circular << 31
};

__global__ __launch_bounds__(256) void k_subrec(int n, const Geo* __restrict__ geo, const Stat* __restrict__ stat,
                                                 const int* __restrict__ sub_ids /* nullptr: id = f */, SubRec* __restrict__ rec,
                                                 SubRec8* __restrict__ rec8 /* nullptr: RF counts are not uniform */,
                                                 unsigned short* __restrict__ lab16 /* with rec8: low 16 bits of the label */)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    const Stat st = stat[f];
    if (st.n == 0) return; // a copy of a repeated bin: no sub-fragments in the sparse path
    const Geo g = geo[f];
    const bool fwd = g.flags & 1;
    int4 ids = make_int4(f, 0, 0, 1);
    if (sub_ids) ids = reinterpret_cast<const int4*>(sub_ids)[f];
    for (int slot = 0; slot < st.n; slot++) {
        SubRec r; r.label = g.id_c; r.centre = centre_kb(g.start_bp, fwd, st, slot);
        r.accu = stat_accu(st, slot) | ((fwd ? stat_accu(st, slot) : stat_accu(st, st.n - 1)) << 16);   // (RF counts are <= 30000)
        r.bin = f | (((g.flags >> 1) & 1) << 31);
        rec[sel3(ids.x, ids.y, ids.z, slot)] = r;
        if (rec8) {
            SubRec8 r8; r8.label = g.id_c | (((g.flags >> 1) & 1) << 31); r8.centre = r.centre; rec8[sel3(ids.x, ids.y, ids.z, slot)] = r8;
            lab16[sel3(ids.x, ids.y, ids.z, slot)] = (unsigned short)(g.id_c & 0xffff);
        }
    }
}

// A contact between two contigs has the expected value v_inter * (float(a_x * a_y) / nfpb): it depends on the product of the
// two RF counts only, so every block first tabulates its ln (same expressions, same device functions as the general path:
// bit-identical) for the products that can occur, and such contacts skip the float64 logarithm.
constexpr int LN_TRANS_LUT = 1024;
// FULL_G = groups of 4 contacts per lane and iteration (template parameter; GRAAL_FULL_G picks 1, 2 or 4 for experiments)
__device__ __forceinline__ int w4(const int4& q, int j) { return j == 0 ? q.x : (j == 1 ? q.y : (j == 2 ? q.z : q.w)); }

// The same when every sub-fragment carries the SAME RF count (level 0: one restriction fragment per bin; any level of a map whose
// bins are complete): the record shrinks to 8 bytes {label | circular << 31, centre} -- twice as many records per cache line for
// the column-side gather, which is what bounds this kernel (random 16-byte reads of an L2-resident table pull a whole line into
// the CU's L1 each) -- and the trans logarithm is one constant.  Bit-identical sums (same expressions on the same values).
template <int FULL_G>
__global__ __launch_bounds__(256) void k_full_nnz_u(const int4* __restrict__ row4, const int4* __restrict__ col4,
                                                     const int4* __restrict__ cnt4, long long nnz, const SubRec8* __restrict__ rec,
                                                     const SubRec* __restrict__ rec_full /* circular contigs only */,
                                                     const int* __restrict__ lcontbp, float nfpb, Par par, int accu,
                                                     long long* __restrict__ out, long long* __restrict__ bad_flag)
{
    const float norm = (float)(accu * accu) / nfpb;
    const double ln_trans = mm_ln(par.v_inter * norm);
    long long acc = 0;
    bool bad = false;
    const int n4 = (int)(nnz >> 2);
    const int stride = (int)(gridDim.x * blockDim.x);
    for (int g0 = (int)(blockIdx.x * blockDim.x + threadIdx.x); g0 <= n4; g0 += FULL_G * stride) {
        int4 r[FULL_G], c[FULL_G], w[FULL_G];
#pragma unroll
        for (int i = 0; i < FULL_G; i++) {
            const int g = g0 + i * stride;
            const int gc = g <= n4 ? g : n4;
            r[i] = ld_stream(row4 + gc); c[i] = ld_stream(col4 + gc); w[i] = ld_stream(cnt4 + gc);
        }
        SubRec8 a[4 * FULL_G], b[4 * FULL_G];
        bool valid[4 * FULL_G];
#pragma unroll
        for (int i = 0; i < FULL_G; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const long long idx = ((long long)(g0 + i * stride) << 2) + j;
                const bool v = (g0 + i * stride) <= n4 && idx < nnz;
                valid[4 * i + j] = v;
                a[4 * i + j] = rec[v ? w4(r[i], j) : 0];
                b[4 * i + j] = rec[v ? w4(c[i], j) : 0];
            }
#pragma unroll
        for (int i = 0; i < FULL_G; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (!valid[4 * i + j]) continue;
                const SubRec8 A = a[4 * i + j], B = b[4 * i + j];
                double ln_ex = ln_trans;
                if (((A.label ^ B.label) & 0x7fffffff) == 0) {
                    const float sd = fabsf(B.centre - A.centre);
                    float ex;
                    if (A.label < 0) {   // circular contig (rare): its length sits in the full record's fragment
                        const int bin_a = rec_full[w4(r[i], j)].bin & 0x7fffffff;
                        ex = rippe_circ(sd, (float)lcontbp[bin_a] / 1000.0f, par) * norm;
                    } else ex = rippe(sd, par) * norm;
                    ln_ex = mm_ln(ex);
                }
                const long long q = to_q((double)__int_as_float(w4(w[i], j)) * ln_ex);
                if (q == Q_BAD) bad = true; else acc += q;
            }
    }
    if (bad) atomicOr((unsigned long long*)bad_flag, 1ull);
    __shared__ long long s_part[4];
    acc = wave_sum_ll(acc);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long v = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        if (v != 0) atomicAdd((unsigned long long*)out, (unsigned long long)v);
    }
}

// The same again with the labels in LDS.  What bounds k_full_nnz_u is the column-side gather: 20 M random 8-byte reads of an
// L2-resident table, each one a 128-byte line fill of the CU's L1 (~75 % of the L1 fill bandwidth at 85 us).  But most
// contacts join two DIFFERENT contigs -- all but a few while contigs are short -- and for those the record is only needed to
// find that out.  Every block therefore keeps the low 16 bits of every sub-fragment's label in LDS (2 bytes x n_sub: 100 KB for
// C5, one 1024-thread block per CU): different 16-bit labels = different contigs = the constant trans logarithm, no gather;
// equal ones (the same contig, or a collision above 65,536 contigs) take the path of k_full_nnz_u.  Same expressions on the same
// values, integer sums: bit-identical.
template <int FULL_G>
__global__ __launch_bounds__(1024) void k_full_nnz_l(const int4* __restrict__ row4, const int4* __restrict__ col4,
                                                      const int4* __restrict__ cnt4, long long nnz, const unsigned short* __restrict__ lab16,
                                                      int n_sub, const SubRec8* __restrict__ rec, const SubRec* __restrict__ rec_full,
                                                      const int* __restrict__ lcontbp, float nfpb, Par par, int accu,
                                                      long long* __restrict__ out, long long* __restrict__ bad_flag)
{
    extern __shared__ unsigned s_lab[];   // two labels per word
    {
        const int nq = (n_sub + 7) >> 3;  // (the device array is padded to a multiple of 16 bytes)
        const uint4* src = reinterpret_cast<const uint4*>(lab16);
        for (int i = threadIdx.x; i < nq; i += blockDim.x) reinterpret_cast<uint4*>(s_lab)[i] = src[i];
    }
    __syncthreads();
    auto lab = [&](int id) -> unsigned { return (s_lab[id >> 1] >> ((id & 1) << 4)) & 0xffffu; };
    const float norm = (float)(accu * accu) / nfpb;
    const double ln_trans = mm_ln(par.v_inter * norm);
    long long acc = 0;
    bool bad = false;
    const int n4 = (int)(nnz >> 2);
    const int stride = (int)(gridDim.x * blockDim.x);
    for (int g0 = (int)(blockIdx.x * blockDim.x + threadIdx.x); g0 <= n4; g0 += FULL_G * stride) {
        int4 r[FULL_G], c[FULL_G], w[FULL_G];
#pragma unroll
        for (int i = 0; i < FULL_G; i++) {
            const int g = g0 + i * stride;
            const int gc = g <= n4 ? g : n4;
            r[i] = ld_stream(row4 + gc); c[i] = ld_stream(col4 + gc); w[i] = ld_stream(cnt4 + gc);
        }
#pragma unroll
        for (int i = 0; i < FULL_G; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const long long idx = ((long long)(g0 + i * stride) << 2) + j;
                if (!((g0 + i * stride) <= n4 && idx < nnz)) continue;
                const int ia = w4(r[i], j), ib = w4(c[i], j);
                double ln_ex = ln_trans;
                if (lab(ia) == lab(ib)) {
                    const SubRec8 A = rec[ia], B = rec[ib];
                    if (((A.label ^ B.label) & 0x7fffffff) == 0) {
                        const float sd = fabsf(B.centre - A.centre);
                        float ex;
                        if (A.label < 0) {   // circular contig (rare): its length sits in the full record's fragment
                            const int bin_a = rec_full[ia].bin & 0x7fffffff;
                            ex = rippe_circ(sd, (float)lcontbp[bin_a] / 1000.0f, par) * norm;
                        } else ex = rippe(sd, par) * norm;
                        ln_ex = mm_ln(ex);
                    }
                }
                const long long q = to_q((double)__int_as_float(w4(w[i], j)) * ln_ex);
                if (q == Q_BAD) bad = true; else acc += q;
            }
    }
    if (bad) atomicOr((unsigned long long*)bad_flag, 1ull);
    __shared__ long long s_part[16];
    acc = wave_sum_ll(acc);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long v = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) v += s_part[i];
        if (v != 0) atomicAdd((unsigned long long*)out, (unsigned long long)v);
    }
}

template <int FULL_G>
__global__ __launch_bounds__(256) void k_full_nnz(const int4* __restrict__ row4, const int4* __restrict__ col4,
                                                   const int4* __restrict__ cnt4, long long nnz, const SubRec* __restrict__ rec,
                                                   const int* __restrict__ lcontbp, float nfpb, Par par,
                                                   int lut_n /* <= LN_TRANS_LUT: products that occur */, int quirk /* ex_pair_ref */,
                                                   long long* __restrict__ out, long long* __restrict__ bad_flag)
{
    __shared__ double s_ln_trans[LN_TRANS_LUT];
    for (int p = threadIdx.x; p < lut_n; p += blockDim.x) s_ln_trans[p] = mm_ln(par.v_inter * ((float)p / nfpb));
    __syncthreads();
    long long acc = 0;
    bool bad = false;
    const int n4 = (int)(nnz >> 2);   // full groups; group n4 is the partial tail (the arrays are padded: in bounds)
    const int stride = (int)(gridDim.x * blockDim.x);
    for (int g0 = (int)(blockIdx.x * blockDim.x + threadIdx.x); g0 <= n4; g0 += FULL_G * stride) {
        int4 r[FULL_G], c[FULL_G], w[FULL_G];
#pragma unroll
        for (int i = 0; i < FULL_G; i++) {
            const int g = g0 + i * stride;
            const int gc = g <= n4 ? g : n4;
            r[i] = ld_stream(row4 + gc); c[i] = ld_stream(col4 + gc); w[i] = ld_stream(cnt4 + gc);
        }
        SubRec a[4 * FULL_G], b[4 * FULL_G];
        bool valid[4 * FULL_G];
#pragma unroll
        for (int i = 0; i < FULL_G; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const long long idx = ((long long)(g0 + i * stride) << 2) + j;
                const bool v = (g0 + i * stride) <= n4 && idx < nnz;
                valid[4 * i + j] = v;
                // (padding words are zero: row/col 0 are valid table indices, the result is discarded)
                a[4 * i + j] = rec[v ? w4(r[i], j) : 0];
                b[4 * i + j] = rec[v ? w4(c[i], j) : 0];
            }
#pragma unroll
        for (int i = 0; i < FULL_G; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (!valid[4 * i + j]) continue;
                const SubRec A = a[4 * i + j], B = b[4 * i + j];
                const int bin_a = A.bin & 0x7fffffff, bin_b = B.bin & 0x7fffffff;
                int prod = (A.accu & 0xffff) * (B.accu & 0xffff);
                if (quirk && A.label != B.label) prod = bin_a < bin_b ? (A.accu >> 16) * (B.accu & 0xffff) : (A.accu & 0xffff) * (B.accu >> 16);
                double ln_ex;
                if (A.label != B.label && (unsigned)prod < (unsigned)lut_n) ln_ex = s_ln_trans[prod];
                else {
                    // ex_pair (kernels3.cu:3062-3078 / 3184-3195) on the precomputed centres
                    const float norm = (float)prod / nfpb;
                    float ex;
                    if (A.label != B.label) ex = par.v_inter * norm;
                    else {
                        const float sd = fabsf(B.centre - A.centre);
                        ex = (A.bin < 0 ? rippe_circ(sd, (float)lcontbp[bin_a] / 1000.0f, par) : rippe(sd, par)) * norm;
                    }
                    ln_ex = mm_ln(ex);
                }
                const long long q = to_q((double)__int_as_float(w4(w[i], j)) * ln_ex); // counts are float32 (the reference's obs type)
                if (q == Q_BAD) bad = true; else acc += q;
            }
    }
    if (bad) atomicOr((unsigned long long*)bad_flag, 1ull);
    // one atomic per BLOCK (same-address atomics go through at ~10 ns each)
    __shared__ long long s_part[4];
    acc = wave_sum_ll(acc);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long v = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        if (v != 0) atomicAdd((unsigned long long*)out, (unsigned long long)v);
    }
}

// cis correction of the expected mass for the current layout: one thread per fragment x (in contig order),
// pairs (x, y later in the same contig) while the gap is below d_max, plus x's own sub-fragment pairs.
__device__ __forceinline__ long long pair_mass_q(const End& X, const Stat& sx, const End& Y, const Stat& sy, float nfpb, const Par& par);
// pair_mass_q from centre coordinates computed once per fragment (centre_kb's float32 values, cached: every pair of a walk would
// otherwise redo two int->float conversions, four divisions and the running sums of BOTH fragments).  Both fragments lie in
// one contig of the layout in question (mass tasks exist for cis relations only).
struct Ctr { float c0, c1, c2; };
__device__ __forceinline__ Ctr centres_of(int start_bp, bool fwd, const Stat& st)
{
    Ctr c; c.c0 = centre_kb(start_bp, fwd, st, 0); c.c1 = st.n > 1 ? centre_kb(start_bp, fwd, st, 1) : 0.0f;
    c.c2 = st.n > 2 ? centre_kb(start_bp, fwd, st, 2) : 0.0f;
    return c;
}
__device__ __forceinline__ long long pair_mass_q_c(const Ctr& cx, const Stat& sx, const Ctr& cy, const Stat& sy, int circ, int lbp,
                                                   float nfpb, const Par& par)
{
    double acc = 0.0;
    for (int a = 0; a < sx.n; a++)
        for (int b = 0; b < sy.n; b++) {
            const float norm = (float)(stat_accu(sx, a) * stat_accu(sy, b)) / nfpb;
            const float sd = fabsf(sel3(cy.c0, cy.c1, cy.c2, b) - sel3(cx.c0, cx.c1, cx.c2, a));
            const float ex = (circ == 1 ? rippe_circ(sd, (float)lbp / 1000.0f, par) : rippe(sd, par)) * norm;
            acc += (double)ex - (double)(par.v_inter * norm);
        }
    return to_q(acc);
}


// SIXTEEN LANES per fragment x (position-index slot i): they take the fragments y behind it in the contig, 16 at a time,
// until all of them are beyond the window.  (One THREAD per fragment left a 1,000-bin genome with 1,000 threads walking
// ~200 x 9 slot pairs each: 1.3 ms per full evaluation, which the nuisance-parameter step pays every MCMC step; a whole wave
// per fragment costs 50,000 nearly empty waves while contigs are short: 58 us against 33.)
// Like the candidate tasks, every fragment PAIR is rounded to Q once, so the sum does not depend on how it is partitioned.
template <int LPF>   // lanes per fragment x: 16, or 64 for maps of a few thousand bins (a contig of 200 bins with nine slot pairs per
                     // fragment pair is a chain of 13 x 9 evaluations per lane at 16 lanes: 80 us per call at the C2 shape, every step
                     // of a reference-arithmetic run with sub-fragments; 64 lanes: a quarter of the chain)
__global__ __launch_bounds__(256) void k_full_mass(int n, const int* __restrict__ perm, const int* __restrict__ contig_off,
                                                    const Geo* __restrict__ geo, const Stat* __restrict__ stat,
                                                    const int* __restrict__ lcont, const int* __restrict__ lcontbp,
                                                    const int* __restrict__ pos, float nfpb, Par par, int reach_bp,
                                                    long long* __restrict__ out, long long* __restrict__ bad_flag)
{
    constexpr int LPF_SHIFT = LPF == 64 ? 6 : 4;
    const int lane = threadIdx.x & 63, sub = lane & (LPF - 1), grp = LPF == 64 ? 0 : lane >> 4;
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> LPF_SHIFT;
    long long accq = 0;
    bool bad = false;
    int remaining = 0;
    Geo gx = {0, 0, 0, 0};
    Stat sx = {0.0f, 0.0f, 0.0f, 0, 0, 0, 0, 0};
    End X = {0, 0, true, 0, 0};
    Ctr cx = {0.0f, 0.0f, 0.0f};
    if (i < n) {
        const int fx = perm[i];
        gx = geo[fx];
        sx = stat[fx];
        X = end_cur(gx, lcontbp, fx);
        cx = centres_of(X.start_bp, X.fwd, sx);
        if (sub == 0) {
            double acc = 0.0;
            for (int a = 0; a < sx.n; a++)
                for (int b = a + 1; b < sx.n; b++) acc += (double)ex_pair(X, sx, a, X, sx, b, nfpb, par);
            const long long q = to_q(acc);
            if (q == Q_BAD) bad = true; else accq += q;
        }
        remaining = lcont[fx] - 1 - pos[fx];
    }
    bool live = remaining > 0; // (uniform within the LPF lanes of a fragment)
    for (int k0 = 1; __ballot(live) != 0; k0 += LPF) {
        bool inside = false;
        const int k = k0 + sub;
        if (live && k <= remaining) {
            const int fy = perm[i + k];
            const Geo gy = geo[fy];
            inside = gy.start_bp - (gx.start_bp + gx.len_bp) <= reach_bp;
            if (inside) {
                const Stat sy = stat[fy];
                const Ctr cy = centres_of(gy.start_bp, (gy.flags & 1) != 0, sy);
                const long long q = pair_mass_q_c(cx, sx, cy, sy, X.circ, X.lbp, nfpb, par);   // (same contig: same circular model)
                if (q == Q_BAD) bad = true; else accq += q;
            }
        }
        // start_bp grows along the contig: once none of a fragment's 16 lanes found a y inside the window, nothing further is
        const unsigned long long bal = __ballot(inside);
        const bool any = LPF == 64 ? bal != 0ull : ((unsigned)(bal >> (16 * grp)) & 0xffffu) != 0u;
        live = live && any && k0 + LPF <= remaining;
    }
    (void)contig_off;
    if (bad) atomicOr((unsigned long long*)bad_flag, 1ull);
    __shared__ long long s_part[4]; // (one atomic per block, as in k_full_nnz)
    const long long q = wave_sum_ll(accq);
    if (lane == 0) s_part[threadIdx.x >> 6] = q;
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long v = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        if (v != 0) atomicAdd((unsigned long long*)out, (unsigned long long)v);
    }
}

// LDS hand-over between the lanes of ONE wave: LDS operations of a wave execute in program order, so no hardware barrier is
// needed -- only the compiler must not move them across lanes' dependencies
#define WAVE_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// The same sum for LONG contigs (thousands of fragments inside the window: C5 on its 7 contigs holds 1.2e8 such pairs), tiled like the
// reference-arithmetic candidate kernel (strict2.h): a wave's lanes are 64 consecutive fragments x of the position index, the fragments y
// behind them are staged in LDS 64 at a time -- record, statistics and sub-fragment centres computed ONCE per y, not once per pair -- and
// every lane walks the staged tile; S waves share an x tile, wave s takes the tiles y number s, s + S, ...  Per pair what is left is the
// contact model.  The same per-pair values, rounded to Q once per pair: bit-identical to k_full_mass.
struct FMTile { int label, start_bp, n, frag; float c0, c1, c2; int a0, a1, a2, pad0, pad1; };   // 48 bytes
template <bool MULTI>
__global__ __launch_bounds__(256) void k_full_mass_t(int n, const int* __restrict__ perm, const Geo* __restrict__ geo, const Stat* __restrict__ stat,
                                                      const int* __restrict__ lcont, const int* __restrict__ lcontbp, const int* __restrict__ pos,
                                                      float nfpb, Par par, int reach_bp, int S, float norm_u, long long* __restrict__ out,
                                                      long long* __restrict__ bad_flag)
{
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int W = blockIdx.x * 4 + wib, tile = W / S, s = W - tile * S;
    __shared__ FMTile s_y[4][64];
    FMTile* const ty = s_y[wib];
    const int i = tile * 64 + lane;
    long long accq = 0;
    bool bad = false;
    Geo gx = {0, 0, 0, 0};
    Stat sx = {0.0f, 0.0f, 0.0f, 0, 0, 0, 0, 0};
    Ctr cx = {0.0f, 0.0f, 0.0f};
    int last = -1, circ = 0, lbp = 0;       // last slot of x's contig; its circular model
    if (i < n) {
        const int fx = perm[i];
        gx = geo[fx];
        sx = stat[fx];
        const End X = end_cur(gx, lcontbp, fx);
        cx = centres_of(X.start_bp, X.fwd, sx);
        circ = X.circ; lbp = X.lbp;
        last = i + (lcont[fx] - 1 - pos[fx]);
        if (s == 0) {   // x's own sub-fragment pairs
            double acc = 0.0;
            for (int a = 0; a < sx.n; a++)
                for (int b = a + 1; b < sx.n; b++) acc += (double)ex_pair(X, sx, a, X, sx, b, nfpb, par);
            const long long q = to_q(acc);
            if (q == Q_BAD) bad = true; else accq += q;
        }
    }
    const float s_tot = (float)lbp / 1000.0f;
    const int x_end = gx.start_bp + gx.len_bp;
    bool live = i < n && last > i;          // (start_bp grows along a contig: once a y of x's contig is beyond the window, all later ones are)
    for (int c = s; ; c += S) {
        const int j0 = tile * 64 + 64 * c;
        // any lane still has fragments of its contig at or behind this tile?
        if (__ballot(live && last >= j0) == 0ull) break;
        {
            const int j = j0 + lane;
            FMTile y;
            y.label = -1; y.start_bp = 0; y.n = 0; y.frag = 0; y.c0 = y.c1 = y.c2 = 0.0f; y.a0 = y.a1 = y.a2 = 0; y.pad0 = y.pad1 = 0;
            if (j < n) {
                const int fy = perm[j];
                const Geo gy = geo[fy];
                const Stat sy = stat[fy];
                const Ctr cy = centres_of(gy.start_bp, (gy.flags & 1) != 0, sy);
                y.label = gy.id_c; y.start_bp = gy.start_bp; y.n = sy.n; y.frag = fy; y.c0 = cy.c0; y.c1 = cy.c1; y.c2 = cy.c2; y.a0 = sy.a0; y.a1 = sy.a1; y.a2 = sy.a2;
            }
            ty[lane] = y;
        }
        WAVE_LDS_SYNC();
        const int cnt = min(64, n - j0);
        if (live && last >= j0) {
            for (int jj = 0; jj < cnt; jj++) {
                const int j = j0 + jj;
                if (j <= i || j > last) continue;                         // every unordered pair once; x's contig only
                const FMTile& y = ty[jj];
                if (y.start_bp - x_end > reach_bp) { live = false; break; } // beyond the window: exactly zero, and so is everything behind it
                double acc = 0.0;
                if (!MULTI) {
                    if (sx.n > 0 && y.n > 0) {
                        const float norm = norm_u >= 0.0f ? norm_u : (float)(sx.a0 * y.a0) / nfpb;
                        const float sd = fabsf(y.c0 - cx.c0);
                        const float ex = (circ == 1 ? rippe_circ(sd, s_tot, par) : rippe(sd, par)) * norm;
                        acc = 0.0 + ((double)ex - (double)(par.v_inter * norm));
                    }
                } else {
                    for (int a = 0; a < sx.n; a++)
                        for (int b = 0; b < y.n; b++) {
                            const float norm = norm_u >= 0.0f ? norm_u : (float)(stat_accu(sx, a) * sel3(y.a0, y.a1, y.a2, b)) / nfpb;
                            const float sd = fabsf(sel3(y.c0, y.c1, y.c2, b) - sel3(cx.c0, cx.c1, cx.c2, a));
                            const float ex = (circ == 1 ? rippe_circ(sd, s_tot, par) : rippe(sd, par)) * norm;
                            acc += (double)ex - (double)(par.v_inter * norm);
                        }
                }
                const long long q = to_q_fast(acc);
                if (q == Q_BAD) bad = true; else accq += q;
            }
        }
        WAVE_LDS_SYNC();
    }
    if (bad) atomicOr((unsigned long long*)bad_flag, 1ull);
    __shared__ long long s_part[4];
    const long long q = wave_sum_ll(accq);
    if (lane == 0) s_part[wib] = q;
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long v = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        if (v != 0) atomicAdd((unsigned long long*)out, (unsigned long long)v);
    }
}

// ------------------------------------------------------------------ candidate tables
struct Task {            // windowed cis sum between piece p and piece q in one layout
    int p, q;            // piece ids; p == q: pairs inside the piece.  The larger piece is p (the parallel axis)
    Xf xp, xq;           // transforms of the two pieces in that layout
    unsigned plus, minus; // 13-bit op masks: layouts (ops) in which this sum is the NEW (+) / OLD (-) value
    int base_p, np, base_q, nq; // position-index ranges of the two pieces (perm[base .. base + n))
};
constexpr int ITEM_CAP = 2048;     // work items per neighbour with a direct item -> task table (else: binary search)
constexpr int INLINE_PAIRS = 1024; // fragment pairs per neighbour that the table block prices itself (k_tm)
constexpr int STRICT_INLINE_M = 20;  // reference arithmetic: affected sets up to this many fragments are priced by the table block (190 pairs x 13)

struct NbTables {        // everything the finishing kernel needs about one neighbour
    PieceKey key;
    int fB;
    int lo[NP], hi[NP], contig[NP];  // old pos range / contig label of every piece (hi < lo: empty)
    Xf xf[N_OPS][NP];
    unsigned long long changed[N_OPS]; // bit p*8+q (both orders): relation (p,q) changed; p==q: intra
    unsigned intra_any;                // bit p: some op changes piece p internally
    int n_tasks, n_items;
    int item_start[MAX_TASKS + 1];     // prefix sum of 64-fragment chunks per task: the mass work list, in a fixed order
    int cw[MAX_TASKS];                 // chunks << 19 | walk; walk = fragments y a chunk of the task is paired with (np inside a piece, else nq)
    long long w_total;                 // sum over tasks of chunks x walk: k_fin sizes its work units by it
    // per piece pair (pair_index): how a CONTACT between the two pieces is priced -- the tasks (distinct cis relations: the old
    // one and the deduplicated new ones) it has to be evaluated under, and the candidates under which the relation changes and
    // is trans afterwards (tplus) / was trans before (tminus)
    unsigned short tplus[N_PAIRS], tminus[N_PAIRS];
    unsigned short pair_old[N_PAIRS];               // the task of the pair's OLD relation (0xffff: it is trans)
    unsigned short pair_n[N_PAIRS];                 // the tasks of its distinct NEW cis relations
    unsigned short pair_task[N_PAIRS][N_OPS];
    unsigned short pair_owner[N_PAIRS][N_OPS];      // per candidate: the task of the pair's new cis relation under it (0xffff: trans, or unchanged)
    unsigned item_tc[ITEM_CAP];        // task | chunk << 16 of item w (valid when n_items <= ITEM_CAP)
    // reference arithmetic (GRAAL_MODE_STRICT): the 13 candidates in classes of equal contact-model inputs per piece pair
    // (same_inputs).  crep[pair][op] = the first candidate with the same inputs as `op` (its class representative: the class is
    // priced once, under it), CREP_OLD = the same inputs as the current layout (nothing to price); cmask[pair][rep] = the
    // candidates of that class
    unsigned char crep[N_PAIRS][N_OPS];
    unsigned short cmask[N_PAIRS][N_OPS];
    int set_m;                         // strict: fragments of contig(fA) u contig(fB) that are left to k_strict (0: priced by k_tm, or fB == fA)
    UEnd endA, endB;                   // the contigs of fA and fB as the union set's builder wants them (k_gprep: one round trip instead of three)
    Task task[MAX_TASKS];
};
constexpr unsigned char CREP_OLD = 0xff;

// "these values are needed HERE": keeps the compiler from sinking independent loads below a data-dependent branch -- two dependent
// memory round trips where one would do
__device__ __forceinline__ void keep_xf(Xf& a, Xf& b) { asm volatile("" : "+v"(a.label), "+v"(a.sigma), "+v"(a.off), "+v"(a.circ), "+v"(a.lbp), "+v"(b.label), "+v"(b.sigma), "+v"(b.off), "+v"(b.circ), "+v"(b.lbp)); }

// one staged fragment y of the mass walk (k_fin): transformed geometry + statistics, 64 bytes
struct YTile { int start_bp, len_bp, flags, label, lbp; float c0, c1, c2; Stat st; };   // (c*: centres of its sub-fragments, kb)
// completion counters of k_scan: N_DONE words on lines of their own take the blocks in turn, so that 496 device-scope atomics
// do not queue on one address (measured: 0.3-0.5 us per step against the single counter, GRAAL_SCAN_DONE_N=1; tools/done_ab.sh)
constexpr int N_DONE = 16, DONE_STRIDE = 16;
constexpr int FLAG_STRIDE = 32; // words between two blocks' completion flags: one 128-byte line each (partial writes to one
                                // line from many XCDs serialise at the memory side)


// exclusive prefix sum of vals[0..n) into out[0..n) and the total into out[n]; vals / out in LDS.  Executed by ONE wave
// (all 64 lanes of it), no block barrier inside: a few hundred entries are cheaper to scan in one wave than to
// synchronise four waves around.  The caller puts __syncthreads() before (vals complete) and after (out visible).
__device__ __forceinline__ void wave_excl_scan(const int* vals, int* out, int n)
{
    const int lane = threadIdx.x & 63;
    int carry = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const int v = i < n ? vals[i] : 0;
        int x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
        if (i < n) out[i] = carry + x - v;
        carry += __shfl(x, 63, 64);
    }
    if (lane == 0) out[n] = carry;
}

__device__ __forceinline__ void pair_of_index(int idx, int& p, int& q)
{
    p = 1; q = 1;
    for (p = 1; p <= MAX_PIECES; p++) { const int row = MAX_PIECES - p + 1; if (idx < row) { q = p + idx; break; } idx -= row; }
}

__device__ __forceinline__ int pair_index(int p, int q) // inverse of pair_of_index, any order of 1 <= p, q <= MAX_PIECES
{
    const int a = p < q ? p : q, b = p < q ? q : p;
    return (a - 1) * (2 * MAX_PIECES + 2 - a) / 2 + (b - a);
}

// short contigs (<= N_MATES fragments each): the geometry and statistics of every fragment of contig(fA) and contig(fB)
// are fetched into LDS while the tables are being built, so the block's own mass pricing needs no global loads
struct SmallCtx {
    int small;                 // both contigs are short
    int base[2], len[2];       // position-index base / length of contig(fA), contig(fB)
    int mates[2][N_MATES];
    Geo geo[2 * N_MATES];
    Stat stat[2 * N_MATES];
};

// one block builds everything about one neighbour; tasks are left in s_task (LDS) too, with the exclusive prefix of
// their fragment-pair counts in s_pp.  Returns the number of tasks.
struct TabViews {   // what the strict pricing of k_tm reads of the block's tables (LDS)
    const Xf* xf;                  // [N_OPS][NP]
    const Xf* xf_old;              // [NP]
    const unsigned char* crep;     // [N_PAIRS][N_OPS]
    const unsigned* cmask;         // [N_PAIRS][N_OPS]
    const Rec *A0, *B0;
};
// (inlined: the pointers are then known to be global -- as arguments of a called function they were generic, the loads FLAT, and the
// compiler ordered every one of them against the LDS stores around it: the four mates-row loads of the first thread went out one
// round trip after the other, 5 us of this block's 13; blockDim.x, re-read from the dispatch packet in every loop, is a constant here)
constexpr int TM_THREADS = 256;
__device__ __forceinline__ int tables_block(const Geo* __restrict__ geo, const Link* __restrict__ link, const int* __restrict__ cbase,
                            const int* __restrict__ mates, const Stat* __restrict__ stat, int fA, int fB, int max_id, NbTables& T,
                            int k, int* __restrict__ step_hdr, Task* s_task, int* s_pp, SmallCtx& sc, int strict, int quirk, TabViews& tv)
{
    __shared__ Rec A0, B0;
    __shared__ int s_baseA, s_baseB;
    __shared__ int rep[NP];
    __shared__ Rec rep_old[NP];
    __shared__ Xf xf_old[NP];
    __shared__ Xf xf[N_OPS][NP];
    __shared__ unsigned long long changed[N_OPS];
    __shared__ unsigned intra_any;
    // dedupe table: entries 0..20 = old relation of (p<=q); then 13*21 new relations
    constexpr int NPAIR = N_PAIRS, NENT = NPAIR * (N_OPS + 1);
    __shared__ int e_valid[NENT], e_owner[NENT], e_slot[NENT], e_flag[NENT];
    __shared__ unsigned t_plus[NPAIR], t_minus[NPAIR];
    __shared__ int p_cnt[NPAIR], p_old[NPAIR];
    __shared__ unsigned short p_list[NPAIR][N_OPS];
    __shared__ unsigned e_plus[NENT], e_minus[NENT];
    __shared__ int s_lo[NP], s_hi[NP], s_contig[NP], s_cbase[NP], s_chunks[NENT], s_start[NENT + 1], s_pairs[NENT];
    const int t = threadIdx.x;
    if (t == 0) {
        const Geo gA = geo[fA], gB = geo[fB];
        const Link lA = link[fA], lB = link[fB];
        const int4 mA0 = reinterpret_cast<const int4*>(mates)[2 * fA], mA1 = reinterpret_cast<const int4*>(mates)[2 * fA + 1];
        const int4 mB0 = reinterpret_cast<const int4*>(mates)[2 * fB], mB1 = reinterpret_cast<const int4*>(mates)[2 * fB + 1];
        s_baseA = cbase[fA]; s_baseB = cbase[fB];
        A0 = rec_gl(gA, lA, fA); B0 = rec_gl(gB, lB, fB);
        sc.base[0] = s_baseA; sc.base[1] = s_baseB; sc.len[0] = lA.l_cont; sc.len[1] = lB.l_cont;
        sc.small = (lA.l_cont <= N_MATES && lB.l_cont <= N_MATES) ? 1 : 0;
        sc.mates[0][0] = mA0.x; sc.mates[0][1] = mA0.y; sc.mates[0][2] = mA0.z; sc.mates[0][3] = mA0.w;
        sc.mates[0][4] = mA1.x; sc.mates[0][5] = mA1.y; sc.mates[0][6] = mA1.z; sc.mates[0][7] = mA1.w;
        sc.mates[1][0] = mB0.x; sc.mates[1][1] = mB0.y; sc.mates[1][2] = mB0.z; sc.mates[1][3] = mB0.w;
        sc.mates[1][4] = mB1.x; sc.mates[1][5] = mB1.y; sc.mates[1][6] = mB1.z; sc.mates[1][7] = mB1.w;
        PieceKey key; key.cA = A0.id_c; key.a = A0.pos; key.cB = B0.id_c; key.b = B0.pos;
        T.key = key; T.fB = fB;
        {
            UEnd ea, eb;
            ea.label = A0.id_c; ea.pos = A0.pos; ea.base = s_baseA; ea.len = lA.l_cont; ea.lbp = lA.l_cont_bp; ea.circ = A0.circ;
            eb.label = B0.id_c; eb.pos = B0.pos; eb.base = s_baseB; eb.len = lB.l_cont; eb.lbp = lB.l_cont_bp; eb.circ = B0.circ;
            T.endA = ea; T.endB = eb;
        }
        piece_representatives(key, fA, fB, A0, B0, rep);
        intra_any = 0;
        for (int p = 0; p < NP; p++) { s_lo[p] = 0; s_hi[p] = -1; s_contig[p] = -1; }
        if (fA != fB) {
            if (key.cA != key.cB) {
                s_lo[1] = 0; s_hi[1] = key.a - 1; s_lo[2] = s_hi[2] = key.a; s_lo[3] = key.a + 1; s_hi[3] = A0.l_cont - 1;
                s_lo[4] = 0; s_hi[4] = key.b - 1; s_lo[5] = s_hi[5] = key.b; s_lo[6] = key.b + 1; s_hi[6] = B0.l_cont - 1;
                for (int p = 1; p <= 3; p++) { s_contig[p] = key.cA; s_contig[p + 3] = key.cB; }
            } else {
                const int lo = key.a < key.b ? key.a : key.b, hi = key.a < key.b ? key.b : key.a;
                s_lo[1] = 0; s_hi[1] = lo - 1; s_lo[2] = s_hi[2] = lo; s_lo[3] = lo + 1; s_hi[3] = hi - 1;
                s_lo[4] = s_hi[4] = hi; s_lo[5] = hi + 1; s_hi[5] = A0.l_cont - 1;
                for (int p = 1; p <= 5; p++) s_contig[p] = key.cA;
            }
        }
        for (int p = 0; p < NP; p++) { T.lo[p] = s_lo[p]; T.hi[p] = s_hi[p]; T.contig[p] = s_contig[p]; }
    }
    STAMP(24, k == 0 && t == 0); // A0 / B0 loaded
    if (t < N_OPS) changed[t] = 0;
    for (int e = t; e < NENT; e += TM_THREADS) { e_valid[e] = 0; e_plus[e] = 0; e_minus[e] = 0; e_owner[e] = e; e_slot[e] = -1; }
    if (t < NPAIR) { t_plus[t] = 0; t_minus[t] = 0; p_cnt[t] = 0; p_old[t] = 0xffff; }
    __syncthreads();
    if (t < NP) {
        if (rep[t] >= 0) { rep_old[t] = rec_gl(geo[rep[t]], link[rep[t]], rep[t]); xf_old[t] = xf_identity(rep_old[t]); }
        else { Xf x; x.label = -1 - t; x.sigma = 1; x.off = 0; x.circ = 0; x.lbp = 0; xf_old[t] = x; }
        s_cbase[t] = s_contig[t] < 0 ? 0 : (s_contig[t] == A0.id_c ? s_baseA : s_baseB);
    }
    if (t >= 64 && t < 64 + 2 * N_MATES && sc.small) { // another wave: these loads overlap the table construction
        const int i = t - 64, w = i / N_MATES, j = i % N_MATES;
        const int f = j < sc.len[w] ? sc.mates[w][j] : -1;
        if (f >= 0) { sc.geo[i] = geo[f]; sc.stat[i] = stat[f]; }
    }
    __syncthreads();
    STAMP(25, k == 0 && t == 0); // representatives loaded
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
    STAMP(26, k == 0 && t == 0); // transforms
    __shared__ unsigned char s_crep[N_PAIRS][N_OPS];
    __shared__ unsigned s_cmask[N_PAIRS][N_OPS];
    tv.xf = &xf[0][0]; tv.xf_old = xf_old; tv.crep = &s_crep[0][0]; tv.cmask = &s_cmask[0][0]; tv.A0 = &A0; tv.B0 = &B0;
    if (strict) {   // classes of equal contact-model inputs (see NbTables::crep): one thread per (piece pair, candidate)
        // what same_inputs (frag_ops.h) compares, packed into 128 bits per (pair, candidate) -- and per pair for the current layout:
        // two keys are equal iff same_inputs says so.  A thread then compares its key with its up to 12 predecessors' at one 16-byte LDS
        // read each (same_inputs on the transforms themselves reads 20 words per comparison: 5-7 us of this block, on the step's
        // critical path -- tools/stamps_step.py, tools/stamps_c4.py)
        __shared__ int4 s_key[N_PAIRS][N_OPS + 1];
        auto key_of = [&](const Xf& a, const Xf& b) {
            const InputsKey ik = inputs_key(a, b, quirk != 0);   // (frag_ops.h)
            return make_int4(ik.x, ik.y, ik.z, ik.w);
        };
        for (int e = t; e < N_PAIRS * (N_OPS + 1); e += TM_THREADS) {
            const int pair = e / (N_OPS + 1), op = e - pair * (N_OPS + 1);
            int p, q;
            pair_of_index(pair, p, q);
            s_key[pair][op] = op == N_OPS ? key_of(xf_old[p], xf_old[q]) : key_of(xf[op][p], xf[op][q]);
            if (op < N_OPS) s_cmask[pair][op] = 0;
        }
        __syncthreads();
        for (int e = t; e < N_PAIRS * N_OPS; e += TM_THREADS) {
            const int pair = e / N_OPS, op = e - pair * N_OPS;
            int p, q;
            pair_of_index(pair, p, q);
            const int4 mine = s_key[pair][op];
            auto eq = [&](const int4& o4) { return o4.x == mine.x && o4.y == mine.y && o4.z == mine.z && o4.w == mine.w; };
            int r = op;
            if (rep[p] < 0 || rep[q] < 0 || eq(s_key[pair][N_OPS])) r = CREP_OLD;
            else
                for (int o = 0; o < op; o++)
                    if (eq(s_key[pair][o])) { r = o; break; }
            s_crep[pair][op] = (unsigned char)r;
            T.crep[pair][op] = (unsigned char)r;
            if (r != CREP_OLD) atomicOr(&s_cmask[pair][r], 1u << op);
        }
        __syncthreads();
        for (int e = t; e < N_PAIRS * N_OPS; e += TM_THREADS) T.cmask[e / N_OPS][e % N_OPS] = (unsigned short)s_cmask[e / N_OPS][e % N_OPS];
        // Reference arithmetic prices classes of inputs, not changed relations: the relation masks, the deduplicated task list and the
        // work lists below serve the exact mode's kernels only (6 of this block's 17 us next to the scan: tools/stamps_step.py)
        if (t == 0) { T.n_tasks = 0; T.n_items = 0; T.intra_any = 0; T.w_total = 0; step_hdr[k] = 0; s_pp[0] = 0; }
        __syncthreads();
        return 0;
    }
    // relations: one thread per (op, p <= q)
    for (int e = t; e < N_OPS * NPAIR; e += TM_THREADS) {
        const int op = e / NPAIR;
        int p, q;
        pair_of_index(e % NPAIR, p, q);
        if (rep[p] < 0 || rep[q] < 0) continue;
        const bool chg = (p == q) ? intra_changed(xf_old[p], xf[op][p]) : rel_changed(xf_old[p], xf_old[q], xf[op][p], xf[op][q]);
        if (!chg) continue;
        atomicOr(&changed[op], (1ull << (p * 8 + q)) | (1ull << (q * 8 + p)));
        if (p == q) atomicOr(&intra_any, 1u << p);
        const int pair = e % NPAIR;
        // old relation entry (shared by all ops) and new relation entry
        if (xf_old[p].label == xf_old[q].label) { e_valid[pair] = 1; atomicOr(&e_minus[pair], 1u << op); }
        else atomicOr(&t_minus[pair], 1u << op);
        if (xf[op][p].label == xf[op][q].label) { e_valid[NPAIR + e] = 1; atomicOr(&e_plus[NPAIR + e], 1u << op); }
        else atomicOr(&t_plus[pair], 1u << op);
    }
    __syncthreads();
    STAMP(27, k == 0 && t == 0); // relations
    // dedupe new-relation entries against every earlier entry with the same relative geometry
    for (int e = NPAIR + t; e < NENT; e += TM_THREADS) {
        if (!e_valid[e]) continue;
        const int ee = e - NPAIR, op = ee / NPAIR, pair = ee % NPAIR;
        int p, q;
        pair_of_index(pair, p, q);
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
    for (int e = NPAIR + t; e < NENT; e += TM_THREADS)
        if (e_valid[e] && e_owner[e] != e) { atomicOr(&e_plus[e_owner[e]], e_plus[e]); }
    __syncthreads();
    STAMP(28, k == 0 && t == 0); // dedupe
    // slots of the surviving (owner) entries, in entry order
    for (int e = t; e < NENT; e += TM_THREADS) e_flag[e] = (e_valid[e] && e_owner[e] == e) ? 1 : 0;
    __syncthreads();
    if (t < 64) wave_excl_scan(e_flag, s_start, NENT); // s_start used as scratch: slot of entry e
    __syncthreads();
    const int n_tasks = s_start[NENT];
    for (int e = t; e < NENT; e += TM_THREADS) if (e_flag[e]) e_slot[e] = s_start[e];
    __syncthreads();
    STAMP(29, k == 0 && t == 0); // slots
    if (t == 0) {
        T.n_tasks = n_tasks;
        T.intra_any = intra_any;
        for (int op = 0; op < N_OPS; op++) T.changed[op] = changed[op];
    }
    for (int e = t; e < NENT; e += TM_THREADS) {
        if (e_slot[e] < 0) continue;
        const int pair = (e < NPAIR) ? e : (e - NPAIR) % NPAIR;
        int p, q;
        pair_of_index(pair, p, q);
        if (s_hi[p] - s_lo[p] < s_hi[q] - s_lo[q]) { const int tp = p; p = q; q = tp; } // the larger piece goes on the parallel (lane) axis
        // (the pieces are swapped BEFORE their transforms are fetched: swapping the 5-word records went through scratch memory)
        Task tk; tk.p = p; tk.q = q;
        if (e < NPAIR) { tk.xp = xf_old[p]; tk.xq = xf_old[q]; }
        else { const int op = (e - NPAIR) / NPAIR; tk.xp = xf[op][p]; tk.xq = xf[op][q]; }
        tk.plus = e_plus[e]; tk.minus = e_minus[e];
        tk.np = s_hi[tk.p] - s_lo[tk.p] + 1; tk.nq = s_hi[tk.q] - s_lo[tk.q] + 1;
        tk.base_p = s_cbase[tk.p] + s_lo[tk.p];
        tk.base_q = s_cbase[tk.q] + s_lo[tk.q];
        const int slot = e_slot[e];
        if (e < NPAIR) p_old[pair] = slot; else p_list[pair][atomicAdd(&p_cnt[pair], 1)] = (unsigned short)slot;
        s_task[slot] = tk;
        s_chunks[slot] = (tk.np + 63) / 64;
        const long long pr = (long long)tk.np * (long long)(tk.p == tk.q ? tk.np : tk.nq);
        s_pairs[slot] = pr > (1 << 20) ? (1 << 20) : (int)pr; // saturated: only compared with INLINE_PAIRS
    }
    __syncthreads();
    STAMP(30, k == 0 && t == 0); // tasks in LDS
    // fragment-pair prefix (for the block's own pricing) and work list (chunks of 64 fragments of the task's first piece,
    // tasks in slot order = a fixed order), one wave each
    if (t < 64) wave_excl_scan(s_pairs, s_pp, n_tasks);
    else if (t < 128) wave_excl_scan(s_chunks, s_start, n_tasks);
    __syncthreads();
    const int n_items = s_start[n_tasks];
    if (t == 0) { T.n_items = n_items; step_hdr[k] = n_items; }
    // (the tasks and the per-pair lists are needed by whoever prices the queued contacts, always)
    for (int i = t; i < n_tasks; i += TM_THREADS) T.task[i] = s_task[i];
    if (t < NPAIR) {
        T.tplus[t] = (unsigned short)t_plus[t]; T.tminus[t] = (unsigned short)t_minus[t]; T.pair_n[t] = (unsigned short)p_cnt[t];
        T.pair_old[t] = (unsigned short)p_old[t];
        for (int op = 0; op < N_OPS; op++) {
            const int e = NPAIR + op * NPAIR + t;
            T.pair_owner[t][op] = (unsigned short)((e_valid[e] && e_slot[e_owner[e]] >= 0) ? e_slot[e_owner[e]] : 0xffff);
        }
        for (int i = 0; i < p_cnt[t]; i++) T.pair_task[t][i] = p_list[t][i];
    }
    if (s_pp[n_tasks] > INLINE_PAIRS) { // k_fin will price this neighbour's mass: it needs the work list in memory
        for (int i = t; i <= n_tasks; i += TM_THREADS) T.item_start[i] = s_start[i];
        for (int i = t; i < n_tasks; i += TM_THREADS) T.cw[i] = (s_chunks[i] << 19) | (s_task[i].p == s_task[i].q ? s_task[i].np : s_task[i].nq);
        if (t >= 128 && t < 192) {
            long long w = 0;
            for (int i = t - 128; i < n_tasks; i += 64) w += (long long)s_chunks[i] * (long long)(s_task[i].p == s_task[i].q ? s_task[i].np : s_task[i].nq);
            w = wave_sum_ll(w);
            if (t == 128) T.w_total = w;
        }
        if (n_items <= ITEM_CAP)
            for (int i = t; i < n_tasks; i += TM_THREADS)
                for (int w = s_start[i]; w < s_start[i + 1]; w++) T.item_tc[w] = (unsigned)i | ((unsigned)(w - s_start[i]) << 16);
    }
    return n_tasks;
}

// Everything the per-step kernels need that does not change from step to step lives in ONE device-resident block
// (one per layout buffer).  Kernel arguments are fetched by serialised scalar loads from the uncached kernarg
// segment (~1 us each); with 20+ arguments that prologue cost more than the kernels' work.  Now each kernel
// takes this pointer plus a handful of per-step scalars.
// one relevant contact, everything k_fin needs to price it without further index loads (32 bytes)
struct QEntry {
    unsigned idx, rel;   // contact index in this shard; mask (bit CODE_BITS*k) of the neighbours it matters to
    unsigned ci, cj;     // relevance codes of its two fragments: CODE_BITS bits per neighbour = piece id
    int fx, fy, cnt;     // fragments (bins) and observed count (float32 bits)
    int slots;           // sub-fragment slots: slx | sly << 2
};
// What k_scan stores per doubly-affected contact (16 bytes): it neither gathers the fragments' records nor waits for its stores to be
// acknowledged -- both used to end the kernel one or two memory round trips after the last such contact turned up (6 us per
// launch: tools/ab.sh, tools/stamps_hits.py).  The consumers load the records anyway; the piece codes and the neighbour mask are
// derived there (q_fetch, q_codes), and a consumer that may run WHILE the scan's last stores are in flight (k_tm's finishing block)
// validates the step's sequence tag carried by both words.
//   w0 = contact index | seq << 32            w1 = fx | fy << 20 | slots << 40 | (seq & 0xfffff) << 44
struct QRaw { unsigned long long w0, w1; };
static_assert(LABEL_BITS <= 20, "fragment ids must fit the 20-bit fields of a queue entry");
struct QSrc {
    const QRaw* queue;
    const int2* geo2;          // (id_c, flags) = first half of a Geo record
    const int* cnt;
    const PieceKey* keys;      // [K] of the step's neighbours (shared memory of the consumer)
    unsigned live;             // bit k: neighbour k is a real pair (fB != fA)
    int K;
    unsigned seq;              // the step's sequence number (low 32 bits)
    int concurrent;            // the producer may still be storing: validate the tags (bounded spin)
    int multi;                 // bins have several sub-fragments: a bin's own pixel is skipped
};
// first half: the entry itself (one round trip; fx < 0: nothing to price).  The caller then requests the two fragments' records,
// their statistics and the contact's count TOGETHER (second round trip) and calls q_codes.
__device__ __forceinline__ QEntry q_fetch(const QSrc& q, unsigned long long e, unsigned long long* __restrict__ err)
{
    QEntry qe;
    qe.rel = 0; qe.ci = 0; qe.cj = 0; qe.cnt = 0; qe.idx = 0; qe.fx = -1; qe.fy = 0; qe.slots = 0;
    unsigned long long w0, w1;
    if (!q.concurrent) {
        typedef unsigned long long v2u __attribute__((ext_vector_type(2)));
        const v2u v = *reinterpret_cast<const v2u*>(q.queue + e);
        w0 = v.x; w1 = v.y;
    } else {
        const unsigned long long* p = reinterpret_cast<const unsigned long long*>(q.queue + e);
        bool ok = false;
        for (int spin = 0; spin < (1 << 20); spin++) {
            w0 = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            w1 = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(w0 >> 32) == q.seq && (unsigned)(w1 >> 44) == (q.seq & 0xfffffu)) { ok = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) { atomicOr(err, 1ull); return qe; }
    }
    qe.idx = (unsigned)w0;
    const int fx = (int)(w1 & 0xfffffull), fy = (int)((w1 >> 20) & 0xfffffull);
    qe.fy = fy; qe.slots = (int)((w1 >> 40) & 0xfull);
    if (q.multi && fx == fy) return qe;   // a bin's own pixel is never revisited (kernels3.cu:3356-3380)
    qe.fx = fx;
    return qe;
}
// second half: piece codes of the two fragments per neighbour and the mask of the neighbours whose two contigs hold both
__device__ __forceinline__ void q_codes(const QSrc& q, QEntry& qe, const Geo& gx, const Geo& gy, int cnt)
{
    unsigned ci = 0, cj = 0;
    for (int k = 0; k < q.K; k++)
        if ((q.live >> k) & 1u) {
            ci |= (unsigned)piece_of(q.keys[k], gx.id_c, geo_pos(gx.flags)) << (CODE_BITS * k);
            cj |= (unsigned)piece_of(q.keys[k], gy.id_c, geo_pos(gy.flags)) << (CODE_BITS * k);
        }
    const unsigned nzi = (ci | (ci >> 1) | (ci >> 2)) & CODE_LSB, nzj = (cj | (cj >> 1) | (cj >> 2)) & CODE_LSB;
    qe.ci = ci; qe.cj = cj; qe.cnt = cnt;
    qe.rel = nzi & nzj;   // both ends in the neighbour's two contigs: the consumers' exact per-candidate masks do the rest
}

struct DevArgs {
    SoaPtr soa;
    long long nnz;
    int n, n_sub_total, bitmap_words, reach_bp;
    const int *row, *col, *cnt, *sub2bin, *sub2bin_multi /* nullptr when every bin has one sub-fragment */, *sub_ids;
    const int *contig_off, *perm;
    const Geo* geo;
    const Link* link;
    const int* cbase;
    const Stat* stat;
    NbTables* tabs;
    int* step_hdr;                // [k] mass work items of neighbour k, [MAXK + k] 1 = k_tm priced them already
    long long* tm_done;           // [k] sequence number of the step whose tables of neighbour k are complete
    QRaw* queue;
    long long* acc;               // K*13 running sums of the step (zeroed again by k_fin once it has read them)
    unsigned long long* counters; // [0] n_rel [1] n_items [2] queue count [5] ticket [6] error; [8..10] copy of [0..2] of the last step
    float nfpb;
    Par par;
};

// windowed mass of one fragment pair in some layout, rounded to Q ONCE per pair: the partition of the work over lanes,
// kernels and ranks is then irrelevant to the sum
__device__ __forceinline__ long long pair_mass_q(const End& X, const Stat& sx, const End& Y, const Stat& sy, float nfpb,
                                                 const Par& par)
{
    double acc = 0.0;
    for (int a = 0; a < sx.n; a++)
        for (int b = 0; b < sy.n; b++)
            acc += (double)ex_pair(X, sx, a, Y, sy, b, nfpb, par) - (double)ex_trans(stat_accu(sx, a), stat_accu(sy, b), nfpb, par);
    return to_q(acc);
}

__device__ __forceinline__ int gap_bp(const End& X, int len_x, const End& Y, int len_y)
{
    return X.start_bp < Y.start_bp ? Y.start_bp - (X.start_bp + len_x) : X.start_bp - (Y.start_bp + len_y);
}


// ---- the queued contacts.  Everything a contact needs is addressable from its queue entry: entry -> {geometry, statistics,
// the piece pair's task list} -> arithmetic.
// what the pricing of a contact needs of a task, and of a piece pair's list: the finishing block of k_tm copies them into LDS
// while it waits for the scan (price_contacts then has no dependent loads behind the queue entry and the two fragment records)
struct PTask { int p; unsigned plus, minus; Xf xp, xq; };
constexpr int PT_CAP = 64;                      // tasks per neighbour held in LDS (more: read from the tables in memory)
constexpr int PR_WORDS = 3 + N_OPS;             // tplus, tminus, old task, the new relation's task per candidate
struct PriceArgs {
    const unsigned short* pr_lds;  // [K][N_PAIRS][PR_WORDS] or nullptr
    const PTask* pt_lds;           // [K][PT_CAP]
    const int* nt_lds;             // [K] number of tasks of neighbour k
    QSrc q;                        // the queue and what expands its entries
    unsigned long long* err;       // counters[6]
    const NbTables* tabs;
    const Geo* geo;
    const Stat* stat;
    const int* lcontbp;
    long long* out;
    unsigned long long* nf;
    float nfpb;
    Par par;
};
// What a queued contact contributes to candidate `op` of neighbour k, in Q30:
//     sum over the cis relations t of its piece pair (the tasks of k_tm's per-pair list: the old one and the distinct new ones)
//         of  ([op in plus_t] - [op in minus_t]) * Q(ob * ln ex_t)
//     + ([op in tplus] - [op in tminus]) * Q(ob * ln ex_trans)
// i.e. every DISTINCT relation is evaluated once per contact -- with the transforms of the task that owns it -- and rounded once;
// candidates that lead to the same relation share the value.  The old relation (the current layout) and the trans value do not
// depend on the neighbour: evaluated once per contact; a new relation that two neighbours share (ejecting fA does not depend on
// the neighbour at all) is recognised by its inputs.  Both routines below compute exactly these terms from the same inputs, so
// their sums are bit-identical (tests/test_engine_gpu.py::test_short_contigs_finished_by_the_table_kernel).
__device__ __forceinline__ double ln_trans_of(const Stat& sx, int slx, const Stat& sy, int sly, const double* __restrict__ ln_tab, int lut_n,
                                              float nfpb, const Par& par)
{
    const int prod = stat_accu(sx, slx) * stat_accu(sy, sly);
    if ((unsigned)prod < (unsigned)lut_n) return ln_tab[prod];   // (k_ln_tab: the same expression)
    return mm_ln(par.v_inter * ((float)prod / nfpb));
}

// one contact's values: Q(ob ln ex) of the current layout, of the trans relation, and of the last new relation evaluated
struct ContactVals {
    const Geo gx, gy;
    const Stat sx, sy;
    const int slx, sly;
    const double ob;
    long long q_old, q_tr, q_memo;
    int have;             // bit 0: q_old, bit 1: q_tr, bit 2: q_memo valid
    int m_xs, m_ys, m_fl, m_lbp;
    __device__ __forceinline__ long long q_cis(const Xf& xa, const Xf& xb, float nfpb, const Par& par)
    {
        const End X = end_xf(gx, xa), Y = end_xf(gy, xb);
        const int fl = (X.fwd ? 1 : 0) | (Y.fwd ? 2 : 0) | (X.circ << 2);
        if ((have & 4) && X.start_bp == m_xs && Y.start_bp == m_ys && fl == m_fl && X.lbp == m_lbp) return q_memo;
        q_memo = to_q(ob * mm_ln(ex_pair(X, sx, slx, Y, sy, sly, nfpb, par)));
        m_xs = X.start_bp; m_ys = Y.start_bp; m_fl = fl; m_lbp = X.lbp; have |= 4;
        return q_memo;
    }
};

template <typename PA>
__device__ __forceinline__ PTask ptask_of(const PA& pa, const NbTables& T, bool in_lds, int k, int ti)
{
    if (in_lds) return pa.pt_lds[k * PT_CAP + ti];
    const Task& g = T.task[ti];
    PTask tk; tk.p = g.p; tk.plus = g.plus; tk.minus = g.minus; tk.xp = g.xp; tk.xq = g.xq;
    return tk;
}

// A few queued contacts (the finishing block of k_tm): 16 lanes per contact, lane = candidate; a lane evaluates the (at most two
// or three) relations its candidate takes part in.
__device__ __forceinline__ void price_contacts(const PriceArgs& pa, unsigned long long nq_total, int first, int n_waves, int lane)
{
    const int op = lane & 15;
    for (unsigned long long e0 = (unsigned long long)first * 4; e0 < nq_total; e0 += (unsigned long long)n_waves * 4) {
        const unsigned long long e = e0 + (lane >> 4);
        if (e >= nq_total || op >= N_OPS) continue;
        QEntry qe = q_fetch(pa.q, e, pa.err);
        if (qe.fx < 0) continue;
        const int fx = qe.fx, fy = qe.fy;
        ContactVals cv = {pa.geo[fx], pa.geo[fy], pa.stat[fx], pa.stat[fy], qe.slots & 3, (qe.slots >> 2) & 3,
                          (double)__int_as_float(pa.q.cnt[qe.idx]), 0, 0, 0, 0, 0, 0, 0, 0};
        q_codes(pa.q, qe, cv.gx, cv.gy, 0);
        if (qe.rel == 0) continue; // (both ends affected, but by no common neighbour)
        unsigned rel = qe.rel;
        while (rel) {
            const int k = (__ffs((int)rel) - 1) / CODE_BITS;
            rel &= rel - 1;
            const int p = (qe.ci >> (CODE_BITS * k)) & 7, q = (qe.cj >> (CODE_BITS * k)) & 7;
            const NbTables& T = pa.tabs[k];
            const int pr = pair_index(p, q);
            const unsigned short* P = pa.pr_lds ? pa.pr_lds + (k * N_PAIRS + pr) * PR_WORDS : nullptr;
            const unsigned tplus = P ? P[0] : T.tplus[pr], tminus = P ? P[1] : T.tminus[pr];
            const int t_old = P ? P[2] : T.pair_old[pr], t_new = P ? P[3 + op] : T.pair_owner[pr][op];
            const bool in_lds = P != nullptr && pa.nt_lds[k] <= PT_CAP;
            long long acc = 0;
            bool bad = false;
            const int sg_tr = (int)((tplus >> op) & 1u) - (int)((tminus >> op) & 1u);
            if (sg_tr != 0) {
                if (!(cv.have & 2)) { cv.q_tr = to_q(cv.ob * ln_trans_of(cv.sx, cv.slx, cv.sy, cv.sly, nullptr, 0, pa.nfpb, pa.par)); cv.have |= 2; }
                if (cv.q_tr == Q_BAD) bad = true; else acc += sg_tr * cv.q_tr;
            }
            if (t_old != 0xffff) {
                const PTask tk = ptask_of(pa, T, in_lds, k, t_old);
                const int sg = (int)((tk.plus >> op) & 1u) - (int)((tk.minus >> op) & 1u);
                if (sg != 0) {
                    if (!(cv.have & 1)) {   // (the current layout: the same value whatever the neighbour)
                        const bool straight = p == tk.p;
                        const End X = end_xf(cv.gx, straight ? tk.xp : tk.xq), Y = end_xf(cv.gy, straight ? tk.xq : tk.xp);
                        cv.q_old = to_q(cv.ob * mm_ln(ex_pair(X, cv.sx, cv.slx, Y, cv.sy, cv.sly, pa.nfpb, pa.par)));
                        cv.have |= 1;
                    }
                    if (cv.q_old == Q_BAD) bad = true; else acc += sg * cv.q_old;
                }
            }
            if (t_new != 0xffff && t_new != t_old) {   // (a new relation merged into the old one carries both signs: nothing)
                const PTask tk = ptask_of(pa, T, in_lds, k, t_new);
                const int sg = (int)((tk.plus >> op) & 1u) - (int)((tk.minus >> op) & 1u);
                if (sg != 0) {
                    const bool straight = p == tk.p;   // (the task lists its pieces larger first)
                    const long long qv = cv.q_cis(straight ? tk.xp : tk.xq, straight ? tk.xq : tk.xp, pa.nfpb, pa.par);
                    if (qv == Q_BAD) bad = true; else acc += sg * qv;
                }
            }
            if (bad) nf_flag(pa.nf, k, op);
            else if (acc != 0) atomicAdd((unsigned long long*)&pa.out[k * N_OPS + op], (unsigned long long)acc);
        }
    }
}

// Reference arithmetic, a few queued contacts (the finishing block of k_tm): 16 lanes per contact, lane = candidate.  The scan
// queued every contact with both ends in some neighbour's affected set; each is priced again under every candidate whose
// inputs differ from the current layout's:  ob (ln ex_new - ln ex_old)  (kernels3.cu:3383-3697).
__device__ __forceinline__ void price_contacts_strict(const QSrc& qs, unsigned long long* __restrict__ err, const NbTables* __restrict__ tabs, const Geo* __restrict__ geo,
                                                      const Stat* __restrict__ stat, const int* __restrict__ lcontbp, long long* __restrict__ out,
                                                      unsigned long long* __restrict__ nf, float nfpb, const Par& par, bool quirk,
                                                      unsigned long long nq_total, int first, int n_waves, int lane,
                                                      const Xf* xf_lds = nullptr, const unsigned char* crep_lds = nullptr)
{
    // (xf_lds / crep_lds: the K tables' transforms [K][N_OPS][NP] and class representatives [K][N_PAIRS][N_OPS] in LDS -- k_tm's finishing block
    // copies them while it waits for the scan; from the tables in memory they are a round trip per (contact, neighbour))
    const int op = lane & 15;
    for (unsigned long long e0 = (unsigned long long)first * 4; e0 < nq_total; e0 += (unsigned long long)n_waves * 4) {
        const unsigned long long e = e0 + (lane >> 4);
        if (e >= nq_total || op >= N_OPS) continue;
        QEntry qe = q_fetch(qs, e, err);
        if (qe.fx < 0) continue;
        const int fx = qe.fx, fy = qe.fy, slx = qe.slots & 3, sly = (qe.slots >> 2) & 3;
        const Geo gx = geo[fx], gy = geo[fy];
        const Stat sx = stat[fx], sy = stat[fy];
        q_codes(qs, qe, gx, gy, qs.cnt[qe.idx]);
        if (qe.rel == 0) continue;
        const End X0 = end_old(gx, ((gx.flags >> 1) & 1) ? lcontbp[fx] : 0), Y0 = end_old(gy, ((gy.flags >> 1) & 1) ? lcontbp[fy] : 0);
        const float ex_old = ex_pair_ref(X0, sx, slx, fx, Y0, sy, sly, fy, nfpb, par, quirk);
        const double ln_old = mm_ln(ex_old), ob = (double)__int_as_float(qe.cnt);
        unsigned rel = qe.rel;
        while (rel) {
            const int k = (__ffs((int)rel) - 1) / CODE_BITS;
            rel &= rel - 1;
            const int p = (qe.ci >> (CODE_BITS * k)) & 7, q = (qe.cj >> (CODE_BITS * k)) & 7;
            unsigned char cls;
            Xf xa, xb;
            if (xf_lds) {
                cls = crep_lds[(k * N_PAIRS + pair_index(p, q)) * N_OPS + op];
                xa = xf_lds[(k * N_OPS + op) * NP + p]; xb = xf_lds[(k * N_OPS + op) * NP + q];
            } else {
                const NbTables& T = tabs[k];
                cls = T.crep[pair_index(p, q)][op];
                xa = T.xf[op][p]; xb = T.xf[op][q];     // (requested together with the class entry)
                keep_xf(xa, xb);
            }
            if (cls == CREP_OLD) continue;   // the inputs of the current layout: the same value
            const End X = end_xf(gx, xa), Y = end_xf(gy, xb);
            const float ex_new = ex_pair_ref(X, sx, slx, fx, Y, sy, sly, fy, nfpb, par, quirk);
            if (ex_new == ex_old) continue;
            const long long qv = to_q(ob * (mm_ln(ex_new) - ln_old));
            if (qv == Q_BAD) nf_flag(nf, k, op);
            else if (qv != 0) atomicAdd((unsigned long long*)&out[k * N_OPS + op], (unsigned long long)qv);
        }
    }
}

// MANY queued contacts (k_fin; long contigs queue millions per step and this pricing was a third of the step): lane = contact,
// and a contact is evaluated once per distinct relation, not once per candidate -- the 13 candidates of a neighbour lead to a
// handful of distinct relations of a piece pair.  The values are added per (neighbour, task) / (neighbour, pair) in LDS (`S`:
// K x (MAX_TASKS + N_PAIRS) words) and folded into the per-candidate sums with the tasks' masks once per block
// (fold_contact_sums).  Short queues: B < 64 contacts per wave and 64 / B lanes per contact, which share its relations.
constexpr int S_PER_K = MAX_TASKS + N_PAIRS;
constexpr int N_WQ = 8, WQ_STRIDE = 16;   // k_fin's work queues: classes of units (U % N_WQ), one counter (on its own line) each
__device__ __forceinline__ int contact_batch_size(unsigned long long nq_total, int n_waves)
{
    return nq_total >= 64ull * (unsigned long long)n_waves ? 64 : (nq_total >= 16ull * (unsigned long long)n_waves ? 16 : 4);
}
// one batch: the B queue entries from b0 on
__device__ __forceinline__ void price_contact_batch(const PriceArgs& pa, long long* __restrict__ S, const double* __restrict__ ln_tab, int lut_n,
                                                    unsigned long long nq_total, unsigned long long b0, int B, int lane)
{
    const int ent = lane % B, slice = lane / B, n_slices = 64 / B;
    {
        const unsigned long long e = b0 + ent;
        if (e >= nq_total) return;
        QEntry qe = q_fetch(pa.q, e, pa.err);
        if (qe.fx < 0) return;
        const int fx = qe.fx, fy = qe.fy;
        ContactVals cv = {pa.geo[fx], pa.geo[fy], pa.stat[fx], pa.stat[fy], qe.slots & 3, (qe.slots >> 2) & 3,
                          (double)__int_as_float(pa.q.cnt[qe.idx]), 0, 0, 0, 0, 0, 0, 0, 0};
        q_codes(pa.q, qe, cv.gx, cv.gy, 0);
        if (qe.rel == 0) return; // (both ends affected, but by no common neighbour)
        unsigned rel = qe.rel;
        while (rel) {
            const int k = (__ffs((int)rel) - 1) / CODE_BITS;
            rel &= rel - 1;
            const int p = (qe.ci >> (CODE_BITS * k)) & 7, q = (qe.cj >> (CODE_BITS * k)) & 7;
            const NbTables& T = pa.tabs[k];
            const int pr = pair_index(p, q);
            if (slice == 0) {   // the two values that do not depend on the neighbour
                const unsigned tmask = (unsigned)(T.tplus[pr] | T.tminus[pr]);
                if (tmask != 0) {
                    if (!(cv.have & 2)) { cv.q_tr = to_q(cv.ob * ln_trans_of(cv.sx, cv.slx, cv.sy, cv.sly, ln_tab, lut_n, pa.nfpb, pa.par)); cv.have |= 2; }
                    if (cv.q_tr == Q_BAD) nf_flag_ops(pa.nf, k, tmask);
                    else if (cv.q_tr != 0) atomicAdd((unsigned long long*)&S[k * S_PER_K + MAX_TASKS + pr], (unsigned long long)cv.q_tr);
                }
                const int t_old = T.pair_old[pr];
                if (t_old != 0xffff) {
                    const Task& tk = T.task[t_old];
                    if (!(cv.have & 1)) {
                        const bool straight = p == tk.p;
                        const End X = end_xf(cv.gx, straight ? tk.xp : tk.xq), Y = end_xf(cv.gy, straight ? tk.xq : tk.xp);
                        cv.q_old = to_q(cv.ob * mm_ln(ex_pair(X, cv.sx, cv.slx, Y, cv.sy, cv.sly, pa.nfpb, pa.par)));
                        cv.have |= 1;
                    }
                    if (cv.q_old == Q_BAD) nf_flag_ops(pa.nf, k, tk.plus | tk.minus);
                    else if (cv.q_old != 0) atomicAdd((unsigned long long*)&S[k * S_PER_K + t_old], (unsigned long long)cv.q_old);
                }
            }
            // (several lanes per contact: lane 0 of them has the two shared values above, the others share the new relations)
            const int n = T.pair_n[pr];
            for (int i = n_slices > 1 ? slice - 1 : 0; i >= 0 && i < n; i += n_slices > 1 ? n_slices - 1 : 1) {
                const int t = T.pair_task[pr][i];
                const Task& tk = T.task[t];
                const bool straight = p == tk.p;   // (the task lists its pieces larger first)
                const long long qv = cv.q_cis(straight ? tk.xp : tk.xq, straight ? tk.xq : tk.xp, pa.nfpb, pa.par);
                if (qv == Q_BAD) nf_flag_ops(pa.nf, k, tk.plus | tk.minus);
                else if (qv != 0) atomicAdd((unsigned long long*)&S[k * S_PER_K + t], (unsigned long long)qv);
            }
        }
    }
}

// the block's per-relation sums of price_contacts_bulk -> its per-candidate sums (acc: K * N_OPS words in LDS)
__device__ __forceinline__ void fold_contact_sums(const NbTables* __restrict__ tabs, const long long* __restrict__ S, long long* __restrict__ acc, int K)
{
    for (int e = threadIdx.x; e < K * S_PER_K; e += TM_THREADS) {
        const long long v = S[e];
        if (v == 0) continue;
        const int k = e / S_PER_K, j = e - k * S_PER_K;
        unsigned plus, minus;
        if (j < MAX_TASKS) { plus = tabs[k].task[j].plus; minus = tabs[k].task[j].minus; }
        else { plus = tabs[k].tplus[j - MAX_TASKS]; minus = tabs[k].tminus[j - MAX_TASKS]; }
        for (int op = 0; op < N_OPS; op++) {
            const long long sg = (long long)((plus >> op) & 1u) - (long long)((minus >> op) & 1u);   // (a contact counts with the NEW layout's sign)
            if (sg != 0) atomicAdd((unsigned long long*)&acc[k * N_OPS + op], (unsigned long long)(sg * v));
        }
    }
}

__global__ void k_ln_tab(double* __restrict__ tab, int n, float nfpb, Par par)
{
    const int p = blockIdx.x * TM_THREADS + threadIdx.x;
    if (p < n) tab[p] = mm_ln(par.v_inter * ((float)p / nfpb));
}

constexpr long long Q_STEP_FAILED = 1ll << 32;   // added to d_q_out's first not-finite flag word by a rank whose step failed (hand_out)
constexpr int X_SLOT_WORDS = 512;         // exchange slot: sequence word + MAXK*13 sums [+ a word of k_strict_flat]; from word X_COARSE the coarse sums; 4 KB
constexpr int X_COARSE = 256;
constexpr int X_WHY = 2 + MAXK * N_OPS;   // a failed step's reason bits (counters[6]), behind the sums and k_strict_flat's word
static_assert(X_WHY < X_COARSE && 2 + MAXK * N_OPS <= X_COARSE && X_COARSE + MAXK * N_OPS < X_SLOT_WORDS, "exchange slot too small");
// the last block of a step: read the K*13 sums, reset the accumulators and counters for the next step, hand the sums
// out -- to d_q_out (device; the caller all-reduces them) or to PINNED HOST memory followed by the step's sequence
// number (the host spins on that word instead of paying for a device->host copy and a stream-synchronise wake-up).
// sync[0] = k_tm ticket, sync[1] = finished k_scan blocks.
__device__ __forceinline__ void hand_out(long long* out, unsigned long long* counters, unsigned long long* sync, int K,
                                         long long* d_q_out, volatile long long* host_res, long long seq)
{
    const unsigned long long why = atomicAdd(&counters[6], 0ull);   // 1 / 4 / 8: an in-kernel wait ran out (tables or scan / relabel flag / k_gprep's word); 2: a work list overflowed
    const bool failed = why != 0ull;
    if (host_res && threadIdx.x == 0) host_res[X_WHY] = (long long)why;
    for (int i = threadIdx.x; i < K * N_OPS; i += TM_THREADS) {
        long long v = (long long)atomicExch((unsigned long long*)&out[i], 0ull); // read the final sum, reset for the next step
        const long long c = (long long)atomicExch((unsigned long long*)&out[MAXK * N_OPS + i], 0ull);   // the coarse sum (to_coarse), zero but for a rare candidate
        const bool flagged = ((atomicAdd(&counters[NF_OFF + (i >> 6)], 0ull) >> (i & 63)) & 1ull) != 0ull; // a term was not finite / out of range
        if (host_res) { host_res[1 + i] = flagged ? Q_NAN : v; host_res[X_COARSE + i] = c; }
        else { d_q_out[i] = flagged ? 0ll : v; d_q_out[MAXK * N_OPS + i] = c; d_q_out[2 * MAXK * N_OPS + i] = flagged ? 1ll : 0ll; }
    }
    __syncthreads();
    // a FAILED step (counters[6]: a kernel of the step gave up waiting, a list overflowed) handed out to a device buffer: the failure must
    // survive the ranks' all-reduce, or the partial sums come back as valid scores.  It travels in the first flag word as 2^32 -- a legitimate
    // flag count is at most the number of ranks -- so the summed buffer says "some rank failed" on EVERY rank (Q_STEP_FAILED: k_qout_pub,
    // graal_amd/sampler.py)
    if (!host_res && failed && threadIdx.x == 0) d_q_out[2 * MAXK * N_OPS] += Q_STEP_FAILED;
    if (threadIdx.x >= 4 && threadIdx.x < 7) counters[NF_OFF + threadIdx.x - 4] = 0;
    if (threadIdx.x < 3) counters[8 + threadIdx.x] = atomicExch(&counters[threadIdx.x], 0ull);
    if (threadIdx.x == 3) { counters[5] = 0; counters[6] = 0; sync[0] = 0; }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0 && host_res) { host_res[0] = failed ? -seq : seq; __threadfence_system(); }
}

constexpr long long NEED_FIN = 1ll << 62; // published instead of the sums: k_tm left work for k_fin, launch it
constexpr long long GAVE_UP = 1ll << 61;  // (with NEED_FIN) k_tm's last block stopped waiting for k_scan
constexpr long long NEED_GEOM = 1ll << 60; // (with NEED_FIN) ... because a neighbour's set is beyond k_tm's own pricing: the layout says so, every rank sees the same
constexpr int FIN_INLINE_Q = 64;          // queued contacts the finishing block of k_tm prices itself

// ------------------------------------------------------------------ per-step kernels
// k_tm (K blocks, launched on the auxiliary stream so that it overlaps k_scan): block k builds the tables of neighbour k
// and, when the expected-mass work of that neighbour is small (<= INLINE_PAIRS fragment pairs: the regime of short
// contigs), prices it on the spot -- one thread per fragment pair -- instead of leaving it to k_fin.
struct Neigh { int fB[MAXK]; };
// fB[j] of a kernel-argument Neigh (a select chain: a dynamically indexed kernel argument is fetched from memory)
__device__ __forceinline__ int sel_nb(const Neigh& nb, int j)
{
    int v = nb.fB[0];
#pragma unroll
    for (int i = 1; i < MAXK; i++) v = j == i ? nb.fB[i] : v;
    return v;
}

struct TmArgs { // first-needed pointers by value (see ScanArgs)
    const int* perm;               // (position index; by value like the others: pointers read from DevArgs in memory are generic -> FLAT loads)
    int reach_bp;
    long long* tm_done;
    const Geo* geo;
    const Link* link;
    const int *cbase, *mates;
    NbTables* tabs;
    int* step_hdr;
    unsigned long long* sync;      // [0] k_tm ticket (+ 2^16 per neighbour left to k_fin)
    const unsigned* flags;         // k_scan's per-block completion flags
    volatile long long* host_res;  // non-null: the last block of k_tm finishes the step itself when the work is small
    int n_scan_blocks;
    const unsigned long long* done; // non-null: k_scan's blocks count themselves here (block b in counter b % n_done, one 128-byte line
                                    // each); the step's scan is complete when every counter has reached its target
    int n_done;
    unsigned long long done_target[N_DONE];
    int wait_ticks;                // how long that block waits for k_scan (100 MHz ticks)
    int strict;                    // GRAAL_MODE_STRICT: reference arithmetic (same_inputs classes; small sets priced here, the rest by k_strict)
    int quirk;                     // GRAAL_MODE_REF_TRANS_ACCU
    const long long* nc;           // number of contigs of the ranked layout, on the device (max_id < 0: read it here)
    // graal_step's deferred flow: the kernel is launched right behind the relabel WITHOUT an event in between (a cross-stream
    // event wait costs ~10 us of start latency here); its blocks spin until k_scan -- stream-ordered behind the relabel --
    // announces that it has started (relabel_flag == relabel_seq).  0 = ordered by the host (event / synchronisation).
    const unsigned long long* relabel_flag;
    unsigned long long relabel_seq;
    unsigned long long relabel_wait_ticks;
    // ... and one extra block publishes the layout statistics the commit kernel left in rows of partials (see k_incr)
    const long long* part;
    int n_part;
    volatile long long* stats_host;
    long long stats_seq;
    long long* stats_dev;
    int strict_inline_m;           // largest affected set k_tm prices itself in reference arithmetic (-1: none, k_strict_dense validation)
    int stage_tables;              // reference arithmetic: the finishing block copies the tables' transforms and classes to LDS while it waits for the scan
    // the finishing block's pointers, by value too
    unsigned long long* counters;
    const QRaw* queue;
    const int* cnt;
    int multi;                     // bins have several sub-fragments
    const Stat* stat;
    const int* lcontbp;
    long long* acc;
    float nfpb;
    Par par;
};

// In the regime of short contigs a step leaves a handful of queued contacts and no mass work beyond what the table
// blocks price themselves.  Then the LAST table block to finish waits for k_scan's blocks (a counter), prices the queued
// contacts and publishes the result: no third kernel, no launch gap.  Otherwise it publishes NEED_FIN and the host launches
// k_fin with the whole chip.
__global__ __launch_bounds__(256) void k_tm(const DevArgs* __restrict__ A, TmArgs ta, int fA, Neigh nb, int K, int max_id, int rank,
                                             int world, long long seq)
{
    __shared__ Task s_task[MAX_TASKS];
    __shared__ int s_pp[MAX_TASKS + 1];
    const int k = blockIdx.x, t = threadIdx.x;
    if (k > K || (k == K && ta.part == nullptr)) return;
    if (ta.relabel_seq != 0ull) {
        // launched without an event behind the relabel: wait until k_scan (stream-ordered behind it) says it has started.  Bounded:
        // if the two kernels cannot be resident together (a tool serialising dispatches) the step is flagged as failed and the
        // host repeats it with an event (eval_sync).
        __shared__ int s_go;
        if (t == 0) {
            int go = 0;
            // (a deadlock guard, not a latency budget: 20 ms -- the first launch of a kernel, an allocation or a profiler can hold
            // the engine's stream for milliseconds)
            const unsigned long long t_end = wall_clock64() + ta.relabel_wait_ticks;
            for (;;) {
                if (__hip_atomic_load(ta.relabel_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == ta.relabel_seq) { go = 1; break; }
                if (wall_clock64() > t_end) break;
                __builtin_amdgcn_s_sleep(2);
            }
            s_go = go;
            if (!go) atomicOr(&ta.counters[6], 4ull);
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (!s_go) {
            // nothing can be built; take the ticket so that the last block still reports (a failed step: hand_out sees counters[6])
            if (ta.host_res != nullptr && t == 0 && k < K) {
                const unsigned long long after = atomicAdd(&ta.sync[0], 1ull) + 1ull;
                if ((after & 0xffffull) == (unsigned long long)K) { ta.sync[0] = 0; __threadfence_system(); ta.host_res[0] = -seq; __threadfence_system(); }
            }
            return;
        }
    }
    if (k == K) {   // the extra block of graal_step's deferred flow: the layout statistics (the commit kernel's rows), off everybody's path
        if (t < 64) publish_partials(ta.part, ta.n_part, ta.stats_dev, ta.stats_host, ta.stats_seq, t);
        return;
    }
    if (max_id < 0) max_id = (int)ta.nc[0] - 1;   // (launched before the host had the statistics: the relabel left the contig count)
    STAMP(0, k == 0 && t == 0);
    NbTables& T = ta.tabs[k];
    const int my_fB = sel_nb(nb, k);
    __shared__ SmallCtx sc;
    __shared__ long long s_acc[N_OPS];
    if (t < N_OPS) s_acc[t] = 0;
    TabViews tv;
    const int n_tasks = tables_block(ta.geo, ta.link, ta.cbase, ta.mates, ta.stat, fA, my_fB, max_id, T, k, ta.step_hdr, s_task, s_pp, sc,
                                     ta.strict, ta.quirk, tv);
    const int total = s_pp[n_tasks];
    // reference arithmetic: the set is contig(fA) u contig(fB), every pair of different bins in it is priced under every class
    const int s_lenA = sc.len[0], s_lenB = (my_fB == fA || tv.B0->id_c == tv.A0->id_c) ? 0 : sc.len[1];
    const int s_m = my_fB == fA ? 0 : s_lenA + s_lenB;
    const int s_pairs = s_m * (s_m - 1) / 2;   // (only used when small)
    const bool inl = ta.strict ? s_m <= ta.strict_inline_m : total <= INLINE_PAIRS;
    if (ta.strict && t == 0) T.set_m = inl ? 0 : s_m;
    STAMP(1, k == 0 && t == 0);
    if (ta.strict && inl && s_m > 1) {
        const int* __restrict__ perm = ta.perm;
        const Geo* __restrict__ geo = ta.geo;
        const Stat* __restrict__ stat = ta.stat;
        const float nfpb = ta.nfpb;
        const Par par = ta.par;
        const int reach_bp = ta.reach_bp;
        const bool quirk = ta.quirk != 0;
        const Rec A0 = *tv.A0, B0 = *tv.B0;
        const PieceKey key = T.key;
        for (int it = rank + world * t; it < s_pairs * N_OPS; it += world * 256) {
            const int e = it / N_OPS, op = it - e * N_OPS;
            int j = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)e)) * 0.5f);   // pair e = (i < j): j (j - 1) / 2 <= e < j (j + 1) / 2
            while (j * (j - 1) / 2 > e) j--;
            while (j * (j + 1) / 2 <= e) j++;
            const int i = e - j * (j - 1) / 2;
            Geo gx, gy;
            Stat sx, sy;
            int fx, fy;
            if (sc.small) {   // both contigs sit in LDS, in position order
                const int wx = i < s_lenA ? 0 : 1, wy = j < s_lenA ? 0 : 1;
                const int ix = i - wx * s_lenA, iy = j - wy * s_lenA;
                fx = sc.mates[wx][ix]; fy = sc.mates[wy][iy];
                gx = sc.geo[wx * N_MATES + ix]; gy = sc.geo[wy * N_MATES + iy]; sx = sc.stat[wx * N_MATES + ix]; sy = sc.stat[wy * N_MATES + iy];
            } else {
                fx = perm[i < s_lenA ? sc.base[0] + i : sc.base[1] + (i - s_lenA)];
                fy = perm[j < s_lenA ? sc.base[0] + j : sc.base[1] + (j - s_lenA)];
                gx = geo[fx]; gy = geo[fy]; sx = stat[fx]; sy = stat[fy];
            }
            if (sx.n == 0 || sy.n == 0) continue;   // (copies of repeated bins: priced by k_rep_delta)
            const int px = piece_of(key, gx.id_c, geo_pos(gx.flags)), py = piece_of(key, gy.id_c, geo_pos(gy.flags));
            const int pr = pair_index(px, py);
            if (tv.crep[pr * N_OPS + op] != op) continue;   // not the representative of its class (or: the inputs of the current layout)
            const End X0 = end_old(gx, ((gx.flags >> 1) & 1) ? (gx.id_c == A0.id_c ? A0.l_cont_bp : B0.l_cont_bp) : 0);
            const End Y0 = end_old(gy, ((gy.flags >> 1) & 1) ? (gy.id_c == A0.id_c ? A0.l_cont_bp : B0.l_cont_bp) : 0);
            const End X = end_xf(gx, tv.xf[op * NP + px]), Y = end_xf(gy, tv.xf[op * NP + py]);
            const unsigned ops = tv.cmask[pr * N_OPS + op];
            const bool near_old = X0.label == Y0.label && gap_bp(X0, gx.len_bp, Y0, gy.len_bp) <= reach_bp;
            const bool near_new = X.label == Y.label && gap_bp(X, gx.len_bp, Y, gy.len_bp) <= reach_bp;
            // beyond the window (or between two contigs) before AND after, every slot pair has the trans value both times: the
            // difference is exactly zero -- unless the reference's trans-branch RF-count indexing is on and a bin's counts differ
            if (!near_old && !near_new && !(quirk && (!stat_uniform(sx) || !stat_uniform(sy)))) continue;
            double v;
            const long long qv = strict_pair_q(X0, Y0, X, Y, sx, fx, sy, fy, nfpb, par, quirk, v);
            if (qv == 0) continue;
            if (qv == Q_BAD) { coarse_add_ops(ta.acc, ta.counters + NF_OFF, k, ops, v); continue; }
            unsigned o2 = ops;
            while (o2) { const int b = __ffs((int)o2) - 1; o2 &= o2 - 1; atomicAdd((unsigned long long*)&s_acc[b], (unsigned long long)qv); }
        }
    }
    if (!ta.strict && inl && total > 0) {
        const int* __restrict__ perm = ta.perm;
        const Geo* __restrict__ geo = ta.geo;
        const Stat* __restrict__ stat = ta.stat;
        const float nfpb = ta.nfpb;
        const Par par = ta.par;
        const int reach_bp = ta.reach_bp;
        // (sums of the block in LDS: a handful of lanes have anything to add, and 13 wave reductions cost microseconds)
        for (int i = rank + world * t; i < total; i += world * 256) {
            int lo_t = 0, hi_t = n_tasks - 1; // last task with s_pp <= i
            while (lo_t < hi_t) { const int mid = (lo_t + hi_t + 1) >> 1; if (s_pp[mid] <= i) lo_t = mid; else hi_t = mid - 1; }
            const Task& tk = s_task[lo_t];
            const bool same = tk.p == tk.q;
            const int li = i - s_pp[lo_t], width = same ? tk.np : tk.nq;
            const int ix = li / width, iy = li - ix * width;
            if (same && iy <= ix) continue;
            const int px = tk.base_p + ix, py = (same ? tk.base_p : tk.base_q) + iy; // slots of the position index
            Geo gx, gy;
            Stat sx, sy;
            if (sc.small) { // both fragments sit in LDS: slot -> (contig, position)
                const int wx = (px >= sc.base[0] && px < sc.base[0] + sc.len[0]) ? 0 : 1;
                const int wy = (py >= sc.base[0] && py < sc.base[0] + sc.len[0]) ? 0 : 1;
                const int jx = wx * N_MATES + (px - sc.base[wx]), jy = wy * N_MATES + (py - sc.base[wy]);
                gx = sc.geo[jx]; gy = sc.geo[jy]; sx = sc.stat[jx]; sy = sc.stat[jy];
            } else {
                const int fx = perm[px], fy = perm[py];
                gx = geo[fx]; gy = geo[fy]; sx = stat[fx]; sy = stat[fy];
            }
            const End X = end_xf(gx, tk.xp), Y = end_xf(gy, same ? tk.xp : tk.xq);
            if (gap_bp(X, gx.len_bp, Y, gy.len_bp) > reach_bp) continue;
            const long long qv = pair_mass_q(X, sx, Y, sy, nfpb, par);
            if (qv == 0) continue;
            if (qv == Q_BAD) { nf_flag_ops(ta.counters + NF_OFF, k, tk.minus ^ tk.plus); continue; }
            // logL = contacts - mass: the NEW layout's mass counts negative, the OLD one positive
            unsigned ops = tk.minus | tk.plus;
            while (ops) {
                const int op = __ffs((int)ops) - 1;
                ops &= ops - 1;
                const long long sgn = (long long)((tk.minus >> op) & 1u) - (long long)((tk.plus >> op) & 1u);
                if (sgn != 0) atomicAdd((unsigned long long*)&s_acc[op], (unsigned long long)(sgn * qv));
            }
        }
    }
    __syncthreads();
    if (t < N_OPS && s_acc[t] != 0) atomicAdd((unsigned long long*)&ta.acc[k * N_OPS + t], (unsigned long long)s_acc[t]);
    __syncthreads();
    STAMP(2, k == 0 && t == 0);
    if (t == 0) {
        if (inl && rank == 0) atomicAdd(&ta.counters[1], (unsigned long long)ta.step_hdr[k]);
        ta.step_hdr[MAXK + k] = inl ? 1 : 0;
        // release: the tables of neighbour k are complete; the word carries the work-list header for k_fin
        const unsigned long long w = ((unsigned long long)(unsigned)seq << 32) | (inl ? 0x80000000ull : 0ull) | (unsigned long long)(unsigned)ta.step_hdr[k];
        __hip_atomic_store((unsigned long long*)&ta.tm_done[k], w, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        STAMP_MAX(3, true);
    }
    if (ta.host_res == nullptr) return;
    // ---- finisher: the last table block ----
    __shared__ int s_fin; // 0 = not the last block, 1 = finish here, 2 = leave it to k_fin
    __shared__ unsigned long long s_nq;
    // what expands the queue's entries (q_fetch + q_codes): the neighbours' piece keys -- fetched while the block waits for the scan
    __shared__ PieceKey s_qkeys[MAXK];
    __shared__ unsigned s_qlive;
    // (dynamic LDS, requested only when a finisher can exist: a k_tm block that k_fin's resident blocks wait for must stay small
    // enough to be placed next to them -- tm_fin_dyn_lds / fin_blocks_no_wait)
    extern __shared__ long long s_tm_dyn[];
    PTask* const s_ptl = reinterpret_cast<PTask*>(s_tm_dyn);
    unsigned short* const s_prl = reinterpret_cast<unsigned short*>(s_ptl + MAXK * PT_CAP);
    int* const s_ntl = reinterpret_cast<int*>(s_prl + MAXK * N_PAIRS * PR_WORDS);
    if (t == 0) {
        int mode = 0;
        __threadfence();
        // one atomic: count of finished table blocks in the low 16 bits, neighbours left to k_fin above
        const unsigned long long ticket = atomicAdd(&ta.sync[0], inl ? 1ull : (1ull + (1ull << 16)));
        const unsigned long long after = ticket + (inl ? 1ull : (1ull + (1ull << 16)));
        if ((after & 0xffffull) == (unsigned long long)K) mode = (after >> 16) ? 2 : 1;
        s_fin = mode;
        s_qlive = 0;
    }
    __syncthreads();
    if (s_fin == 0) return;
    if (s_fin == 2) {
        // some neighbour's work is left to k_fin / k_strict whatever the scan finds: say so AT ONCE -- the host's launches then
        // queue up behind the scan on its stream while it is still running, instead of starting their trip when it has ended
        // (contigs of 20-100 bins, the middle of a run: ~10 us of a 130 us step)
        if (t == 0) { ta.sync[0] = 0; __threadfence_system(); ta.host_res[0] = seq | NEED_FIN | NEED_GEOM; __threadfence_system(); }
        return;
    }
    {   // all blocks of k_scan done?  Every thread polls its share of the flags, for a bounded TIME: when the two kernels
        // do not actually run concurrently (a profiler or debugger serialising dispatches, streams sharing a hardware
        // queue) the scan cannot even start before this block exits -- then the step is handed to k_fin, which the host
        // launches behind the scan.
        // (only wave 0 polls -- all its flag loads of a round are in flight together, no block barrier inside the loop; the
        // other waves wait at the barrier below)
        __shared__ int s_seen;
        // the other waves meanwhile copy what the pricing needs of all K tables into LDS (the other blocks released their
        // tables before they took their tickets)
        if (t >= 64 && t < 64 + K && s_fin == 1) {   // (the other blocks released their tables before they took their tickets)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            s_qkeys[t - 64] = ta.tabs[t - 64].key;
            if (ta.tabs[t - 64].fB != fA) atomicOr(&s_qlive, 1u << (t - 64));
        }
        if (t >= 64 && s_fin == 1 && ta.strict && ta.stage_tables) {   // reference arithmetic: the transforms and class representatives of all K tables (22 KB at K = 10)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const int u = t - 64, nu = 256 - 64;
            Xf* const sx = reinterpret_cast<Xf*>(s_tm_dyn);
            unsigned char* const sc = reinterpret_cast<unsigned char*>(sx + MAXK * N_OPS * NP);
            for (int e = u; e < K * N_OPS * NP; e += nu) { const int k = e / (N_OPS * NP), r = e - k * (N_OPS * NP); sx[e] = ta.tabs[k].xf[r / NP][r % NP]; }
            for (int e = u; e < K * N_PAIRS * N_OPS; e += nu) { const int k = e / (N_PAIRS * N_OPS), r = e - k * (N_PAIRS * N_OPS); sc[e] = ta.tabs[k].crep[r / N_OPS][r % N_OPS]; }
        }
        if (t >= 64 && s_fin == 1 && !ta.strict) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const int u = t - 64, nu = 256 - 64;
            for (int e = u; e < K * N_PAIRS; e += nu) {
                const int k = e / N_PAIRS, pr = e - k * N_PAIRS;
                const NbTables& T = ta.tabs[k];
                unsigned short* P = s_prl + e * PR_WORDS;
                P[0] = T.tplus[pr]; P[1] = T.tminus[pr]; P[2] = T.pair_old[pr];
                for (int op = 0; op < N_OPS; op++) P[3 + op] = T.pair_owner[pr][op];
            }
            for (int e = u; e < K * PT_CAP; e += nu) {
                const int k = e / PT_CAP, i = e - k * PT_CAP;
                const NbTables& T = ta.tabs[k];
                if (i < T.n_tasks) { const Task& g = T.task[i]; PTask tk; tk.p = g.p; tk.plus = g.plus; tk.minus = g.minus; tk.xp = g.xp; tk.xq = g.xq; s_ptl[e] = tk; }
            }
            if (u < K) s_ntl[u] = ta.tabs[u].n_tasks;
        }
        if (t < 64) {
            bool ok0 = false;
            const unsigned long long t_end = wall_clock64() + (unsigned long long)ta.wait_ticks;
            for (;;) {
                unsigned missing = 0; // (no short-circuit: the loads of a round must not wait for each other)
                if (ta.done) { if (t < ta.n_done) missing = __hip_atomic_load(ta.done + DONE_STRIDE * t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ta.done_target[t]; }
                else
                    for (int b = t; b < ta.n_scan_blocks; b += 64)
                        missing |= __hip_atomic_load(&ta.flags[FLAG_STRIDE * b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ^ (unsigned)seq;
                if (__ballot(missing != 0) == 0) { ok0 = true; break; }
                if (__ballot(wall_clock64() > t_end) != 0) break; // wave-uniform exit
            }
            if (t == 0) s_seen = ok0 ? 1 : 0;
        }
        __syncthreads();
        const bool ok = s_seen != 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (t == 0) {
            s_nq = ok ? __hip_atomic_load(&ta.counters[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            if (ok && s_nq > (unsigned long long)FIN_INLINE_Q) s_fin = 2;
            if (!ok) s_fin = 3; // gave up waiting
        }
    }
    __syncthreads();
    STAMP(4, t == 0);
    if (s_fin >= 2) { // (nothing was reset: k_fin finds the step as k_scan and k_tm left it)
        if (t == 0) { ta.sync[0] = 0; __threadfence_system(); ta.host_res[0] = seq | NEED_FIN | (s_fin == 3 ? GAVE_UP : 0ll); __threadfence_system(); }
        return;
    }
    // the queue's entries are expanded here (q_fetch + q_codes): the neighbours' piece keys (above), and a validated sequence tag per
    // entry -- the scan's last stores may still be on their way
    QSrc qs;
    qs.queue = ta.queue; qs.geo2 = reinterpret_cast<const int2*>(ta.geo); qs.cnt = ta.cnt; qs.keys = s_qkeys; qs.live = s_qlive; qs.K = K;
    qs.seq = (unsigned)seq; qs.concurrent = 1; qs.multi = ta.multi;
    if (ta.strict)
        price_contacts_strict(qs, ta.counters + 6, ta.tabs, ta.geo, ta.stat, ta.lcontbp, ta.acc, ta.counters + NF_OFF, ta.nfpb, ta.par, ta.quirk != 0,
                              s_nq, t >> 6, 4, t & 63, ta.stage_tables ? reinterpret_cast<const Xf*>(s_tm_dyn) : nullptr,
                              reinterpret_cast<const unsigned char*>(reinterpret_cast<const Xf*>(s_tm_dyn) + MAXK * N_OPS * NP));
    else {
        PriceArgs pa;
        pa.pr_lds = s_prl; pa.pt_lds = s_ptl; pa.nt_lds = s_ntl;
        pa.q = qs; pa.err = ta.counters + 6; pa.tabs = ta.tabs; pa.geo = ta.geo; pa.stat = ta.stat; pa.lcontbp = ta.lcontbp;
        pa.out = ta.acc; pa.nf = ta.counters + NF_OFF; pa.nfpb = ta.nfpb; pa.par = ta.par;
        price_contacts(pa, s_nq, t >> 6, 4, t & 63);
    }
    __syncthreads(); // (waits for this block's atomics: they are complete, at the memory side, before hand_out reads the sums)
    STAMP(5, t == 0);
    hand_out(ta.acc, ta.counters, ta.sync, K, nullptr, ta.host_res, seq);
    STAMP(6, t == 0);
}

// k_scan: ONE streaming pass over this rank's contact list for all 13*K candidates of the step.
// A contact matters to neighbour k iff its two ends lie in DIFFERENT pieces of k (or in one piece whose circular model
// may change).  Every block first marks the "affected" ids -- the fragments of contig(fA) and of the contigs of the
// neighbours -- in a 1-bit-per-id LDS bitmap (6 KB for 50k fragments) straight from the position index, so the pass
// needs no preparation kernel.  Then, cheapest test first:
//   1. row id in the bitmap?  The list is sorted by row, so whole waves fail this test together and never load their
//      `col` words;
//   2. col id in the bitmap?
//   3. piece ids of both fragments (8-byte gathers of the geometry records) -> nibble mask of neighbours.
// Survivors are appended to a queue (wave-aggregated atomics) for k_fin.  Counts are not read here at all.
#ifndef GRAAL_SCAN_PRE
#define GRAAL_SCAN_PRE 4
#endif
// groups of row words requested ABOVE the prologue: 4 = the first batch, 8 = the first two, 0 = none (HBM idle during the
// prologue).
constexpr int SCAN_PRE = GRAAL_SCAN_PRE;

struct ScanArgs { // by value: kernel-argument pointers are known to be GLOBAL (pointers read from a struct in memory are
                  // generic -> FLAT instructions, whose out-of-order completion forces full counter drains on the hot path)
    const Geo* geo;
    const Link* link;
    const int *cbase, *perm, *mates, *sub_ids, *sub2bin, *cnt;
    const int4 *row4, *col4;
    const int2* geo2;             // (id_c, flags) = first half of a Geo record
    QRaw* queue;
    unsigned long long* counters;
    unsigned* flags;              // [block] sequence number of the last step this block finished
    unsigned seq32;
    unsigned long long* done;     // non-null: completion = one fire-and-forget atomic per block on one of n_done counters instead of a flag
    int n_done;
    long long nnz;
    int bitmap_words;
    unsigned bm_wmask;            // word index mask of the affected bitmap: all ones when every id has its own bit; 2^b - 1 when the ids are
                                  // FOLDED onto 2^b words (more ids than LDS bits): a folded bit may be set by another id, the contact is then
                                  // queued for nothing and the consumers' own membership test (q_codes: piece 0) drops it
    int strict;                   // GRAAL_MODE_STRICT: queue every contact with both ends in a neighbour's affected set
    int wt_queue;                 // queue entries are read by k_tm's finishing block (concurrent kernel): write-through stores
    unsigned token;               // unique per launch: "wave 1's keys of THIS launch are in LDS"
    const long long* nc;          // number of contigs of the ranked layout (max_id < 0: read it here)
    unsigned long long* relabel_flag;   // block 0 stores relabel_seq here at once: "the relabel in front of this kernel is complete" (k_tm waits for it)
    unsigned long long relabel_seq;     // 0: nobody waits
};

// word j (0 .. 4G-1) of G groups held in registers (select chain over constant indices: stays in registers)
template <int G> __device__ __forceinline__ int sel_words(const int4 (&a)[G], int j)
{
    int v = 0;
#pragma unroll
    for (int i = 0; i < G; i++) {
        const int4 q = a[i];
        const int w = (j & 3) == 0 ? q.x : ((j & 3) == 1 ? q.y : ((j & 3) == 2 ? q.z : q.w));
        v = (j >> 2) == i ? w : v;
    }
    return v;
}

// G = groups of 4 contacts per thread and iteration.  G = 4 with two 1024-thread blocks per CU (8 waves/SIMD, <= 64 VGPRs);
// G = 8 with one block per CU (half the waves to launch, the same bytes in flight).
template <bool SINGLE_SUB, int G>
__global__ __launch_bounds__(1024, (G <= 4 ? 8 : 4)) void k_scan(ScanArgs sa, int fA, Neigh nb, int K, int max_id,
                                                int dry /* timing replays: count, do not queue */)
{
    extern __shared__ unsigned s_bm[];
    __shared__ Rec s_rec[MAXK + 1];
    __shared__ int s_cbase[MAXK + 1], s_clen[MAXK + 1], s_pref[MAXK + 2], s_fB[MAXK];
    __shared__ int s_waves_done, s_long;
    __shared__ int s_hitbuf[16][64 * 3];   // per wave: the doubly-affected contacts of one iteration (row id, col id, contact index), dealt one per lane
    __shared__ unsigned s_keys_ready;
    __shared__ unsigned long long s_nrel;
    const int t = threadIdx.x;
    const int lane = t & 63;
    if (sa.relabel_seq != 0ull && blockIdx.x == 0 && t == 0)
        __hip_atomic_store(sa.relabel_flag, sa.relabel_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the relabel's writes were released when it ended)
    (void)max_id;   // (the candidates' geometry is k_tm's and the consumers' business; kept in the signature for the timing replays)
    STAMP(8, blockIdx.x == 0 && t == 0 && !dry);
    STAMP_BLK(0, t == 0 && !dry);
    // ---- prologue.  Only the affected BITMAP stands between a block and its stream, so wave 0 builds nothing else: one
    // 32-byte load per fragment (fA and the K neighbours: their mates rows = the fragments of their contigs when those hold
    // <= N_MATES fragments, a marker row otherwise), LDS bit sets, the block's one barrier.  Everything the rare later tests
    // need (piece keys, circular-model flags, the long contigs' index ranges) is derived by WAVE 1 from its own loads, off the
    // critical path; a wave that finds an affected row waits for s_keys_ready first.
    // (round 1 built keys, flags and prefix sums in wave 0 before the barrier: 3.5-4 us of a 20 us launch)
    int4 row0 = make_int4(-1, -1, -1, -1), row1 = row0;
    if (t <= K) {   // (select chains, not nb.fB[t]: a dynamically indexed kernel argument is fetched from memory)
        const int f = t == 0 ? fA : sel_nb(nb, t - 1);
        row0 = reinterpret_cast<const int4*>(sa.mates)[2 * f]; row1 = reinterpret_cast<const int4*>(sa.mates)[2 * f + 1];
    }
    Geo my_geo = {0, 0, 0, 0};
    Link my_link = {0, 0, 0, 0};
    int my_cbase = 0, my_f = 0;
    if (t >= 64 && t - 64 <= K) {
        my_f = t == 64 ? fA : sel_nb(nb, t - 65);
        my_geo = sa.geo[my_f]; my_link = sa.link[my_f]; my_cbase = sa.cbase[my_f];
    }
    const long long nnz = sa.nnz;
    const int n4 = (int)(nnz >> 2);           // groups of 4 contacts: 0 .. n4 (the last one partial or empty; nnz < 2^33)
    const int4* __restrict__ row4 = sa.row4;
    const int stride = (int)(gridDim.x * blockDim.x);
    int g0 = (int)(blockIdx.x * blockDim.x) + t;
    // unconditional loads with a clamped group index (group n4 is in bounds: the arrays are padded): branch-free, so the
    // compiler keeps all of them in flight together.  The first batch is requested here, ABOVE the barrier: 40 % of the list is
    // on its way while the bitmap is built.  Waves 0 and 1 too: their prologue loads were issued first (just above), so they
    // come back first (vector-memory results return in issue order) and the two waves hold their first batch like everybody else
    // instead of starting their stream a round trip late -- which made them the tail of every block.
    auto ldg = [&](int g) { return ld_stream(row4 + (g < n4 ? g : n4)); }; // (group indices fit 32 bits: nnz < 2^32)
    int4 f[G];
    static_assert(SCAN_PRE == 0 || SCAN_PRE == 4, "");
    if (SCAN_PRE) {
#pragma unroll
        for (int i = 0; i < G; i++) f[i] = ldg(g0 + i * stride);
    }
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#ifdef GRAAL_EXP_NOFOLD   // (A/B build, tools/ab.sh: what the fold's one AND per id costs)
    const unsigned wm = 0xffffffffu;
#else
    const unsigned wm = sa.bm_wmask;
#endif
    auto set_bits = [&](int fr) {   // mark fragment fr: its id, or the ids of its sub-fragments
        if (SINGLE_SUB) atomicOr(&s_bm[((unsigned)fr >> 5) & wm], 1u << (fr & 31));
        else {
            const int4 ids = reinterpret_cast<const int4*>(sa.sub_ids)[fr];
            atomicOr(&s_bm[((unsigned)ids.x >> 5) & wm], 1u << (ids.x & 31));
            if (ids.w > 1) atomicOr(&s_bm[((unsigned)ids.y >> 5) & wm], 1u << (ids.y & 31));
            if (ids.w > 2) atomicOr(&s_bm[((unsigned)ids.z >> 5) & wm], 1u << (ids.z & 31));
        }
    };
    if (t < 64) {
        // (all three are first touched by anybody else BEHIND the block's barrier -- wave 1 sets s_keys_ready there -- so this
        // clear is ordered before every use.  LDS keeps what earlier launches left: another engine of this process, or another
        // process on the same GPU, may have left the very same token value behind)
        if (t == 0) { s_waves_done = 0; s_nrel = 0; s_keys_ready = 0; }
        for (int i = t; i < sa.bitmap_words; i += 64) s_bm[i] = 0;
        WSYNC();
        const bool is_long = t <= K && row0.x == MATES_LONG;
        if (t <= K && !is_long) {
            if (row0.x >= 0) set_bits(row0.x);
            if (row0.y >= 0) set_bits(row0.y);
            if (row0.z >= 0) set_bits(row0.z);
            if (row0.w >= 0) set_bits(row0.w);
            if (row1.x >= 0) set_bits(row1.x);
            if (row1.y >= 0) set_bits(row1.y);
            if (row1.z >= 0) set_bits(row1.z);
            if (row1.w >= 0) set_bits(row1.w);
        }
        const unsigned long long lm = __ballot(is_long);
        if (t == 0) s_long = lm != 0 ? 1 : 0;
    }
    auto wait_keys = [&]() {   // bounded: a wave always gets out (the step is then flagged as failed)
        for (int spin = 0; spin < (1 << 24); spin++) {
            if (__hip_atomic_load(&s_keys_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == sa.token) return;
            __builtin_amdgcn_s_sleep(1);
        }
        atomicOr(&sa.counters[6], 1ull);
    };
    __syncthreads();
    if (t >= 64 && t < 128) {   // wave 1, BEHIND the barrier (in front of it, the whole block would wait for this work)
        const int u = t - 64;
        if (u <= K) { s_rec[u] = rec_gl(my_geo, my_link, my_f); s_cbase[u] = my_cbase; }
        if (u < MAXK) s_fB[u] = sel_nb(nb, u);
        WSYNC();
        if (u < K) {
            const Rec& A0 = s_rec[0];
            const Rec& B0 = s_rec[u + 1];
            const int fB = s_fB[u];
            // (the piece keys, the circular-model flags and the neighbour masks of the queued contacts are the consumers' business
            // now: q_codes)
            bool dup = (fB == fA) || (B0.id_c == A0.id_c);
            if (fB != fA)
                for (int j = 0; j < u; j++) if (s_fB[j] != fA && s_rec[j + 1].id_c == B0.id_c) dup = true;
            s_clen[u + 1] = (dup || B0.l_cont <= N_MATES) ? 0 : B0.l_cont;    // long contigs only: the short ones are marked from their rows
        }
        if (u == 0) s_clen[0] = s_rec[0].l_cont <= N_MATES ? 0 : s_rec[0].l_cont;
        WSYNC();
        if (u == 0) {
            int acc = 0;
            for (int j = 0; j <= K; j++) { s_pref[j] = acc; acc += s_clen[j]; }
            s_pref[K + 1] = acc;
            __hip_atomic_store(&s_keys_ready, sa.token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
#undef WSYNC
    if (s_long) { // some affected contig holds more than N_MATES fragments: everybody marks it from the position index (two
                  // more barriers are nothing next to that regime's work)
        wait_keys();
        const int total = s_pref[K + 1];
        const int* __restrict__ perm = sa.perm;
        for (int e = t; e < total; e += (int)blockDim.x) {
            int j = 0;
            while (j < K && e >= s_pref[j + 1]) j++;
            set_bits(perm[s_cbase[j] + (e - s_pref[j])]);
        }
        __syncthreads();
    }
    const int4* __restrict__ col4 = sa.col4;
    const int* __restrict__ sub2bin = sa.sub2bin;
    QRaw* __restrict__ queue = sa.queue;
    unsigned long long* __restrict__ counters = sa.counters;
    unsigned long long n_rel = 0;
    STAMP_BLK(1, t == 0 && !dry);
    STAMP(9, blockIdx.x == 0 && t == 0 && !dry);
    // first test of one batch = G groups of 4 contacts (ga + i * stride): bit j of the result = contact j of this lane is valid
    // and has an affected row.  No memory operations other than LDS reads: the main loop's fast path stays a pure stream.
    auto test_rows = [&](const int4 (&rr)[G], const int ga) -> unsigned {
        unsigned vmask = 0xffffffffu >> (32 - 4 * G); // valid contacts of the G groups: all of them, except at the very end
        if (ga + (G - 1) * stride >= n4) {
            vmask = 0;
#pragma unroll
            for (int i = 0; i < G; i++) {
                const long long rem = nnz - ((long long)(ga + i * stride) << 2);
                const int v = rem >= 4 ? 4 : (rem > 0 ? (int)rem : 0);
                vmask |= ((1u << v) - 1u) << (4 * i);
            }
        }
        unsigned hit = 0;   // bit j: contact j of this lane has an affected row
#pragma unroll
        for (int i = 0; i < G; i++) {
            const int4 q = rr[i];
            hit |= (((s_bm[((unsigned)q.x >> 5) & wm] >> (q.x & 31)) & 1u) | (((s_bm[((unsigned)q.y >> 5) & wm] >> (q.y & 31)) & 1u) << 1)
                    | (((s_bm[((unsigned)q.z >> 5) & wm] >> (q.z & 31)) & 1u) << 2) | (((s_bm[((unsigned)q.w >> 5) & wm] >> (q.w & 31)) & 1u) << 3)) << (4 * i);
        }
        return hit & vmask;
    };
    // the rest of a batch in which some lane of the wave found an affected row (rare while contigs are short)
    auto process_hits = [&](const int4 (&rr)[G], const int ga, unsigned hit) {
        // second test, still wide: the col words of the groups with an affected row (up to G 16-byte loads in flight
        // together), all bitmap tests at once
        int4 cc[G];
        {
            unsigned hit2 = 0;
#pragma unroll
            for (int i = 0; i < G; i++) {
                cc[i] = make_int4(0, 0, 0, 0);
                if ((hit >> (4 * i)) & 0xfu) cc[i] = ld_stream(col4 + (ga + i * stride));
            }
#pragma unroll
            for (int i = 0; i < G; i++) {
                const int4 q = cc[i];
                hit2 |= (((s_bm[((unsigned)q.x >> 5) & wm] >> (q.x & 31)) & 1u) | (((s_bm[((unsigned)q.y >> 5) & wm] >> (q.y & 31)) & 1u) << 1)
                         | (((s_bm[((unsigned)q.z >> 5) & wm] >> (q.z & 31)) & 1u) << 2) | (((s_bm[((unsigned)q.w >> 5) & wm] >> (q.w & 31)) & 1u) << 3)) << (4 * i);
            }
            hit &= hit2;
        }
        // Queue slots for ALL doubly-affected contacts of this wave iteration are reserved with one atomic (the few that the
        // third test rejects leave entries with an empty neighbour mask behind, which the pricing skips).  One reservation per
        // PASS of the loop below -- up to 16 per iteration, 10^5 per step when two long contigs are affected -- made the tail
        // of the queue the bottleneck of the late stage: same-address atomics with return, ~0.7 ms of a 0.8 ms scan.
#ifdef GRAAL_EXP_COLONLY   // (bisecting build: stop behind the second test -- wrong results)
        n_rel += __popc(hit);
        return;
#endif
        // ---- third test.  The doubly-affected contacts of this wave iteration are first DEALT to the lanes, one each, through a
        // per-wave LDS list: they come in clusters -- the rows of a contig's fragments hold contacts with their contig mates next
        // to each other, so one lane owns four of them while the others own none -- and a lane used to take its contacts one per
        // pass, a dependent round trip each: four passes in a wave's last iteration ended the kernel 6 us late, every step
        // (tools/ab.sh: 23.2 us per launch in a step, 16.8 without the third test).  Now one pass per 64 contacts: the gathers of
        // all of them and the slot reservation are in flight together.
        const unsigned mine = (unsigned)__popc(hit);
        unsigned incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
        const unsigned total = (unsigned)__shfl((int)incl, 63, 64);
        if (total == 0) return;
        HITSTAT_BEGIN();
        // Queue slots for ALL of them are reserved with one atomic (the few that the third test rejects leave entries with an empty
        // neighbour mask behind, which the pricing skips)
        unsigned long long base = 0;
        if (!dry && lane == 63) base = atomicAdd(&counters[2], (unsigned long long)total);
        int* const hb = s_hitbuf[t >> 6];
        for (unsigned n0 = 0; n0 < total; n0 += 64) {
            {   // owners: my contacts number incl - mine .. incl - 1 of the iteration; those of this round go to the list
                unsigned hh = hit, idx = incl - mine;
                while (hh) {
                    const int j = __ffs((int)hh) - 1;
                    hh &= hh - 1;
                    if (idx >= n0 && idx < n0 + 64) {
                        const unsigned e = (idx - n0) * 3;
                        hb[e] = sel_words<G>(rr, j); hb[e + 1] = sel_words<G>(cc, j);
                        hb[e + 2] = (int)((((unsigned)(ga + (j >> 2) * stride)) << 2) + (unsigned)(j & 3));   // (contact index: < 2^32)
                    }
                    idx += 1;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const unsigned n = n0 + (unsigned)lane;
            unsigned cidx = 0;
            int q_slots = 0, fx = 0, fy = 0;
            if (n < total) {
                const int rj = hb[3 * lane], cj_ = hb[3 * lane + 1];
                cidx = (unsigned)hb[3 * lane + 2];
                if (SINGLE_SUB) { fx = rj; fy = cj_; }
                else { const int a = sub2bin[rj], b = sub2bin[cj_]; fx = a >> 2; fy = b >> 2; q_slots = (a & 3) | ((b & 3) << 2); }
            }
            // (all lanes: the reservation's result)
            const unsigned slot_base = (unsigned)__shfl(base, 63, 64);
            if (n < total && !dry) {
                // 16 bytes per contact, tagged with the step's sequence number in both words (QRaw): no record gathers here, and nobody
                // waits for these stores -- a consumer that runs next to this kernel checks the tags, one behind it needs nothing
                unsigned long long* qw = reinterpret_cast<unsigned long long*>(queue + (slot_base + n));
                const unsigned long long w0 = (unsigned long long)cidx | ((unsigned long long)sa.seq32 << 32);
                const unsigned long long w1 = (unsigned long long)(unsigned)fx | ((unsigned long long)(unsigned)fy << 20) | ((unsigned long long)(unsigned)q_slots << 40)
                                              | ((unsigned long long)(sa.seq32 & 0xfffffu) << 44);
                if (sa.wt_queue) {   // device-scope (write-through) stores: the reader may be a block of k_tm on another XCD
                    __hip_atomic_store(qw + 0, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(qw + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    typedef unsigned long long v2u __attribute__((ext_vector_type(2)));
                    v2u a; a.x = w0; a.y = w1;
                    *reinterpret_cast<v2u*>(qw) = a;
                }
            }
            const unsigned rel = n < total ? 1u : 0u;   // (statistics: doubly-affected contacts)
            n_rel += __popc(rel);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
#ifdef GRAAL_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the stamp includes this wave's queue stores)
#endif
        HITSTAT_END();
    };
    // ---- main loop: G loads in flight, first test, (rarely) the rest.  (A two-buffer software pipeline of this loop was
    // measured in tools/scan_micro.hip: 15.4 -> 14.6 us per isolated launch with 2 x 2 groups through raw buffer loads, slower
    // with 2 x 4; in this kernel two buffers of 4 groups do not fit next to process_hits in 64 VGPRs.)
    if (!SCAN_PRE) {
#pragma unroll
        for (int i = 0; i < G; i++) f[i] = ldg(g0 + i * stride);
    }
#ifdef GRAAL_EXP_NOHITS   // (bisecting build, tools/ab_scan2.sh: the stream and the first test only -- wrong results)
#define PROCESS_HITS(a, b, c) do { n_rel += __popc(c); } while (0)
#else
#define PROCESS_HITS(a, b, c) process_hits(a, b, c)
#endif
    {
        const unsigned hit = test_rows(f, g0);
        if (__ballot(hit != 0) != 0) PROCESS_HITS(f, g0, hit);
    }
    for (int g = g0 + G * stride; g <= n4; g += G * stride) {
        int4 q[G];
#pragma unroll
        for (int i = 0; i < G; i++) q[i] = ldg(g + i * stride);
        const unsigned hit = test_rows(q, g);
        if (__ballot(hit != 0) != 0) PROCESS_HITS(q, g, hit);
    }
#undef PROCESS_HITS
    STAMP(10, blockIdx.x == 0 && t == 0 && !dry);
    STAMP_BLK(2, t == 0 && !dry);
    n_rel = (unsigned long long)wave_sum_ll((long long)n_rel);
    // (the count of relevant pairs is a statistic: it goes through LDS and the block's last wave adds it once -- one global
    // atomic per WAVE on that one word was ~80 us of a late-stage scan)
    if (lane == 0 && n_rel && !dry) atomicAdd(&s_nrel, n_rel);
    if (dry == 0) {
        // completion signal for k_tm's finishing block.  No block barrier here (a trailing __syncthreads measurably costs this
        // kernel 5 us): each wave counts itself in LDS, the last wave of the block signals.  The slot reservations (atomics with
        // return) have completed by then, so the queue's length is final when the last block has signalled; the entries themselves
        // are agent-scope (write-through) stores that may still be in flight.
        // (nobody waits for the queue stores any more: a consumer that may run before they have landed validates the sequence
        // tag of every entry, QRaw)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        int last = 0;
        if (lane == 0) last = (atomicAdd(&s_waves_done, 1) == (int)(blockDim.x >> 6) - 1);
        if (last) {
            const unsigned long long nr = atomicAdd(&s_nrel, 0ull);
            if (nr) atomicAdd(&counters[0], nr);
            if (sa.done) __hip_atomic_fetch_add(sa.done + DONE_STRIDE * (blockIdx.x % sa.n_done), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (result unused: no-return atomic)
            else __hip_atomic_store(sa.flags + FLAG_STRIDE * blockIdx.x, sa.seq32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__device__ __forceinline__ int sel_base(const int (&a)[MAXK + 1], int i)
{
    int v = a[0];
#pragma unroll
    for (int j = 1; j <= MAXK; j++) v = (i == j) ? a[j] : v;
    return v;
}

// k_fin: (0) waits for k_tm's tables (normally long complete: k_tm was launched before k_scan); (1) the mass tasks k_tm
// left over -- unit = (neighbour, task, chunk of 64 fragments of the task's larger piece, strided run of segments of the
// other piece), one wave per unit; (2) the queued contacts -- lane = contact, one evaluation per distinct relation.
// Completion: the last block to finish (ticket counter) reads the K*13 sums, zeroes the accumulators for the next step,
// and hands the sums out: to d_q_out (device; the caller all-reduces them) or, if host_res is given, to PINNED HOST
// memory followed by the step's sequence number -- the host spins on that word instead of paying for a device->host
// copy launch and a stream-synchronise wake-up.
struct FinArgs { // first-needed pointers by value (see ScanArgs)
    long long* tm_done;
    const int* step_hdr;
    unsigned long long* counters;
    const QRaw* queue;
    const int* cnt;                // the contacts' counts (a queue entry holds the contact's index)
    int multi;                     // bins have several sub-fragments
    const NbTables* tabs;
    const Geo* geo;
    const Stat* stat;
    long long* acc;
    unsigned long long* sync;
    const double* ln_tab;          // ln of the trans value by RF-count product (k_ln_tab), lut_n entries
    int lut_n;
    int skip;                      // diagnostics (GRAAL_FIN_SKIP): 1 = no mass units, 2 = no queued contacts
    int seg;                       // fragments y per mass unit (0: chosen per step)
    float norm_u;                  // >= 0: every sub-fragment has the same RF count a, and this is float(a * a) / nfpb
    int upw;                       // mass units per wave of the grid the unit size aims at
    unsigned long long* wq;        // work-queue counters: N_WQ for the mass units, N_WQ for the contact batches, WQ_STRIDE words apart (nullptr: static deal)
};

__global__ __launch_bounds__(256) void k_fin(const DevArgs* __restrict__ A, FinArgs fa, int fA, int K, int rank, int world,
                                              long long* __restrict__ d_q_out, volatile long long* host_res, long long seq)
{
    const NbTables* __restrict__ tabs = fa.tabs;
    const Geo* __restrict__ geo = fa.geo;
    const Stat* __restrict__ stat = fa.stat;
    const QRaw* __restrict__ queue = fa.queue;
    long long* __restrict__ out = fa.acc;
    unsigned long long* __restrict__ counters = fa.counters;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = gridDim.x * (blockDim.x >> 6);
    __shared__ YTile s_tile[4][64];
    // the block's K*13 sums: every mass unit and every (queued contact, candidate) adds here, the block flushes once.  Straight
    // to the global accumulators, a step with millions of queued contacts issued ~10^8 atomics on 65 addresses -- the VALUs
    // sat idle 91 % of a 12 ms step behind them (rocprofv3: SQ_INSTS_VALU against the kernel's duration).
    __shared__ long long s_accb[MAXK * N_OPS];
    __shared__ long long s_items;
    // dynamic LDS, sized by K (fin_dyn_lds): the per-relation sums of the queued contacts, then the unit list's prefix sums
    extern __shared__ long long s_dyn[];
    long long* const S = s_dyn;
    int* const s_ust = reinterpret_cast<int*>(s_dyn + K * S_PER_K);
#define USTART(k_, ti_) s_ust[(k_) * (MAX_TASKS + 1) + (ti_)]
    for (int i = threadIdx.x; i < MAXK * N_OPS; i += blockDim.x) s_accb[i] = 0;
    for (int i = threadIdx.x; i < K * S_PER_K; i += blockDim.x) S[i] = 0;
    if (threadIdx.x == 0) s_items = 0;
    STAMP(16, blockIdx.x == 0 && threadIdx.x == 0);
    STAMP_FBLK(0, threadIdx.x == 0);
    const unsigned long long nq_total = counters[2]; // written by k_scan, the previous kernel on the stream
    // ---- wait for the tables (bounded spin: every wave reaches the exit even if k_tm never ran).  The word k_tm
    // releases carries the neighbour's work list header with the sequence number: seq << 32 | priced << 31 | n_items ----
    __shared__ int s_ok;
    __shared__ unsigned s_hdr[MAXK];
    __shared__ int s_nt[MAXK];
    __shared__ long long s_wt[MAXK];
    if (threadIdx.x == 0) s_ok = 1;
    if (threadIdx.x < MAXK) { s_hdr[threadIdx.x] = 0x80000000u; s_nt[threadIdx.x] = 0; s_wt[threadIdx.x] = 0; }
    __syncthreads();
    // k_tm's compact per-task list (chunks << 19 | walk), every thread a few (neighbour, task) entries -- requested together
    // with the task counts (entries beyond them are leftovers of earlier steps and ignored): one memory round trip less
    constexpr int NE = (MAXK * MAX_TASKS + 255) / 256;
    int cw[NE];
    auto load_cw = [&]() {
#pragma unroll
        for (int j = 0; j < NE; j++) {
            const int e = (int)threadIdx.x + j * 256;
            cw[j] = 0;
            if (e < K * MAX_TASKS) { const int k = e / MAX_TASKS; cw[j] = tabs[k].cw[e - k * MAX_TASKS]; }
        }
    };
    if ((int)threadIdx.x < K) {
        bool ok = false;
        for (int spin = 0; spin < (1 << 22); spin++) {
            const unsigned long long w = (unsigned long long)__hip_atomic_load(&fa.tm_done[threadIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(w >> 32) == (unsigned)seq) { s_hdr[threadIdx.x] = (unsigned)w; ok = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok) { s_ok = 0; atomicOr((unsigned long long*)&counters[6], 1ull); }
        else if (!(s_hdr[threadIdx.x] >> 31)) { // (bit 31: priced by k_tm already, nothing written for k_fin)
            s_nt[threadIdx.x] = tabs[threadIdx.x].n_tasks;
            s_wt[threadIdx.x] = tabs[threadIdx.x].w_total;
        }
    }
    __shared__ PieceKey s_qkeys[MAXK];   // what expands the queue's entries (q_fetch + q_codes)
    __shared__ unsigned s_qlive;
    if (threadIdx.x == 0) s_qlive = 0;
    __syncthreads();
    if ((int)threadIdx.x < K && s_ok) { s_qkeys[threadIdx.x] = tabs[threadIdx.x].key; if (tabs[threadIdx.x].fB != fA) atomicOr(&s_qlive, 1u << threadIdx.x); }
    load_cw();
    STAMP(17, blockIdx.x == 0 && threadIdx.x == 0);
    const int* __restrict__ perm = A->perm;
    const int* __restrict__ lcontbp = A->soa.p[F_LCONTBP];
    const float nfpb = A->nfpb;
    const Par par = A->par;
    const int reach_bp = A->reach_bp;
    if (s_ok) {
        unsigned long long items = 0;
        // ---- work units.  An ITEM is (task, chunk of 64 fragments x of its larger piece); its walk over the other piece is cut
        // into segments of SEG fragments y, and (item, segment) is the unit a wave takes.  One wave per item, as it used to be,
        // made a step last as long as its longest item: with pieces of thousands of fragments next to pieces of one, rocprofv3
        // showed the VALUs busy 9 % of the time (C5 with its 7 contigs: 12 ms per step).  Every block derives the same unit
        // list -- per neighbour the exclusive prefix of units per task -- in LDS; unit U of the flat list belongs to rank
        // U % world.  (Sums are integers: any partition of the pairs gives the same result.)
        // (USTART(k, ti) = first unit of task ti among neighbour k's units: dynamic LDS, sized by K like the contact sums S)
        __shared__ int s_ubase[MAXK + 1];
        // SEG: fragments y per unit.  At most 128 (16 with sub-fragments: up to 9 slot pairs per fragment pair), and small enough
        // for the step to have ~4 units per wave of the grid: a unit is one long dependent chain of float32 powf / expf, ~25 us
        // for 16 y with sub-fragments, and a step of the C2 stand-in has only ~1,500 such units for 2,048 waves -- its k_fin lasted
        // 140 us for 12 us worth of VALU work, the few waves that got two or three units carried it (tools/fin_seg_ab.sh:
        // SEG 16 -> 2 takes the scoring phase from 138 to 79 us there, from 178 to 126 us at the C3 shape).
        const bool multi_sub = A->sub2bin_multi != nullptr;
        // ... and a unit is a RUN of segments: run r of a (task, chunk) takes the segments r, r + R, r + 2R, ... of the walk (R
        // runs per chunk).  The segments near the chunk are the expensive ones and those beyond the window cost nothing, so
        // contiguous runs would be a lottery; strided ones cost about the same, end at the window (a lane that has left the
        // window stays out: distances only grow along the walk), and load x once per run.  LRUN = segments per run.
        int SEG = fa.seg, LRUN = 1;
        bool long_units = false;   // units of hundreds of fragments y each: worth a draw from a work queue
        if (SEG <= 0) {
            long long w_all = 0;
            for (int k = 0; k < K; k++) w_all += s_wt[k];
            const long long per_unit = w_all / ((long long)fa.upw * n_waves * world);   // fragments y a unit should walk
            SEG = multi_sub ? 2 : 16;
            while (SEG < (multi_sub ? 16 : 64) && 2 * SEG <= per_unit / 2) SEG <<= 1;
            LRUN = (int)(per_unit / SEG > 1 ? (per_unit / SEG > 4096 ? 4096 : per_unit / SEG) : 1);
            long_units = per_unit >= 512;
        }
        auto runs_of = [&](int walk) { const int n_seg = max(1, (walk + SEG - 1) / SEG); return min(n_seg, (n_seg + LRUN - 1) / LRUN); };
#pragma unroll
        for (int j = 0; j < NE; j++) {
            const int e = (int)threadIdx.x + j * 256;
            if (e < K * MAX_TASKS) {
                const int k = e / MAX_TASKS, ti = e - k * MAX_TASKS;
                if (ti < s_nt[k]) USTART(k, ti) = (int)((unsigned)cw[j] >> 19) * runs_of(cw[j] & 0x7ffff);
            }
        }
        __syncthreads();
        for (int k = threadIdx.x >> 6; k < K; k += (int)(blockDim.x >> 6)) // one wave per neighbour, side by side
            wave_excl_scan(&USTART(k, 0), &USTART(k, 0), s_nt[k]); // (in place: a lane reads its entry before it writes it)
        __syncthreads();
        if (threadIdx.x == 0) {
            s_ubase[0] = 0;
            for (int k = 0; k < K; k++) s_ubase[k + 1] = s_ubase[k] + USTART(k, s_nt[k]);
        }
        __syncthreads();
        const int total_units = s_ubase[K];
        STAMP(11, blockIdx.x == 0 && threadIdx.x == 0);
        STAMP_FBLK(1, threadIdx.x == 0);
        // (static deal.  Handing the units out through one global counter -- they differ by orders of magnitude, and the slowest
        // block of a C2 step ends 100 us after the typical one -- was measured: 2,048 waves queueing on one address cost more
        // than the imbalance, 175 -> 201 us per step.)
        // Hand-out.  Units differ in cost by an order of magnitude (beyond the window: nothing; circular contig, nine slot pairs: a
        // long chain) and a wave gets only ~4 of them: dealt statically, the slowest block of a C2 step worked five times as long as
        // the median one.  So the units of this rank (v = 0, 1, ...: U = rank + world v) form N_WQ classes v % N_WQ, each with a
        // counter of its own; a wave draws from its class, then from the others.  The next draw is in flight while a unit is
        // priced.  Only when units are long (the late stage: 0.95 -> 0.90 ms per step): a draw is a device-scope atomic with
        // return, 2-3 us, and with units of a few microseconds the queue costs far more than the imbalance (C2 stand-in 88 -> 140
        // us, C4 122 -> 165 us per step with every unit drawn; one counter for everything, round 1: 175 -> 201 us).  Otherwise the
        // static deal: wave w takes v = w, w + n_waves, ...
        const long long n_virtual = (fa.skip & 1) ? 0 : ((long long)total_units - rank + world - 1) / world;
        int cls = wave % N_WQ, tried = 0;
        unsigned long long* const wq = long_units ? fa.wq : nullptr;
        long long pend = wq ? 0 : wave;
        if (wq && lane == 0) pend = (long long)atomicAdd(&fa.wq[cls * WQ_STRIDE], 1ull);
        for (;;) {
            long long v;
            if (wq) {
                const long long j = __shfl(pend, 0, 64);
                v = cls + (long long)N_WQ * j;
                if (v >= n_virtual) {   // this class is used up: on to the next one
                    if (++tried == N_WQ) break;
                    cls = (cls + 1) % N_WQ;
                    if (lane == 0) pend = (long long)atomicAdd(&wq[cls * WQ_STRIDE], 1ull);
                    continue;
                }
                if (lane == 0) pend = (long long)atomicAdd(&wq[cls * WQ_STRIDE], 1ull);
            } else {
                v = pend;
                if (v >= n_virtual) break;
                pend += n_waves;
            }
            const int U = (int)(rank + world * v);
            int k = 0;
            for (int j = 1; j < K; j++) k += (U >= s_ubase[j]) ? 1 : 0;
            const int u = U - s_ubase[k];
            const NbTables& T = tabs[k];
            int lo_t = 0, hi_t = s_nt[k] - 1; // task of unit u: last task with s_ustart <= u
            while (lo_t < hi_t) { const int mid = (lo_t + hi_t + 1) >> 1; if (USTART(k, mid) <= u) lo_t = mid; else hi_t = mid - 1; }
            const int ti = lo_t;
            const Task tk = T.task[ti];
            const int np = tk.np, nq = tk.nq, base_p = tk.base_p, base_q = tk.base_q;
            const int n_seg = max(1, ((tk.p == tk.q ? np : nq) + SEG - 1) / SEG), n_runs = runs_of(tk.p == tk.q ? np : nq);
            const int within = u - USTART(k, ti), chunk = within / n_runs, run = within - chunk * n_runs;
            items += run == 0;
            // lane = one fragment x of the chunk.  The fragments y it is paired with are walked AWAY from the chunk, 64 at a
            // time: the wave stages their transformed geometry and statistics in LDS once (one dependent-load chain per 64
            // y instead of one per pair -- the loop used to be bound by that latency), then every lane runs over the tile
            // reading LDS broadcasts.  A lane is done at its first y beyond the window (distances only grow from there).
            const int ix = chunk * 64 + lane;
            const bool has_x = ix < np;
            const bool same = tk.p == tk.q;
            Geo gx = {0, 0, 0, 0};
            Stat sx = {0.0f, 0.0f, 0.0f, 0, 0, 0, 0, 0};
            if (has_x) { const int fx = perm[base_p + ix]; gx = geo[fx]; sx = stat[fx]; }
            const End X = end_xf(gx, tk.xp);
            const Ctr cx = centres_of(X.start_bp, X.fwd, sx);
            bool x_below = true, asc = true;
            if (!same) {
                // q's fragment nearest to x in this layout: pieces map to disjoint intervals, so the side is fixed by
                // comparing x with q's first fragment (the same for every x of the piece: taken from lane 0)
                const End Q0 = end_xf(geo[perm[base_q]], tk.xq);
                x_below = __shfl((int)(X.start_bp < Q0.start_bp), 0, 64) != 0;
                asc = (tk.xq.sigma > 0) == x_below; // walk q by increasing old position?
            }
            const int ny_all = same ? np : nq;             // length of the walk
            const int walk_lo = same ? chunk * 64 + 1 : 0;
            YTile* tile = s_tile[threadIdx.x >> 6];
            bool done = !has_x, bad = false;
            long long accq = 0;
            for (int seg = run; seg < n_seg && __ballot(!done) != 0; seg += n_runs) {
            const int seg_lo = max(walk_lo, seg * SEG);
            const int ny = min(ny_all, (seg + 1) * SEG);   // this segment is [seg_lo, ny) of the walk
            for (int tb = seg_lo; tb < ny; tb += 64) {
                const int st = tb + lane;
                if (st < ny) {
                    const int iy = same ? st : (asc ? st : nq - 1 - st);
                    const int fy = perm[(same ? base_p : base_q) + iy];
                    const Geo gy = geo[fy];
                    const End Y = end_xf(gy, same ? tk.xp : tk.xq);
                    YTile y;
                    y.start_bp = Y.start_bp; y.len_bp = gy.len_bp; y.flags = (Y.fwd ? 1 : 0) | (Y.circ << 1); y.label = Y.label;
                    y.lbp = Y.lbp; y.st = stat[fy];
                    const Ctr cy = centres_of(Y.start_bp, Y.fwd, y.st);
                    y.c0 = cy.c0; y.c1 = cy.c1; y.c2 = cy.c2;
                    tile[lane] = y;
                }
                WAVE_LDS_SYNC();
                const int cnt = ny - tb < 64 ? ny - tb : 64;
                for (int j = 0; j < cnt; j++) {
                    // (x's own sub-fragment pairs and earlier fragments of the same piece are left out: candidates never
                    // revisit a bin's own pixel, and every pair is priced once, from its lower fragment)
                    if (done || (same && tb + j <= ix)) continue;
                    const YTile& y = tile[j];
                    End Y; Y.label = y.label; Y.start_bp = y.start_bp; Y.fwd = y.flags & 1; Y.circ = (y.flags >> 1) & 1; Y.lbp = y.lbp;
                    const int gap = same ? gap_bp(X, gx.len_bp, Y, y.len_bp)
                                         : (x_below ? Y.start_bp - (X.start_bp + gx.len_bp) : X.start_bp - (Y.start_bp + y.len_bp));
                    if (gap > reach_bp) { done = true; continue; }
                    long long q1;
                    if (!multi_sub) {   // one sub-fragment per bin: pair_mass_q_c without its slot loops (the same operations on the same values)
                        const float norm = fa.norm_u >= 0.0f ? fa.norm_u : (float)(sx.a0 * y.st.a0) / nfpb;
                        const float sd = fabsf(y.c0 - cx.c0);
                        const float ex = (X.circ == 1 ? rippe_circ(sd, (float)X.lbp / 1000.0f, par) : rippe(sd, par)) * norm;
                        q1 = to_q((double)ex - (double)(par.v_inter * norm));
                    } else {
                        Ctr cy; cy.c0 = y.c0; cy.c1 = y.c1; cy.c2 = y.c2;
                        q1 = pair_mass_q_c(cx, sx, cy, y.st, X.circ, X.lbp, nfpb, par);
                    }
                    if (q1 == Q_BAD) bad = true; else accq += q1;
                }
                if (__ballot(!done) == 0) break;
                WAVE_LDS_SYNC();
            }
            WAVE_LDS_SYNC();   // (the next segment stages its tile over this one)
            }
            const long long qv = __shfl(wave_sum_ll(accq), 0, 64);
            if (__ballot(bad) != 0 && lane == 0) nf_flag_ops(counters + NF_OFF, k, tk.minus ^ tk.plus);
            if (lane < N_OPS && qv != 0) {
                // logL = contacts - mass: the NEW layout's mass counts negative, the OLD one positive (lane = candidate)
                const long long sgn = (long long)((tk.minus >> lane) & 1u) - (long long)((tk.plus >> lane) & 1u);
                if (sgn != 0) atomicAdd((unsigned long long*)&s_accb[k * N_OPS + lane], (unsigned long long)(sgn * qv));
            }
        }
        if (lane == 0 && items) atomicAdd((unsigned long long*)&s_items, items); // (a statistic: one global atomic per block, below)
        STAMP(12, blockIdx.x == 0 && threadIdx.x == 0);
        // ---- queued contacts ----
        PriceArgs pa;
        pa.pr_lds = nullptr; pa.pt_lds = nullptr; pa.nt_lds = nullptr;
        pa.q.queue = queue; pa.q.geo2 = reinterpret_cast<const int2*>(geo); pa.q.cnt = fa.cnt; pa.q.keys = s_qkeys; pa.q.live = s_qlive; pa.q.K = K;
        pa.q.seq = (unsigned)seq; pa.q.concurrent = 0; pa.q.multi = fa.multi; pa.err = counters + 6;
        pa.tabs = tabs; pa.geo = geo; pa.stat = stat; pa.lcontbp = lcontbp; pa.out = s_accb; pa.nf = counters + NF_OFF; pa.nfpb = nfpb; pa.par = par;
        // (batches of queued contacts continue the round robin where this rank's mass units ended)
        const int mass_slots = (int)(((long long)total_units - rank + world - 1) / world % n_waves);
        const int B = contact_batch_size(nq_total, n_waves);
        const long long n_batches = (fa.skip & 2) ? 0 : (long long)((nq_total + (unsigned long long)B - 1ull) / (unsigned long long)B);
        if (fa.wq && B == 64) {   // (millions of queued contacts: drawn like the units, from the second set of counters)
            int cls2 = wave % N_WQ, tried2 = 0;
            long long pend2 = 0;
            if (lane == 0) pend2 = (long long)atomicAdd(&fa.wq[(N_WQ + cls2) * WQ_STRIDE], 1ull);
            for (;;) {
                const long long b = cls2 + (long long)N_WQ * __shfl(pend2, 0, 64);
                if (b >= n_batches) {
                    if (++tried2 == N_WQ) break;
                    cls2 = (cls2 + 1) % N_WQ;
                    if (lane == 0) pend2 = (long long)atomicAdd(&fa.wq[(N_WQ + cls2) * WQ_STRIDE], 1ull);
                    continue;
                }
                if (lane == 0) pend2 = (long long)atomicAdd(&fa.wq[(N_WQ + cls2) * WQ_STRIDE], 1ull);
                price_contact_batch(pa, S, fa.ln_tab, fa.lut_n, nq_total, (unsigned long long)b * (unsigned long long)B, B, lane);
            }
        } else {
            for (long long b = (wave - mass_slots + n_waves) % n_waves; b < n_batches; b += n_waves)
                price_contact_batch(pa, S, fa.ln_tab, fa.lut_n, nq_total, (unsigned long long)b * (unsigned long long)B, B, lane);
        }
        STAMP(13, blockIdx.x == 0 && threadIdx.x == 0);
        STAMP_FBLK(2, threadIdx.x == 0);
        __syncthreads();
        fold_contact_sums(tabs, S, s_accb, K);
        __syncthreads();
        STAMP(14, blockIdx.x == 0 && threadIdx.x == 0);
        STAMP_FBLK(3, threadIdx.x == 0);
        for (int i = threadIdx.x; i < K * N_OPS; i += blockDim.x) {
            const long long v = s_accb[i];
            if (v != 0) atomicAdd((unsigned long long*)&out[i], (unsigned long long)v);
        }
        if (threadIdx.x == 255 && s_items) atomicAdd(&counters[1], (unsigned long long)s_items);
    }
    // ---- completion ticket: every block releases its atomics, the last one hands the sums out ----
    __shared__ int s_last;
    __syncthreads();
    STAMP(18, blockIdx.x == gridDim.x - 1 && threadIdx.x == 0);
    ATOMICS_DONE();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long ticket = atomicAdd(&counters[5], 1ull);
        s_last = (ticket == (unsigned long long)gridDim.x - 1ull);
    }
    __syncthreads();
    if (!s_last) return;
    STAMP(19, threadIdx.x == 0);
    __threadfence();
    if (fa.wq && threadIdx.x < 2 * N_WQ) atomicExch(&fa.wq[threadIdx.x * WQ_STRIDE], 0ull);   // (every wave has made its last draw)
    hand_out(out, counters, fa.sync, K, d_q_out, host_res, seq);
    STAMP(21, threadIdx.x == 0);
#undef USTART
}

// ------------------------------------------------------------------ reference-arithmetic validation mode
// GRAAL_MODE_STRICT: a candidate's delta is computed the way the reference's sub_compute_likelihood defines it
// (kernels3.cu:3259-3718): EVERY pixel between two different bins of contig(fA) u contig(fB) is priced again from the float32 kb
// coordinates of the candidate layout and compared with its value in the current layout -- also the pairs whose geometry the
// move leaves mathematically unchanged (their float32 coordinates shift, so their values move by rounding noise: DESIGN.md
// section 2, deviation 1) and, with GRAAL_MODE_REF_TRANS_ACCU, with the reference's RF-count indexing in the trans branch.
//   delta = sum_{contacts inside the set} ob (ln ex_new - ln ex_old)  -  sum_{ALL slot pairs inside the set} (ex_new - ex_old)
// No window, no piece relations, no deduplication: O(m^2) per candidate for m affected bins, like the reference.  It exists for
// parity (tests/test_strict_gpu.py: bit-exact traces against the reference-arithmetic oracle on generic coordinates and
// non-uniform RF counts), not for speed.
struct STile { Geo g; int lbp, piece, frag, pad; Stat st; };   // one staged fragment of the strict mass walk, 64 bytes
struct StrictArgs {
    const int *perm, *cbase, *lcontbp;
    const Link* link;
    float nfpb;
    Par par;
    int quirk;
    int reach_bp;
    unsigned long long list_cap;
    int seg;                      // fragments y per work unit (a power of two <= 64: the y tile of a unit is cut into 64 / seg segments)
};

// correction of the layout independent all-trans mass T_all for the reference's trans-branch RF-count indexing: pairs of
// different contigs whose lower-id bin is reversed and has non-uniform RF counts (ubins lists the bins with non-uniform counts)
__global__ __launch_bounds__(256) void k_quirk_mass(int n_u, const int* __restrict__ ubins, int n_bins, const Geo* __restrict__ geo,
                                                     const Stat* __restrict__ stat, float nfpb, Par par, long long* __restrict__ out,
                                                     long long* __restrict__ bad_flag)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long q = 0;
    if (i < (long long)n_u * n_bins) {
        const int x = ubins[i / n_bins], y = (int)(i % n_bins);
        if (y > x) {
            const Geo gx = geo[x], gy = geo[y];
            const Stat sx = stat[x], sy = stat[y];
            if ((gx.flags & 1) == 0 && gx.id_c != gy.id_c && sx.n > 0 && sy.n > 0) {
                const int alast = stat_accu(sx, sx.n - 1);
                double acc = 0.0;
                for (int a = 0; a < sx.n; a++)
                    for (int b = 0; b < sy.n; b++)
                        acc += (double)ex_trans(alast, stat_accu(sy, b), nfpb, par) - (double)ex_trans(stat_accu(sx, a), stat_accu(sy, b), nfpb, par);
                q = to_q(acc);
                if (q == Q_BAD) { q = 0; atomicOr((unsigned long long*)bad_flag, 1ull); }
            }
        }
    }
    q = wave_sum_ll(q);
    if ((threadIdx.x & 63) == 0 && q != 0) atomicAdd((unsigned long long*)out, (unsigned long long)q);
}

__global__ __launch_bounds__(256) void k_strict_dense(FinArgs fa, StrictArgs sa, int fA, int K, int rank, int world,
                                                       long long* __restrict__ d_q_out, volatile long long* host_res, long long seq)
{
    const NbTables* __restrict__ tabs = fa.tabs;
    const Geo* __restrict__ geo = fa.geo;
    const Stat* __restrict__ stat = fa.stat;
    unsigned long long* __restrict__ counters = fa.counters;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int wave = blockIdx.x * (blockDim.x >> 6) + wib, n_waves = gridDim.x * (blockDim.x >> 6);
    __shared__ STile s_tile[4][64];
    __shared__ long long s_accb[MAXK * N_OPS];
    __shared__ Xf s_xf[MAXK][N_OPS][NP];
    __shared__ unsigned s_ident[MAXK][N_OPS];      // bit p: candidate op leaves piece p exactly where it is
    __shared__ int s_m[MAXK], s_lenA[MAXK], s_baseA[MAXK], s_baseB[MAXK], s_ubase[MAXK + 1];
    __shared__ int s_ok;
    __shared__ PieceKey s_qkeys[MAXK];
    __shared__ unsigned s_qlive;
    for (int i = threadIdx.x; i < MAXK * N_OPS; i += blockDim.x) s_accb[i] = 0;
    if (threadIdx.x == 0) { s_ok = 1; s_qlive = 0; }
    __syncthreads();
    if ((int)threadIdx.x < K) { // the tables come from k_tm on another stream (bounded spin, as in k_fin)
        bool ok = false;
        for (int spin = 0; spin < (1 << 22); spin++) {
            const unsigned long long w = (unsigned long long)__hip_atomic_load(&fa.tm_done[threadIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(w >> 32) == (unsigned)seq) { ok = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok) { s_ok = 0; atomicOr((unsigned long long*)&counters[6], 1ull); }
    }
    __syncthreads();
    const unsigned long long nq_total = counters[2];
    if (s_ok) {
        for (int i = threadIdx.x; i < K * N_OPS * NP; i += blockDim.x) {
            const int k = i / (N_OPS * NP), op = (i / NP) % N_OPS, p = i % NP;
            s_xf[k][op][p] = tabs[k].xf[op][p];
        }
        if ((int)threadIdx.x < K) {
            const int k = threadIdx.x, fB = tabs[k].fB;
            s_qkeys[k] = tabs[k].key;
            if (fB != fA) atomicOr(&s_qlive, 1u << k);
            const Geo gA = geo[fA], gB = geo[fB];
            const int lenA = sa.link[fA].l_cont, lenB = (fB == fA || gB.id_c == gA.id_c) ? 0 : sa.link[fB].l_cont;
            s_lenA[k] = lenA; s_baseA[k] = sa.cbase[fA]; s_baseB[k] = sa.cbase[fB];
            s_m[k] = fB == fA ? 0 : lenA + lenB;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < K * N_OPS; i += blockDim.x) {
            const int k = i / N_OPS, op = i % N_OPS;
            unsigned m = 0;
            for (int p = 1; p < NP; p++) {
                const Xf x = s_xf[k][op][p];
                const int c = tabs[k].contig[p];      // the piece's contig in the current layout (-1: empty piece)
                // identity: same label, no mirror, no shift, same circular model (xf_identity of any of its fragments)
                if (c >= 0 && x.label == c && x.sigma == 1 && x.off == 0) {
                    const int rep_f = c == geo[fA].id_c ? fA : tabs[k].fB;
                    const Geo gr = geo[rep_f];
                    const int circ_old = (gr.flags >> 1) & 1;
                    if (x.circ == circ_old && (circ_old == 0 || x.lbp == sa.lcontbp[rep_f])) m |= 1u << p;
                }
            }
            s_ident[k][op] = m;
        }
        if (threadIdx.x == 0) {
            s_ubase[0] = 0;
            for (int k = 0; k < K; k++) { const int nt = (s_m[k] + 63) >> 6; s_ubase[k + 1] = s_ubase[k] + nt * (nt + 1) / 2; }
        }
        __syncthreads();
        // ---- (1) every slot pair between two different bins of the affected set: unit = (neighbour, tile of 64 x, tile of 64 y)
        const int total_units = s_ubase[K];
        STile* tile = s_tile[wib];
        for (int U = rank + world * wave; U < total_units; U += world * n_waves) {
            int k = 0;
            for (int j = 1; j < K; j++) k += (U >= s_ubase[j]) ? 1 : 0;
            int u = U - s_ubase[k];
            const int m = s_m[k], nt = (m + 63) >> 6;
            int ti = 0;
            while (u >= nt - ti) { u -= nt - ti; ti++; }
            const int tj = ti + u;
            const NbTables& T = tabs[k];
            auto frag_at = [&](int i) { return sa.perm[i < s_lenA[k] ? s_baseA[k] + i : s_baseB[k] + (i - s_lenA[k])]; };
            const int ix = ti * 64 + lane;
            const bool has_x = ix < m;
            Geo gx = {0, 0, 0, 0};
            Stat sx = {0.0f, 0.0f, 0.0f, 0, 0, 0, 0, 0};
            int fx = 0, px = 0, lbpx = 0;
            if (has_x) {
                fx = frag_at(ix); gx = geo[fx]; sx = stat[fx];
                px = piece_of(T.key, gx.id_c, geo_pos(gx.flags));
                lbpx = ((gx.flags >> 1) & 1) ? sa.lcontbp[fx] : 0;
            }
            {
                const int iy = tj * 64 + lane;
                if (iy < m) {
                    STile y; y.frag = frag_at(iy); y.g = geo[y.frag]; y.st = stat[y.frag];
                    y.piece = piece_of(T.key, y.g.id_c, geo_pos(y.g.flags));
                    y.lbp = ((y.g.flags >> 1) & 1) ? sa.lcontbp[y.frag] : 0; y.pad = 0;
                    tile[lane] = y;
                }
            }
            WAVE_LDS_SYNC();
            const End X0 = end_old(gx, lbpx);
            long long accq[N_OPS];
#pragma unroll
            for (int op = 0; op < N_OPS; op++) accq[op] = 0;
            unsigned bad = 0;
            const int cnt = m - tj * 64 < 64 ? m - tj * 64 : 64;
            for (int j = 0; j < cnt; j++) {
                if (!has_x || ix >= tj * 64 + j) continue;       // every unordered pair once; never a bin with itself
                const STile& y = tile[j];
                const Stat sy = y.st;
                const End Y0 = end_old(y.g, y.lbp);
                float exo[3][3];
#pragma unroll
                for (int a = 0; a < 3; a++)
#pragma unroll
                    for (int b = 0; b < 3; b++)
                        exo[a][b] = (a < sx.n && b < sy.n) ? ex_pair_ref(X0, sx, a, fx, Y0, sy, b, y.frag, sa.nfpb, sa.par, sa.quirk) : 0.0f;
#pragma unroll
                for (int op = 0; op < N_OPS; op++) {
                    const unsigned idm = s_ident[k][op];
                    if (((idm >> px) & 1u) && ((idm >> y.piece) & 1u)) continue;   // both pieces stay where they are: same inputs, same values
                    const End X = end_xf(gx, s_xf[k][op][px]), Y = end_xf(y.g, s_xf[k][op][y.piece]);
                    double acc = 0.0;
#pragma unroll
                    for (int a = 0; a < 3; a++)
#pragma unroll
                        for (int b = 0; b < 3; b++)
                            if (a < sx.n && b < sy.n)
                                acc += (double)exo[a][b] - (double)ex_pair_ref(X, sx, a, fx, Y, sy, b, y.frag, sa.nfpb, sa.par, sa.quirk);
                    const long long q1 = to_q(acc);
                    if (q1 == Q_BAD) coarse_add_ops(fa.acc, counters + NF_OFF, k, 1u << op, acc); else accq[op] += q1;
                }
            }
#pragma unroll
            for (int op = 0; op < N_OPS; op++) {
                const long long qv = wave_sum_ll(accq[op]);
                if (lane == 0 && qv != 0) atomicAdd((unsigned long long*)&s_accb[k * N_OPS + op], (unsigned long long)qv);
            }
            (void)bad;
            WAVE_LDS_SYNC();
        }
        // ---- (2) the queued contacts (k_scan queued every contact with both ends in some neighbour's affected set): 16 lanes per
        // contact, lane = candidate; every candidate of every such neighbour is priced again
        {
            QSrc qs;
            qs.queue = fa.queue; qs.geo2 = reinterpret_cast<const int2*>(geo); qs.cnt = fa.cnt; qs.keys = s_qkeys; qs.live = s_qlive; qs.K = K;
            qs.seq = (unsigned)seq; qs.concurrent = 0; qs.multi = fa.multi;
            const int op = lane & 15;
            for (unsigned long long e0 = (unsigned long long)(n_waves - 1 - wave) * 4; e0 < nq_total; e0 += (unsigned long long)n_waves * 4) {
                const unsigned long long e = e0 + (lane >> 4);
                if (e >= nq_total || op >= N_OPS) continue;
                QEntry qe = q_fetch(qs, e, counters + 6);
                if (qe.fx < 0) continue;
                const int fx = qe.fx, fy = qe.fy, slx = qe.slots & 3, sly = (qe.slots >> 2) & 3;
                const Geo gx = geo[fx], gy = geo[fy];
                const Stat sx = stat[fx], sy = stat[fy];
                q_codes(qs, qe, gx, gy, qs.cnt[qe.idx]);
                if (qe.rel == 0) continue;
                const End X0 = end_old(gx, ((gx.flags >> 1) & 1) ? sa.lcontbp[fx] : 0), Y0 = end_old(gy, ((gy.flags >> 1) & 1) ? sa.lcontbp[fy] : 0);
                const float ex_old = ex_pair_ref(X0, sx, slx, fx, Y0, sy, sly, fy, sa.nfpb, sa.par, sa.quirk);
                const double ln_old = mm_ln(ex_old), ob = (double)__int_as_float(qe.cnt);
                unsigned rel = qe.rel;
                while (rel) {
                    const int k = (__ffs((int)rel) - 1) / CODE_BITS;
                    rel &= rel - 1;
                    const int p = (qe.ci >> (CODE_BITS * k)) & 7, q = (qe.cj >> (CODE_BITS * k)) & 7;
                    const unsigned idm = s_ident[k][op];
                    if (((idm >> p) & 1u) && ((idm >> q) & 1u)) continue;
                    const End X = end_xf(gx, s_xf[k][op][p]), Y = end_xf(gy, s_xf[k][op][q]);
                    const float ex_new = ex_pair_ref(X, sx, slx, fx, Y, sy, sly, fy, sa.nfpb, sa.par, sa.quirk);
                    if (ex_new == ex_old) continue;
                    const long long qv = to_q(ob * (mm_ln(ex_new) - ln_old));
                    if (qv == Q_BAD) nf_flag(counters + NF_OFF, k, op);
                    else if (qv != 0) atomicAdd((unsigned long long*)&s_accb[k * N_OPS + op], (unsigned long long)qv);
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < K * N_OPS; i += blockDim.x) {
            const long long v = s_accb[i];
            if (v != 0) atomicAdd((unsigned long long*)&fa.acc[i], (unsigned long long)v);
        }
        if (threadIdx.x == 255 && blockIdx.x == 0 && rank == 0) atomicAdd(&counters[1], (unsigned long long)total_units);
    }
    __shared__ int s_last;
    ATOMICS_DONE();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long ticket = atomicAdd(&counters[5], 1ull);
        s_last = (ticket == (unsigned long long)gridDim.x - 1ull);
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    hand_out(fa.acc, counters, fa.sync, K, d_q_out, host_res, seq);
}

// ------------------------------------------------------------------ reference arithmetic, windowed (k_strict_cull + k_strict)
// The same sums as k_strict_dense -- every term it adds that is not exactly zero -- without its O(m^2) evaluations:
//   * a fragment pair that is beyond the window (or in two contigs) both before and under a candidate has the trans value both
//     times, slot by slot: the difference is exactly 0 and the pair is skipped (unless the reference's trans-branch RF-count
//     indexing is on and RF counts differ inside a bin: then nothing is skipped);
//   * the 13 candidates of a neighbour fall into classes of equal contact-model inputs per piece pair (same_inputs, built by
//     k_tm): a class is priced once, under its first candidate, and its value added to every candidate of the class; the class
//     of the current layout's own inputs is never priced.
// Work unit = (neighbour, tile of 64 fragments x, tile of 64 fragments y) of the affected set, tiles never straddling the two
// contigs.  k_strict_cull lists the units that can hold a pair inside the window under some layout (interval arithmetic on
// the tiles' and pieces' bp extents: one thread per candidate unit); k_strict takes the listed units round robin, one wave
// each: lane = fragment x, the y tile staged in LDS, every (x, y) rounded to Q once per class.  Then the queued contacts (the
// scan queued every contact with both ends in a neighbour's set): lane = contact, one evaluation per class.
struct SetGeo { int m, lenA, lenB, baseA, baseB, tilesA, nt, lbpA, lbpB, cA, cB, pad; };   // one neighbour's affected set

// (the layout's part: needs nothing of k_tm's tables; `live` = the set is left to the strict kernels, NbTables::set_m > 0)
__device__ __forceinline__ SetGeo set_geo_of(int fA, int fB, bool live, const Geo* __restrict__ geo, const Link* __restrict__ link,
                                             const int* __restrict__ cbase)
{
    SetGeo g;
    const Geo gA = geo[fA], gB = geo[fB];
    const Link lA = link[fA], lB = link[fB];
    g.cA = gA.id_c; g.cB = gB.id_c;
    g.lenA = lA.l_cont; g.lenB = (fB == fA || gB.id_c == gA.id_c) ? 0 : lB.l_cont;
    g.m = live ? g.lenA + g.lenB : 0;      // (0: fB == fA, or priced by k_tm already)
    g.baseA = cbase[fA]; g.baseB = cbase[fB];
    g.lbpA = lA.l_cont_bp; g.lbpB = lB.l_cont_bp;
    g.tilesA = (g.lenA + 63) >> 6;
    g.nt = g.m > 0 ? g.tilesA + ((g.lenB + 63) >> 6) : 0;
    g.pad = 0;
    return g;
}
__device__ __forceinline__ SetGeo set_geo(const NbTables& T, const Geo* __restrict__ geo, const Link* __restrict__ link,
                                          const int* __restrict__ cbase, int fA)
{
    return set_geo_of(fA, T.fB, T.set_m > 0, geo, link, cbase);
}

// old bp interval [lo, hi) of piece p of the neighbour and its contig (0 = contig(fA), 1 = contig(fB)); lo >= hi: empty
__device__ __forceinline__ void piece_extent(const PieceKey& key, const Geo& gA, const Geo& gB, const SetGeo& sg, int p, int& lo, int& hi, int& side)
{
    lo = 0; hi = 0; side = 0;
    if (key.cA != key.cB) {
        const Geo& g = p <= 3 ? gA : gB;
        side = p <= 3 ? 0 : 1;
        const int lbp = p <= 3 ? sg.lbpA : sg.lbpB, w = p <= 3 ? p : p - 3;
        if (w == 1) { lo = 0; hi = g.start_bp; } else if (w == 2) { lo = g.start_bp; hi = g.start_bp + g.len_bp; } else { lo = g.start_bp + g.len_bp; hi = lbp; }
        return;
    }
    const bool a_first = key.a < key.b;
    const Geo& L = a_first ? gA : gB;
    const Geo& H = a_first ? gB : gA;
    if (p == 1) { lo = 0; hi = L.start_bp; }
    else if (p == 2) { lo = L.start_bp; hi = L.start_bp + L.len_bp; }
    else if (p == 3) { lo = L.start_bp + L.len_bp; hi = H.start_bp; }
    else if (p == 4) { lo = H.start_bp; hi = H.start_bp + H.len_bp; }
    else if (p == 5) { lo = H.start_bp + H.len_bp; hi = sg.lbpA; }
}


__global__ __launch_bounds__(256) void k_strict_cull(const NbTables* __restrict__ tabs, const Geo* __restrict__ geo, const Link* __restrict__ link,
                                                      const int* __restrict__ cbase, const int* __restrict__ perm, int fA, int K, int rank, int world,
                                                      int reach_bp, int no_window, int seg_fixed /* 0: chosen here */, int seg_min,
                                                      unsigned long long target_units, unsigned long long* __restrict__ list,
                                                      unsigned long long* __restrict__ list_n, unsigned long long cap,
                                                      unsigned long long* __restrict__ counters)
{
    __shared__ SetGeo s_sg[MAXK];
    __shared__ int s_plo[MAXK][NP], s_phi[MAXK][NP], s_pside[MAXK][NP];
    __shared__ Xf s_xf[MAXK][N_OPS][NP];
    __shared__ unsigned char s_crep[MAXK][N_PAIRS][N_OPS];
    __shared__ int s_rbase[MAXK + 1];
    __shared__ int s_row[4];      // the row's x tile: lo, hi (old bp), side
    const int t = threadIdx.x;
    STAMP(29, blockIdx.x == 0 && t == 0);
    if (t < K) {
        const SetGeo sg = set_geo(tabs[t], geo, link, cbase, fA);
        s_sg[t] = sg;
        const Geo gA = geo[fA], gB = geo[tabs[t].fB];
        for (int p = 0; p < NP; p++) { int lo, hi, side; piece_extent(tabs[t].key, gA, gB, sg, p, lo, hi, side); s_plo[t][p] = p ? lo : 0; s_phi[t][p] = p ? hi : 0; s_pside[t][p] = side; }
    }
    for (int i = t; i < K * N_OPS * NP; i += blockDim.x) { const int k = i / (N_OPS * NP), r = i - k * (N_OPS * NP); s_xf[k][r / NP][r % NP] = tabs[k].xf[r / NP][r % NP]; }
    for (int i = t; i < K * N_PAIRS * N_OPS; i += blockDim.x) { const int k = i / (N_PAIRS * N_OPS), r = i - k * (N_PAIRS * N_OPS); s_crep[k][r / N_OPS][r % N_OPS] = tabs[k].crep[r / N_OPS][r % N_OPS]; }
    __syncthreads();
    if (t == 0) { s_rbase[0] = 0; for (int k = 0; k < K; k++) s_rbase[k + 1] = s_rbase[k] + s_sg[k].nt; }
    __syncthreads();
    const int n_rows = s_rbase[K];
    // fragments y per unit (a power of two): the y tiles are cut so that the step has ~target_units units -- from the sets of THIS step
    // (every block finds the same value).  (The host used to choose it from the longest contig of the layout: a step between two
    // contigs of 400 bins in a layout that also holds one of 2,000 got units of 32 fragments y x ~5 classes, a few hundred units
    // for 2,048 waves.)
    int seg = seg_fixed;
    if (seg <= 0) {
        unsigned long long est = 0;
        for (int k = 0; k < K; k++) { const unsigned long long nt = (unsigned long long)s_sg[k].nt; est += nt * (nt + 1ull) / 2ull; }
        seg = 64;
        while (seg > seg_min && est * (unsigned long long)(64 / seg) < target_units) seg >>= 1;
    }
    const unsigned long long lg_seg = (unsigned long long)(31 - __clz(seg));
    // old bp extent of tile tt of neighbour k
    auto tile_extent = [&](int k, int tt, int& lo, int& hi, int& side) {
        const SetGeo& sg = s_sg[k];
        side = tt < sg.tilesA ? 0 : 1;
        const int t0 = side ? tt - sg.tilesA : tt, len = side ? sg.lenB : sg.lenA, base = side ? sg.baseB : sg.baseA;
        const int first = t0 * 64, next = first + 64;
        lo = geo[perm[base + first]].start_bp;
        hi = next >= len ? (side ? sg.lbpB : sg.lbpA) : geo[perm[base + next]].start_bp;
    };
    for (int row = blockIdx.x; row < n_rows; row += gridDim.x) {
        int k = 0;
        for (int j = 1; j < K; j++) k += row >= s_rbase[j] ? 1 : 0;
        const int ti = row - s_rbase[k];
        const SetGeo sg = s_sg[k];
        __syncthreads();
        if (t == 0) { int lo, hi, side; tile_extent(k, ti, lo, hi, side); s_row[0] = lo; s_row[1] = hi; s_row[2] = side; }
        __syncthreads();
        const int xlo = s_row[0], xhi = s_row[1], xside = s_row[2];
        for (int tj0 = ti; tj0 < sg.nt; tj0 += blockDim.x) {
            const int tj = tj0 + t;
            bool alive = false;
            if (tj < sg.nt && ((ti + tj + k) % world) == rank) {
                int ylo, yhi, yside;
                tile_extent(k, tj, ylo, yhi, yside);
                for (int p = 1; p < NP && !alive; p++) {
                    if (s_pside[k][p] != xside) continue;
                    const int xs = max(xlo, s_plo[k][p]), xe = min(xhi, s_phi[k][p]);
                    if (xs >= xe) continue;
                    for (int q = 1; q < NP && !alive; q++) {
                        if (s_pside[k][q] != yside) continue;
                        const int ys = max(ylo, s_plo[k][q]), ye = min(yhi, s_phi[k][q]);
                        if (ys >= ye) continue;
                        const int pr = pair_index(p, q);
                        // the current layout
                        const bool near_old = xside == yside && max(ys - xe, xs - ye) <= reach_bp;
                        for (int op = 0; op < N_OPS && !alive; op++) {
                            if (s_crep[k][pr][op] != op) continue;   // a class is priced under its first candidate only
                            if (no_window || near_old) { alive = true; break; }
                            const Xf a = s_xf[k][op][p], b = s_xf[k][op][q];
                            if (a.label != b.label) continue;
                            const int xs2 = a.sigma > 0 ? xs + a.off : a.off - xe, xe2 = a.sigma > 0 ? xe + a.off : a.off - xs;
                            const int ys2 = b.sigma > 0 ? ys + b.off : b.off - ye, ye2 = b.sigma > 0 ? ye + b.off : b.off - ys;
                            if (max(ys2 - xe2, xs2 - ye2) <= reach_bp) alive = true;
                        }
                    }
                }
            }
            // an alive tile pair is listed as ceil(count of its y tile / seg) units: (k, ti, tj, segment)
            int ne = 0;
            if (alive) {
                const int side = tj < sg.tilesA ? 0 : 1, t0 = side ? tj - sg.tilesA : tj;
                const int cnt_y = min(64, (side ? sg.lenB : sg.lenA) - t0 * 64);
                ne = (cnt_y + seg - 1) / seg;
            }
            if (__ballot(alive)) {
                const int lane = t & 63;
                int incl = ne;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
                const int total = __shfl(incl, 63, 64);
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(list_n, (unsigned long long)total);
                base = __shfl(base, 0, 64);
                for (int e = 0; e < ne; e++) {
                    const unsigned long long at = base + (unsigned long long)(incl - ne + e);
                    if (at < cap) list[at] = ((unsigned long long)k << 56) | (lg_seg << 52) | ((unsigned long long)ti << 32) | ((unsigned long long)tj << 8) | (unsigned long long)e;
                    else atomicOr(&counters[6], 2ull);   // (cannot happen: the host sizes the list for the longest contig)
                }
            }
        }
    }
}

struct STileW { Geo g; int lbp, piece, frag, nonuni; Stat st; };   // one staged fragment y, 64 bytes
constexpr int STRICT_ACC_COPIES = 8;   // copies of the block's K*13 sums (lane & 7 picks one): LDS atomics of a wave spread over them

// dynamic LDS of k_strict for K neighbours: the transforms, the classes and their candidate masks
__host__ __device__ constexpr size_t strict_dyn_lds(int K) { return (size_t)K * (N_OPS * NP * sizeof(Xf) + N_PAIRS * N_OPS * sizeof(unsigned short) + ((N_PAIRS * N_OPS + 3) & ~3)); }

template <bool MULTI>
__global__ __launch_bounds__(256) void k_strict(FinArgs fa, StrictArgs sa, int fA, int K, const unsigned long long* __restrict__ list,
                                                 unsigned long long* __restrict__ list_n, long long* __restrict__ d_q_out,
                                                 volatile long long* host_res, long long seq)
{
    const NbTables* __restrict__ tabs = fa.tabs;
    const Geo* __restrict__ geo = fa.geo;
    const Stat* __restrict__ stat = fa.stat;
    unsigned long long* __restrict__ counters = fa.counters;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wib, n_waves = gridDim.x * 4;
    __shared__ STileW s_tile[4][64];
    __shared__ long long s_accb[STRICT_ACC_COPIES][MAXK * N_OPS];
    __shared__ SetGeo s_sg[MAXK];
    extern __shared__ long long s_dyn_strict[];
    Xf* const s_xf = reinterpret_cast<Xf*>(s_dyn_strict);                                          // [K][N_OPS][NP]
    unsigned short* const s_cmask = reinterpret_cast<unsigned short*>(s_xf + K * N_OPS * NP);     // [K][N_PAIRS][N_OPS]
    unsigned char* const s_crep = reinterpret_cast<unsigned char*>(s_cmask + K * N_PAIRS * N_OPS); // [K][N_PAIRS][N_OPS]
#define XF_(k_, op_, p_) s_xf[((k_) * N_OPS + (op_)) * NP + (p_)]
#define CREP_(k_, pr_, op_) s_crep[((k_) * N_PAIRS + (pr_)) * N_OPS + (op_)]
#define CMASK_(k_, pr_, op_) s_cmask[((k_) * N_PAIRS + (pr_)) * N_OPS + (op_)]
    STAMP(16, blockIdx.x == 0 && threadIdx.x == 0);
    for (int i = threadIdx.x; i < STRICT_ACC_COPIES * MAXK * N_OPS; i += 256u) (&s_accb[0][0])[i] = 0;
    // (the tables are complete: the launch is ordered behind k_tm by an event)
    for (int i = threadIdx.x; i < K * N_OPS * NP; i += 256u) { const int k = i / (N_OPS * NP), r = i - k * (N_OPS * NP); s_xf[i] = tabs[k].xf[r / NP][r % NP]; }
    for (int i = threadIdx.x; i < K * N_PAIRS * N_OPS; i += 256u) {
        const int k = i / (N_PAIRS * N_OPS), r = i - k * (N_PAIRS * N_OPS);
        s_crep[i] = tabs[k].crep[r / N_OPS][r % N_OPS]; s_cmask[i] = tabs[k].cmask[r / N_OPS][r % N_OPS];
    }
    __shared__ PieceKey s_qkeys[MAXK];   // what expands the queue's entries (q_fetch + q_codes)
    __shared__ unsigned s_qlive;
    if (threadIdx.x == 255) s_qlive = 0;
    __syncthreads();
    if ((int)threadIdx.x < K) {
        s_sg[threadIdx.x] = set_geo(tabs[threadIdx.x], geo, sa.link, sa.cbase, fA);
        s_qkeys[threadIdx.x] = tabs[threadIdx.x].key;
        if (tabs[threadIdx.x].fB != fA) atomicOr(&s_qlive, 1u << threadIdx.x);
    }
    __syncthreads();
    const unsigned long long nq_total = counters[2];          // written by k_scan, an earlier kernel on the stream
    const unsigned long long n_units = min(*list_n, sa.list_cap); // written by k_strict_cull, the previous kernel on the stream
    const float nfpb = sa.nfpb;
    const Par par = sa.par;
    const bool quirk = sa.quirk != 0;
    const int reach_bp = sa.reach_bp;
    long long* const my_acc = s_accb[lane & (STRICT_ACC_COPIES - 1)];
    auto add_ops = [&](int k, unsigned ops, long long v) {
        while (ops) { const int b = __ffs((int)ops) - 1; ops &= ops - 1; atomicAdd((unsigned long long*)&my_acc[k * N_OPS + b], (unsigned long long)v); }
    };
    STAMP(17, blockIdx.x == 0 && threadIdx.x == 0);
    // ---- (1) the listed units
    STileW* tile = s_tile[wib];
    for (unsigned long long u = (unsigned long long)wave; u < n_units; u += (unsigned long long)n_waves) {
        const unsigned long long ent = list[u];
        const int seg = 1 << (int)((ent >> 52) & 7ull);   // (fragments y of this unit: chosen per step by k_strict_cull)
        const int k = (int)(ent >> 56), ti = (int)((ent >> 32) & 0xfffffull), tj = (int)((ent >> 8) & 0xffffffull), j0 = (int)(ent & 0xffull) * seg;
        const SetGeo sg = s_sg[k];
        const PieceKey key = tabs[k].key;
        auto frag_at = [&](int tt, int l, bool& ok) {
            const int side = tt < sg.tilesA ? 0 : 1, t0 = side ? tt - sg.tilesA : tt, pos = t0 * 64 + l;
            ok = l < 64 && pos < (side ? sg.lenB : sg.lenA);
            return ok ? sa.perm[(side ? sg.baseB : sg.baseA) + pos] : 0;
        };
        bool has_x;
        const int fx = frag_at(ti, lane, has_x);
        Geo gx = {0, 0, 0, 0};
        Stat sx = {0.0f, 0.0f, 0.0f, 0, 0, 0, 0, 0};
        int px = 0, lbpx = 0;
        if (has_x) {
            gx = geo[fx]; sx = stat[fx];
            px = piece_of(key, gx.id_c, geo_pos(gx.flags));
            lbpx = ((gx.flags >> 1) & 1) ? (gx.id_c == sg.cA ? sg.lbpA : sg.lbpB) : 0;
        }
        const bool nonuni_x = !stat_uniform(sx);
        int cnt = 0;
        {   // the unit's segment of the y tile: fragments j0 .. j0 + seg of it
            bool has_y;
            const int fy = frag_at(tj, j0 + lane, has_y);
            has_y = has_y && lane < seg;
            if (has_y) {
                STileW y; y.frag = fy; y.g = geo[fy]; y.st = stat[fy];
                y.piece = piece_of(key, y.g.id_c, geo_pos(y.g.flags));
                y.lbp = ((y.g.flags >> 1) & 1) ? (y.g.id_c == sg.cA ? sg.lbpA : sg.lbpB) : 0;
                y.nonuni = stat_uniform(y.st) ? 0 : 1;
                tile[lane] = y;
            }
            cnt = __popcll(__ballot(has_y));
        }
        WAVE_LDS_SYNC();
        const End X0 = end_old(gx, lbpx);
        long long accq[N_OPS];
#pragma unroll
        for (int op = 0; op < N_OPS; op++) accq[op] = 0;
        unsigned bad = 0;
        int cur_pr = -1;
        auto flush = [&]() {
            if (cur_pr < 0) return;
#pragma unroll
            for (int op = 0; op < N_OPS; op++)
                if (accq[op] != 0) { add_ops(k, CMASK_(k, cur_pr, op), accq[op]); accq[op] = 0; }
        };
        const bool live_x = has_x && sx.n > 0;
        for (int j = 0; j < cnt; j++) {
            if (!live_x || (ti == tj && j0 + j <= lane)) continue;  // every unordered pair once; never a bin with itself
            const STileW& y = tile[j];
            const Stat sy = y.st;
            if (sy.n == 0) continue;                                  // (a copy of a repeated bin: priced by k_rep_delta)
            const int py = y.piece, pr = pair_index(px, py);
            if (pr != cur_pr) { flush(); cur_pr = pr; }
            const End Y0 = end_old(y.g, y.lbp);
            const bool near_old = X0.label == Y0.label && gap_bp(X0, gx.len_bp, Y0, y.g.len_bp) <= reach_bp;
            const bool always = quirk && (nonuni_x || y.nonuni != 0);
            bool have_old = false;
            float exo[3][3];
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b < 3; b++) exo[a][b] = 0.0f;
            for (int op = 0; op < N_OPS; op++) {
                if (CREP_(k, pr, op) != op) continue;
                const End X = end_xf(gx, XF_(k, op, px)), Y = end_xf(y.g, XF_(k, op, py));
                const bool near_new = X.label == Y.label && gap_bp(X, gx.len_bp, Y, y.g.len_bp) <= reach_bp;
                if (!near_old && !near_new && !always) continue;      // the trans value both times, slot by slot: exactly zero
                if (!have_old) {
                    have_old = true;
                    if (MULTI) {
#pragma unroll
                        for (int a = 0; a < 3; a++)
#pragma unroll
                            for (int b = 0; b < 3; b++)
                                if (a < sx.n && b < sy.n) exo[a][b] = ex_pair_ref(X0, sx, a, fx, Y0, sy, b, y.frag, nfpb, par, quirk);
                    } else exo[0][0] = ex_pair_ref(X0, sx, 0, fx, Y0, sy, 0, y.frag, nfpb, par, quirk);
                }
                double acc = 0.0;
                if (MULTI) {
#pragma unroll
                    for (int a = 0; a < 3; a++)
#pragma unroll
                        for (int b = 0; b < 3; b++)
                            if (a < sx.n && b < sy.n) acc += (double)exo[a][b] - (double)ex_pair_ref(X, sx, a, fx, Y, sy, b, y.frag, nfpb, par, quirk);
                } else acc += (double)exo[0][0] - (double)ex_pair_ref(X, sx, 0, fx, Y, sy, 0, y.frag, nfpb, par, quirk);
                const long long q1 = to_q(acc);
                if (q1 == Q_BAD) coarse_add_ops(fa.acc, counters + NF_OFF, k, (unsigned)CMASK_(k, pr, op), acc);
                else {
                    // (`op` is wave-uniform -- the loop counter -- so this is a scalar jump to one 64-bit add, not 13 selects)
                    switch (op) {
                    case 0: accq[0] += q1; break; case 1: accq[1] += q1; break; case 2: accq[2] += q1; break; case 3: accq[3] += q1; break;
                    case 4: accq[4] += q1; break; case 5: accq[5] += q1; break; case 6: accq[6] += q1; break; case 7: accq[7] += q1; break;
                    case 8: accq[8] += q1; break; case 9: accq[9] += q1; break; case 10: accq[10] += q1; break; case 11: accq[11] += q1; break;
                    default: accq[12] += q1; break;
                    }
                }
            }
        }
        flush();
        for (int o = 32; o > 0; o >>= 1) bad |= __shfl_down(bad, o, 64);
        if (lane == 0 && bad) nf_flag_ops(counters + NF_OFF, k, bad & 0xffffu);
        WAVE_LDS_SYNC();   // (the next unit stages its tile over this one)
    }
    STAMP_MAX(18, lane == 0);
    // ---- (2) the queued contacts: lane = contact, one evaluation per class
    QSrc qs;
    qs.queue = fa.queue; qs.geo2 = reinterpret_cast<const int2*>(geo); qs.cnt = fa.cnt; qs.keys = s_qkeys; qs.live = s_qlive; qs.K = K;
    qs.seq = (unsigned)seq; qs.concurrent = 0; qs.multi = fa.multi;
    for (unsigned long long b0 = (unsigned long long)(n_waves - 1 - wave) * 64ull; b0 < nq_total; b0 += (unsigned long long)n_waves * 64ull) {
        const unsigned long long e = b0 + (unsigned long long)lane;
        if (e >= nq_total) continue;
        QEntry qe = q_fetch(qs, e, counters + 6);
        if (qe.fx < 0) continue;
        const int fx = qe.fx, fy = qe.fy, slx = qe.slots & 3, sly = (qe.slots >> 2) & 3;
        const Geo gx = geo[fx], gy = geo[fy];
        const Stat sx = stat[fx], sy = stat[fy];
        q_codes(qs, qe, gx, gy, qs.cnt[qe.idx]);
        if (qe.rel == 0) continue;
        const End X0 = end_old(gx, ((gx.flags >> 1) & 1) ? sa.lcontbp[fx] : 0), Y0 = end_old(gy, ((gy.flags >> 1) & 1) ? sa.lcontbp[fy] : 0);
        const float ex_old = ex_pair_ref(X0, sx, slx, fx, Y0, sy, sly, fy, nfpb, par, quirk);
        const double ln_old = mm_ln(ex_old), ob = (double)__int_as_float(qe.cnt);
        unsigned rel = qe.rel;
        while (rel) {
            const int k = (__ffs((int)rel) - 1) / CODE_BITS;
            rel &= rel - 1;
            const int p = (qe.ci >> (CODE_BITS * k)) & 7, q = (qe.cj >> (CODE_BITS * k)) & 7;
            const int pr = pair_index(p, q);
            for (int op = 0; op < N_OPS; op++) {
                if (CREP_(k, pr, op) != op) continue;
                const End X = end_xf(gx, XF_(k, op, p)), Y = end_xf(gy, XF_(k, op, q));
                const float ex_new = ex_pair_ref(X, sx, slx, fx, Y, sy, sly, fy, nfpb, par, quirk);
                if (ex_new == ex_old) continue;
                const long long qv = to_q(ob * (mm_ln(ex_new) - ln_old));
                if (qv == Q_BAD) nf_flag_ops(counters + NF_OFF, k, CMASK_(k, pr, op));
                else if (qv != 0) add_ops(k, CMASK_(k, pr, op), qv);
            }
        }
    }
    STAMP_MAX(19, lane == 0);
    __syncthreads();
    for (int i = threadIdx.x; i < K * N_OPS; i += 256u) {
        long long v = 0;
#pragma unroll
        for (int c = 0; c < STRICT_ACC_COPIES; c++) v += s_accb[c][i];
        if (v != 0) atomicAdd((unsigned long long*)&fa.acc[i], (unsigned long long)v);
    }
    if (threadIdx.x == 255 && blockIdx.x == 0) atomicAdd(&counters[1], n_units);
    __shared__ int s_last;
    ATOMICS_DONE();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long ticket = atomicAdd(&counters[5], 1ull);
        s_last = (ticket == (unsigned long long)gridDim.x - 1ull);
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (threadIdx.x == 0) *list_n = 0;   // (every block has read it: the list is empty again for the next step)
    hand_out(fa.acc, counters, fa.sync, K, d_q_out, host_res, seq);
    STAMP(20, threadIdx.x == 0);
#undef XF_
#undef CREP_
#undef CMASK_
}

// k_strict_flat: the same sums as k_strict_cull + k_strict for SMALL affected sets -- the middle of a run, contigs of 20-100 bins:
// a step there has a handful of tile-pair units and a few hundred queued contacts, and k_strict spends 40 us on each part with
// one wave walking 4 fragments y x ~9 classes, or 5 neighbours x ~9 classes of one contact, one evaluation after the other
// (tools/stamps_c4.py).  Here every (fragment pair, class) and every (contact, neighbour, class) is a LANE: two evaluations deep,
// whatever the set's size; no unit list, no culling kernel, no event -- the kernel sits behind the scan on its stream and waits for
// k_tm's tables by itself (its grid cannot fill the chip: k_tm always finds room).  Sets with more than FLAT_CAP_PAIRS pairs
// are left to the tiled kernels: the step's last block then publishes NEED_FIN and touches nothing.
// Same per-pair / per-contact terms, rounded to Q the same way: bit-identical to k_strict (tests/test_strict_windowed_gpu.py).
constexpr long long FLAT_CAP_PAIRS = 40000;     // x 13 class slots = 520 k lanes = ~11 rounds of the grid
constexpr int FLAT_BLOCKS = 192;                // < 256 CUs: blocks that wait for k_tm can never keep it off the chip
template <bool MULTI>
__global__ __launch_bounds__(256) void k_strict_flat(FinArgs fa, StrictArgs sa, int fA, Neigh nb, int K, int rank, int world, long long* __restrict__ d_q_out,
                                                      volatile long long* host_res, long long seq)
{
    const NbTables* __restrict__ tabs = fa.tabs;
    const Geo* __restrict__ geo = fa.geo;
    const Stat* __restrict__ stat = fa.stat;
    unsigned long long* __restrict__ counters = fa.counters;
    const int t = threadIdx.x, lane = t & 63;
    __shared__ long long s_accb[STRICT_ACC_COPIES][MAXK * N_OPS];
    __shared__ SetGeo s_sg[MAXK];
    __shared__ long long s_pbase[MAXK + 1];
    __shared__ PieceKey s_qkeys[MAXK];
    __shared__ unsigned s_qlive;
    __shared__ int s_ok, s_last;
    // (the transforms and classes are read where they are needed, from k_tm's tables in L2: a lane wants two transforms and one
    // class entry -- staging all K x 13 x 6 of them in LDS first, as k_strict does for its long loops, was 6 us of a 30 us kernel)
#define XF_(k_, op_, p_) tabs[k_].xf[op_][p_]
#define CREP_(k_, pr_, op_) tabs[k_].crep[pr_][op_]
#define CMASK_(k_, pr_, op_) tabs[k_].cmask[pr_][op_]
    STAMP(16, blockIdx.x == 0 && t == 0);
    const unsigned long long nq_total = counters[2];          // written by k_scan, an earlier kernel on the stream (on its way during the wait below)
    if (t == 0) { s_ok = 1; s_qlive = 0; }
    for (int i = t; i < STRICT_ACC_COPIES * MAXK * N_OPS; i += 256u) (&s_accb[0][0])[i] = 0;
    __syncthreads();
    if (t < K) {   // k_tm's tables of neighbour t (and its own pricing of the small sets): complete?  Bounded wait.
        // (what the layout alone says about the set is on its way meanwhile)
        const int fB_t = sel_nb(nb, t);
        s_sg[t] = set_geo_of(fA, fB_t, true, geo, sa.link, sa.cbase);
        if (fB_t != fA) atomicOr(&s_qlive, 1u << t);
        bool ok = false;
        for (int spin = 0; spin < (1 << 22); spin++) {
            const unsigned long long w = (unsigned long long)__hip_atomic_load(&fa.tm_done[t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(w >> 32) == (unsigned)seq) { ok = true; break; }
            if ((spin & 63) == 63 && (__hip_atomic_load(&counters[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4ull)) break;   // k_tm gave up (see there)
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok) { s_ok = 0; atomicOr(&counters[6], 1ull); }
        else {   // (the acquire above covers this thread's reads of neighbour t's tables)
            s_qkeys[t] = tabs[t].key;
            if (!(tabs[t].set_m > 0)) { s_sg[t].m = 0; s_sg[t].nt = 0; }   // (fB == fA, or priced by k_tm already)
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    STAMP(21, blockIdx.x == 0 && t == 0);
    const bool ok = s_ok != 0;
    if (t == 0) {
        long long acc = 0;
        for (int k = 0; k < K; k++) { s_pbase[k] = acc; const long long m = ok ? s_sg[k].m : 0; acc += m * (m - 1) / 2; }
        s_pbase[K] = acc;
    }
    __syncthreads();
    const long long total_pairs = s_pbase[K];
    const bool too_big = total_pairs > FLAT_CAP_PAIRS;   // (every block finds the same: the tables are the same)
    STAMP(17, blockIdx.x == 0 && t == 0);
    if (ok && !too_big) {
        const float nfpb = sa.nfpb;
        const Par par = sa.par;
        const bool quirk = sa.quirk != 0;
        const int reach_bp = sa.reach_bp;
        long long* const my_acc = s_accb[lane & (STRICT_ACC_COPIES - 1)];
        auto add_ops = [&](int k, unsigned ops, long long v) {
            while (ops) { const int b = __ffs((int)ops) - 1; ops &= ops - 1; atomicAdd((unsigned long long*)&my_acc[k * N_OPS + b], (unsigned long long)v); }
        };
        const long long gi = (long long)blockIdx.x * 256u + t, stride = (long long)gridDim.x * 256u;
        QSrc qs;
        qs.queue = fa.queue; qs.geo2 = reinterpret_cast<const int2*>(geo); qs.cnt = fa.cnt; qs.keys = s_qkeys; qs.live = s_qlive; qs.K = K;
        qs.seq = (unsigned)seq; qs.concurrent = 0; qs.multi = fa.multi;
        const long long per_contact = (long long)K * N_OPS;
        // one item space: the pairs' lanes first (whole waves), the contacts' lanes behind them -- both kinds are in flight together
        const long long pair_items = (total_pairs * N_OPS + 63) & ~63ll, all_items = pair_items + (long long)nq_total * per_contact;
        for (long long it0 = gi; it0 < all_items; it0 += stride) {
          if (it0 < pair_items) {
            // ---- (1) lane = (fragment pair of a neighbour's set, class slot)
            const long long it = it0;
            if (it >= total_pairs * N_OPS) continue;
            if ((int)((it0 >> 6) % (long long)world) != rank) continue;   // the pairs' lanes are dealt to the ranks wave by wave (the queued contacts are
                                                                          // this rank's own: its shard of the list)
            const long long pi = it / N_OPS;
            const int op = (int)(it - pi * N_OPS);
            int k = 0;
            for (int j = 1; j < K; j++) k += pi >= s_pbase[j] ? 1 : 0;
            const long long local = pi - s_pbase[k];
            // pair number -> (i < j): local = j (j - 1) / 2 + i
            int j = (int)((1.0 + sqrt(1.0 + 8.0 * (double)local)) * 0.5);
            while ((long long)j * (j - 1) / 2 > local) j -= 1;
            while ((long long)(j + 1) * j / 2 <= local) j += 1;
            const int i = (int)(local - (long long)j * (j - 1) / 2);
            const SetGeo sg = s_sg[k];
            const int fx = i < sg.lenA ? sa.perm[sg.baseA + i] : sa.perm[sg.baseB + i - sg.lenA];
            const int fy = j < sg.lenA ? sa.perm[sg.baseA + j] : sa.perm[sg.baseB + j - sg.lenA];
            const Geo gx = geo[fx], gy = geo[fy];
            const Stat sx = stat[fx], sy = stat[fy];
            if (sx.n == 0 || sy.n == 0) continue;                     // (a copy of a repeated bin: priced by k_rep_delta)
            const PieceKey key = s_qkeys[k];
            const int px = piece_of(key, gx.id_c, geo_pos(gx.flags)), py = piece_of(key, gy.id_c, geo_pos(gy.flags));
            const int pr = pair_index(px, py);
            const int cls = CREP_(k, pr, op);
            Xf xa = XF_(k, op, px), xb = XF_(k, op, py);              // (requested together with the class entry)
            const unsigned cm = CMASK_(k, pr, op);
            keep_xf(xa, xb);
            if (cls != op) continue;                                  // a class is priced once, under its first candidate
            const int lbpx = ((gx.flags >> 1) & 1) ? (gx.id_c == sg.cA ? sg.lbpA : sg.lbpB) : 0;
            const int lbpy = ((gy.flags >> 1) & 1) ? (gy.id_c == sg.cA ? sg.lbpA : sg.lbpB) : 0;
            const End X0 = end_old(gx, lbpx), Y0 = end_old(gy, lbpy);
            const End X = end_xf(gx, xa), Y = end_xf(gy, xb);
            const bool near_old = X0.label == Y0.label && gap_bp(X0, gx.len_bp, Y0, gy.len_bp) <= reach_bp;
            const bool near_new = X.label == Y.label && gap_bp(X, gx.len_bp, Y, gy.len_bp) <= reach_bp;
            const bool always = quirk && (!stat_uniform(sx) || !stat_uniform(sy));
            if (!near_old && !near_new && !always) continue;          // the trans value both times, slot by slot: exactly zero
            double acc = 0.0;
            if (MULTI) {
#pragma unroll
                for (int a = 0; a < 3; a++)
#pragma unroll
                    for (int b = 0; b < 3; b++)
                        if (a < sx.n && b < sy.n)
                            acc += (double)ex_pair_ref(X0, sx, a, fx, Y0, sy, b, fy, nfpb, par, quirk) - (double)ex_pair_ref(X, sx, a, fx, Y, sy, b, fy, nfpb, par, quirk);
            } else acc += (double)ex_pair_ref(X0, sx, 0, fx, Y0, sy, 0, fy, nfpb, par, quirk) - (double)ex_pair_ref(X, sx, 0, fx, Y, sy, 0, fy, nfpb, par, quirk);
            const long long q1 = to_q(acc);
            if (q1 == Q_BAD) coarse_add_ops(fa.acc, counters + NF_OFF, k, cm, acc);
            else if (q1 != 0) add_ops(k, cm, q1);
            STAMP_MAX(18, lane == 0);
          } else {
            // ---- (2) lane = (queued contact, neighbour, class slot)
            const long long it = it0 - pair_items;
            const unsigned long long e = (unsigned long long)(it / per_contact);
            const int r = (int)(it - (long long)e * per_contact), k = r / N_OPS, op = r - k * N_OPS;
            QEntry qe = q_fetch(qs, e, counters + 6);
            if (qe.fx < 0) continue;
            const int fx = qe.fx, fy = qe.fy, slx = qe.slots & 3, sly = (qe.slots >> 2) & 3;
            const Geo gx = geo[fx], gy = geo[fy];
            const Stat sx = stat[fx], sy = stat[fy];
            q_codes(qs, qe, gx, gy, qs.cnt[qe.idx]);
            if (!((qe.rel >> (CODE_BITS * k)) & 1u)) continue;
            const int p = (qe.ci >> (CODE_BITS * k)) & 7, q = (qe.cj >> (CODE_BITS * k)) & 7;
            const int pr = pair_index(p, q);
            const int cls = CREP_(k, pr, op);
            Xf xa = XF_(k, op, p), xb = XF_(k, op, q);
            const unsigned cm = CMASK_(k, pr, op);
            keep_xf(xa, xb);
            if (cls != op) continue;
            const End X0 = end_old(gx, ((gx.flags >> 1) & 1) ? sa.lcontbp[fx] : 0), Y0 = end_old(gy, ((gy.flags >> 1) & 1) ? sa.lcontbp[fy] : 0);
            const float ex_old = ex_pair_ref(X0, sx, slx, fx, Y0, sy, sly, fy, nfpb, par, quirk);
            const End X = end_xf(gx, xa), Y = end_xf(gy, xb);
            const float ex_new = ex_pair_ref(X, sx, slx, fx, Y, sy, sly, fy, nfpb, par, quirk);
            if (ex_new == ex_old) continue;
            const double ob = (double)__int_as_float(qe.cnt);
            const long long qv = to_q(ob * (mm_ln(ex_new) - mm_ln(ex_old)));
            if (qv == Q_BAD) nf_flag_ops(counters + NF_OFF, k, cm);
            else if (qv != 0) add_ops(k, cm, qv);
            STAMP_MAX(19, lane == 0);
          }
        }
    }
    __syncthreads();
    for (int i = t; i < K * N_OPS; i += 256u) {
        long long v = 0;
#pragma unroll
        for (int c = 0; c < STRICT_ACC_COPIES; c++) v += s_accb[c][i];
        if (v != 0) atomicAdd((unsigned long long*)&fa.acc[i], (unsigned long long)v);
    }
    if (t == 255 && blockIdx.x == 0 && ok && !too_big && rank == 0) atomicAdd(&counters[1], (unsigned long long)total_pairs);
    ATOMICS_DONE();
    __syncthreads();
    if (t == 0) {
        const unsigned long long ticket = atomicAdd(&counters[5], 1ull);
        s_last = (ticket == (unsigned long long)gridDim.x - 1ull);
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (ok && too_big) {   // nothing was priced, nothing is reset: the tiled kernels find the step as k_scan and k_tm left it
        if (t == 0) { counters[5] = 0; __threadfence_system(); if (host_res) host_res[0] = seq | NEED_FIN; __threadfence_system(); }
        return;
    }
    // (was this kernel needed at all?  The host keeps a running mean of the answer and lets k_tm finish the steps by itself again
    // when it is mostly "no": eval_sync)
    // (bit 1: what the sets' geometry alone says -- some set is beyond what k_tm prices itself -- the same on every rank: with an exchange
    // attached the ranks must switch between the two flows together, see eval_sync)
    if (t == 0 && host_res) {
        long long big = 0;
        for (int k = 0; k < K; k++) big += s_sg[k].m > STRICT_INLINE_M ? 1 : 0;
        host_res[1 + MAXK * N_OPS] = ((total_pairs > 0 || nq_total > (unsigned long long)FIN_INLINE_Q) ? 1 : 0) | (big ? 2 : 0);
    }
    hand_out(fa.acc, counters, fa.sync, K, d_q_out, host_res, seq);
    STAMP(20, t == 0);
#undef XF_
#undef CREP_
#undef CMASK_
}

#include "strict2.h"

// ------------------------------------------------------------------ repeated bins (allow_repeats)
// A repeated ("duplicated") bin has several fragment copies (frag_dispatcher / collector_id_repeats,
// simulation_loader.py:258-277); the expected value of a pixel is the float32 sum over the ACTIVE copy pairs of its two
// bins (kernels3.cu:2915-2930) and there is no sparse shortcut for it.  Like the reference (kernels3.cu:3356-3380: the
// "repeats x all unique bins", "repeats x repeats" and "repeat diagonal" ranges) the engine prices every pixel of a
// repeated bin densely -- full likelihood and candidates -- while the sparse path above sees those bins with zero
// sub-fragments (no mass, no contacts).
struct RepArgs {
    int n_dup, n_bins, n_sub_total;
    const int* dup_bins;        // [n_dup] unique-bin ids
    const int* dup_index;       // [n_bins] index into dup_bins, -1 = not repeated
    const int* dispatcher;      // [n_bins][2]
    const int* collector;       // fragment ids, copies of a bin contiguous
    const float* obs;           // [n_dup][3][n_sub_total] observation rows of the repeated bins' sub-fragments
    const int* sub_ids;         // [n_bins][4]
    const Stat* stat_bin;       // [n_bins] sub-fragment lengths / accu of the BIN (shared by its copies)
    const Geo* geo;
    const int* lcontbp;
    float nfpb;
    Par par;
    int quirk;                  // GRAAL_MODE_REF_TRANS_ACCU: the reference's RF-count indexing in the trans branch
};

// ex_pair with the reference's trans-branch RF-count indexing when X is the pixel's FIRST copy (a copy of its lower-id bin;
// of the bin itself on the diagonal): reversed, every slot of X is priced with the count of its last sub-fragment
// (kernels3.cu:3155; oracle/graal_oracle.c:walk_trans)
__device__ __forceinline__ float ex_pair_first(const End& X, const Stat& sx, int slx, const End& Y, const Stat& sy, int sly, float nfpb,
                                               const Par& p, bool quirk)
{
    if (X.label != Y.label) {
        int ax = stat_accu(sx, slx);
        if (quirk && !X.fwd) ax = stat_accu(sx, sx.n - 1);
        return p.v_inter * ((float)(ax * stat_accu(sy, sly)) / nfpb);
    }
    return ex_pair(X, sx, slx, Y, sy, sly, nfpb, p);
}

// float factorial of kernels3.cu:80-93 and evaluate_likelihood_double (kernels3.cu:191-210)
__device__ __forceinline__ float factorial_f(float n)
{
    float result = 1.0f;
    n = floorf(n);
    if (n < 10.0f) { for (int c = 1; c <= (int)n; c++) result = result * (float)c; }
    else result = powf(n, n) * expf(-n) * sqrtf((float)(2.0 * M_PI * (double)n));
    return result;
}
__device__ __forceinline__ double lik_double(double ex, double ob)
{
    double res = 0.0;
    if (ex != 0.0) {
        const double ln_ex = mm_ln((float)ex);   // (ex is a float32 expected value in double)
        if (ob >= 15.0) res = ob * ln_ex - ex - (ob * log(ob) - ob + log(sqrt(ob * 2.0 * M_PI)));
        else if (ob > 0.0) res = ob * ln_ex - ex - log((double)factorial_f((float)ob));
        else if (ob == 0.0) res = -ex;
    }
    return res;
}

// one copy of a bin in some layout
struct CopyView { End e; bool active; };

// View of fragment f in the CURRENT layout (VIEW = 0) or under candidate `op` of a neighbour (VIEW = 1)
struct CandCtx { const NbTables* T; int op, fA; };
template <int VIEW>
__device__ __forceinline__ CopyView copy_view(const RepArgs& R, int f, const CandCtx& C)
{
    const Geo g = R.geo[f];
    CopyView v;
    v.active = geo_active(g.flags);
    if (VIEW == 0) { v.e = end_cur(g, R.lcontbp, f); return v; }
    const int p = piece_of(C.T->key, g.id_c, geo_pos(g.flags));
    if (p > 0) v.e = end_xf(g, C.T->xf[C.op][p]); else v.e = end_cur(g, R.lcontbp, f);
    if (C.op == 8 && f == C.fA && ((g.flags >> 3) & 1)) v.active = !v.active; // swap_activity_frag (kernels3.cu:283)
    return v;
}

// Poisson log-likelihood of the pixel (bin lo < bin hi), or of bin lo's own upper triangle (lo == hi); body of
// evaluate_likelihood / sub_compute_likelihood for bins with copies (kernels3.cu:2895-3220): float32 expected values
// accumulated over the active copy pairs in dispatcher order, float64 log-likelihood summed over the slots.
template <int VIEW>
__device__ double pixel_lik(const RepArgs& R, int lo, int hi, const CandCtx& C)
{
    const bool diag = lo == hi;
    const Stat si = R.stat_bin[lo], sj = R.stat_bin[hi];
    float ex[3][3] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
    const int i0 = R.dispatcher[2 * lo], i1 = R.dispatcher[2 * lo + 1], j0 = R.dispatcher[2 * hi], j1 = R.dispatcher[2 * hi + 1];
    for (int ri = i0; ri < i1; ri++) {
        const CopyView vi = copy_view<VIEW>(R, R.collector[ri], C);
        if (!vi.active) continue;
        for (int rj = j0; rj < j1; rj++) {
            const CopyView vj = copy_view<VIEW>(R, R.collector[rj], C);
            if (!vj.active) continue;
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b < 3; b++)
                    if (a < si.n && b < sj.n) ex[a][b] = ex[a][b] + ex_pair_first(vi.e, si, a, vj.e, sj, b, R.nfpb, R.par, R.quirk != 0);
        }
    }
    // observations: symmetric matrix, stored by rows of the repeated bins' sub-fragments
    const int di = R.dup_index[lo], dj = R.dup_index[hi];
    const int* ids_i = R.sub_ids + 4 * lo;
    const int* ids_j = R.sub_ids + 4 * hi;
    double val = 0.0;
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
            if (a >= si.n || b >= sj.n || (diag && b <= a)) continue;
            const float ob = di >= 0 ? R.obs[((size_t)di * 3 + a) * R.n_sub_total + ids_j[b]]
                                     : R.obs[((size_t)dj * 3 + b) * R.n_sub_total + ids_i[a]];
            val = lik_double((double)ex[a][b], (double)ob) + val;
        }
    return val;
}

// which pixel does thread (ui, v) own?  Every pixel with at least one repeated bin exactly once.
__device__ __forceinline__ bool rep_pixel(const RepArgs& R, int ui, int v, int& lo, int& hi)
{
    const int u = R.dup_bins[ui];
    if (v != u && R.dup_index[v] >= 0 && v < u) return false; // pair of two repeated bins: owned by the smaller one
    lo = u < v ? u : v; hi = u < v ? v : u;
    return true;
}

// full likelihood of the repeated bins' pixels in the current layout (Q sum)
__global__ __launch_bounds__(256) void k_rep_full(RepArgs R, int rank, int world, long long* __restrict__ out, long long* __restrict__ bad_flag)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long q = 0;
    // (several ranks: the repeated bins' pixels are dealt to them one by one -- each is rounded to Q by itself, so the ranks' sums add up to the
    // single rank's bit for bit)
    if (i < (long long)R.n_dup * R.n_bins && (int)(i % (long long)world) == rank) {
        int lo, hi;
        CandCtx C; C.T = nullptr; C.op = 0; C.fA = -1;
        if (rep_pixel(R, (int)(i / R.n_bins), (int)(i % R.n_bins), lo, hi)) q = to_q(pixel_lik<0>(R, lo, hi, C));
        if (q == Q_BAD) { q = 0; atomicOr((unsigned long long*)bad_flag, 1ull); }
    }
    q = wave_sum_ll(q);
    if ((threadIdx.x & 63) == 0 && q != 0) atomicAdd((unsigned long long*)out, (unsigned long long)q);
}

// candidate deltas of the repeated bins' pixels: thread = (neighbour k, repeated bin, other bin); skipped when no copy of the
// repeated bin lies in contig(fA) u contig(fB_k) (the reference's list of repeats in the sub-index, cuda_lib_gl.py:2466)
__global__ __launch_bounds__(256) void k_rep_delta(RepArgs R, const NbTables* __restrict__ tabs, const long long* tm_done,
                                                    long long seq, int fA, int K, int rank, int world,
                                                    long long* __restrict__ acc, unsigned long long* counters)
{
    __shared__ long long s_acc[MAXK * N_OPS];
    __shared__ int s_ok;
    const int t = threadIdx.x;
    for (int i = t; i < MAXK * N_OPS; i += blockDim.x) s_acc[i] = 0;
    if (t == 0) s_ok = 1;
    __syncthreads();
    if (t < K) { // the tables come from k_tm on another stream
        bool ok = false;
        for (int spin = 0; spin < (1 << 22); spin++) {
            const unsigned long long w = (unsigned long long)__hip_atomic_load(&tm_done[t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(w >> 32) == (unsigned)seq) { ok = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok) { s_ok = 0; atomicOr(&counters[6], 1ull); }
    }
    __syncthreads();
    const long long per_k = (long long)R.n_dup * R.n_bins;
    const long long i = (long long)blockIdx.x * blockDim.x + t;
    if (s_ok && i < per_k * K && (i % world) == rank) {
        const int k = (int)(i / per_k);
        const long long j = i % per_k;
        const NbTables* T = tabs + k;
        int lo, hi;
        if (T->fB != fA && rep_pixel(R, (int)(j / R.n_bins), (int)(j % R.n_bins), lo, hi)) {
            const int u = R.dup_bins[j / R.n_bins];
            bool touched = false;
            for (int r = R.dispatcher[2 * u]; r < R.dispatcher[2 * u + 1]; r++) {
                const Geo g = R.geo[R.collector[r]];
                touched = touched || piece_of(T->key, g.id_c, geo_pos(g.flags)) > 0;
            }
            if (hi != lo && R.dup_index[lo] >= 0 && R.dup_index[hi] >= 0) {
                // both repeated: the reference re-evaluates the pixel only if BOTH bins have a copy in the two contigs (its
                // "repeats x repeats" range runs over the repeats of the sub-index, kernels3.cu:3368-3373) -- also when an
                // activity swap of one of them would change it: kept, so that candidate scores are the reference's
                const int w = lo == u ? hi : lo;
                bool touched_w = false;
                for (int r = R.dispatcher[2 * w]; r < R.dispatcher[2 * w + 1]; r++) {
                    const Geo g = R.geo[R.collector[r]];
                    touched_w = touched_w || piece_of(T->key, g.id_c, geo_pos(g.flags)) > 0;
                }
                touched = touched && touched_w;
            }
            if (touched) {
                CandCtx C; C.T = T; C.op = 0; C.fA = fA;
                const long long q_old = to_q(pixel_lik<0>(R, lo, hi, C));
                for (int op = 0; op < N_OPS; op++) {
                    C.op = op;
                    const long long q_new = to_q(pixel_lik<1>(R, lo, hi, C));
                    if (q_new == Q_BAD || q_old == Q_BAD) { nf_flag(counters + NF_OFF, k, op); continue; }
                    const long long dq = q_new - q_old;
                    if (dq != 0) atomicAdd((unsigned long long*)&s_acc[k * N_OPS + op], (unsigned long long)dq);
                }
            }
        }
    }
    __syncthreads();
    for (int i2 = t; i2 < K * N_OPS; i2 += blockDim.x)
        if (s_acc[i2] != 0) atomicAdd((unsigned long long*)&acc[i2], (unsigned long long)s_acc[i2]);
}

// ------------------------------------------------------------------ host side
struct HostStep;                 // host_step.h: the sampler's per-step host logic
void hs_free(HostStep* p);

struct Ctx {
    int device = 0;
    HostStep* hs = nullptr;
    std::string err;
    hipStream_t stream = nullptr;
    hipStream_t aux = nullptr;    // k_tm runs here, concurrently with k_scan on the main stream
    hipStream_t fstream = nullptr; // a full evaluation that runs NEXT TO a step's scoring kernels (graal_step, flag 8)
    hipEvent_t ev_full = nullptr;  // the relabel in front of it
    hipEvent_t ev_fin = nullptr;  // end of the last asynchronous evaluation (orders the next k_tm after it)
    hipEvent_t ev_relabel = nullptr; // end of the relabel kernels graal_begin_step left running (orders the next k_tm, on aux)
    hipEvent_t ev_tm = nullptr;   // end of the step's k_tm (orders a chip-filling k_fin behind it, see launch_fin)
    hipEvent_t ev_par = nullptr, ev_par2 = nullptr;   // graal_set_params: the parameter update fenced against the auxiliary streams
    // the commit's own-pixel correction (k_apply, own_pixel_q): observed counts of every bin's own sub-fragment pairs; what the statistics'
    // publications have delivered since somebody last took it (graal_take_carry_correction / graal_step with flag 16)
    float* d_own_obs = nullptr;
    bool apply_had_own = false;      // the last commit computed it
    bool all_uniform = false;        // every bin's sub-fragments carry ONE RF count: the trans-branch indexing (GRAAL_MODE_REF_TRANS_ACCU) changes nothing
    int corr_src = 0;                // the statistics in flight come from: 1 = exactly one commit (k_apply's rows), 2 = no change, 0 = neither (unknown)
    long long carry_q = 0;
    bool carry_bad = false;
    long long* d_own_acc = nullptr;  // k_own_corr: [0] sum, [1] unknown terms, [2] ticket
    long long* h_guard = nullptr;    // pinned host, 8 words: k_incr's out-of-range record ([0] != 0: tripped)
    long long* h_own = nullptr;      // pinned host: [0] the commit's number, [1] its correction (Q), [2] its unknown terms
    long long own_seq = 0;
    bool own_pending = false;        // a correction kernel is out and its result has not been taken from h_own
    int peer_repeats = 0;            // eval_sync: repeats of the step in progress because ANOTHER rank's part of it failed (host exchange)
    bool carry_wanted = false;       // somebody takes the corrections (graal_take_carry_correction was called): only then are they computed
    bool own_complete = false;       // d_own_obs was handed in by the caller (graal_upload_own_obs): it covers the WHOLE contact list, not this rank's shard
    bool corr_inflight = false;      // between begin_step_launch and begin_step_collect
    bool corr_skip = false;          // the caller holds a full evaluation of the layout whose commit has not been relabelled yet: its correction is void
    bool full_rep_sharded = false; // the full evaluation in flight dealt the repeated bins' pixels to the ranks (full_launch -> full_collect)
    bool args_synced = false;     // the device-resident argument blocks are complete (sync_args ran with parameters and sub-fragment tables in place)
    bool relabel_pending = false;
    // graal_step's deferred flow (begin_step_launch(defer)): the next k_tm is ordered behind the relabel by a device flag that the
    // next k_scan sets, not by an event; and it publishes the layout statistics in an extra block
    bool relabel_spin_pending = false, stats_pub_pending = false;
    bool spin_ok = getenv("GRAAL_NO_TM_SPIN") == nullptr;   // switched off when k_tm and k_scan turn out not to run concurrently
    bool spin_ok_saved = true;                                // ... and while an RCCL communicator is attached (graal_attach_rccl / graal_detach_rccl)
    unsigned long long relabel_flag_seq = 0, scan_relabel_seq = 0;
    bool pub_in_flight = false;   // ... and its k_tm carries the statistics' publication block (re-armed if the evaluation is repeated)
    unsigned long long tm_spin_ticks = getenv("GRAAL_TM_SPIN_TICKS") ? strtoull(getenv("GRAAL_TM_SPIN_TICKS"), nullptr, 10) : 2000000ull;   // 100 MHz ticks
    bool spin_used = false;       // the evaluation in flight relies on the flag (eval_sync repeats it with an event if k_tm gives up)
    // k_strict2 behind k_gprep through a word in memory instead of an event (launch_strict): GRAAL_STRICT_GWAIT=0 orders them by the event,
    // GRAAL_GP_WAIT_TICKS bounds the in-kernel wait (100 MHz ticks; 1 = give up at once: the test hook that forces the repeat-behind-events path),
    // GRAAL_GP_ACQUIRE=0: only a block that had to wait runs the agent-scope acquire (round 4's form).  Default 1: every block's first wave runs
    // it behind its poll -- poll, acquire, wait for the invalidate, barrier, plain loads: the consumer form the memory model asks for, whatever
    // the caches held (MI355X_MICROARCH.md, inter-workgroup visibility); not measurable at the C2 stand-in (117-127 us per step either way)
    bool gwait_env = getenv("GRAAL_STRICT_GWAIT") == nullptr || atoi(getenv("GRAAL_STRICT_GWAIT")) != 0;
    int gp_wait_ticks = getenv("GRAAL_GP_WAIT_TICKS") ? std::max(1, atoi(getenv("GRAAL_GP_WAIT_TICKS"))) : 200000;   // 2 ms
    int gp_acquire = getenv("GRAAL_GP_ACQUIRE") ? atoi(getenv("GRAAL_GP_ACQUIRE")) : 1;
    // run counters (graal_run_counters): evaluations, steps repeated behind events after an in-kernel wait ran out, k_strict2 launches that
    // followed k_gprep through the word / behind the event, k_strict_flat launches, steps k_tm's finisher handed to a finishing kernel
    long long rc_evals = 0, rc_repeats = 0, rc_gwait = 0, rc_gevent = 0, rc_flat = 0, rc_need_fin = 0;
    bool begin_launched = false;  // graal_begin_step_launch ran for the current layout; graal_begin_step only has to wait
    bool stats_from_apply = false; // the last commit published the statistics of the layout it produced (sequence stats_seq)
    bool fin_pending = false;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> ring; // pairs of events around k_scan, one pair per call (graal_scan_times)
    long long ring_calls = 0;
    std::vector<hipEvent_t> sring; // pairs of events around the tiled reference-arithmetic kernel of a call (k_strict2 / k_strict; graal_strict_times)
    long long sring_calls = 0;
    bool ev_this_call = false;     // the evaluation being launched carries the event pairs
    bool timing_valid = false;
    bool scan_ready = false;
    bool publish = false;         // k_post publishes the sums to pinned host memory (synchronous single-GPU path)
    bool want_events = true;      // record the per-kernel HIP events (graal_last_timing)
    // problem
    int n = 0, n_bins = 0, n_sub_total = 0;
    long long nnz = 0;
    bool single_sub = true;
    float nfpb = 1.0f;
    Par par{};
    bool have_par = false, have_sub = false, have_frags = false, have_contacts = false, order_valid = false;
    int n_contigs = 0;
    double t_all = 0.0;         // layout independent all-trans expected mass
    std::vector<long long> t_hist;          // its inputs: number of sub-fragments per RF count,
    std::map<long long, long long> t_self;  // and of slot pairs inside one bin per product of RF counts
    long long c_lf_q = 0;       // sum of the log-factorial terms of this shard's contacts, each rounded to Q: an integer sum, so
                                // the full likelihood is bit-identical for any sharding of the list (a double sum depends on the order)
    std::vector<int> h_accu;    // [n_bins][3]
    int ln_lut_n = 0;           // entries of k_full_nnz's ln(trans) table (0: none)
    std::vector<int> h_nsub;
    std::vector<Stat> h_stat;   // per bin (host copy: the per-fragment table is derived from it at upload_frags)
    std::vector<int> h_sub_id;  // [n_bins][4]
    std::vector<int> h_sub2bin; // [n_sub_total] bin << 2 | slot
    // repeated bins (graal_upload_repeats)
    bool has_rep = false;
    int n_dup = 0;
    std::vector<int> h_dup_index; // [n_bins] -> index into the repeated bins, -1
    int *d_dup_bins = nullptr, *d_dup_index = nullptr, *d_dispatcher = nullptr, *d_collector = nullptr, *d_sub_ids_all = nullptr;
    float* d_rep_obs = nullptr;
    Stat* stat_frag = nullptr;    // [n] statistics per FRAGMENT (copies of repeated bins: zero sub-fragments)
    // device
    int* soa_mem[2] = {nullptr, nullptr};
    SoaPtr soa[2];
    int cur = 0;
    Geo* geo = nullptr;
    Link* link = nullptr;
    int* mates = nullptr;
    Stat* stat = nullptr;
    SubRec* sub_rec = nullptr;    // [n_sub_total] per sub-fragment record of the full evaluation (k_subrec)
    SubRec8* sub_rec8 = nullptr;  // the compact form, used when every sub-fragment has the same RF count (uniform_accu > 0)
    unsigned short* sub_lab16 = nullptr; // low 16 bits of every sub-fragment's label (k_full_nnz_l keeps them in LDS)
    int uniform_accu = 0;
    int* sub2bin = nullptr;
    int *row = nullptr, *col = nullptr, *cnt = nullptr;
    QRaw* queue = nullptr;
    int last_fA = 0, last_K = 0, last_max_id = 0; // proposal of the last evaluation (timing replays of the scan)
    int last_fA_launch = 0;       // fA of the evaluation being launched (k_fin may be launched later, from eval_sync)
    int last_fB[MAXK] = {};
    int max_lcont = 0;            // longest contig at the last graal_begin_step (sizes k_fin's grid)
    int lcont_bound = 0;          // upper bound of the longest contig NOW (graal_step scores before it has the statistics)
    int* d_sub_ids = nullptr;     // [n_bins][4], only when some bin has more than one sub-fragment
    unsigned long long *keys = nullptr, *keys_sorted = nullptr;
    int *o2n = nullptr, *perm = nullptr, *cbase = nullptr;
    int* pstart = nullptr;         // start_bp in position order (the position index's twin: written wherever it is)
    int *len_of2[2] = {nullptr, nullptr}, *contig_off2[2] = {nullptr, nullptr}; // per layout buffer (see k_incr)
    long long* d_part = nullptr;  // [<= 1024][N_STAT] the commit kernel's per-block statistics (published by the next k_incr)
    Changed* d_chg = nullptr;     // [2] commit records: a commit fills one, the relabel that consumes it clears the other
    int chg_w = 0, chg_last = 0;
    int apply_blocks = 0;         // grid of the last k_apply = rows of d_part
    bool ranks_valid = false;     // labels of buffer `cur` are ranks and its len/offset arrays are current
    int pending_commits = 0;      // commits since the last graal_begin_step
    bool incr_ok = false;         // the single pending commit started from a ranked layout with the right max_id
    void* cub_tmp = nullptr;
    size_t cub_tmp_bytes = 0;
    NbTables* tabs = nullptr;
    int* step_hdr = nullptr;      // [MAXK] mass work items per neighbour of the current step, [MAXK..] priced by k_tm
    long long* tm_done = nullptr; // [MAXK] sequence number of the step whose tables are complete
    long long* d_acc = nullptr;   // K*13 running sums (self-cleaning: the step's last block zeroes them after reading)
    unsigned long long* d_sync = nullptr; // [0] k_tm ticket [8] finished blocks of k_scan
    unsigned* d_flags = nullptr;          // k_scan's per-block completion flags
    unsigned long long scan_done_total[N_DONE] = {}; // per completion counter: blocks of all non-dry scans launched so far
    unsigned long long* d_done = nullptr;    // the N_DONE completion counters of k_scan, DONE_STRIDE words apart
    unsigned long long* d_wq = nullptr;      // k_fin's work-queue counters (2 x N_WQ, WQ_STRIDE words apart), zero at rest
    int mode = 0;                 // GRAAL_MODE_* flags (graal_set_mode)
    unsigned long long* d_slist = nullptr;   // k_strict's unit list (k_strict_cull fills it), slist_cap entries
    unsigned long long* d_slist_n = nullptr; // its length, a word of d_scalars (zero at rest: k_strict's last block clears it)
    unsigned long long slist_cap = 0;
    unsigned long long slist_floor = 0;      // entries the list holds at least: raised when a step's list overflowed (eval_sync grows it and repeats the step)
    unsigned long long slist_worst = 0;      // the last launch's worst case (every tile pair of the union set listed)
    long long rc_list_grown = 0;
    long long rc_carry_repairs = 0;  // graal_step (flag 16): steps whose own-pixel correction was unknown and that evaluated the layout in full instead
    unsigned long long slist_soft_cap = getenv("GRAAL_SLIST_SOFT_CAP") ? std::max<unsigned long long>(64ull, strtoull(getenv("GRAAL_SLIST_SOFT_CAP"), nullptr, 10))
                                                                        : (1ull << 23);   // 8 M entries = 64 MB (C5's 7 contigs list ~1e5); tests set it small
    USet* d_uset = nullptr;       // reference arithmetic over the step's union set (strict2.h): the set, the classes per pair of global pieces
    GClass* d_cls = nullptr;
    int* d_cls_n = nullptr;
    unsigned scan_token = 0x5ca90000u; // k_scan launches so far (ScanArgs.token)
    double* d_ln_tab = nullptr;   // [LN_TRANS_LUT] ln of the trans value by RF-count product (k_ln_tab; rebuilt by sync_args)
    int* d_ubins = nullptr;       // bins whose sub-fragments carry different RF counts (k_quirk_mass)
    int n_ubins = 0;
    bool finisher_ok = true;      // k_tm's last block may finish short-contig steps (switched off when the kernels turn
                                  // out not to run concurrently, e.g. under a profiler that serialises dispatches)
    int gave_up = 0;
    int event_every = 8;          // a HIP event pair around k_scan on every n-th evaluation (they cost a few us of gaps)
    long long eval_calls = 0;
    DevArgs* d_args = nullptr;    // [2]: one argument block per layout buffer
    // GRAAL_EVAL_TIMING: host clock of the synchronous evaluation, printed when the context is destroyed (us per step: launches, until
    // k_tm asked for the finishing kernels, their launches, until the result) -- [0] steps, [1] steps that needed them
    // reference arithmetic, one rank: `mid_run` = k_strict_flat goes out behind every scan and k_tm's last block does not try to
    // finish the step (set while most steps need more than k_tm: need_ema, a running mean of "this step did"); flat_tried = this
    // step's flat kernel has been launched (if it also says NEED_FIN, the tiled kernels follow)
    bool mid_run = false, flat_tried = false, step_needed_fin = false, step_needed_geom = false;
    double need_ema = 0.0;
    bool eval_timing = getenv("GRAAL_EVAL_TIMING") != nullptr;
    double et[6] = {0, 0, 0, 0, 0, 0};
    long long et_n[2] = {0, 0};
    long long* h_res = nullptr;   // pinned host: [0] sequence number of the published step, [1..] K*13 sums
    long long* h_stats = nullptr; // pinned host: [0] sequence number, [1..16] the statistics words of k_stats_fin
    long long* h_full = nullptr;  // pinned host: [0] sequence number, [1..4] the sums / flags of the last full evaluation (k_full_pub)
    long long full_seq = 0;
    // where the step's last block publishes: h_res, or -- with an exchange attached -- this rank's slot of the step's
    // parity in a host segment shared by the ranks of the node (host and device views of the same memory)
    long long* res_host = nullptr;
    long long* res_dev = nullptr;
    long long* x_host = nullptr;  // exchange segment: [2 parities][world ranks][X_SLOT_WORDS], registered with HIP
    long long* x_dev = nullptr;
    size_t x_bytes = 0;
    int x_rank = 0, x_world = 1;
    void* nccl_comm = nullptr;    // graal_attach_rccl: the ranks' Q vectors are summed by ONE ncclAllReduce on the engine's stream (eval_sync)
    int n_rank = 0, n_world = 1;
    long long stats_seq = 0;
    int4* d_dref = nullptr;       // genome-distance reference (graal_upload_distance_ref)
    unsigned long long* d_dist = nullptr; // [0] sum, [1] ticket
    long long* h_dist = nullptr;  // pinned host: [0] sequence number, [1] half units
    long long dist_seq = 0;
    long long seq = 0;
    long long* d_scalars = nullptr; // [0..7] stats, [8..9] full q, [10..12] step counters, [13] stale (int), [14] #circ,
                                    // [15] ticket, [16] error, [18..20] counters of the last finished step
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

// histograms behind T_all (they depend on the RF counts and on which bins are repeats, not on the parameters)
void build_t_hist(Ctx* h)
{
    h->t_hist.clear(); h->t_self.clear();
    if (!h->have_sub) return;
    auto nsub = [&](int b) { return (h->has_rep && h->h_dup_index[b] >= 0) ? 0 : h->h_nsub[b]; }; // repeated bins: priced densely
    for (int b = 0; b < h->n_bins; b++)
        for (int s = 0; s < nsub(b); s++) {
            const int a = h->h_accu[3 * b + s];
            if ((size_t)a >= h->t_hist.size()) h->t_hist.resize(a + 1, 0);
            h->t_hist[a]++;
            for (int t = 0; t < nsub(b); t++) h->t_self[(long long)a * h->h_accu[3 * b + t]]++;
        }
}

void compute_t_all(Ctx* h, bool rebuild = true)
{
    // T_all = sum over pairs of DIFFERENT bins, all slot pairs, of float32(v * float32(float32(ax*ay)/nfpb)); a nuisance-
    // parameter step changes v_inter only: the histograms are kept (the per-bin loops were 0.2 ms of each graal_set_params at C5)
    if (rebuild) build_t_hist(h);
    if (!h->have_par || !h->have_sub) return;
    const std::vector<long long>& hist = h->t_hist;
    auto c = [&](long long m) { return (double)(h->par.v_inter * ((float)(int)m / h->nfpb)); };
    double all = 0.0;
    for (size_t u = 0; u < hist.size(); u++)
        if (hist[u])
            for (size_t w = 0; w < hist.size(); w++)
                if (hist[w]) all += (double)hist[u] * (double)hist[w] * c((long long)u * (long long)w);
    double self = 0.0;
    for (const auto& kv : h->t_self) self += (double)kv.second * c(kv.first);
    h->t_all = 0.5 * (all - self);
}

int reach_bp(const Ctx* h) { return (int)ceil((double)h->par.d_max * 1000.0) + 1000; }

// GRAAL_DEBUG_ADDR=1: where the engine's buffers are (stderr, whenever a pointer changes) -- so that the address of a "Memory access fault by GPU"
// can be matched to a buffer (the open fault of the two-ranks-on-one-GPU rehearsal)
void debug_print_buffers(const Ctx* h, const char* when)
{
    static const bool on = getenv("GRAAL_DEBUG_ADDR") != nullptr;
    if (!on) return;
    const size_t n = (size_t)h->n, nnz = (size_t)h->nnz;
    fprintf(stderr, "[graal addr, pid %d, %s] n %zu nnz %zu | soa0 %p soa1 %p | row %p col %p cnt %p (%zu B each) queue %p (%zu B) | geo %p link %p perm %p pstart %p cbase %p mates %p | "
                    "stat_frag %p sub2bin %p sub_rec %p | tabs %p d_args %p d_scalars %p d_acc %p d_sync %p d_done %p d_flags %p tm_done %p d_part %p d_chg %p | "
                    "d_slist %p (%llu entries) d_uset %p d_cls %p d_cls_n %p | x_dev %p x_host %p res_dev %p h_res %p h_stats %p h_full %p h_own %p\n",
            (int)getpid(), when, n, nnz, (void*)h->soa_mem[0], (void*)h->soa_mem[1], (void*)h->row, (void*)h->col, (void*)h->cnt, nnz * 4, (void*)h->queue, (nnz + 8) * sizeof(QRaw),
            (void*)h->geo, (void*)h->link, (void*)h->perm, (void*)h->pstart, (void*)h->cbase, (void*)h->mates, (void*)h->stat_frag, (void*)h->sub2bin, (void*)h->sub_rec,
            (void*)h->tabs, (void*)h->d_args, (void*)h->d_scalars, (void*)h->d_acc, (void*)h->d_sync, (void*)h->d_done, (void*)h->d_flags, (void*)h->tm_done, (void*)h->d_part, (void*)h->d_chg,
            (void*)h->d_slist, (unsigned long long)h->slist_cap, (void*)h->d_uset, (void*)h->d_cls, (void*)h->d_cls_n, (void*)h->x_dev, (void*)h->x_host, (void*)h->res_dev, (void*)h->h_res,
            (void*)h->h_stats, (void*)h->h_full, (void*)h->h_own);
}

// (re)write the two device-resident argument blocks; called whenever a pointer, size or parameter changes
int sync_args(Ctx* h)
{
    debug_print_buffers(h, "sync_args");
    DevArgs a[2];
    for (int b = 0; b < 2; b++) {
        memset(&a[b], 0, sizeof(DevArgs));
        a[b].soa = h->soa[b];
        a[b].nnz = h->nnz; a[b].n = h->n; a[b].n_sub_total = h->n_sub_total;
        a[b].bitmap_words = (h->n_sub_total + 31) / 32 + 2;
        a[b].reach_bp = h->have_par ? reach_bp(h) : 0;
        a[b].row = h->row; a[b].col = h->col; a[b].cnt = h->cnt; a[b].sub2bin = h->sub2bin;
        a[b].sub2bin_multi = h->single_sub ? nullptr : h->sub2bin; a[b].sub_ids = h->d_sub_ids;
        a[b].contig_off = h->contig_off2[b]; a[b].perm = h->perm; a[b].geo = h->geo; a[b].stat = h->stat_frag;
        a[b].link = h->link; a[b].cbase = h->cbase;
        a[b].tabs = h->tabs; a[b].step_hdr = h->step_hdr; a[b].tm_done = h->tm_done; a[b].acc = h->d_acc;
        a[b].queue = h->queue; a[b].counters = (unsigned long long*)(h->d_scalars + 10);
        a[b].nfpb = h->nfpb; a[b].par = h->par;
    }
    CK(hipMemcpy(h->d_args, a, sizeof a, hipMemcpyHostToDevice));
    h->args_synced = h->have_par && h->have_sub && h->d_args != nullptr;
    if (h->have_par && h->have_sub && h->ln_lut_n > 0) {
        if (!h->d_ln_tab) CK(hipMalloc(&h->d_ln_tab, sizeof(double) * LN_TRANS_LUT));
        k_ln_tab<<<blocks_for(h->ln_lut_n, 256), 256, 0, h->stream>>>(h->d_ln_tab, h->ln_lut_n, h->nfpb, h->par);
        CK(hipGetLastError());
        CK(hipStreamSynchronize(h->stream));
    }
    return GRAAL_OK;
}

int refresh(Ctx* h)
{
    k_refresh_geo<<<blocks_for(h->n, 256), 256, 0, h->stream>>>(h->soa[h->cur], h->geo, h->link, h->n);
    CK(hipGetLastError());
    return GRAAL_OK;
}

constexpr int MAX_SCAN_BLOCKS = 4096;
constexpr int FULL_BAD = 27; // d_scalars[FULL_BAD]: a term of the last full evaluation was not finite / out of range
constexpr int SLIST_N = 28;  // d_scalars[SLIST_N]: length of k_strict's unit list (zero at rest)
constexpr int RELABEL_FLAG = 30; // d_scalars[RELABEL_FLAG]: sequence number of the last relabel k_scan has announced as complete (k_tm spins on it)
constexpr int SCAN_LDS_MAX = 48 * 1024; // affected bitmap of k_scan: 1 bit per contact-list id up to 393,216 ids, folded beyond (launch_scan)

// threads per block of the streaming pass: 1024 (two blocks per CU) for the lists it is built for; a list of a few hundred
// thousand contacts (a yeast-sized genome at 3 sub-fragments per bin) would fill only a handful of such blocks, each marking
// the whole affected set in its prologue: 256-thread blocks there (C2 stand-in: 177 -> 165 us per step)
int scan_threads_cfg(const Ctx* h)
{
    static const int v = getenv("GRAAL_SCAN_THREADS") ? atoi(getenv("GRAAL_SCAN_THREADS")) : 0;
    return v > 0 ? v : (h->nnz < 2000000 ? 256 : 1024);
}

int scan_groups_cfg()
{
    static const int e = getenv("GRAAL_SCAN_G") ? atoi(getenv("GRAAL_SCAN_G")) : 4;
    static const int v = (e == 8 || e == 2) ? e : 4;
    return v;
}

// How k_tm's finishing block learns that the scan is complete.  Default: every scan block adds itself to one counter with a
// fire-and-forget atomic and the finisher polls that one word.  GRAAL_SCAN_DONE=flags: one flag line per block, all 496 of
// them polled -- measured 1.5 us slower per step (each polling round is 8 uncached loads per lane behind the scan's own
// stream in the memory queues: scan complete -> seen took ~7 us).
bool scan_done_counter()
{
    static const bool v = getenv("GRAAL_SCAN_DONE") ? (strcmp(getenv("GRAAL_SCAN_DONE"), "flags") != 0) : true;
    return v;
}

int scan_done_n()
{
    static const int e = getenv("GRAAL_SCAN_DONE_N") ? atoi(getenv("GRAAL_SCAN_DONE_N")) : N_DONE;
    return e >= 1 && e <= N_DONE ? e : N_DONE;
}

int scan_grid(const Ctx* h)
{
    // two 1024-thread blocks per CU fill the 256 CUs; 16 fewer leave room for k_tm's blocks, which run at the same time (a
    // CU that hosts one of them takes only one scan block, and a scan block that has to wait for a slot ends 8 us late)
    static const int scan_blocks = getenv("GRAAL_SCAN_BLOCKS") ? atoi(getenv("GRAAL_SCAN_BLOCKS"))
                                                               : (scan_groups_cfg() == 8 ? 256 - 8 : 256 * 2 - 16);
    const long long groups = (h->nnz >> 2) + 1;
    const long long per_block = (long long)scan_groups_cfg() * scan_threads_cfg(h); // groups one block takes per iteration
    return (int)std::max<long long>(1, std::min<long long>((groups + per_block - 1) / per_block, scan_blocks));
}

// the streaming pass (see k_scan); dry = timing replay that counts relevant contacts but queues nothing
int launch_scan(Ctx* h, int fA, const Neigh& nb, int K, int max_id, int dry, hipStream_t st, bool finisher_reads = true)
{
    const int nbk = scan_grid(h), scan_threads = scan_threads_cfg(h);
    // one bit per contact-list id while they fit the LDS budget (393,216 ids); beyond that the ids are folded onto 2^18 bits
    // (ScanArgs::bm_wmask): with a' affected ids a fraction a' / 2^18 of the rows takes the second test for nothing and
    // (a' / 2^18)^2 of the contacts is queued for nothing -- dropped by the consumers' membership test, results unchanged.
    // GRAAL_SCAN_FOLD_BITS = b folds onto 2^b bits whatever the size (tests: b = 10 makes every other row a false positive)
    static const int fold_bits = getenv("GRAAL_SCAN_FOLD_BITS") ? std::min(18, std::max(5, atoi(getenv("GRAAL_SCAN_FOLD_BITS")))) : 0;
    size_t shm = (size_t)((h->n_sub_total + 31) / 32 + 2) * 4;
    unsigned wmask = 0xffffffffu;
    if (shm > (size_t)SCAN_LDS_MAX || fold_bits) {
        const int b = fold_bits ? fold_bits : 18;
        if ((size_t)1 << b < (size_t)h->n_sub_total) { wmask = (1u << (b - 5)) - 1u; shm = ((size_t)1 << (b - 5)) * 4; }
    }
    ScanArgs sa;
    sa.geo = h->geo; sa.link = h->link; sa.mates = h->mates; sa.cnt = h->cnt;
    sa.cbase = h->cbase; sa.perm = h->perm; sa.sub_ids = h->d_sub_ids;
    sa.row4 = reinterpret_cast<const int4*>(h->row); sa.nnz = h->nnz; sa.bitmap_words = (int)(shm / 4); sa.bm_wmask = wmask;
    sa.col4 = reinterpret_cast<const int4*>(h->col); sa.geo2 = reinterpret_cast<const int2*>(h->geo); sa.sub2bin = h->sub2bin;
    sa.queue = h->queue; sa.counters = (unsigned long long*)(h->d_scalars + 10);
    sa.flags = h->d_flags; sa.seq32 = (unsigned)h->seq;
    sa.strict = (h->mode & GRAAL_MODE_STRICT) ? 1 : 0;
    sa.wt_queue = finisher_reads ? 1 : 0;
    sa.token = ++h->scan_token;
    sa.nc = h->d_scalars + NC_WORD;
    sa.relabel_flag = (unsigned long long*)(h->d_scalars + RELABEL_FLAG);
    sa.relabel_seq = dry ? 0ull : h->scan_relabel_seq;
    sa.done = (scan_done_counter() && !dry) ? h->d_done : nullptr;
    sa.n_done = scan_done_n();
    if (sa.done) for (int c = 0; c < sa.n_done; c++) h->scan_done_total[c] += (unsigned long long)((nbk - c + sa.n_done - 1) / sa.n_done);
    if (nbk > MAX_SCAN_BLOCKS) return fail(h, GRAAL_E_ARG, "GRAAL_SCAN_BLOCKS too large");
    if (scan_groups_cfg() == 8) {
        if (h->single_sub) k_scan<true, 8><<<nbk, scan_threads, shm, st>>>(sa, fA, nb, K, max_id, dry);
        else k_scan<false, 8><<<nbk, scan_threads, shm, st>>>(sa, fA, nb, K, max_id, dry);
    } else if (scan_groups_cfg() == 2) {
        if (h->single_sub) k_scan<true, 2><<<nbk, scan_threads, shm, st>>>(sa, fA, nb, K, max_id, dry);
        else k_scan<false, 2><<<nbk, scan_threads, shm, st>>>(sa, fA, nb, K, max_id, dry);
    } else {
        if (h->single_sub) k_scan<true, 4><<<nbk, scan_threads, shm, st>>>(sa, fA, nb, K, max_id, dry);
        else k_scan<false, 4><<<nbk, scan_threads, shm, st>>>(sa, fA, nb, K, max_id, dry);
    }
    CK(hipGetLastError());
    return GRAAL_OK;
}

// how long a kernel waits for the scan's completion: a generous multiple of the time the streaming pass needs at 2 TB/s, plus launch slack
int fin_wait_ticks(const Ctx* h) { return (int)std::min<long long>(100ll * 50 + (long long)(4.0 * 4.0 * (double)h->nnz / 2.0e12 * 1.0e8), 1ll << 30); }

// dynamic LDS of k_tm's finishing block: the pricing records of all K tables (see k_tm)
constexpr size_t tm_fin_dyn_lds()
{
    return sizeof(PTask) * MAXK * PT_CAP + sizeof(unsigned short) * MAXK * N_PAIRS * PR_WORDS + sizeof(int) * MAXK + 16;
}
size_t fin_dyn_lds(int K) { return (size_t)K * (S_PER_K * sizeof(long long) + (MAX_TASKS + 1) * sizeof(int)); }   // K = 10: 37 KB

// k_fin's blocks spin until k_tm has released the tables.  k_tm is launched first, on the other stream, but nothing guarantees
// that its blocks are PLACED first: if k_fin's grid gets there first (a short scan) and fills every CU, k_tm's blocks have
// nowhere to go, every block of k_fin spins to its bound and the step fails (seen: 2,048 blocks on the C2 stand-in -- registers;
// and once in ~100 runs with 768 blocks after k_tm's LDS had grown -- LDS).  So the largest grid that may spin is the one that
// leaves room for a block of k_tm on every CU, in registers AND in LDS, computed from the kernels' own attributes; a larger
// grid is ordered behind k_tm by an event instead (launch_fin).
int fin_blocks_no_wait(int K)
{
    static size_t lds_tm = 0, lds_fin = 0;
    static int regs_tm = 0, regs_fin = 0;
    if (lds_tm == 0) {
        hipFuncAttributes a;
        if (hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_tm)) == hipSuccess) { lds_tm = a.sharedSizeBytes; regs_tm = a.numRegs; }
        if (hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_fin)) == hipSuccess) { lds_fin = a.sharedSizeBytes; regs_fin = a.numRegs; }
        if (lds_tm == 0) lds_tm = 64 * 1024;
        if (lds_fin == 0) lds_fin = 32 * 1024;
        if (regs_tm <= 0) regs_tm = 128;
        if (regs_fin <= 0) regs_fin = 128;
    }
    const size_t lds_cu = 160 * 1024;   // gfx950
    const int regs_simd = 512;          // VGPRs per lane and SIMD; a 256-thread block puts one wave on each SIMD
    const auto up8 = [](int r) { return (r + 7) & ~7; };
    const int by_lds = (int)((lds_cu - std::min(lds_cu, lds_tm)) / (lds_fin + fin_dyn_lds(K)));
    const int by_regs = (regs_simd - up8(regs_tm)) / up8(regs_fin);
    return 256 * std::max(1, std::min(by_lds, by_regs));
}

int fin_blocks_cfg(const Ctx* h, int K)
{
    static const int fin_blocks_env = getenv("GRAAL_FIN_BLOCKS") ? atoi(getenv("GRAAL_FIN_BLOCKS")) : 0;
    // short contigs leave k_fin a handful of contacts: a small grid keeps its launch and completion ticket cheap.  Contigs of a
    // few hundred fragments: the largest grid that may spin for k_tm's tables (3 blocks per CU unless LDS allows fewer).
    // Contigs of thousands of fragments: long dependent chains, 4 resident blocks per CU and fresh ones as they retire keep the
    // VALUs busiest (2.8 -> 2.2 ms per step on C5's 7 contigs in round 1) -- that grid waits for k_tm through an event.
    if (fin_blocks_env > 0) return fin_blocks_env;
    if (h->max_lcont > 0 && h->max_lcont <= 16) return 32;
    return h->max_lcont > 1024 ? 2048 : std::min(768, fin_blocks_no_wait(K));
}

// left-over mass items, queued contacts, hand-out.  Short contigs leave it a handful of contacts: a small grid keeps its
// launch and completion ticket cheap; long contigs get the whole chip.
int launch_fin(Ctx* h, int K, int rank, int world, long long* d_q_out, bool publish, hipStream_t st)
{
    const int fin_blocks = fin_blocks_cfg(h, K);
    if (fin_blocks > fin_blocks_no_wait(K)) { CK(hipEventRecord(h->ev_tm, h->aux)); CK(hipStreamWaitEvent(st, h->ev_tm, 0)); }   // (see fin_blocks_no_wait)
    static const int fin_skip = getenv("GRAAL_FIN_SKIP") ? atoi(getenv("GRAAL_FIN_SKIP")) : 0;   // (diagnostics: wrong sums)
    FinArgs fa;
    fa.tm_done = h->tm_done; fa.step_hdr = h->step_hdr; fa.counters = (unsigned long long*)(h->d_scalars + 10); fa.queue = h->queue;
    fa.cnt = h->cnt; fa.multi = h->single_sub ? 0 : 1;
    fa.tabs = h->tabs; fa.geo = h->geo; fa.stat = h->stat_frag; fa.acc = h->d_acc; fa.sync = h->d_sync;
    static const int fin_seg = getenv("GRAAL_FIN_SEG") ? atoi(getenv("GRAAL_FIN_SEG")) : 0;
    fa.ln_tab = h->d_ln_tab; fa.lut_n = h->d_ln_tab ? h->ln_lut_n : 0; fa.skip = fin_skip; fa.seg = fin_seg;
    fa.norm_u = h->uniform_accu > 0 ? (float)(h->uniform_accu * h->uniform_accu) / h->nfpb : -1.0f;
    static const int fin_upw = getenv("GRAAL_FIN_UPW") ? std::max(1, atoi(getenv("GRAAL_FIN_UPW"))) : 4;
    fa.upw = fin_upw;
    static const bool fin_static = getenv("GRAAL_FIN_STATIC") != nullptr;   // (units and contact batches dealt statically)
    fa.wq = fin_static ? nullptr : h->d_wq;
    k_fin<<<fin_blocks, 256, fin_dyn_lds(K), st>>>(h->d_args + h->cur, fa, h->last_fA_launch, K, rank, world, d_q_out, publish ? h->res_dev : nullptr, h->seq);
    CK(hipGetLastError());
    return GRAAL_OK;
}

// GRAAL_STRICT_DENSE=1: the O(m^2) validation kernel (k_strict_dense) instead of k_strict_cull + k_strict (tests compare the two)
bool strict_dense_cfg()
{
    static const bool v = getenv("GRAAL_STRICT_DENSE") != nullptr && atoi(getenv("GRAAL_STRICT_DENSE")) != 0;
    return v;
}

// reference arithmetic: what k_tm left (sets larger than STRICT_INLINE_M, the queued contacts when there are many), hand-out
int launch_strict(Ctx* h, int fA, int K, int rank, int world, long long* d_q_out, bool publish, hipStream_t st)
{
    // k_strict_cull needs k_tm's tables and nothing of the scan: it goes out on the AUXILIARY stream, behind k_tm, and runs under
    // the scan; k_strict waits for both (the event on the auxiliary stream, the scan in front of it on its own)
    FinArgs fa;
    fa.tm_done = h->tm_done; fa.step_hdr = h->step_hdr; fa.counters = (unsigned long long*)(h->d_scalars + 10); fa.queue = h->queue;
    fa.cnt = h->cnt; fa.multi = h->single_sub ? 0 : 1;
    fa.tabs = h->tabs; fa.geo = h->geo; fa.stat = h->stat_frag; fa.acc = h->d_acc; fa.sync = h->d_sync;
    fa.ln_tab = nullptr; fa.lut_n = 0; fa.skip = 0; fa.seg = 0; fa.norm_u = -1.0f; fa.upw = 4; fa.wq = nullptr;
    StrictArgs sx;
    sx.perm = h->perm; sx.cbase = h->cbase; sx.lcontbp = h->soa[h->cur].p[F_LCONTBP]; sx.link = h->link; sx.nfpb = h->nfpb; sx.par = h->par;
    sx.quirk = (h->mode & GRAAL_MODE_REF_TRANS_ACCU) ? 1 : 0;
    sx.reach_bp = reach_bp(h);
    sx.list_cap = 0;
    sx.seg = 64;
    if (strict_dense_cfg()) {
        CK(hipEventRecord(h->ev_tm, h->aux));      // (k_tm is on the auxiliary stream, done or not: the event completes behind it)
        CK(hipStreamWaitEvent(st, h->ev_tm, 0));   // the tables are complete before the kernel starts: nobody spins for them
        k_strict_dense<<<1024, 256, 0, st>>>(fa, sx, fA, K, rank, world, d_q_out, publish ? h->res_dev : nullptr, h->seq);
        CK(hipGetLastError());
        return GRAAL_OK;
    }
    static const bool v1 = getenv("GRAAL_STRICT_V1") != nullptr && atoi(getenv("GRAAL_STRICT_V1")) != 0;   // (A/B: the per-neighbour kernels of round 3)
    if (!v1) {
        // the union set's kernels (strict2.h): k_gprep (classes per pair of global pieces + the unit list) on the auxiliary stream behind
        // k_tm, under the scan; k_strict2 waits for both
        if (!h->d_uset) {
            CK(hipMalloc(&h->d_uset, sizeof(USet)));
            CK(hipMalloc(&h->d_cls, sizeof(GClass) * (size_t)US_MAXPAIRS * US_NCAND));
            CK(hipMalloc(&h->d_cls_n, sizeof(int) * US_MAXPAIRS + 1024));   // (+ the units' draw counter, k_gprep's ticket and completion word: a line of its own each)
            CK(hipMemset(h->d_cls_n, 0, sizeof(int) * US_MAXPAIRS + 1024));
            CK(hipDeviceSynchronize());
        }
        const int lc = std::max(std::max(h->max_lcont, h->lcont_bound), 1);
        static const int blocks_env = getenv("GRAAL_STRICT_BLOCKS") ? atoi(getenv("GRAAL_STRICT_BLOCKS")) : 0;
        // (the GRID by the longest contig as last seen -- one commit stale: a performance choice; everything that must HOLD the step is sized by
        // the bound `lc`, twice that + 2.  By the bound, contigs of 130-256 bins went to the 1,024-block grid behind an event instead of the
        // 512-block one that follows k_gprep through its word: GRAAL_S2_GRID_BY_BOUND=1 for A/B)
        static const bool grid_by_bound = getenv("GRAAL_S2_GRID_BY_BOUND") != nullptr;
        const int lg = grid_by_bound ? lc : std::max(h->max_lcont, 1);
        const int blocks = blocks_env > 0 ? blocks_env : (lg <= 64 ? 32 : (lg <= 256 ? 512 : 1024));
        // fragments per tile: 64 (one per lane); with several sub-fragments per bin 32 -- the halves of a wave hold the same 32 fragments and
        // take two fragments of the segment at a time (k_strict2): a unit is a 32 x 4 block instead of a 64 x 2 strip, which wastes fewer lanes
        // on pieces of a few dozen bins and at the window's edge (GRAAL_S2_TILE=64: the strips, for A/B)
        static const int tile_env = getenv("GRAAL_S2_TILE") ? atoi(getenv("GRAAL_S2_TILE")) : 0;
        const int TILE = tile_env == 64 ? 64 : (tile_env == 32 ? 32 : (h->single_sub ? 64 : 32));
        // tiles of the union: at most K + 1 contigs, at most every fragment; + one partial tile per global piece
        const unsigned long long nt = std::min<unsigned long long>((unsigned long long)(K + 1) * (unsigned long long)((lc + TILE - 1) / TILE),
                                                                   (unsigned long long)((h->n + TILE - 1) / TILE + K + 1)) + (unsigned long long)US_MAXP;
        if (nt >= 65536ull) return fail(h, GRAAL_E_STATE, "reference arithmetic: more than 65,535 tiles in a step's union set");
        // the unit list's entries: 4 fragments of the segment side (one sub-fragment per bin; k_strict2 merges up to 4 of them) or 1 (several:
        // up to 2); GRAAL_STRICT_SEG fixes the entry size (entries of one fragment at one sub-fragment per bin: the list's traffic cost the C4
        // stand-in 20 % of its run), GRAAL_STRICT_REP the waves that may share one unit's classes (1, 2, 4, 8)
        static const int seg_env = getenv("GRAAL_STRICT_SEG") ? atoi(getenv("GRAAL_STRICT_SEG")) : 0;
        const int seg_max = h->single_sub ? 16 : (TILE == 32 ? 4 : 2);   // (k_strict2's segment: SEG, seg_cap)
        // (one sub-fragment per bin: entries of 4 fragments -- of 16, a whole unit, once a contig may exceed 512 bins: a unit of 64 x 4 pairs is
        // 13 us of set-up for ~1 us per class, and the kernel merges neighbouring entries only from 49,000 of them on; C4 stand-in, 4 cycles:
        // 214 us per step against 224, the late stage unchanged)
        // (tiles of 32: entries of TWO fragments, one per half of the wave -- FOUR once the longest contig may hold more than 512 bins (the bound,
        // one commit stale: twice the longest + 2): a step there has a few thousand units, more than half the grid's waves, so no two waves share
        // one; with half as many, twice as long, every unit is shared by two waves and none idles.  C3 stand-in (contigs of 350 bins): 142 us
        // per step against 157; C2 stand-in (contigs of 150-220): 108 against 99)
        const int seg_unit = ((seg_env == 1 || seg_env == 2 || seg_env == 4 || seg_env == 8 || seg_env == 16) && seg_env <= seg_max) ? seg_env
                                                                                                 : (h->single_sub ? (lc > 512 ? 16 : 4) : (TILE == 32 ? (lc > 512 ? 4 : 2) : 1));
        static const int rep_env = getenv("GRAAL_STRICT_REP") ? atoi(getenv("GRAAL_STRICT_REP")) : 8;
        const int rep_max = rep_env >= 16 ? 16 : (rep_env >= 8 ? 8 : (rep_env >= 4 ? 4 : (rep_env >= 2 ? 2 : 1)));
        // units the grid wants before k_strict2 merges neighbouring entries into longer units (per wave: 6 at one sub-fragment per bin; with
        // several, a unit's fragment pairs are nine evaluations each -- GRAAL_S2_TARGET_X4: the figure in quarters, for A/B)
        static const int target_env = getenv("GRAAL_S2_TARGET_X4") ? atoi(getenv("GRAAL_S2_TARGET_X4")) : 0;
        const unsigned long long target_x4 = target_env > 0 ? (unsigned long long)target_env : (h->single_sub ? 24ull : 24ull);
        const unsigned long long target = std::max<unsigned long long>(1ull, target_x4 * 4ull * (unsigned long long)blocks / 4ull);
        const unsigned long long pairs_max = nt * (nt + 1ull) / 2ull;
        // The list's worst case -- EVERY tile pair of the union listed -- is quadratic in the union's size (2e9 entries for 1e6 fragments in a few
        // contigs), while the interval cull lists the pairs within reach of each other under some candidate: orders of magnitude fewer.
        // It is sized ONCE per layout size, for the largest union n fragments and MAXK neighbours can form -- not for this step's longest
        // contig: growing it with the contigs meant a hipFree / hipMalloc behind two stream synchronizes in the middle of a run, again and
        // again while an assembly's contigs grow.  One rank: SLIST_SOFT_CAP entries at most to begin with; if a step's list overflows,
        // k_gprep says so (counters[6] bit 1), the step ends as failed, eval_sync raises the floor and repeats it.  Several ranks: the worst
        // case (a repeated step on ONE rank would leave the ranks out of step).
        const unsigned long long nt_n = (unsigned long long)((h->n + TILE - 1) / TILE + MAXK + 1) + (unsigned long long)US_MAXP;
        const unsigned long long worst = (nt_n * (nt_n + 1ull) / 2ull) * (unsigned long long)(TILE / seg_unit) + 64ull;
        (void)pairs_max;
        const unsigned long long SLIST_SOFT_CAP = h->slist_soft_cap;   // (GRAAL_SLIST_SOFT_CAP, read when the handle is created)
        h->slist_worst = worst;
        const unsigned long long need = (world == 1 && publish) ? std::min(worst, std::max(SLIST_SOFT_CAP, h->slist_floor)) : worst;
        if (need > h->slist_cap) {
            CK(hipDeviceSynchronize());   // (rare: the first tiled step of a layout size, or a list that has just overflowed)
            if (h->d_slist) CK(hipFree(h->d_slist));
            h->d_slist = nullptr;
            h->slist_cap = 0;
            const unsigned long long cap = std::max<unsigned long long>(need, 64ull);
            CK(hipMalloc(&h->d_slist, cap * sizeof(unsigned long long)));
            h->slist_cap = cap;
        }
        h->d_slist_n = (unsigned long long*)(h->d_scalars + SLIST_N);
        sx.list_cap = h->slist_cap;
        sx.seg = 0;
        fa.norm_u = h->uniform_accu > 0 ? (float)(h->uniform_accu * h->uniform_accu) / h->nfpb : -1.0f;
        static const int s2_skip = getenv("GRAAL_S2_SKIP") ? atoi(getenv("GRAAL_S2_SKIP")) : 0;   // (diagnostics: wrong sums.  1 = no units, 2 = no queued contacts)
        fa.skip = s2_skip;
        const int no_window = (sx.quirk && h->n_ubins > 0) ? 1 : 0;
        S2Args s2;
        s2.uset = h->d_uset; s2.cls = h->d_cls; s2.cls_n = h->d_cls_n; s2.seg_unit = seg_unit; s2.rep_max = rep_max; s2.target = target;
        static const int draw_env = getenv("GRAAL_STRICT_DRAW") ? atoi(getenv("GRAAL_STRICT_DRAW")) : 8;
        s2.draw_min = draw_env > 0 ? draw_env : 8;
        s2.next = reinterpret_cast<unsigned long long*>(h->d_cls_n + US_MAXPAIRS + (US_MAXPAIRS & 1));
        // k_strict2 behind k_gprep WITHOUT an event (S2Args::gp): a kernel behind an event of another stream starts ~11 us after the event
        // completes (tools/stamps_s2.py, C2 stand-in: k_gprep done 23 us, k_strict2 started 37; without any ordering -- GRAAL_DEBUG runs -- 26, as
        // soon as the host has submitted it).  k_gprep's results go out as device-scope stores, its last block stores the step's number, and
        // k_strict2's blocks wait for that word.  Only a grid that leaves room for k_gprep's blocks on every CU may wait for them in the kernel
        // (512 blocks: two per CU; cf. fin_blocks_no_wait), only one rank (a repeated step must not leave the ranks out of step), only while
        // the engine's streams are known to run side by side (spin_ok).  (768 blocks -- three per CU, still room -- with the wait instead of 1,024
        // behind the event: C3 / C4 stand-ins 214-221 / 238-243 us per step against 200-203 / 222-225.)
        // (round 5 tried the word with the 1,024-block grids too: the wait ran out in every run -- C3 and C4 stand-ins -- and the engine went back to
        // events, as the argument above predicts)
        const bool gwait = h->gwait_env && publish && world == 1 && h->spin_ok && blocks <= 512;
        s2.gp = gwait ? s2.next + 32 : nullptr;   // (ticket: 256 bytes behind the draw counter; the completion word 256 bytes behind the ticket: GP_DONE)
        s2.gp_seq = (unsigned long long)h->seq;
        s2.gp_wait_ticks = h->gp_wait_ticks;
        s2.gp_acquire = h->gp_acquire;
        if (gwait) { h->spin_used = true; h->rc_gwait += 1; } else h->rc_gevent += 1;
        const int cull_blocks = (int)std::min<unsigned long long>(1024ull, std::max<unsigned long long>(1ull, nt));
        // Where k_gprep goes.  A long scan (millions of contacts): on the auxiliary stream behind k_tm, under the scan; k_strict2 waits for both
        // through an event.  A short one (a map of a few thousand bins: the scan is over before k_tm's tables are): on the MAIN stream, behind
        // the scan and an event of k_tm that has long completed when the stream gets there -- k_strict2 then follows k_gprep in stream order,
        // without the ~10 us a kernel waits behind an event that completes right in front of it (tools/stamps_s2.py, C2 stand-in:
        // k_gprep done 27 us, k_strict2 started 38.7 us)
        static const int inorder_env = getenv("GRAAL_STRICT_INORDER") ? atoi(getenv("GRAAL_STRICT_INORDER")) : -1;
        const bool inorder = inorder_env > 0;   // (measured: no gain -- the event in front of k_gprep costs what the one in front of k_strict2 did; kept as a switch)
        if (inorder) {
            CK(hipEventRecord(h->ev_tm, h->aux));
            CK(hipStreamWaitEvent(st, h->ev_tm, 0));
        }
        k_gprep<<<GPREP_CLS_BLOCKS + cull_blocks, 256, 0, inorder ? st : h->aux>>>(h->tabs, h->pstart, fA, K, rank, world, sx.reach_bp, no_window, sx.quirk,
                                                                     seg_unit, TILE, h->d_slist, h->d_slist_n, h->slist_cap,
                                                                     (unsigned long long*)(h->d_scalars + 10), s2);
        CK(hipGetLastError());
        // (k_strict2 on the auxiliary stream right behind k_gprep -- stream order instead of the event, next to the scan, its waves waiting for the
        // scan's completion counters before the queued contacts -- was tried: it starts 9 us earlier (a kernel behind an event of another stream
        // starts ~10 us late whether the event completes right in front of it or has long completed), but full runs gained nothing (C2 stand-in,
        // 100 cycles: 130 us per step without, 134-146 with) and on the C4 stand-in the waiting waves once kept the scan off the CUs until their
        // bound ran out.  Not kept.)
        if (!inorder && !gwait) {
            CK(hipEventRecord(h->ev_tm, h->aux));      // (behind k_tm and k_gprep: tables, classes and unit list complete -- nobody spins for them)
            CK(hipStreamWaitEvent(st, h->ev_tm, 0));
        }
        hipStream_t ks = st;
        const size_t sslot = (size_t)(h->sring_calls % (long long)(h->sring.size() / 2));
        if (h->ev_this_call) CK(hipEventRecord(h->sring[2 * sslot], ks));   // (behind the wait: the pair spans the kernel, not the scan in front of it)
        if (h->single_sub) k_strict2<false><<<blocks, 256, 0, ks>>>(fa, sx, s2, K, h->d_slist, h->d_slist_n, d_q_out, publish ? h->res_dev : nullptr, h->seq);
        else k_strict2<true><<<blocks, 256, 0, ks>>>(fa, sx, s2, K, h->d_slist, h->d_slist_n, d_q_out, publish ? h->res_dev : nullptr, h->seq);
        CK(hipGetLastError());
        if (h->ev_this_call) { CK(hipEventRecord(h->sring[2 * sslot + 1], ks)); h->sring_calls += 1; }
        return GRAAL_OK;
    }
    // the unit list holds at most K * nt (nt + 1) / 2 entries, nt = tiles of the two longest contigs (grow-only)
    const unsigned long long nt = 2ull * (unsigned long long)((std::max(std::max(h->max_lcont, h->lcont_bound), 1) + 63) / 64);
    static const int blocks_env = getenv("GRAAL_STRICT_BLOCKS") ? atoi(getenv("GRAAL_STRICT_BLOCKS")) : 0;
    const int lc = std::max(h->max_lcont, h->lcont_bound);
    const int blocks = blocks_env > 0 ? blocks_env : (lc > 0 && lc <= 64 ? 32 : (lc <= 1024 ? 512 : 1024));
    // fragments y per unit: a 64 x 64 tile pair under half a dozen classes with nine slot pairs each is a millisecond of dependent
    // float32 powf / expf in ONE wave -- with contigs of a few hundred bins a step has a few dozen tile pairs, and the C2 stand-in
    // ran at 3.2 ms per step.  The y tile is cut into segments so that the grid has ~6 units per wave (bounded below: the x tile
    // is loaded once per unit) -- by k_strict_cull, which knows the sets of THIS step; GRAAL_STRICT_SEG fixes it.
    static const int seg_env = getenv("GRAAL_STRICT_SEG") ? atoi(getenv("GRAAL_STRICT_SEG")) : 0;
    const int seg_fixed = (seg_env == 1 || seg_env == 2 || seg_env == 4 || seg_env == 8 || seg_env == 16 || seg_env == 32 || seg_env == 64) ? seg_env : 0;
    const int seg_min = h->single_sub ? 4 : 1;
    const unsigned long long target = 6ull * 4ull * (unsigned long long)blocks;
    sx.seg = 0;
    // units: at most one per tile pair when the segments stay whole tiles, below 2 x target once they are cut (the cut stops at the
    // first size that reaches the target), or every tile pair cut to the fixed / smallest size
    const unsigned long long pairs_max = (unsigned long long)K * nt * (nt + 1ull) / 2ull;
    const unsigned long long need = (seg_fixed ? pairs_max * (unsigned long long)(64 / seg_fixed)
                                               : std::max(pairs_max, std::min(pairs_max * (unsigned long long)(64 / seg_min), 2ull * target))) + 64ull;
    if (need > h->slist_cap) {
        CK(hipStreamSynchronize(st));
        CK(hipStreamSynchronize(h->aux));
        if (h->d_slist) CK(hipFree(h->d_slist));
        h->d_slist = nullptr;
        const unsigned long long cap = std::max<unsigned long long>(need + need / 2ull, 1ull << 16);
        CK(hipMalloc(&h->d_slist, cap * sizeof(unsigned long long)));
        h->slist_cap = cap;
    }
    // (the list's length lives in the scalars block -- zeroed, synchronously, when the context was created: a hipMemset issued
    // here could still be in flight when k_strict_cull counts into it)
    h->d_slist_n = (unsigned long long*)(h->d_scalars + SLIST_N);
    sx.list_cap = h->slist_cap;
    const int no_window = (sx.quirk && h->n_ubins > 0) ? 1 : 0;
    // rows of candidate units = tiles of the affected sets: a block per row up to the chip's width
    const int cull_blocks = (int)std::min<unsigned long long>(1024ull, std::max<unsigned long long>(1ull, (unsigned long long)K * nt));
    k_strict_cull<<<cull_blocks, 256, 0, h->aux>>>(h->tabs, h->geo, h->link, h->cbase, h->perm, fA, K, rank, world, sx.reach_bp, no_window, seg_fixed, seg_min, target,
                                                   h->d_slist, h->d_slist_n, h->slist_cap, (unsigned long long*)(h->d_scalars + 10));
    CK(hipGetLastError());
    CK(hipEventRecord(h->ev_tm, h->aux));      // (behind k_tm and the cull: tables and unit list complete -- nobody spins for them)
    CK(hipStreamWaitEvent(st, h->ev_tm, 0));
    if (h->single_sub) k_strict<false><<<blocks, 256, strict_dyn_lds(K), st>>>(fa, sx, fA, K, h->d_slist, h->d_slist_n, d_q_out, publish ? h->res_dev : nullptr, h->seq);
    else k_strict<true><<<blocks, 256, strict_dyn_lds(K), st>>>(fa, sx, fA, K, h->d_slist, h->d_slist_n, d_q_out, publish ? h->res_dev : nullptr, h->seq);
    CK(hipGetLastError());
    return GRAAL_OK;
}

// reference arithmetic, small affected sets (k_strict_flat): behind the scan on its stream, no event, no unit list
int launch_flat(Ctx* h, int fA, const Neigh* nbp /* nullptr: the neighbours of the evaluation in flight */, int K, int rank, int world, long long* d_q_out, bool publish, hipStream_t st)
{
    FinArgs fa;
    fa.tm_done = h->tm_done; fa.step_hdr = h->step_hdr; fa.counters = (unsigned long long*)(h->d_scalars + 10); fa.queue = h->queue;
    fa.cnt = h->cnt; fa.multi = h->single_sub ? 0 : 1;
    fa.tabs = h->tabs; fa.geo = h->geo; fa.stat = h->stat_frag; fa.acc = h->d_acc; fa.sync = h->d_sync;
    fa.ln_tab = nullptr; fa.lut_n = 0; fa.skip = 0; fa.seg = 0; fa.norm_u = -1.0f; fa.upw = 4; fa.wq = nullptr;
    StrictArgs sx;
    sx.perm = h->perm; sx.cbase = h->cbase; sx.lcontbp = h->soa[h->cur].p[F_LCONTBP]; sx.link = h->link; sx.nfpb = h->nfpb; sx.par = h->par;
    sx.quirk = (h->mode & GRAAL_MODE_REF_TRANS_ACCU) ? 1 : 0;
    sx.reach_bp = reach_bp(h);
    sx.list_cap = 0;
    sx.seg = 64;
    static const int blocks_env = getenv("GRAAL_FLAT_BLOCKS") ? atoi(getenv("GRAAL_FLAT_BLOCKS")) : 0;
    const int blocks = blocks_env > 0 ? std::min(blocks_env, FLAT_BLOCKS) : FLAT_BLOCKS;
    Neigh nb;
    for (int k = 0; k < MAXK; k++) nb.fB[k] = nbp ? nbp->fB[k] : h->last_fB[k];
    h->rc_flat += 1;
    if (h->single_sub) k_strict_flat<false><<<blocks, 256, 0, st>>>(fa, sx, fA, nb, K, rank, world, d_q_out, publish ? h->res_dev : nullptr, h->seq);
    else k_strict_flat<true><<<blocks, 256, 0, st>>>(fa, sx, fA, nb, K, rank, world, d_q_out, publish ? h->res_dev : nullptr, h->seq);
    CK(hipGetLastError());
    return GRAAL_OK;
}

// may this evaluation use k_strict_flat?  With several ranks (an exchange attached) too: the flat and the tiled kernels deal the fragment
// pairs to the ranks differently, so every rank must pick the same one -- and it does: the choice between them is the sets' geometry
// (k_tm says at once that a set is beyond its own pricing, k_strict_flat that the pairs are too many), the same on every rank; what a
// rank's own scan found -- its shard's queued contacts -- only decides WHO prices those contacts, which are that rank's alone
bool flat_allowed(const Ctx* h, int world)
{
    static const bool no_flat = getenv("GRAAL_NO_FLAT") != nullptr;
    return !no_flat && (h->mode & GRAAL_MODE_STRICT) && !strict_dense_cfg() && (world == 1 || h->x_host != nullptr) && h->publish;
}

} // namespace

// statistics of the layout (k_stats ran earlier on the stream): publish + wait; res[0..15] on return
static int wait_stats(Ctx* h, long long res[16])
{
    volatile long long* p = h->h_stats;
    bool seen = false;
    for (long long spin = 0; spin < 400000000ll; spin++) {
        if (p[0] == h->stats_seq) { seen = true; break; }
        if ((spin & 0xfffff) == 0xfffff && hipStreamQuery(h->stream) != hipErrorNotReady) { seen = (p[0] == h->stats_seq); break; }
        __builtin_ia32_pause();
    }
    if (!seen) {
        CK(hipStreamSynchronize(h->stream));
        if (p[0] != h->stats_seq) return fail(h, GRAAL_E_HIP, "the layout statistics were not published");
    }
    __sync_synchronize();
    for (int i = 0; i < 16; i++) res[i] = p[1 + i];
    return GRAAL_OK;
}

// the last commit's own-pixel correction (k_own_corr, on the full evaluation's stream): published long before anybody asks
static int wait_own(Ctx* h, long long* q, long long* bad)
{
    if (!h->own_pending) { if (q) *q = 0; if (bad) *bad = 1; return GRAAL_OK; }   // (nothing out: unknown)
    volatile long long* p = h->h_own;
    bool seen = false;
    for (long long spin = 0; spin < 400000000ll; spin++) {
        if (p[0] == h->own_seq) { seen = true; break; }
        if ((spin & 0xfffff) == 0xfffff && hipStreamQuery(h->fstream) != hipErrorNotReady) { seen = (p[0] == h->own_seq); break; }
        __builtin_ia32_pause();
    }
    if (!seen) {
        CK(hipStreamSynchronize(h->fstream));
        if (p[0] != h->own_seq) return fail(h, GRAAL_E_HIP, "the commit's own-pixel correction was not published");
    }
    __sync_synchronize();
    if (q) *q = p[1];
    if (bad) *bad = p[2];
    h->own_pending = false;
    return GRAAL_OK;
}

static int fetch_stats(Ctx* h, long long res[16], bool reset_stale)
{
    h->stats_seq += 1;
    k_stats_fin<<<1, 64, 0, h->stream>>>(h->d_scalars, h->h_stats, h->stats_seq, reset_stale ? 1 : 0);
    CK(hipGetLastError());
    return wait_stats(h, res);
}

// end of a full evaluation: the two sums, the not-finite flag and the repeats' sum go to pinned host memory followed by the
// evaluation's sequence number (the host spins on it: no device->host copies, no stream synchronise), and the accumulators
// are zero again for the next one (they are zero at rest: no memsets in front of an evaluation either)
__global__ void k_full_pub(long long* __restrict__ sc, volatile long long* host, long long seq)
{
    const int t = threadIdx.x;
    const int src = t == 0 ? 8 : (t == 1 ? 9 : (t == 2 ? FULL_BAD : 17));
    if (t < 4) { host[1 + t] = sc[src]; sc[src] = 0; }
    __threadfence_system();
    __syncthreads();
    if (t == 0) { host[0] = seq; __threadfence_system(); }
}

struct graal_ctx : Ctx {};

// ---- RCCL, resolved at run time: the library carries no link-time dependency on it (the CPU build and a single-rank run never load it)
struct NcclId { char b[128]; };   // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128), passed by value like the original
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
static Rccl* rccl_load(std::string* err)
{
    static Rccl R;
    if (R.lib) return &R;
    const char* names[] = {getenv("GRAAL_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* lib = nullptr;
    for (const char* n : names) if (n && !lib) lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!lib) { const char* e = dlerror(); if (err) *err = std::string("RCCL not found: ") + (e ? e : "dlopen failed"); return nullptr; }   // (dlerror() clears its state: once)
    R.GetUniqueId = (int (*)(NcclId*))dlsym(lib, "ncclGetUniqueId");
    R.CommInitRank = (int (*)(void**, int, NcclId, int))dlsym(lib, "ncclCommInitRank");
    R.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(lib, "ncclAllReduce");
    R.CommDestroy = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
    R.GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
    if (!R.GetUniqueId || !R.CommInitRank || !R.AllReduce || !R.CommDestroy) { if (err) *err = "RCCL: a symbol is missing"; dlclose(lib); return nullptr; }
    R.lib = lib;
    return &R;
}
constexpr int NCCL_INT64 = 4, NCCL_SUM = 0;   // rccl.h: ncclInt64, ncclSum


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
    {
        // k_tm's last block may WAIT for k_scan (see k_tm), so the two must never share a hardware queue (where they would
        // run one after the other): the runtime keeps separate queue pools per priority, and a high-priority queue also
        // gets the small latency-bound kernel dispatched ahead of the streaming one
        int least = 0, greatest = 0;
        CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        CK(hipStreamCreateWithPriority(&h->aux, hipStreamNonBlocking, greatest));
    }
    CK(hipStreamCreateWithFlags(&h->fstream, hipStreamNonBlocking));
    CK(hipEventCreateWithFlags(&h->ev_full, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&h->ev_fin, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&h->ev_relabel, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&h->ev_tm, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&h->ev_par, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&h->ev_par2, hipEventDisableTiming));
    for (auto& ev : h->ev) CK(hipEventCreate(&ev));
    h->ring.resize(2 * 1024, nullptr);
    for (auto& ev : h->ring) CK(hipEventCreate(&ev));
    h->sring.resize(2 * 256, nullptr);
    for (auto& ev : h->sring) CK(hipEventCreate(&ev));
    CK(hipMalloc(&h->d_scalars, 32 * sizeof(long long)));
    CK(hipMemset(h->d_scalars, 0, 32 * sizeof(long long)));
    CK(hipMalloc(&h->d_acc, 2 * MAXK * N_OPS * sizeof(long long)));   // (fine sums, coarse sums: to_coarse)
    CK(hipMemset(h->d_acc, 0, 2 * MAXK * N_OPS * sizeof(long long)));
    CK(hipMalloc(&h->tm_done, MAXK * sizeof(long long)));
    CK(hipMemset(h->tm_done, 0, MAXK * sizeof(long long)));
    CK(hipMalloc(&h->d_wq, 2 * N_WQ * WQ_STRIDE * sizeof(unsigned long long)));
    CK(hipMemset(h->d_wq, 0, 2 * N_WQ * WQ_STRIDE * sizeof(unsigned long long)));
    CK(hipMalloc(&h->d_done, N_DONE * DONE_STRIDE * sizeof(unsigned long long)));
    CK(hipMemset(h->d_done, 0, N_DONE * DONE_STRIDE * sizeof(unsigned long long)));
    CK(hipMalloc(&h->d_sync, 32 * sizeof(unsigned long long)));
    CK(hipMemset(h->d_sync, 0, 32 * sizeof(unsigned long long)));
    CK(hipMalloc(&h->d_flags, (size_t)MAX_SCAN_BLOCKS * FLAG_STRIDE * sizeof(unsigned)));
    CK(hipMemset(h->d_flags, 0, (size_t)MAX_SCAN_BLOCKS * FLAG_STRIDE * sizeof(unsigned)));
    if (getenv("GRAAL_EVENT_EVERY")) h->event_every = std::max(1, atoi(getenv("GRAAL_EVENT_EVERY")));
    CK(hipMalloc(&h->d_qout, 3 * MAXK * N_OPS * sizeof(long long)));   // (fine sums, coarse sums, not-finite flags: hand_out)
    CK(hipMalloc(&h->tabs, MAXK * sizeof(NbTables)));
    CK(hipMalloc(&h->step_hdr, 2 * MAXK * sizeof(int)));
    CK(hipMemset(h->step_hdr, 0, 2 * MAXK * sizeof(int)));
    CK(hipMalloc(&h->d_args, 2 * sizeof(DevArgs)));
    CK(hipMalloc(&h->d_part, 1024 * N_STAT * sizeof(long long)));
    CK(hipMemset(h->d_part, 0, 1024 * N_STAT * sizeof(long long)));
    CK(hipMalloc(&h->d_chg, 2 * sizeof(Changed)));
    CK(hipMemset(h->d_chg, 0, 2 * sizeof(Changed)));
    CK(hipHostMalloc((void**)&h->h_res, X_SLOT_WORDS * sizeof(long long), hipHostMallocDefault));
    memset(h->h_res, 0, X_SLOT_WORDS * sizeof(long long));
    h->res_host = h->res_dev = h->h_res;
    CK(hipHostMalloc((void**)&h->h_dist, 2 * sizeof(long long), hipHostMallocDefault));
    memset(h->h_dist, 0, 2 * sizeof(long long));
    CK(hipMalloc(&h->d_dist, 2 * sizeof(unsigned long long)));
    CK(hipMemset(h->d_dist, 0, 2 * sizeof(unsigned long long)));
    CK(hipHostMalloc((void**)&h->h_full, 8 * sizeof(long long), hipHostMallocDefault));
    memset(h->h_full, 0, 8 * sizeof(long long));
    CK(hipHostMalloc((void**)&h->h_stats, 17 * sizeof(long long), hipHostMallocDefault));
    memset(h->h_stats, 0, 17 * sizeof(long long));
    CK(hipHostMalloc((void**)&h->h_guard, 8 * sizeof(long long), hipHostMallocDefault));
    memset(h->h_guard, 0, 8 * sizeof(long long));
    CK(hipHostMalloc((void**)&h->h_own, 4 * sizeof(long long), hipHostMallocDefault));
    memset(h->h_own, 0, 4 * sizeof(long long));
    CK(hipMalloc(&h->d_own_acc, 4 * sizeof(long long)));
    CK(hipMemset(h->d_own_acc, 0, 4 * sizeof(long long)));
    {
        const long long init[8] = {0, 0, 0, 0, 0, 0x7fffffff, 0, -1}; // k_stats accumulators (re-armed by k_stats_fin)
        CK(hipMemcpy(h->d_scalars, init, sizeof init, hipMemcpyHostToDevice));
    }
    return GRAAL_OK;
}

void graal_destroy(graal_ctx* h)
{
    if (!h) return;
    if (h->eval_timing && h->et_n[0]) {
        const double n = (double)h->et_n[0], m = (double)std::max<long long>(h->et_n[1], 1);
        fprintf(stderr, "graal eval timing: %lld synchronous evaluations, launches %.1f us each; %lld finished by k_tm (%.1f us from launch to result); "
                        "%lld needed k_fin / k_strict: %.1f us until k_tm said so, %.1f us to launch them, %.1f us until the result (%.1f us in all)\n",
                h->et_n[0], h->et[0] / n, h->et_n[0] - h->et_n[1], h->et[4] / (double)std::max<long long>(h->et_n[0] - h->et_n[1], 1),
                h->et_n[1], h->et[1] / m, h->et[2] / m, h->et[3] / m, h->et[5] / m);
    }
    if (h->stream) {
        (void)hipSetDevice(h->device);
        (void)hipStreamSynchronize(h->stream);
        if (h->aux) (void)hipStreamSynchronize(h->aux);
        if (h->fstream) (void)hipStreamSynchronize(h->fstream);
        // ... and whatever this handle's kernels were launched on besides its own streams (graal_eval_candidates_q takes the CALLER'S stream: the
        // torch path of exchange="rccl"): nothing of this process may still touch the buffers, pinned words and the registered segment freed below
        (void)hipDeviceSynchronize();
        void* ptrs[] = {h->soa_mem[0], h->soa_mem[1], h->geo, h->stat, h->sub_rec, h->sub_rec8, h->sub_lab16, h->d_ubins, h->d_ln_tab, h->sub2bin, h->row, h->col, h->cnt, h->queue, h->d_sub_ids, h->d_acc, h->tm_done, h->d_sync, h->d_done, h->d_wq, h->d_flags, h->d_slist, h->d_uset, h->d_cls, h->d_cls_n, h->stat_frag, h->d_dup_bins, h->d_dup_index,
                        h->d_dispatcher, h->d_collector, h->d_sub_ids_all, h->d_rep_obs,
                        h->keys, h->keys_sorted, h->o2n, h->len_of2[0], h->len_of2[1], h->contig_off2[0], h->contig_off2[1], h->perm, h->pstart, h->cbase, h->link, h->mates, h->cub_tmp, h->tabs, h->step_hdr, h->d_args, h->d_chg, h->d_part,
                        h->d_scalars, h->d_qout, h->d_dref, h->d_dist};
        for (void* p : ptrs) if (p) (void)hipFree(p);
        if (h->nccl_comm) { Rccl* R = rccl_load(nullptr); if (R) (void)R->CommDestroy(h->nccl_comm); h->nccl_comm = nullptr; }
        if (h->d_own_obs) (void)hipFree(h->d_own_obs);
        if (h->x_host) (void)hipHostUnregister(h->x_host);
        if (h->h_res) (void)hipHostFree(h->h_res);
        if (h->h_stats) (void)hipHostFree(h->h_stats);
        if (h->h_own) (void)hipHostFree(h->h_own);
        if (h->h_guard) (void)hipHostFree(h->h_guard);
        if (h->d_own_acc) (void)hipFree(h->d_own_acc);
        if (h->h_full) (void)hipHostFree(h->h_full);
        if (h->h_dist) (void)hipHostFree(h->h_dist);
        for (auto& ev : h->ev) if (ev) (void)hipEventDestroy(ev);
        for (auto& ev : h->ring) if (ev) (void)hipEventDestroy(ev);
        for (auto& ev : h->sring) if (ev) (void)hipEventDestroy(ev);
        if (h->ev_fin) (void)hipEventDestroy(h->ev_fin);
        if (h->ev_relabel) (void)hipEventDestroy(h->ev_relabel);
        if (h->ev_tm) (void)hipEventDestroy(h->ev_tm);
        if (h->ev_par) (void)hipEventDestroy(h->ev_par);
        if (h->ev_par2) (void)hipEventDestroy(h->ev_par2);
        if (h->aux) (void)hipStreamDestroy(h->aux);
        if (h->fstream) (void)hipStreamDestroy(h->fstream);
        if (h->ev_full) (void)hipEventDestroy(h->ev_full);
        (void)hipStreamDestroy(h->stream);
    }
    hs_free(h->hs);
    delete h;
}

const char* graal_last_error(const graal_ctx* h) { return h ? h->err.c_str() : "null handle"; }

// parameters only: the two device-resident argument blocks get the new Par (and the window derived from it) from a one-block kernel in
// stream order -- no blocking copy, no synchronize (a nuisance-parameter step sends parameters twice: cuda_lib_gl.py:2073-2107)
__global__ void k_set_par(DevArgs* a, Par par, int reach)
{
    if (threadIdx.x < 2) { a[threadIdx.x].par = par; a[threadIdx.x].reach_bp = reach; }
}

int graal_set_params(graal_ctx* h, const float* p)
{
    if (!h || !p) return GRAAL_E_ARG;
    Par np_;   // (validated before it is compared or committed: h->par never holds rejected values)
    memcpy(&np_, p, sizeof(Par));
    if (!(np_.v_inter > 0.0f)) return fail(h, GRAAL_E_ARG, "v_inter must be > 0 (the sparse form prices every pixel at >= v_inter)");
    if (!(np_.d_max > 0.0f) || !(np_.d_max < 2.0e6f)) return fail(h, GRAAL_E_ARG, "d_max out of range");
    if (h->have_par && memcmp(&h->par, &np_, sizeof(Par)) == 0) return GRAAL_OK;   // (re-sending the parameters in force)
    const bool first = !h->have_par;
    const bool ln_same = !first && h->par.v_inter == np_.v_inter;      // (k_ln_tab: ln of v_inter x the products of RF counts / nfpb)
    h->par = np_;
    h->have_par = true;
    compute_t_all(h, h->t_hist.empty());
    CK(hipSetDevice(h->device));
    if (first || !h->args_synced) {   // (the blocks hold nothing yet, or an upload is pending: the whole blocks, synchronously)
        if (h->stream) (void)hipStreamSynchronize(h->stream);
        return sync_args(h);
    }
    // In stream order on the engine's stream, fenced against the two other streams both ways: no kernel of theirs still reads the old
    // parameters when they are overwritten (their last work is waited for), none launched from now on starts before the new ones are in place
    CK(hipEventRecord(h->ev_par, h->aux));
    CK(hipStreamWaitEvent(h->stream, h->ev_par, 0));
    CK(hipEventRecord(h->ev_par2, h->fstream));
    CK(hipStreamWaitEvent(h->stream, h->ev_par2, 0));
    k_set_par<<<1, 64, 0, h->stream>>>(h->d_args, h->par, reach_bp(h));
    CK(hipGetLastError());
    if (!ln_same && h->have_sub && h->ln_lut_n > 0 && h->d_ln_tab) {
        k_ln_tab<<<blocks_for(h->ln_lut_n, 256), 256, 0, h->stream>>>(h->d_ln_tab, h->ln_lut_n, h->nfpb, h->par);
        CK(hipGetLastError());
    }
    CK(hipEventRecord(h->ev_par, h->stream));
    CK(hipStreamWaitEvent(h->aux, h->ev_par, 0));
    CK(hipStreamWaitEvent(h->fstream, h->ev_par, 0));
    return GRAAL_OK;
}

int graal_upload_subfrags(graal_ctx* h, const int32_t* sub_id, const float* sub_len, const int32_t* sub_accu, int32_t n_bins,
                          int32_t n_sub_total, float nfpb)
{
    if (h && h->stream) (void)hipStreamSynchronize(h->stream); // (graal_begin_step may have left its relabel kernels running)
    if (!h || !sub_id || !sub_len || !sub_accu || n_bins <= 0 || n_sub_total < n_bins || !(nfpb > 0)) return GRAAL_E_ARG;
    CK(hipSetDevice(h->device));
    std::vector<Stat> st(n_bins);
    std::vector<int> s2b(n_sub_total, -1);
    h->h_accu.assign(sub_accu, sub_accu + 3 * (size_t)n_bins);
    {   // products of two RF counts that can occur: size of k_full_nnz's ln(trans) table
        long long amax = 0;
        for (int v : h->h_accu) amax = std::max<long long>(amax, v);
        h->ln_lut_n = (int)std::min<long long>(LN_TRANS_LUT, amax * amax + 1);
    }
    h->h_nsub.resize(n_bins);
    bool single = true;
    bool uniform = true;   // every bin's sub-fragments carry one RF count (stat_uniform)
    for (int b = 0; b < n_bins; b++) {
        const int ns = sub_id[4 * b + 3];
        if (ns < 1 || ns > 3) return fail(h, GRAAL_E_ARG, "n_sub must be 1..3 (sub-sampling factor 3 is baked in, simulation_loader.py:682-698)");
        single &= (ns == 1);
        h->h_nsub[b] = ns;
        Stat s{};
        s.n = ns;
        s.l0 = sub_len[3 * b]; s.l1 = ns > 1 ? sub_len[3 * b + 1] : 0.0f; s.l2 = ns > 2 ? sub_len[3 * b + 2] : 0.0f;
        s.a0 = sub_accu[3 * b]; s.a1 = ns > 1 ? sub_accu[3 * b + 1] : 0; s.a2 = ns > 2 ? sub_accu[3 * b + 2] : 0;
        for (int k = 0; k < ns; k++) {
            const int sid = sub_id[4 * b + k];
            if (sid < 0 || sid >= n_sub_total || s2b[sid] != -1) return fail(h, GRAAL_E_ARG, "sub_id must map sub-fragments to bins one-to-one");
            if (sub_accu[3 * b + k] <= 0 || sub_accu[3 * b + k] > 30000) return fail(h, GRAAL_E_ARG, "sub_accu out of range");
            s2b[sid] = b * 4 + k;
        }
        st[b] = s;
        uniform &= (ns < 2 || s.a1 == s.a0) && (ns < 3 || s.a2 == s.a0);
    }
    h->all_uniform = uniform;
    for (int v : s2b) if (v < 0) return fail(h, GRAAL_E_ARG, "every sub-fragment must belong to a bin");
    h->h_stat = st;
    h->h_sub2bin = s2b;
    h->h_sub_id.assign(sub_id, sub_id + 4 * (size_t)n_bins);
    h->has_rep = false; h->n_dup = 0; h->h_dup_index.assign((size_t)n_bins, -1); // (graal_upload_repeats comes after)
    // single_sub additionally needs sub id == bin id so that the scan can skip the sub2bin gather
    for (int b = 0; single && b < n_bins; b++) single = (sub_id[4 * b] == b);
    if (h->stat) { (void)hipFree(h->stat); (void)hipFree(h->sub2bin); (void)hipFree(h->d_sub_ids); (void)hipFree(h->sub_rec); (void)hipFree(h->sub_rec8); (void)hipFree(h->sub_lab16); h->d_sub_ids = nullptr; h->sub_rec = nullptr; h->sub_rec8 = nullptr; h->sub_lab16 = nullptr; }
    if (!single) {
        CK(hipMalloc(&h->d_sub_ids, sizeof(int) * 4 * (size_t)n_bins));
        CK(hipMemcpy(h->d_sub_ids, sub_id, sizeof(int) * 4 * (size_t)n_bins, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&h->stat, sizeof(Stat) * (size_t)n_bins));
    CK(hipMalloc(&h->sub2bin, sizeof(int) * (size_t)n_sub_total));
    CK(hipMalloc(&h->sub_rec, sizeof(SubRec) * (size_t)n_sub_total));
    CK(hipMemset(h->sub_rec, 0, sizeof(SubRec) * (size_t)n_sub_total));
    CK(hipMalloc(&h->sub_rec8, sizeof(SubRec8) * (size_t)n_sub_total));
    CK(hipMemset(h->sub_rec8, 0, sizeof(SubRec8) * (size_t)n_sub_total));
    CK(hipMalloc(&h->sub_lab16, 2 * (size_t)(((size_t)n_sub_total + 7) & ~(size_t)7)));
    CK(hipMemset(h->sub_lab16, 0, 2 * (size_t)(((size_t)n_sub_total + 7) & ~(size_t)7)));
    {
        h->uniform_accu = sub_accu[0];
        for (int b = 0; b < n_bins && h->uniform_accu; b++)
            for (int k = 0; k < sub_id[4 * b + 3]; k++) if (sub_accu[3 * b + k] != h->uniform_accu) { h->uniform_accu = 0; break; }
        if (h->uniform_accu > 30000) h->uniform_accu = 0;
    }
    {   // bins with non-uniform RF counts: the only ones the reference's trans-branch indexing prices differently
        std::vector<int> ub;
        for (int b = 0; b < n_bins; b++) {
            const int ns = sub_id[4 * b + 3];
            bool uni = true;
            for (int k = 1; k < ns; k++) uni = uni && sub_accu[3 * b + k] == sub_accu[3 * b];
            if (!uni) ub.push_back(b);
        }
        if (h->d_ubins) { (void)hipFree(h->d_ubins); h->d_ubins = nullptr; }
        h->n_ubins = (int)ub.size();
        if (h->n_ubins) {
            CK(hipMalloc(&h->d_ubins, sizeof(int) * ub.size()));
            CK(hipMemcpy(h->d_ubins, ub.data(), sizeof(int) * ub.size(), hipMemcpyHostToDevice));
        }
    }
    CK(hipMemcpy(h->stat, st.data(), sizeof(Stat) * (size_t)n_bins, hipMemcpyHostToDevice));
    CK(hipMemcpy(h->sub2bin, s2b.data(), sizeof(int) * (size_t)n_sub_total, hipMemcpyHostToDevice));
    h->n_bins = n_bins; h->n_sub_total = n_sub_total; h->nfpb = nfpb; h->single_sub = single; h->have_sub = true;
    compute_t_all(h);
    return sync_args(h);
}

// bin of a sub-fragment id (host copy of the sub-fragment table)
static int s2b_host(const Ctx* h, int sid)
{
    if (h->h_sub2bin.empty()) return 0;
    return h->h_sub2bin[(size_t)sid] >> 2;
}

static RepArgs rep_args(const Ctx* h)
{
    RepArgs R;
    R.n_dup = h->n_dup; R.n_bins = h->n_bins; R.n_sub_total = h->n_sub_total;
    R.dup_bins = h->d_dup_bins; R.dup_index = h->d_dup_index; R.dispatcher = h->d_dispatcher; R.collector = h->d_collector;
    R.obs = h->d_rep_obs; R.sub_ids = h->d_sub_ids_all; R.stat_bin = h->stat; R.geo = h->geo;
    R.lcontbp = h->soa[h->cur].p[F_LCONTBP]; R.nfpb = h->nfpb; R.par = h->par;
    R.quirk = (h->mode & GRAAL_MODE_REF_TRANS_ACCU) ? 1 : 0;
    return R;
}

int graal_upload_repeats(graal_ctx* h, const int32_t* dup_bins, int32_t n_dup, const int32_t* dispatcher, const int32_t* collector,
                         int32_t n_collector, const float* obs_rows)
{
    if (h && h->stream) (void)hipStreamSynchronize(h->stream); // (graal_begin_step may have left its relabel kernels running)
    if (!h || n_dup < 0 || (n_dup > 0 && (!dup_bins || !dispatcher || !collector || !obs_rows))) return GRAAL_E_ARG;
    if (!h->have_sub) return fail(h, GRAAL_E_STATE, "upload_subfrags first");
    CK(hipSetDevice(h->device));
    void* old[] = {h->d_dup_bins, h->d_dup_index, h->d_dispatcher, h->d_collector, h->d_sub_ids_all, h->d_rep_obs};
    for (void* p : old) if (p) (void)hipFree(p);
    h->d_dup_bins = h->d_dup_index = h->d_dispatcher = h->d_collector = h->d_sub_ids_all = nullptr; h->d_rep_obs = nullptr;
    h->h_dup_index.assign((size_t)h->n_bins, -1);
    h->has_rep = n_dup > 0; h->n_dup = n_dup;
    h->have_frags = false; h->have_contacts = false; h->order_valid = false; // both depend on which bins are repeated
    if (n_dup == 0) { compute_t_all(h); return sync_args(h); }
    for (int i = 0; i < n_dup; i++) {
        if (dup_bins[i] < 0 || dup_bins[i] >= h->n_bins || h->h_dup_index[dup_bins[i]] >= 0) return fail(h, GRAAL_E_ARG, "repeated bins must be distinct bin ids");
        h->h_dup_index[dup_bins[i]] = i;
    }
    for (int b = 0; b < h->n_bins; b++) {
        const int lo = dispatcher[2 * b], hi = dispatcher[2 * b + 1];
        if (lo < 0 || hi > n_collector || hi <= lo) return fail(h, GRAAL_E_ARG, "dispatcher ranges must be non-empty and inside the collector");
        if (h->h_dup_index[b] < 0 && (hi - lo != 1 || collector[lo] != b)) return fail(h, GRAAL_E_ARG, "a non-repeated bin has exactly one copy: itself");
    }
    CK(hipMalloc(&h->d_dup_bins, sizeof(int) * (size_t)n_dup));
    CK(hipMalloc(&h->d_dup_index, sizeof(int) * (size_t)h->n_bins));
    CK(hipMalloc(&h->d_dispatcher, sizeof(int) * 2 * (size_t)h->n_bins));
    CK(hipMalloc(&h->d_collector, sizeof(int) * (size_t)n_collector));
    CK(hipMalloc(&h->d_sub_ids_all, sizeof(int) * 4 * (size_t)h->n_bins));
    CK(hipMalloc(&h->d_rep_obs, sizeof(float) * 3 * (size_t)n_dup * (size_t)h->n_sub_total));
    CK(hipMemcpy(h->d_dup_bins, dup_bins, sizeof(int) * (size_t)n_dup, hipMemcpyHostToDevice));
    CK(hipMemcpy(h->d_dup_index, h->h_dup_index.data(), sizeof(int) * (size_t)h->n_bins, hipMemcpyHostToDevice));
    CK(hipMemcpy(h->d_dispatcher, dispatcher, sizeof(int) * 2 * (size_t)h->n_bins, hipMemcpyHostToDevice));
    CK(hipMemcpy(h->d_collector, collector, sizeof(int) * (size_t)n_collector, hipMemcpyHostToDevice));
    CK(hipMemcpy(h->d_sub_ids_all, h->h_sub_id.data(), sizeof(int) * 4 * (size_t)h->n_bins, hipMemcpyHostToDevice));
    CK(hipMemcpy(h->d_rep_obs, obs_rows, sizeof(float) * 3 * (size_t)n_dup * (size_t)h->n_sub_total, hipMemcpyHostToDevice));
    compute_t_all(h);
    return sync_args(h);
}

// counts as float32: what the reference's dense observation matrix holds (cuda_lib_gl.py:153-156).  Integer counts are
// exact up to 2^24; the blacklist fill (cuda_lib_gl.py:161-172) makes them non-integer.
static int upload_contacts_impl(graal_ctx* h, const int32_t* row, const int32_t* col, const float* count, int64_t nnz)
{
    if (h && h->stream) (void)hipStreamSynchronize(h->stream); // (graal_begin_step may have left its relabel kernels running)
    if (!h->have_sub) return fail(h, GRAAL_E_STATE, "upload_subfrags first");
    if (nnz >= (1ll << 32)) return fail(h, GRAAL_E_ARG, "at most 2^32 - 1 contacts per shard");
    CK(hipSetDevice(h->device));
    long long c_lf_q = 0;
    long long lf_small[16];
    for (int i = 0; i < 16; i++) lf_small[i] = llrint(lf_term((double)i) * Q_SCALE);
    for (int64_t i = 0; i < nnz; i++) {
        if (row[i] < 0 || col[i] >= h->n_sub_total || row[i] >= col[i]) return fail(h, GRAAL_E_ARG, "contacts need 0 <= row < col < n_sub_total");
        const float c = count[i];
        if (!(c > 0.0f) || !(c < 1.0e30f)) return fail(h, GRAAL_E_ARG, "contact counts must be > 0 and finite");
        if (h->has_rep && (h->h_dup_index[s2b_host(h, row[i])] >= 0 || h->h_dup_index[s2b_host(h, col[i])] >= 0))
            return fail(h, GRAAL_E_ARG, "contacts of repeated bins belong to graal_upload_repeats (observation rows), not to the contact list");
        const int ci = (int)c;
        c_lf_q += (c < 16.0f && (float)ci == c) ? lf_small[ci] : llrint(lf_term((double)c) * Q_SCALE);
    }
    if (h->row) { (void)hipFree(h->row); (void)hipFree(h->col); (void)hipFree(h->cnt); (void)hipFree(h->queue); h->row = h->col = h->cnt = nullptr; h->queue = nullptr; }
    const size_t bytes = sizeof(int) * (size_t)(nnz + 8); // +8: int4 tail reads stay in bounds
    CK(hipMalloc(&h->row, bytes)); CK(hipMalloc(&h->col, bytes)); CK(hipMalloc(&h->cnt, bytes));
    CK(hipMemset(h->row, 0, bytes)); CK(hipMemset(h->col, 0, bytes));
    CK(hipMalloc(&h->queue, sizeof(QRaw) * (size_t)(nnz + 8))); // every contact may have both ends affected in the worst case
    // (recycled device memory may hold another engine's entries, tags and all: a consumer that validates tags must never meet them)
    CK(hipMemset(h->queue, 0, sizeof(QRaw) * (size_t)(nnz + 8)));
    if (nnz) {
        CK(hipMemcpy(h->row, row, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice));
        CK(hipMemcpy(h->col, col, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice));
        CK(hipMemcpy(h->cnt, count, sizeof(float) * (size_t)nnz, hipMemcpyHostToDevice));
    }
    h->nnz = nnz; h->c_lf_q = c_lf_q; h->have_contacts = true;
    // the observed counts of every bin's OWN sub-fragment pairs (own_pixel_q): contacts whose two sub-fragments belong to one bin
    if (h->d_own_obs) { (void)hipFree(h->d_own_obs); h->d_own_obs = nullptr; }
    if (!h->single_sub && !h->has_rep) {
        std::vector<float> own(3 * (size_t)h->n_bins, 0.0f);
        for (int64_t i = 0; i < nnz; i++) {
            const int ra = h->h_sub2bin[(size_t)row[i]], rb = h->h_sub2bin[(size_t)col[i]];
            if ((ra >> 2) != (rb >> 2)) continue;
            const int sa = std::min(ra & 3, rb & 3), sb = std::max(ra & 3, rb & 3);
            if (sa == sb) continue;
            own[3 * (size_t)(ra >> 2) + (sa == 0 ? sb - 1 : 2)] = count[i];
        }
        CK(hipMalloc(&h->d_own_obs, sizeof(float) * own.size()));
        CK(hipMemcpy(h->d_own_obs, own.data(), sizeof(float) * own.size(), hipMemcpyHostToDevice));
    }
    h->own_complete = false;
    h->carry_q = 0; h->carry_bad = true;   // (whatever was pending belonged to another problem)
    return sync_args(h);
}

int graal_upload_contacts(graal_ctx* h, const int32_t* row, const int32_t* col, const int32_t* count, int64_t nnz)
{
    if (!h || nnz < 0 || (nnz > 0 && (!row || !col || !count))) return GRAAL_E_ARG;
    std::vector<float> f((size_t)nnz);
    for (int64_t i = 0; i < nnz; i++) {
        if (count[i] <= 0) return fail(h, GRAAL_E_ARG, "contact counts must be > 0");
        if (count[i] > (1 << 24)) return fail(h, GRAAL_E_ARG, "integer contact counts above 2^24 are not exact in float32");
        f[(size_t)i] = (float)count[i];
    }
    return upload_contacts_impl(h, row, col, f.data(), nnz);
}

int graal_upload_contacts_f32(graal_ctx* h, const int32_t* row, const int32_t* col, const float* count, int64_t nnz)
{
    if (!h || nnz < 0 || (nnz > 0 && (!row || !col || !count))) return GRAAL_E_ARG;
    return upload_contacts_impl(h, row, col, count, nnz);
}

int graal_upload_frags(graal_ctx* h, const int32_t* const soa[GRAAL_N_FIELDS], int32_t n)
{
    if (h && h->stream) (void)hipStreamSynchronize(h->stream); // (graal_begin_step may have left its relabel kernels running)
    if (h && h->fstream) (void)hipStreamSynchronize(h->fstream); // (... and the last commit its own-pixel correction, which reads the layout)
    if (!h || !soa || n <= 0) return GRAAL_E_ARG;
    if (!h->have_sub) return fail(h, GRAAL_E_STATE, "upload_subfrags first");
    if (n != h->n_bins && !h->has_rep) return fail(h, GRAAL_E_STATE, "n differs from the number of bins: upload the repeats (graal_upload_repeats) first");
    if (n < h->n_bins) return fail(h, GRAAL_E_ARG, "fewer fragments than bins");
    if ((long long)n >= (1ll << LABEL_BITS) / 2 - 4) return fail(h, GRAAL_E_ARG, "too many fragments for the relabel key");
    CK(hipSetDevice(h->device));
    for (int i = 0; i < n; i++) {
        if (soa[F_IDC][i] < 0 || soa[F_IDC][i] >= 2 * n + 4) return fail(h, GRAAL_E_ARG, "id_c must be in [0, 2n+4)");
        const int b = soa[F_IDD][i];
        if (b < 0 || b >= h->n_bins) return fail(h, GRAAL_E_ARG, "id_d out of range");
        const bool special = h->has_rep && h->h_dup_index[b] >= 0;
        if (!special && (soa[F_REP][i] != 0 || soa[F_ACTIV][i] != 1 || b != i))
            return fail(h, GRAAL_E_ARG, "rep / activ / id_d of a fragment of a non-repeated bin must be 0 / 1 / its own index");
        if ((soa[F_REP][i] != 0 && soa[F_REP][i] != 1) || (soa[F_ACTIV][i] != 0 && soa[F_ACTIV][i] != 1))
            return fail(h, GRAAL_E_ARG, "rep and activ must be 0 or 1");
    }
    if (h->n != n) {
        void* old[] = {h->soa_mem[0], h->soa_mem[1], h->geo, h->keys, h->keys_sorted, h->o2n, h->len_of2[0], h->len_of2[1], h->contig_off2[0], h->contig_off2[1], h->perm, h->pstart, h->cbase, h->link, h->mates, h->cub_tmp};
        for (void* p : old) if (p) (void)hipFree(p);
        for (int b = 0; b < 2; b++) {
            CK(hipMalloc(&h->soa_mem[b], sizeof(int) * (size_t)n * GRAAL_N_FIELDS));
            for (int k = 0; k < GRAAL_N_FIELDS; k++) h->soa[b].p[k] = h->soa_mem[b] + (size_t)k * n;
        }
        CK(hipMalloc(&h->geo, sizeof(Geo) * (size_t)n));
        CK(hipMalloc(&h->keys, sizeof(unsigned long long) * (size_t)n));
        CK(hipMalloc(&h->keys_sorted, sizeof(unsigned long long) * (size_t)n));
        CK(hipMalloc(&h->o2n, sizeof(int) * (size_t)(2 * n + 8)));
        for (int b = 0; b < 2; b++) {
            CK(hipMalloc(&h->len_of2[b], sizeof(int) * (size_t)(n + 2)));
            CK(hipMalloc(&h->contig_off2[b], sizeof(int) * (size_t)(n + 2)));
        }
        CK(hipMalloc(&h->perm, sizeof(int) * (size_t)n));
        CK(hipMalloc(&h->pstart, sizeof(int) * (size_t)n));
        CK(hipMalloc(&h->cbase, sizeof(int) * (size_t)n));
        CK(hipMalloc(&h->link, sizeof(Link) * (size_t)n));
        CK(hipMalloc(&h->mates, sizeof(int) * N_MATES * (size_t)n));
        size_t b1 = 0, b2 = 0;
        (void)hipcub::DeviceRadixSort::SortKeys(nullptr, b1, h->keys, h->keys_sorted, n, 0, 2 * LABEL_BITS, h->stream);
        (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b2, h->len_of2[0], h->contig_off2[0], n + 1, h->stream);
        h->cub_tmp_bytes = b1 > b2 ? b1 : b2;
        CK(hipMalloc(&h->cub_tmp, h->cub_tmp_bytes));
        h->n = n;
    }
    h->cur = 0; h->ranks_valid = false; h->pending_commits = 0; h->begin_launched = false; h->stats_from_apply = false;
    for (int k = 0; k < GRAAL_N_FIELDS; k++)
        CK(hipMemcpy(h->soa[0].p[k], soa[k], sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
    {   // statistics per fragment: those of its bin; copies of repeated bins (original included) carry no sub-fragments in
        // the sparse path -- their pixels are priced by the dense repeat kernels
        std::vector<Stat> sf((size_t)n);
        for (int i = 0; i < n; i++) {
            const int b = soa[F_IDD][i];
            sf[(size_t)i] = h->h_stat[(size_t)b];
            if (h->has_rep && h->h_dup_index[b] >= 0) sf[(size_t)i].n = 0;
        }
        if (h->stat_frag) (void)hipFree(h->stat_frag);
        CK(hipMalloc(&h->stat_frag, sizeof(Stat) * (size_t)n));
        CK(hipMemcpy(h->stat_frag, sf.data(), sizeof(Stat) * (size_t)n, hipMemcpyHostToDevice));
    }
    h->have_frags = true; h->order_valid = false;
    int rc = refresh(h);
    if (rc) return rc;
    CK(hipStreamSynchronize(h->stream));
    return sync_args(h);
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
    k_stats<<<std::min(blocks_for(h->n, 256), 64), 256, 0, h->stream>>>(h->soa[h->cur], h->n, h->d_scalars);
    CK(hipGetLastError());
    long long res[16];
    { int rc = fetch_stats(h, res, false); if (rc) return rc; }
    out[0] = res[0]; out[1] = res[1]; out[2] = res[2]; out[3] = res[3]; out[4] = res[4]; out[5] = res[5];
    out[6] = res[14]; out[7] = 0;
    // (h->n_contigs is the contig count of the RANKED layout -- what the next incremental relabel counts from (k_incr's nc_old) and what
    // graal_apply_move checks max_id against.  With a commit pending these statistics are of the layout BEHIND it: taking them over made the
    // relabel that followed count from the wrong number whenever that commit had changed the number of contigs -- bench.py's
    // layout_stats() / modify_gl_cuda_buffer(0) pair behind its MCMC warm-up; found in round 5 by a guard in k_incr, after a year of
    // silently misplaced entries of the position index in that one flow and four GPU faults in its two-rank rehearsal)
    if (h->pending_commits == 0 && h->ranks_valid) h->n_contigs = (int)res[0];
    return GRAAL_OK;
}

// first half of graal_begin_step: everything it launches; the statistics are on their way to pinned host memory afterwards
static int begin_step_launch(graal_ctx* h, bool defer = false)
{
    CK(hipSetDevice(h->device));
    const int n = h->n, bs = 256, nb = blocks_for(n, bs);
    const int cur = h->cur;
    SoaPtr s = h->soa[cur];
    static const bool no_incr = getenv("GRAAL_NO_INCREMENTAL_RELABEL") != nullptr;
    // The statistics do not depend on the relabel.  After exactly one commit they are in pinned host memory already (the
    // commit kernel published them); when nothing changed, the statistics kernel publishes them itself.  Either way the host
    // reads them -- and goes on to draw the step's proposal -- while the relabel kernels launched here still run (everything
    // the host launches next is ordered after them: same stream, or the event recorded below).
    const bool incr = h->ranks_valid && h->pending_commits == 1 && h->incr_ok && !no_incr;
    const bool early = h->ranks_valid && (h->pending_commits == 0 || incr);
    bool full_relabel = false;
    const bool from_apply = incr && h->stats_from_apply;
    const bool defer_stats = defer && h->spin_ok && from_apply;   // k_tm's extra block publishes them (graal_step reads them with the scores)
    h->corr_src = (from_apply && h->apply_had_own) ? 1 : ((h->ranks_valid && h->pending_commits == 0) ? 2 : 0);
    if (h->corr_skip) { h->corr_src = 2; h->corr_skip = false; }
    h->corr_inflight = true;
    if (from_apply) {
        // nothing to launch for the statistics: the commit kernel left its rows, k_incr (below) publishes them
    } else if (early) {
        h->stats_seq += 1;
        k_stats<<<std::min(nb, 64), bs, 0, h->stream>>>(s, n, h->d_scalars, h->h_stats, h->stats_seq, incr ? 0 : 1);
    } else k_stats<<<std::min(nb, 64), bs, 0, h->stream>>>(s, n, h->d_scalars);
    h->stats_from_apply = false;
    if (h->ranks_valid && h->pending_commits == 0) {
        // nothing changed since the last call: labels are ranks already
    } else if (incr) {
        // exactly one commit since the last ranking: count instead of sort (rank arrays of buffer 1-cur -> buffer cur)
        k_incr<<<blocks_for(n + 1, bs) + 1, bs, 0, h->stream>>>(s, n, h->d_chg + h->chg_last, h->len_of2[1 - cur], h->contig_off2[1 - cur], h->n_contigs,
                                                           h->len_of2[cur], h->contig_off2[cur], h->perm, h->pstart, h->cbase, h->geo, h->link, h->d_scalars,
                                                           h->mates, (int*)(h->d_chg + (1 - h->chg_last)), (int)(sizeof(Changed) / sizeof(int)),
                                                           (defer_stats ? nullptr : h->d_part), h->apply_blocks, from_apply ? h->h_stats : nullptr, h->stats_seq, h->h_guard);
        CK(hipGetLastError());
        if (defer && h->spin_ok) {
            // graal_step: k_tm goes out next, without an event: it spins until the step's k_scan (behind this kernel on the stream) starts
            h->relabel_spin_pending = true;
            h->stats_pub_pending = defer_stats;
        } else {
            // the host does not wait for this kernel: whatever runs on the OTHER stream next (k_tm) must
            CK(hipEventRecord(h->ev_relabel, h->stream));
            h->relabel_pending = true;
        }
    } else {
        k_relabel_keys<<<nb, bs, 0, h->stream>>>(s, n, h->keys);
        CK(hipGetLastError());
        size_t tb = h->cub_tmp_bytes;
        CK(hipcub::DeviceRadixSort::SortKeys(h->cub_tmp, tb, h->keys, h->keys_sorted, n, 0, 2 * LABEL_BITS, h->stream));
        k_relabel_o2n<<<nb, bs, 0, h->stream>>>(h->keys_sorted, n, h->d_scalars, h->o2n, h->len_of2[cur]);
        k_relabel_apply<<<nb, bs, 0, h->stream>>>(s, n, h->o2n);
        CK(hipGetLastError());
        tb = h->cub_tmp_bytes;
        CK(hipcub::DeviceScan::ExclusiveSum(h->cub_tmp, tb, h->len_of2[cur], h->contig_off2[cur], n + 1, h->stream));
        k_build_perm<<<nb, bs, 0, h->stream>>>(s, n, h->contig_off2[cur], h->perm, h->cbase, h->pstart);
        CK(hipGetLastError());
        int rc = refresh(h);
        if (rc) return rc;
        k_mates<<<nb, bs, 0, h->stream>>>(n, h->perm, h->cbase, h->link, h->mates, (int*)h->d_chg, (int)(2 * sizeof(Changed) / sizeof(int)));
        CK(hipGetLastError());
        full_relabel = true;
    }
    if (!early) { // the sorting path reads the statistics on the device, so they are published (and re-armed) behind it
        h->stats_seq += 1;
        k_stats_fin<<<1, 64, 0, h->stream>>>(h->d_scalars, h->h_stats, h->stats_seq, 1);
        CK(hipGetLastError());
    }
    if (defer && full_relabel) {
        // graal_step will not wait for the statistics (which, on this path, are published behind the relabel and so imply it):
        // whatever it launches on the other stream is ordered behind these kernels by the event
        CK(hipEventRecord(h->ev_relabel, h->stream));
        h->relabel_pending = true;
    }
    h->begin_launched = true;
    return GRAAL_OK;
}

int graal_begin_step_launch(graal_ctx* h)
{
    if (!h) return GRAAL_E_ARG;
    if (!h->have_frags) return fail(h, GRAAL_E_STATE, "no fragments uploaded");
    return h->begin_launched ? GRAAL_OK : begin_step_launch(h);
}

// second half of graal_begin_step: wait for the statistics, check them, take them over
static int begin_step_collect(graal_ctx* h, int64_t stats[8], int32_t* max_id);

__global__ void k_stats_pub(const long long* __restrict__ part, int n_part, long long* __restrict__ stats, volatile long long* host, long long seq)
{
    publish_partials(part, n_part, stats, host, seq, (int)threadIdx.x);
}

int graal_begin_step(graal_ctx* h, int64_t stats[8], int32_t* max_id)
{
    if (!h) return GRAAL_E_ARG;
    if (!h->have_frags) return fail(h, GRAAL_E_STATE, "no fragments uploaded");
    if (!h->begin_launched) { int rc = begin_step_launch(h); if (rc) return rc; }
    h->begin_launched = false;
    if (h->stats_pub_pending) {   // (graal_step deferred the publication to the scoring kernels and then handed the step back)
        k_stats_pub<<<1, 64, 0, h->stream>>>(h->d_part, h->apply_blocks, h->d_scalars, h->h_stats, h->stats_seq);
        CK(hipGetLastError());
        h->stats_pub_pending = false;
    }
    return begin_step_collect(h, stats, max_id);
}

// graal_step launches the scoring kernels BEFORE it has the statistics (they are stream-ordered behind the relabel and read
// the contig count on the device) and collects the statistics together with the scores: what graal_begin_step establishes on
// the host side, without the wait.  The longest contig is then one commit stale: a commit at most joins two contigs.
static void begin_step_assume(graal_ctx* h)
{
    h->begin_launched = false;
    h->order_valid = true; h->ranks_valid = true; h->pending_commits = 0; h->incr_ok = false;
    h->lcont_bound = (int)std::min<long long>(h->n, 2ll * h->max_lcont + 2);
}

static int begin_step_collect(graal_ctx* h, int64_t stats[8], int32_t* max_id)
{
    const int n = h->n;
    long long res[16];
    { int rc = wait_stats(h, res); if (rc) return rc; }
    if (h->h_guard && ((volatile long long*)h->h_guard)[0] != 0) {
        volatile long long* g = h->h_guard;
        char msg[320];
        snprintf(msg, sizeof msg, "relabel: an index out of range was caught (fragment %lld, old label %lld, new rank %lld, offset %lld, position %lld, length %lld, "
                 "new contigs %lld, contigs %lld, n %d): the layout's position index is not to be trusted", g[1], g[2], g[3], g[4], g[5], g[6], g[7] >> 32, g[7] & 0xffffffffll, n);
        g[0] = 0;
        return fail(h, GRAAL_E_STATE, msg);
    }
    const int nc = (int)res[0];
    // (a corrupt layout could have made the kernels above index out of range; the uploads validate labels and the
    // mutations keep them in [0, n_contigs + 2], so this is a consistency check, not a guard)
    if (nc <= 0 || nc > n || res[7] >= 2 * n + 8) return fail(h, GRAAL_E_STATE, "corrupt layout: contig heads / labels out of range");
    if (res[6] != 0 && !h->has_rep) return fail(h, GRAAL_E_STATE, "corrupt layout: rep / activ / id_d changed without repeats");
    h->n_contigs = nc; h->order_valid = true; h->ranks_valid = true; h->pending_commits = 0; h->incr_ok = false;
    // the commit's own-pixel correction rides on the statistics of the layout it produced
    if (h->corr_src == 1) {
        long long cq = 0, cb = 0;
        { const int rc = wait_own(h, &cq, &cb); if (rc) return rc; }
        if (cb != 0) h->carry_bad = true; else h->carry_q += cq;
    }
    else if (h->corr_src == 0) h->carry_bad = true;    // (a layout that is not one commit away from the last one: the correction is unknown)
    h->corr_src = 2; h->corr_inflight = false;
    h->max_lcont = (int)res[4];
    h->lcont_bound = h->max_lcont;
    if (max_id) *max_id = nc - 1;
    if (stats) {
        for (int i = 0; i < 6; i++) stats[i] = res[i];
        stats[6] = res[14];                      // #(circ == 1)
        stats[7] = (long long)*(int*)&res[13];   // fragments that hit the unwritten paste branch since the last begin_step
    }
    return GRAAL_OK;
}

int graal_relabel_contigs(graal_ctx* h, int32_t* max_id) { return graal_begin_step(h, nullptr, max_id); }

// the full evaluation in two halves: the kernels (and the publication of the sums) on stream `fs`, and the wait for them.  A caller
// that launches on another stream than the engine's orders it behind the relabel itself (graal_step, flag 8).
static int full_launch(graal_ctx* h, hipStream_t fs)
{
    if (!(h->have_frags && h->have_contacts && h->have_par)) return fail(h, GRAAL_E_STATE, "upload fragments, contacts and parameters first");
    if (!h->order_valid) return fail(h, GRAAL_E_STATE, "call graal_relabel_contigs after changing the layout");
    CK(hipSetDevice(h->device));
    SoaPtr s = h->soa[h->cur];
    const bool quirk = (h->mode & GRAAL_MODE_REF_TRANS_ACCU) != 0;
    if (h->nnz) {
        static const bool no_compact = getenv("GRAAL_FULL_NO_COMPACT") != nullptr;
        // (equal RF counts everywhere: the reference's trans-branch indexing picks the same count whatever the orientation, so
        // the compact records serve that mode too)
        const bool compact = h->uniform_accu > 0 && !no_compact;
        k_subrec<<<blocks_for(h->n, 256), 256, 0, fs>>>(h->n, h->geo, h->stat_frag, h->d_sub_ids, h->sub_rec, compact ? h->sub_rec8 : nullptr, h->sub_lab16);
        // 8 blocks of 256 threads per CU; every lane takes FULL_G groups of 4 contacts per iteration
        static const int full_g = getenv("GRAAL_FULL_G") ? atoi(getenv("GRAAL_FULL_G")) : 2;
        static const int full_bpc = getenv("GRAAL_FULL_BPC") ? atoi(getenv("GRAAL_FULL_BPC")) : 8;   // blocks per CU
        const int FG = full_g == 1 ? 1 : (full_g == 4 ? 4 : 2);
        const long long groups = (h->nnz >> 2) + 1;
        const int nb = (int)std::max<long long>(1, std::min<long long>((groups + 256 * FG - 1) / (256 * FG), 256 * full_bpc));
#define FULL_NNZ_ARGS reinterpret_cast<const int4*>(h->row), reinterpret_cast<const int4*>(h->col), reinterpret_cast<const int4*>(h->cnt), \
                      h->nnz, h->sub_rec, s.p[F_LCONTBP], h->nfpb, h->par, h->ln_lut_n, quirk ? 1 : 0, h->d_scalars + 8, h->d_scalars + FULL_BAD
        // labels in LDS (k_full_nnz_l): uniform RF counts, a list worth it, and 2 bytes per sub-fragment within 150 KB of LDS
        static const bool no_lds = getenv("GRAAL_FULL_NO_LDS") != nullptr;
        const size_t lab_bytes = 2 * (((size_t)h->n_sub_total + 7) & ~(size_t)7);
        if (compact && !no_lds && h->nnz >= 2000000 && lab_bytes <= 150 * 1024) {
            static bool attr_set = false;
            if (!attr_set) {
                CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_full_nnz_l<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
                CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_full_nnz_l<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
                attr_set = true;
            }
            const int nbl = (int)std::max<long long>(1, std::min<long long>((groups + 1024 * FG - 1) / (1024 * FG), 256));
#define FULL_NNZ_L_ARGS reinterpret_cast<const int4*>(h->row), reinterpret_cast<const int4*>(h->col), reinterpret_cast<const int4*>(h->cnt), \
                        h->nnz, h->sub_lab16, h->n_sub_total, h->sub_rec8, h->sub_rec, s.p[F_LCONTBP], h->nfpb, h->par, h->uniform_accu,        \
                        h->d_scalars + 8, h->d_scalars + FULL_BAD
            if (FG == 2) k_full_nnz_l<2><<<nbl, 1024, lab_bytes, fs>>>(FULL_NNZ_L_ARGS);
            else k_full_nnz_l<4><<<nbl, 1024, lab_bytes, fs>>>(FULL_NNZ_L_ARGS);
#undef FULL_NNZ_L_ARGS
        }
        else if (compact) {
#define FULL_NNZ_U_ARGS reinterpret_cast<const int4*>(h->row), reinterpret_cast<const int4*>(h->col), reinterpret_cast<const int4*>(h->cnt), \
                        h->nnz, h->sub_rec8, h->sub_rec, s.p[F_LCONTBP], h->nfpb, h->par, h->uniform_accu, h->d_scalars + 8, h->d_scalars + FULL_BAD
            if (FG == 1) k_full_nnz_u<1><<<nb, 256, 0, fs>>>(FULL_NNZ_U_ARGS);
            else if (FG == 4) k_full_nnz_u<4><<<nb, 256, 0, fs>>>(FULL_NNZ_U_ARGS);
            else k_full_nnz_u<2><<<nb, 256, 0, fs>>>(FULL_NNZ_U_ARGS);
#undef FULL_NNZ_U_ARGS
        }
        else if (FG == 1) k_full_nnz<1><<<nb, 256, 0, fs>>>(FULL_NNZ_ARGS);
        else if (FG == 4) k_full_nnz<4><<<nb, 256, 0, fs>>>(FULL_NNZ_ARGS);
        else k_full_nnz<2><<<nb, 256, 0, fs>>>(FULL_NNZ_ARGS);
#undef FULL_NNZ_ARGS
    }
    if (quirk && h->n_ubins) // T_all prices every pair of different bins with the plain trans value: add the indexing's difference
        k_quirk_mass<<<blocks_for((long long)h->n_ubins * h->n_bins, 256), 256, 0, fs>>>(h->n_ubins, h->d_ubins, h->n_bins, h->geo, h->stat_frag,
                                                                                             h->nfpb, h->par, h->d_scalars + 9, h->d_scalars + FULL_BAD);
    static const int fmt_env = getenv("GRAAL_FULL_MASS_TILED") ? atoi(getenv("GRAAL_FULL_MASS_TILED")) : -1;   // (A/B: 0 = never, 1 = always)
    const int lc_full = std::max(std::max(h->max_lcont, h->lcont_bound), 1);
    // (maps of a few thousand bins keep the kernel with one WAVE per fragment x: 3,500 bins are 55 tiles -- the C3 stand-in, which evaluates the
    // full likelihood every step, went from 237 to 522 us per step with the tiled kernel)
    if (fmt_env == 1 || (fmt_env != 0 && lc_full > 256 && h->n > 16384)) {
        // long contigs: the tiled kernel; S waves share an x tile so that the grid has a few thousand waves whatever the contigs' length
        const int n_tiles = (h->n + 63) / 64;
        const int S = std::min(16, std::max(1, ((std::min(lc_full, h->n) + 63) / 64 + 7) / 8));
        const float norm_u = h->uniform_accu > 0 ? (float)(h->uniform_accu * h->uniform_accu) / h->nfpb : -1.0f;
        const int nb = (n_tiles * S + 3) / 4;
        if (h->single_sub) k_full_mass_t<false><<<nb, 256, 0, fs>>>(h->n, h->perm, h->geo, h->stat_frag, s.p[F_LCONT], s.p[F_LCONTBP], s.p[F_POS], h->nfpb, h->par,
                                                                   reach_bp(h), S, norm_u, h->d_scalars + 9, h->d_scalars + FULL_BAD);
        else k_full_mass_t<true><<<nb, 256, 0, fs>>>(h->n, h->perm, h->geo, h->stat_frag, s.p[F_LCONT], s.p[F_LCONTBP], s.p[F_POS], h->nfpb, h->par,
                                                     reach_bp(h), S, norm_u, h->d_scalars + 9, h->d_scalars + FULL_BAD);
    } else if (h->n <= 16384)
        k_full_mass<64><<<blocks_for(h->n, 4), 256, 0, fs>>>(h->n, h->perm, h->contig_off2[h->cur], h->geo, h->stat_frag, s.p[F_LCONT],
                                                                    s.p[F_LCONTBP], s.p[F_POS], h->nfpb, h->par, reach_bp(h),
                                                                    h->d_scalars + 9, h->d_scalars + FULL_BAD);
    else
        k_full_mass<16><<<blocks_for(h->n, 16), 256, 0, fs>>>(h->n, h->perm, h->contig_off2[h->cur], h->geo, h->stat_frag, s.p[F_LCONT],
                                                                     s.p[F_LCONTBP], s.p[F_POS], h->nfpb, h->par, reach_bp(h),
                                                                     h->d_scalars + 9, h->d_scalars + FULL_BAD);
    if (h->has_rep) { // every pixel of a repeated bin, densely.  With an exchange attached the pixels are dealt to the ranks and the sum travels
                      // with the contacts' part (full_collect), which the callers add up over the ranks; otherwise every rank computes all of it
        const RepArgs R = rep_args(h);
        const int fw = h->x_host ? h->x_world : (h->nccl_comm ? h->n_world : 1), fr = h->x_host ? h->x_rank : (h->nccl_comm ? h->n_rank : 0);
        h->full_rep_sharded = fw > 1;
        k_rep_full<<<blocks_for((long long)h->n_dup * h->n_bins, 256), 256, 0, fs>>>(R, fr, fw, h->d_scalars + 17, h->d_scalars + FULL_BAD);
    }
    // the accumulators (d_scalars[8], [9], [17], [FULL_BAD]) are zero at rest because k_full_pub clears them behind the sums it
    // publishes: a call that fails between its first kernel and that publication must not leave partial sums to the next one
    auto reset_acc = [&]() {
        (void)hipStreamSynchronize(fs);
        (void)hipGetLastError();
        (void)hipMemset(h->d_scalars + 8, 0, 2 * sizeof(long long));
        (void)hipMemset(h->d_scalars + 17, 0, sizeof(long long));
        (void)hipMemset(h->d_scalars + FULL_BAD, 0, sizeof(long long));
    };
    { const hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { reset_acc(); h->err = hipGetErrorString(e_); return GRAAL_E_HIP; } }
    h->full_seq += 1;
    k_full_pub<<<1, 64, 0, fs>>>(h->d_scalars, h->h_full, h->full_seq);
    { const hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { reset_acc(); h->err = hipGetErrorString(e_); return GRAAL_E_HIP; } }
    return GRAAL_OK;
}

static int full_collect(graal_ctx* h, hipStream_t fs, int64_t q_out[2])
{
    auto reset_acc = [&]() {
        (void)hipStreamSynchronize(fs);
        (void)hipGetLastError();
        (void)hipMemset(h->d_scalars + 8, 0, 2 * sizeof(long long));
        (void)hipMemset(h->d_scalars + 17, 0, sizeof(long long));
        (void)hipMemset(h->d_scalars + FULL_BAD, 0, sizeof(long long));
    };
    {
        volatile long long* p = h->h_full;
        bool seen = false;
        for (long long spin = 0; spin < 400000000ll; spin++) {
            if (p[0] == h->full_seq) { seen = true; break; }
            if ((spin & 0xfffff) == 0xfffff && hipStreamQuery(fs) != hipErrorNotReady) { seen = (p[0] == h->full_seq); break; }
            __builtin_ia32_pause();
        }
        if (!seen) {
            const hipError_t e_ = hipStreamSynchronize(fs);
            if (e_ != hipSuccess || p[0] != h->full_seq) { reset_acc(); return fail(h, GRAAL_E_HIP, "the full evaluation did not publish its sums"); }
        }
        __sync_synchronize();
    }
    const long long res[2] = {h->h_full[1], h->h_full[2]};
    const long long bad = h->h_full[3], rep_q = h->has_rep ? h->h_full[4] : 0;
    // (q[0]: what the callers add up over the ranks -- the contacts of this rank's shard and, when the repeated bins' pixels were dealt to the
    // ranks, this rank's share of them; q[1]: what every rank computes alike)
    const bool rep_sharded = h->has_rep && h->full_rep_sharded;
    q_out[0] = res[0] - (int64_t)h->c_lf_q + (rep_sharded ? rep_q : 0);
    q_out[1] = -(res[1] + (int64_t)llrint(h->t_all * Q_SCALE)) + (rep_sharded ? 0 : rep_q);
    if (bad) { q_out[0] = Q_BAD; q_out[1] = 0; } // a term was not finite / out of range: INT64_MIN exactly, the host reports NaN
    return GRAAL_OK;
}

// The full likelihood of a sharded contact list inside graal_step (flag 8): every rank has evaluated its shard (q[0]: the contacts' part,
// rounded term by term: an integer sum, the same for any sharding) and the whole mass part (q[1], computed by every rank alike).  The
// ranks' q[0] are summed through spare words of the exchange slots -- written and read by the HOSTS (the segment is shared memory),
// tagged with the sequence number of the step's last candidate evaluation, which every rank has published by now; a slot is written
// again two steps later, when every rank has read it (the argument of eval_sync).  Q_BAD (a term was not finite) is sticky.
constexpr int X_FULL = 400;   // slot words [X_FULL] = q[0], [X_FULL + 1] = tag
static_assert(X_COARSE + MAXK * N_OPS <= X_FULL && X_FULL + 2 < X_SLOT_WORDS - 1, "exchange slot layout");
static int full_exchange(graal_ctx* h, int64_t q[2])
{
    const long long tag = h->seq;
    const size_t world = (size_t)h->x_world, par = (size_t)(tag & 1);
    volatile long long* mine = h->x_host + (par * world + (size_t)h->x_rank) * X_SLOT_WORDS;
    mine[X_FULL] = q[0];
    __sync_synchronize();
    mine[X_FULL + 1] = tag;
    __sync_synchronize();
    long long sum = q[0];
    for (int r = 0; r < h->x_world; r++) {
        if (r == h->x_rank) continue;
        volatile long long* o = h->x_host + (par * world + (size_t)r) * X_SLOT_WORDS;
        const auto t0 = std::chrono::steady_clock::now();
        long long spin = 0;
        while (o[X_FULL + 1] != tag) {
            if ((++spin & 0xfffff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60))
                return fail(h, GRAAL_E_STATE, "exchange: another rank did not publish its full likelihood within 60 s (ranks out of step, or a rank died)");
            __builtin_ia32_pause();
        }
        __sync_synchronize();
        const long long oq = o[X_FULL];
        sum = (sum == Q_BAD || oq == Q_BAD) ? Q_BAD : (long long)((unsigned long long)sum + (unsigned long long)oq);
    }
    q[0] = sum;
    return GRAAL_OK;
}

int graal_eval_full_q(graal_ctx* h, int64_t q_out[2])
{
    if (!h || !q_out) return GRAAL_E_ARG;
    const int rc = full_launch(h, h->stream);
    return rc ? rc : full_collect(h, h->stream, q_out);
}

// set parameters + relabel + full evaluation with ONE wait: what compute_likelihood_4_nuisance (cuda_lib_gl.py:1986-2017) does between two
// MCMC steps -- memcpy_htod of the test parameters, the relabel of the layout the last step committed, evaluate_likelihood + sum -- as one
// call: the relabel and the evaluation's kernels go out back to back (the longest contig the launch shapes are sized with is one commit
// stale, as in graal_step), the statistics are collected behind the sums (published long before them)
int graal_eval_full_params(graal_ctx* h, const float* param8, int64_t stats[8], int32_t* max_id, int64_t q_out[2])
{
    if (!h || !q_out) return GRAAL_E_ARG;
    if (!h->have_frags) return fail(h, GRAAL_E_STATE, "no fragments uploaded");
    if (param8) { const int rc = graal_set_params(h, param8); if (rc) return rc; }
    if (!h->begin_launched) { const int rc = begin_step_launch(h); if (rc) return rc; }
    if (h->stats_pub_pending) {   // (graal_step deferred the publication to the scoring kernels and then handed the step back)
        k_stats_pub<<<1, 64, 0, h->stream>>>(h->d_part, h->apply_blocks, h->d_scalars, h->h_stats, h->stats_seq);
        CK(hipGetLastError());
        h->stats_pub_pending = false;
    }
    begin_step_assume(h);
    int rc = full_launch(h, h->stream);
    if (rc) { (void)begin_step_collect(h, stats, max_id); return rc; }
    rc = full_collect(h, h->stream, q_out);
    const int rc2 = begin_step_collect(h, stats, max_id);
    return rc ? rc : rc2;
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
    const DevArgs* A = h->d_args + h->cur;
    h->seq += 1;
    h->last_fA_launch = fA;
    // (1) tables + small mass work on the auxiliary stream: overlaps the scan.  A previous asynchronous evaluation must
    // have finished with the tables first (the synchronous path has waited for its results already).  In the synchronous
    // single-rank path (h->publish) its last block also finishes the step when the work is small.
    if (h->fin_pending) { CK(hipStreamWaitEvent(h->aux, h->ev_fin, 0)); h->fin_pending = false; }
    if (h->relabel_pending) { // graal_begin_step left its relabel kernels running on the engine's stream
        CK(hipStreamWaitEvent(h->aux, h->ev_relabel, 0));
        if (st != h->stream) CK(hipStreamWaitEvent(st, h->ev_relabel, 0));
        h->relabel_pending = false;
    }
    // graal_step's deferred flow: no event -- k_tm spins until this step's k_scan, behind the relabel on the engine's stream, starts
    const bool spin = h->relabel_spin_pending && st == h->stream;
    if (h->relabel_spin_pending && !spin) { CK(hipEventRecord(h->ev_relabel, h->stream)); CK(hipStreamWaitEvent(h->aux, h->ev_relabel, 0)); CK(hipStreamWaitEvent(st, h->ev_relabel, 0)); }
    h->relabel_spin_pending = false;
    const bool strict = (h->mode & GRAAL_MODE_STRICT) != 0;
    TmArgs ta;
    ta.strict = strict ? 1 : 0;
    ta.quirk = (h->mode & GRAAL_MODE_REF_TRANS_ACCU) ? 1 : 0;
    ta.nc = h->d_scalars + NC_WORD;

    ta.geo = h->geo; ta.link = h->link; ta.cbase = h->cbase; ta.mates = h->mates; ta.tabs = h->tabs; ta.step_hdr = h->step_hdr;
    ta.perm = h->perm; ta.reach_bp = reach_bp(h); ta.tm_done = h->tm_done;
    static const bool no_finisher = getenv("GRAAL_NO_FINISHER") != nullptr; // always finish with k_fin (diagnostics)
    ta.flags = h->d_flags;
    ta.sync = h->d_sync; ta.n_scan_blocks = scan_grid(h);
    ta.done = scan_done_counter() ? h->d_done : nullptr; ta.n_done = scan_done_n();
    for (int c = 0; c < N_DONE; c++) ta.done_target[c] = c < ta.n_done ? h->scan_done_total[c] + (unsigned long long)((scan_grid(h) - c + ta.n_done - 1) / ta.n_done) : 0ull;
    // (late stage -- a few long contigs hold nearly every fragment: nearly every step needs k_fin anyway, so it is launched
    // right behind the scan instead of after k_tm's verdict has made the round trip through the host, ~10 us per step)
    const bool late_stage = h->max_lcont > 128 && (long long)h->n_contigs * 64 < (long long)h->n;
    // (one rank only: in this flow k_tm prices no small sets itself, and the ranks would have to enter and leave it together -- a rank whose
    // finisher is off, or whose running mean differs, must not deal a small set's pairs one way while its peers deal them the other.  With
    // several ranks k_strict_flat goes out on k_tm's word instead: one host round trip later)
    const bool mid = flat_allowed(h, world) && world == 1 && h->mid_run && !late_stage && h->finisher_ok && !no_finisher;
    h->flat_tried = false;
    // (with k_strict_flat behind the scan k_tm prices nothing itself: one thread per pair walking the classes is 30-50 us for a set
    // of 20 fragments, and the flat kernel would wait for it)
    ta.strict_inline_m = strict_dense_cfg() ? -1 : (mid ? 0 : STRICT_INLINE_M);
    // (a long scan -- millions of contacts: the copy is over before the scan is.  A short one -- the C2 / C3 stand-ins -- is complete before the
    // tables are: the copy would stand in front of the contacts, two round trips for a handful of them; C2 stand-in 130 -> 137 us per step)
    ta.stage_tables = h->nnz >= 4000000 ? 1 : 0;
    ta.host_res = (h->publish && (world == 1 || h->x_host) && !no_finisher && h->finisher_ok && !h->has_rep && !late_stage && !mid && !(strict && strict_dense_cfg())) ? h->res_dev : nullptr;
    ta.wait_ticks = fin_wait_ticks(h);
    ta.counters = (unsigned long long*)(h->d_scalars + 10); ta.queue = h->queue; ta.cnt = h->cnt; ta.multi = h->single_sub ? 0 : 1; ta.stat = h->stat_frag;
    ta.lcontbp = h->soa[h->cur].p[F_LCONTBP]; ta.acc = h->d_acc; ta.nfpb = h->nfpb; ta.par = h->par;
    ta.relabel_flag = (const unsigned long long*)(h->d_scalars + RELABEL_FLAG);
    ta.relabel_seq = spin ? ++h->relabel_flag_seq : 0ull;
    ta.relabel_wait_ticks = h->tm_spin_ticks;
    h->scan_relabel_seq = ta.relabel_seq;
    h->spin_used = spin;
    const bool pub = h->stats_pub_pending;
    h->stats_pub_pending = false;
    h->pub_in_flight = pub;
    ta.part = pub ? h->d_part : nullptr; ta.n_part = h->apply_blocks; ta.stats_host = h->h_stats; ta.stats_seq = h->stats_seq; ta.stats_dev = h->d_scalars;
    k_tm<<<K + (pub ? 1 : 0), 256, ta.host_res ? tm_fin_dyn_lds() : 0, h->aux>>>(A, ta, fA, nb, K, max_id, rank, world, h->seq);
    CK(hipGetLastError());
    // (the event that orders a chip-filling k_fin / the strict kernels behind k_tm is recorded where they are launched)
    // (2) the streaming pass, with a HIP event pair around it on every event_every-th call
    const bool ev = h->want_events && (h->eval_calls % h->event_every == 0);
    h->ev_this_call = ev;
    h->eval_calls += 1;
    const size_t slot = (size_t)(h->ring_calls % (long long)(h->ring.size() / 2));
    if (ev) CK(hipEventRecord(h->ring[2 * slot], st));
    { int rc_ = launch_scan(h, fA, nb, K, max_id, 0, st, ta.host_res != nullptr); if (rc_) return rc_; }
    if (ev) { CK(hipEventRecord(h->ring[2 * slot + 1], st)); h->ring_calls += 1; }
    // (3) finishing kernel -- unless k_tm's last block does that job (it asks for k_fin through the result word if not)
    if (h->has_rep) { // the repeated bins' pixels, densely, for all 13 K candidates
        // (its blocks wait for k_tm's tables too, and with 167 VGPRs three of them fill a CU's register files: ordered behind
        // k_tm by the event rather than trusted to be placed after it -- see fin_blocks_no_wait)
        CK(hipEventRecord(h->ev_tm, h->aux));
        CK(hipStreamWaitEvent(st, h->ev_tm, 0));
        const RepArgs R = rep_args(h);
        k_rep_delta<<<blocks_for((long long)h->n_dup * h->n_bins * K, 256), 256, 0, st>>>(R, h->tabs, h->tm_done, h->seq, fA, K, rank, world,
                                                                                         h->d_acc, (unsigned long long*)(h->d_scalars + 10));
        CK(hipGetLastError());
    }
    if (strict) {
        if (ta.host_res == nullptr) {
            int rc_ = 0;
            // (several ranks: whoever finishes a step that is not in the late stage -- this rank's finisher may be off while its peers' are on --
            // goes through k_strict_flat first, like a rank that k_tm sent there: the flat and the tiled kernels deal the pairs differently)
            if (mid || (world > 1 && !late_stage && flat_allowed(h, world))) { rc_ = launch_flat(h, fA, &nb, K, rank, world, (long long*)d_q_out, h->publish, st); h->flat_tried = true; }
            else rc_ = launch_strict(h, fA, K, rank, world, (long long*)d_q_out, h->publish, st);
            if (rc_) return rc_;
            if (!h->publish) { CK(hipEventRecord(h->ev_fin, st)); h->fin_pending = true; }
        }
    } else if (ta.host_res == nullptr) {
        int rc_ = launch_fin(h, K, rank, world, (long long*)d_q_out, h->publish, st);
        if (rc_) return rc_;
        if (!h->publish) { CK(hipEventRecord(h->ev_fin, st)); h->fin_pending = true; }
    }
    h->timing_valid = h->want_events;
    h->scan_ready = true;
    h->last_fA = fA; h->last_K = K; h->last_max_id = max_id;
    for (int k = 0; k < MAXK; k++) h->last_fB[k] = nb.fB[k];
    return GRAAL_OK;
}

// behind the all-reduce: the summed device buffer (hand_out's layout: Q sums, coarse sums, not-finite flags) to pinned host memory in the
// layout the single-rank step publishes, the sequence word last
__global__ void k_qout_pub(const long long* __restrict__ q, volatile long long* host, int K, long long seq)
{
    const bool failed = q[2 * MAXK * N_OPS] >= Q_STEP_FAILED;   // (some rank's step failed: every rank reads the same sum and publishes -seq)
    for (int i = threadIdx.x; i < K * N_OPS; i += blockDim.x) {
        host[1 + i] = (q[2 * MAXK * N_OPS + i] & (Q_STEP_FAILED - 1)) != 0 ? Q_NAN : q[i];
        host[X_COARSE + i] = q[MAXK * N_OPS + i];
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { host[0] = failed ? -seq : seq; __threadfence_system(); }
}

// synchronous evaluation: launch the step, spin on the sequence word its last block writes into pinned host memory (fall
// back to a stream synchronise if it does not show up -- it always does unless the launch failed); q_sum[K*13] = the Q sums
// of this rank, plus -- with an exchange attached -- those the other ranks of the node published for the same step
static int eval_sync(graal_ctx* h, int32_t fA, const int32_t* fB, int32_t K, int32_t max_id, int rank, int world, long long* q_sum, long long* c_sum)
{
    const long long want = h->seq + 1;
    if (h->nccl_comm) {
        // the north star's exchange: every rank's finishing kernel leaves its sums in a device buffer, ONE ncclAllReduce over xGMI on the
        // engine's stream sums them, a last tiny kernel publishes the total to this rank's pinned host memory -- all on the GPU timeline, the
        // host waits once.  (k_tm's last block cannot finish such a step: the finishing kernel always runs; k_strict_flat -- which may hand
        // the step back through the host -- is not used.)
        Rccl* R = rccl_load(&h->err);
        if (!R) return GRAAL_E_STATE;
        h->res_host = h->res_dev = h->h_res;
        h->publish = false;
        int rc = graal_eval_candidates_q(h, fA, fB, K, max_id, rank, world, (int64_t*)h->d_qout, nullptr);
        if (rc) return rc;
        const int nrc = R->AllReduce(h->d_qout, h->d_qout, (size_t)3 * MAXK * N_OPS, NCCL_INT64, NCCL_SUM, h->nccl_comm, h->stream);
        if (nrc != 0) { h->err = std::string("ncclAllReduce failed: ") + (R->GetErrorString ? R->GetErrorString(nrc) : "?"); return GRAAL_E_HIP; }
        k_qout_pub<<<1, 256, 0, h->stream>>>(h->d_qout, h->h_res, K, want);
        CK(hipGetLastError());
        volatile long long* res = h->h_res;
        bool seen = false;
        for (long long spin = 0; spin < 2000000000ll; spin++) {
            if (res[0] == want || res[0] == -want) { seen = true; break; }
            if ((spin & 0xfffff) == 0xfffff && hipStreamQuery(h->stream) != hipErrorNotReady) { seen = res[0] == want || res[0] == -want; break; }
            __builtin_ia32_pause();
        }
        if (!seen) {   // (bounded: a collective that a peer never enters must not hang this rank for ever)
            const auto t0 = std::chrono::steady_clock::now();
            while (res[0] != want && res[0] != -want && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(60)) __builtin_ia32_pause();
            if (res[0] != want && res[0] != -want) return fail(h, GRAAL_E_HIP, "RCCL exchange: the all-reduced sums were not published within 60 s (a rank out of step, or the collective hangs)");
        }
        // A failed step on ANY rank (hand_out: Q_STEP_FAILED travels through the all-reduce) is an error on EVERY rank, the same one, at the same
        // step: no rank returns partial sums as scores, none repeats alone.  (With a communicator attached the engine uses no in-kernel waits --
        // graal_attach_rccl switches them off -- so a failure here is a kernel that did not run, not a wait that a repeat behind events would cure.)
        if (res[0] == -want) return fail(h, GRAAL_E_HIP, "RCCL exchange: the step failed on some rank (a kernel of the step gave up or a list overflowed); no scores were returned on any rank");
        __sync_synchronize();
        for (int i = 0; i < K * N_OPS; i++) { q_sum[i] = res[1 + i]; c_sum[i] = res[X_COARSE + i]; }
        return GRAAL_OK;
    }
    if (world > 1) { // this rank's slot of the step's parity (two steps later the slot is reused: every rank has read it by then,
                     // because nobody finishes step s+1 before everybody has published it, i.e. has finished reading step s)
        const size_t off = ((size_t)(want & 1) * (size_t)world + (size_t)rank) * X_SLOT_WORDS;
        h->res_host = h->x_host + off;
        h->res_dev = h->x_dev + off;
    } else { h->res_host = h->res_dev = h->h_res; }
    auto now_us = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = h->eval_timing ? now_us() : 0.0;
    double t1 = 0.0, t2 = 0.0, t3 = 0.0;
    h->publish = true;
    h->step_needed_fin = false;
    h->step_needed_geom = false;
    h->rc_evals += 1;
    int rc = graal_eval_candidates_q(h, fA, fB, K, max_id, rank, world, (int64_t*)h->d_qout, nullptr);
    h->publish = false;
    if (rc) return rc;
    const bool launched_mid = h->flat_tried;   // (k_strict_flat went out with the step: it reports whether it was needed)
    if (h->eval_timing) t1 = now_us();
    volatile long long* res = h->res_host;
    bool seen = false;
    for (long long spin = 0; spin < 200000000ll; spin++) {
        const long long v = res[0];
        if (v == want || v == -want) { seen = true; break; }
        if ((v & ~(GAVE_UP | NEED_GEOM)) == (want | NEED_FIN)) { // k_tm left (the heavy part of) the step to k_fin
            if ((v & GAVE_UP) && ++h->gave_up >= 3) h->finisher_ok = false;
            if (v & NEED_GEOM) h->step_needed_geom = true;
            res[0] = 0;
            if (h->eval_timing) t2 = now_us();
            h->publish = true;
            h->step_needed_fin = true;
            h->rc_need_fin += 1;
            if ((h->mode & GRAAL_MODE_STRICT) && flat_allowed(h, world) && !h->flat_tried) { // small sets first; if they are not, it says NEED_FIN again
                h->flat_tried = true;
                rc = launch_flat(h, fA, nullptr, K, rank, world, (long long*)h->d_qout, true, h->stream);
            } else
                rc = (h->mode & GRAAL_MODE_STRICT) ? launch_strict(h, fA, K, rank, world, (long long*)h->d_qout, true, h->stream)
                                                   : launch_fin(h, K, rank, world, (long long*)h->d_qout, true, h->stream);
            h->publish = false;
            if (rc) return rc;
            if (h->eval_timing) t3 = now_us();
            continue;
        }
        if ((spin & 0xfffff) == 0xfffff && hipStreamQuery(h->stream) != hipErrorNotReady && hipStreamQuery(h->aux) != hipErrorNotReady) {
            seen = (res[0] == want || res[0] == -want);
            if (!seen && (res[0] & ~(GAVE_UP | NEED_GEOM)) == (want | NEED_FIN)) continue;
            break;
        }
        __builtin_ia32_pause();
    }
    if (!seen) {
        CK(hipStreamSynchronize(h->stream));
        CK(hipStreamSynchronize(h->aux));
        if (res[0] != want && res[0] != -want) return fail(h, GRAAL_E_HIP, "the step's last block did not publish its results");
    }
    if ((h->mode & GRAAL_MODE_STRICT) && (res[0] == want)) {
        // running mean of "k_tm alone would not have finished this step": above 1/2 the flat kernel goes out with every step and
        // k_tm does not wait for the scan; below 1/4 k_tm finishes the steps by itself again.  With several ranks only what the LAYOUT says
        // counts (a set beyond k_tm's own pricing), not the rank's own queue: the ranks must change flows together -- in one flow k_tm prices
        // the small sets (dealt to the ranks its way), in the other k_strict_flat does (dealt its way)
        const bool needed = world > 1 ? (launched_mid ? (res[1 + MAXK * N_OPS] & 2) != 0 : h->step_needed_geom)
                                      : (launched_mid ? ((res[1 + MAXK * N_OPS] & 1) != 0 || h->step_needed_fin) : h->step_needed_fin);
        h->need_ema = 0.9 * h->need_ema + (needed ? 0.1 : 0.0);
        if (!h->mid_run && h->need_ema > 0.5) h->mid_run = true;
        else if (h->mid_run && h->need_ema < 0.25) h->mid_run = false;
    }
    if (h->eval_timing) {
        const double t4 = now_us();
        h->et_n[0] += 1; h->et[0] += t1 - t0;
        if (t2 > 0.0) { h->et_n[1] += 1; h->et[1] += t2 - t1; h->et[2] += t3 - t2; h->et[3] += t4 - t3; h->et[5] += t4 - t0; }
        else h->et[4] += t4 - t1;
    }
    const unsigned long long why = res[0] == -want ? (unsigned long long)res[X_WHY] : 0ull;
    // (a step's unit list overflowed its soft cap -- launch_strict -- and can still grow: one rank only, see there)
    const bool grow = (why & 2ull) != 0ull && world == 1 && h->slist_cap < h->slist_worst;
    const bool waits = res[0] == -want && h->spin_used && ((why & ~2ull) != 0ull || !grow);
    if (res[0] == -want && (waits || grow)) {
        // k_tm gave up waiting for k_scan's announcement (or k_strict2 for k_gprep's word): the two kernels did not run side by side (a tool
        // serialising dispatches).  Everything of the step has ended by now; put the step's accumulators back to rest and repeat it ordered
        // by the host -- and stay with events from here on.  Or the step's unit list was too short: repeat with a longer one.
        if (waits) {
            h->spin_ok = false;
            h->rc_repeats += 1;
            fprintf(stderr, "graal: kernels of the engine's two streams did not run side by side (a tool that serialises dispatches?): step repeated, "
                            "the steps are ordered through events from here on\n");
        }
        h->spin_used = false;
        if (grow) { h->slist_floor = std::min(h->slist_worst, std::max<unsigned long long>(h->slist_cap, 64ull) * 8ull); h->rc_list_grown += 1; }
        if (h->pub_in_flight) h->stats_pub_pending = true;   // (its publication block gave up with the others)
        CK(hipStreamSynchronize(h->stream));
        CK(hipStreamSynchronize(h->aux));
        CK(hipMemset(h->d_scalars + 10, 0, 3 * sizeof(long long)));   // step counters [0..2]; NOT words 13 / 14 (layout statistics: stale flag, circular contigs)
        CK(hipMemset(h->d_scalars + 15, 0, 2 * sizeof(long long)));   // ticket, error
        CK(hipMemset(h->d_scalars + 10 + NF_OFF, 0, 3 * sizeof(long long)));
        CK(hipMemset(h->d_scalars + SLIST_N, 0, sizeof(long long)));
        CK(hipMemset(h->d_acc, 0, 2 * MAXK * N_OPS * sizeof(long long)));
        CK(hipMemset(h->d_sync, 0, 32 * sizeof(unsigned long long)));
        CK(hipDeviceSynchronize());
        res[0] = 0;
        return eval_sync(h, fA, fB, K, max_id, rank, world, q_sum, c_sum);
    }
    if (res[0] == -want) return fail(h, GRAAL_E_HIP, "timed out waiting for the candidate tables / the scan (a kernel of the step did not run)");
    __sync_synchronize();
    for (int i = 0; i < K * N_OPS; i++) { q_sum[i] = res[1 + i]; c_sum[i] = res[X_COARSE + i]; }
    // the other ranks' slots: their GPUs write them, this host reads them (coherent host memory; 8-byte words, the
    // sequence word last).  A rank that died leaves its word behind: give up after ~60 s instead of spinning for ever.
    for (int r = 0; r < world; r++) {
        if (r == rank) continue;
        volatile long long* o = h->x_host + ((size_t)(want & 1) * (size_t)world + (size_t)r) * X_SLOT_WORDS;
        const auto t0 = std::chrono::steady_clock::now();
        long long spin = 0;
        bool peer_failed = false;
        for (;;) {
            const long long v = o[0];
            if (v == want) break;
            if (v == -want) { peer_failed = true; break; }
            if ((++spin & 0xfffff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60))
                return fail(h, GRAAL_E_STATE, "exchange: another rank did not publish this step within 60 s (ranks out of step, or a rank died)");
            __builtin_ia32_pause();
        }
        if (peer_failed) {
            // That rank's step ended as failed -- an in-kernel wait of its engine ran out (its two kernels did not run side by side: several
            // ranks on ONE GPU in a rehearsal, a tool that serialises dispatches, a hiccup) -- and it repeats the step behind events under the
            // NEXT step number.  Every rank must repeat with it, or the ranks are out of step: its -want stays in this parity's slot until
            // the step after next, so every peer sees it, whenever it looks.  (This rank's own part ended regularly: nothing to put back.)
            // A rank that fails for good returns an error and never publishes the repeat: its peers time out above.
            if (h->peer_repeats >= 3) return fail(h, GRAAL_E_HIP, "another rank's step failed again and again (exchange)");
            h->peer_repeats += 1;
            h->rc_repeats += 1;
            const int rc2 = eval_sync(h, fA, fB, K, max_id, rank, world, q_sum, c_sum);
            h->peer_repeats -= 1;
            return rc2;
        }
        __sync_synchronize();
        for (int i = 0; i < K * N_OPS; i++) {
            const long long oq = o[1 + i];
            q_sum[i] = (q_sum[i] == Q_NAN || oq == Q_NAN) ? Q_NAN : (long long)((unsigned long long)q_sum[i] + (unsigned long long)oq);   // (the marker is sticky)
            c_sum[i] += o[X_COARSE + i];
        }
    }
    return GRAAL_OK;
}

int graal_eval_candidates(graal_ctx* h, int32_t fA, const int32_t* fB, int32_t K, int32_t max_id, double* delta)
{
    if (!h || !delta) return GRAAL_E_ARG;
    long long q[MAXK * N_OPS], c[MAXK * N_OPS];
    const int rc = eval_sync(h, fA, fB, K, max_id, h->nccl_comm ? h->n_rank : 0, h->nccl_comm ? h->n_world : 1, q, c);   // (RCCL attached: the ranks' sum)
    if (rc) return rc;
    for (int i = 0; i < K * N_OPS; i++) delta[i] = q_value(q[i], c[i]);
    return GRAAL_OK;
}

int graal_exchange_bytes(int32_t world, int64_t* bytes)
{
    if (!bytes || world < 1) return GRAAL_E_ARG;
    *bytes = (int64_t)(2 * (size_t)world * X_SLOT_WORDS * sizeof(long long));
    return GRAAL_OK;
}

int graal_attach_exchange(graal_ctx* h, void* segment, int64_t bytes, int32_t rank, int32_t world, int64_t seq_floor, int64_t* seq_now)
{
    if (!h) return GRAAL_E_ARG;
    if (seq_now) *seq_now = h->seq;
    if (!segment) return GRAAL_OK; // query only
    if (world < 2 || rank < 0 || rank >= world) return fail(h, GRAAL_E_ARG, "exchange: bad rank / world");
    if (bytes < (int64_t)(2 * (size_t)world * X_SLOT_WORDS * sizeof(long long))) return fail(h, GRAAL_E_ARG, "exchange: segment too small (graal_exchange_bytes)");
    if (h->x_host) return fail(h, GRAAL_E_STATE, "exchange: already attached");
    CK(hipSetDevice(h->device));
    CK(hipStreamSynchronize(h->stream));
    CK(hipStreamSynchronize(h->aux));
    CK(hipStreamSynchronize(h->fstream));
    CK(hipHostRegister(segment, (size_t)bytes, hipHostRegisterMapped | hipHostRegisterPortable));
    void* dp = nullptr;
    hipError_t e = hipHostGetDevicePointer(&dp, segment, 0);
    if (e != hipSuccess) { (void)hipHostUnregister(segment); h->err = std::string("hipHostGetDevicePointer failed: ") + hipGetErrorString(e); return GRAAL_E_HIP; }
    h->x_host = (long long*)segment; h->x_dev = (long long*)dp; h->x_bytes = (size_t)bytes; h->x_rank = rank; h->x_world = world;
    // the published word is the step's sequence number: all ranks continue from the same one
    if (seq_floor > h->seq) h->seq = seq_floor;
    if (seq_now) *seq_now = h->seq;
    debug_print_buffers(h, "attach_exchange");
    return GRAAL_OK;
}

int graal_exchange_selftest(graal_ctx* h, int64_t tag, int32_t phase)
{
    if (!h) return GRAAL_E_ARG;
    if (!h->x_host) return fail(h, GRAAL_E_STATE, "graal_attach_exchange first");
    const size_t w = (size_t)h->x_world;
    if (phase == 0) { // write: this rank's GPU tags its two slots
        CK(hipSetDevice(h->device));
        k_exchange_tag<<<1, 64, 0, h->stream>>>(h->x_dev + (size_t)h->x_rank * X_SLOT_WORDS, h->x_dev + (w + (size_t)h->x_rank) * X_SLOT_WORDS,
                                                 X_SLOT_WORDS - 1, (long long)tag + h->x_rank);
        CK(hipGetLastError());
        CK(hipStreamSynchronize(h->stream));
        return GRAAL_OK;
    }
    // check (after a barrier of the ranks): every rank's tag must be visible to this host in both slots
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        bool all = true;
        for (size_t r = 0; r < w; r++)
            for (size_t par = 0; par < 2; par++)
                all = all && ((volatile long long*)h->x_host)[(par * w + r) * X_SLOT_WORDS + (X_SLOT_WORDS - 1)] == (long long)tag + (long long)r;
        if (all) return GRAAL_OK;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2))
            return fail(h, GRAAL_E_STATE, "exchange self-test: another rank's GPU write is not visible in the shared segment");
        __builtin_ia32_pause();
    }
}

int graal_detach_exchange(graal_ctx* h)
{
    if (!h) return GRAAL_E_ARG;
    if (!h->x_host) return GRAAL_OK;
    CK(hipSetDevice(h->device));
    CK(hipStreamSynchronize(h->stream));
    CK(hipStreamSynchronize(h->aux));
    CK(hipStreamSynchronize(h->fstream));   // (the in-step full evaluation publishes the ranks' contact parts into the segment from ITS stream: one still
                                            // in flight at the unregistration was a GPU memory fault at the segment's host address -- seen once, two ranks)
    CK(hipDeviceSynchronize());             // (and a caller's stream that graal_eval_candidates_q was given)
    CK(hipHostUnregister(h->x_host));
    h->x_host = nullptr; h->x_dev = nullptr; h->x_bytes = 0; h->x_rank = 0; h->x_world = 1;
    h->res_host = h->res_dev = h->h_res;
    return GRAAL_OK;
}

int graal_rccl_unique_id(void* id128)
{
    if (!id128) return GRAAL_E_ARG;
    Rccl* R = rccl_load(nullptr);
    if (!R) return GRAAL_E_STATE;
    return R->GetUniqueId((NcclId*)id128) == 0 ? GRAAL_OK : GRAAL_E_HIP;
}

int graal_attach_rccl(graal_ctx* h, const void* id128, int32_t rank, int32_t world)
{
    if (!h || !id128 || world < 1 || rank < 0 || rank >= world) return GRAAL_E_ARG;
    if (h->nccl_comm) return fail(h, GRAAL_E_STATE, "RCCL: already attached");
    if (h->x_host) return fail(h, GRAAL_E_STATE, "RCCL: the host exchange is attached (one exchange at a time)");
    Rccl* R = rccl_load(&h->err);
    if (!R) return GRAAL_E_STATE;
    CK(hipSetDevice(h->device));
    CK(hipStreamSynchronize(h->stream));
    CK(hipStreamSynchronize(h->aux));
    NcclId id;
    memcpy(&id, id128, sizeof id);
    void* comm = nullptr;
    const int nrc = R->CommInitRank(&comm, world, id, rank);
    if (nrc != 0 || !comm) { h->err = std::string("ncclCommInitRank failed: ") + (R->GetErrorString ? R->GetErrorString(nrc) : "?"); return GRAAL_E_HIP; }
    h->nccl_comm = comm; h->n_rank = rank; h->n_world = world;
    // no in-kernel waits between the kernels of a step while a communicator is attached (k_tm behind the relabel through the scan's flag, the
    // deferred statistics, k_strict2 behind k_gprep's word): their remedy -- one rank repeating its step behind events -- would leave the ranks
    // out of step inside a collective.  Everything is ordered by events.
    h->spin_ok_saved = h->spin_ok;
    h->spin_ok = false;
    h->relabel_spin_pending = false;
    return GRAAL_OK;
}

int graal_detach_rccl(graal_ctx* h)
{
    if (!h) return GRAAL_E_ARG;
    if (!h->nccl_comm) return GRAAL_OK;
    CK(hipSetDevice(h->device));
    CK(hipStreamSynchronize(h->stream));
    CK(hipStreamSynchronize(h->aux));
    Rccl* R = rccl_load(nullptr);
    if (R) (void)R->CommDestroy(h->nccl_comm);
    h->nccl_comm = nullptr; h->n_rank = 0; h->n_world = 1;
    h->spin_ok = h->spin_ok_saved;
    return GRAAL_OK;
}

int graal_eval_candidates_x(graal_ctx* h, int32_t fA, const int32_t* fB, int32_t K, int32_t max_id, int64_t* q_sum, int64_t* c_sum)
{
    if (!h || !q_sum || !c_sum) return GRAAL_E_ARG;
    if (!h->x_host && !h->nccl_comm) return fail(h, GRAAL_E_STATE, "graal_attach_exchange or graal_attach_rccl first");
    long long q[MAXK * N_OPS], c[MAXK * N_OPS];
    const int rc = eval_sync(h, fA, fB, K, max_id, h->nccl_comm ? h->n_rank : h->x_rank, h->nccl_comm ? h->n_world : h->x_world, q, c);
    if (rc) return rc;
    for (int i = 0; i < K * N_OPS; i++) { q_sum[i] = q[i]; c_sum[i] = c[i]; }
    return GRAAL_OK;
}

int graal_upload_distance_ref(graal_ctx* h, const int32_t* init_prev, const int32_t* init_next, const int32_t* init_ori,
                              const int32_t* orientable, const uint8_t* counted, int32_t n)
{
    if (!h || !init_prev || !init_next || !init_ori || !orientable || !counted) return GRAAL_E_ARG;
    if (!h->have_frags || n != h->n) return fail(h, GRAAL_E_STATE, "upload the fragments first (same n)");
    CK(hipSetDevice(h->device));
    std::vector<int4> ref((size_t)n);
    for (int i = 0; i < n; i++) ref[(size_t)i] = make_int4(init_prev[i], init_next[i], init_ori[i], (orientable[i] != 0 ? 1 : 0) | (counted[i] ? 2 : 0));
    if (h->d_dref) { CK(hipFree(h->d_dref)); h->d_dref = nullptr; }
    CK(hipMalloc(&h->d_dref, (size_t)n * sizeof(int4)));
    CK(hipMemcpy(h->d_dref, ref.data(), (size_t)n * sizeof(int4), hipMemcpyHostToDevice));
    return GRAAL_OK;
}

int graal_genome_distance(graal_ctx* h, int64_t* half_units)
{
    if (!h || !half_units) return GRAAL_E_ARG;
    if (!h->d_dref) return fail(h, GRAAL_E_STATE, "graal_upload_distance_ref first");
    CK(hipSetDevice(h->device));
    h->dist_seq += 1;
    k_dist<<<blocks_for(h->n, 256), 256, 0, h->stream>>>(h->soa[h->cur], h->n, h->d_dref, h->d_dist, h->h_dist, h->dist_seq);
    CK(hipGetLastError());
    volatile long long* p = h->h_dist;
    bool seen = false;
    for (long long spin = 0; spin < 400000000ll; spin++) {
        if (p[0] == h->dist_seq) { seen = true; break; }
        if ((spin & 0xfffff) == 0xfffff && hipStreamQuery(h->stream) != hipErrorNotReady) { seen = (p[0] == h->dist_seq); break; }
        __builtin_ia32_pause();
    }
    if (!seen) {
        CK(hipStreamSynchronize(h->stream));
        if (p[0] != h->dist_seq) return fail(h, GRAAL_E_HIP, "k_dist did not publish the genome distance");
    }
    __sync_synchronize();
    *half_units = p[1];
    return GRAAL_OK;
}

int graal_set_mode(graal_ctx* h, int32_t flags)
{
    if (!h || (flags & ~(GRAAL_MODE_REF_TRANS_ACCU | GRAAL_MODE_STRICT))) return GRAAL_E_ARG;
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->mode = flags;
    return GRAAL_OK;
}

int graal_set_finisher(graal_ctx* h, int32_t enabled)
{
    if (!h) return GRAAL_E_ARG;
    h->finisher_ok = enabled != 0;
    h->gave_up = 0;
    return GRAAL_OK;
}

/* per-kernel HIP events (graal_last_timing) cost ~2 us each on the host: a driver may switch them off */
int graal_set_timing(graal_ctx* h, int32_t enabled)
{
    if (!h) return GRAAL_E_ARG;
    h->want_events = enabled != 0;
    if (enabled > 0) h->event_every = enabled; // an event pair around k_scan on every enabled-th evaluation
    if (!h->want_events) h->timing_valid = false;
    return GRAAL_OK;
}

int graal_apply_move(graal_ctx* h, int32_t fA, int32_t fB, int32_t op, int32_t max_id, int32_t* n_stale)
{
    if (!h || op < 0 || op >= N_OPS) return GRAAL_E_ARG;
    if (!h->have_frags) return fail(h, GRAAL_E_STATE, "no fragments uploaded");
    if (fA < 0 || fA >= h->n || fB < 0 || fB >= h->n) return fail(h, GRAAL_E_ARG, "fragment index out of range");
    CK(hipSetDevice(h->device));
    int* d_stale = (int*)(h->d_scalars + 13);
    // (the commit record d_chg is clear: the last kernel of every relabel clears it; a second commit without a relabel in
    // between finds the first one's entries, and the relabel then sorts instead of counting)
    h->stats_seq += 1;
    // (one fragment per thread up to 256 blocks: the kernel is a dependent chain -- records of fA / fB, the fragment's own 13
    // words, the stores -- and no longer pays atomics per block, so more, shorter blocks win: 12.4 -> ~6 us at 50k fragments)
    static const int apply_blocks_env = getenv("GRAAL_APPLY_BLOCKS") ? atoi(getenv("GRAAL_APPLY_BLOCKS")) : 256;
    h->apply_blocks = std::min(blocks_for(h->n, 256), std::max(1, std::min(apply_blocks_env, 1024)));
    // (several ranks: a rank's own table covers its shard of the list only -- unless the caller handed in the whole one, graal_upload_own_obs:
    // then every rank computes the same correction and nothing has to be exchanged; with an RCCL communicator of several ranks a repair could
    // not be exchanged inside the step: not offered)
    const bool own_on = !h->single_sub && !h->has_rep && h->d_own_obs != nullptr && h->stat_frag != nullptr && h->have_par &&
                        (h->own_complete || !h->x_host) && !(h->nccl_comm && h->n_world > 1) && h->carry_wanted;
    // (the layout the last correction kernel reads is the buffer THIS commit writes: it has published long ago -- the host collected it with the
    // step's statistics -- unless the caller commits without stepping; then wait here)
    if (h->own_pending) { const int rc = wait_own(h, nullptr, nullptr); if (rc) return rc; }
    k_apply<<<h->apply_blocks, 256, 0, h->stream>>>(h->soa[h->cur], h->soa[1 - h->cur], h->n, op, fA, fB, max_id, d_stale, h->d_chg + h->chg_w, h->d_part);
    if (own_on) {
        // (on the full evaluation's stream -- idle but for the rare resyncs --, next to the commit: it reads what the commit reads, and the host has
        // seen this step's scores, so everything that wrote that layout is complete)
        h->own_seq += 1;
        k_own_corr<<<std::min(blocks_for(h->n * 8, 256), 1024), 256, 0, h->fstream>>>(h->soa[h->cur], h->n, op, fA, fB, max_id, h->stat_frag, h->d_own_obs, h->nfpb, h->par,
                                                                                     (h->mode & GRAAL_MODE_REF_TRANS_ACCU) && !h->all_uniform ? 1 : 0,
                                                                                     h->d_own_acc, h->h_own, h->own_seq);
        h->own_pending = true;
    }
    h->apply_had_own = own_on;
    h->chg_last = h->chg_w; h->chg_w ^= 1;
    CK(hipGetLastError());
    h->begin_launched = false;
    h->corr_skip = false;
    h->stats_from_apply = true; // statistics of the new layout are on their way to pinned host memory (sequence stats_seq)
    h->incr_ok = h->ranks_valid && h->pending_commits == 0 && max_id == h->n_contigs - 1;
    h->pending_commits += 1;
    h->cur = 1 - h->cur;
    h->order_valid = false; // graal_begin_step (relabel) rebuilds the index and the geometry records
    if (n_stale) { // asking for the count costs a synchronisation; pass NULL and read it from graal_begin_step's stats[7]
        int stale = 0;
        CK(hipMemcpyAsync(&stale, d_stale, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        CK(hipStreamSynchronize(h->stream));
        *n_stale = stale;
    }
    return GRAAL_OK;
}

int graal_last_timing(graal_ctx* h, float out[4])
{
    if (!h || !out) return GRAAL_E_ARG;
    out[0] = out[1] = out[2] = out[3] = 0.0f;
    return graal_scan_times(h, 1, &out[1]);
}

int graal_scan_times(graal_ctx* h, int32_t n, float* out_ms)
{
    if (!h || !out_ms || n < 1) return GRAAL_E_ARG;
    const long long cap = (long long)(h->ring.size() / 2);
    if (!h->timing_valid || h->ring_calls < n || n > cap) return fail(h, GRAAL_E_STATE, "not enough timed candidate evaluations");
    CK(hipSetDevice(h->device));
    for (int i = 0; i < n; i++) {
        const size_t slot = (size_t)((h->ring_calls - n + i) % cap);
        CK(hipEventSynchronize(h->ring[2 * slot + 1]));
        CK(hipEventElapsedTime(&out_ms[i], h->ring[2 * slot], h->ring[2 * slot + 1]));
    }
    return GRAAL_OK;
}

int graal_strict_times(graal_ctx* h, int32_t n, float* out_ms)
{
    if (!h || !out_ms || n < 1) return GRAAL_E_ARG;
    const long long cap = (long long)(h->sring.size() / 2);
    if (h->sring_calls < n || n > cap) return fail(h, GRAAL_E_STATE, "not enough timed launches of the tiled reference-arithmetic kernel");
    CK(hipSetDevice(h->device));
    for (int i = 0; i < n; i++) {
        const size_t slot = (size_t)((h->sring_calls - n + i) % cap);
        CK(hipEventSynchronize(h->sring[2 * slot + 1]));
        CK(hipEventElapsedTime(&out_ms[i], h->sring[2 * slot], h->sring[2 * slot + 1]));
    }
    return GRAAL_OK;
}

int graal_time_scan(graal_ctx* h, int32_t K, int32_t reps, float* avg_ms)
{
    if (h && avg_ms && reps < 0 && K >= 1 && K <= MAXK) {   // ISOLATED replays: an event pair around each launch, the device idle in between
        if (!h->scan_ready) return fail(h, GRAAL_E_STATE, "run graal_eval_candidates first (the replays reuse its tables)");
        CK(hipSetDevice(h->device));
        Neigh nb;
        for (int k = 0; k < MAXK; k++) nb.fB[k] = h->last_fB[k];
        std::vector<float> t;
        for (int i = 0; i < -reps + 3; i++) {
            CK(hipStreamSynchronize(h->stream));
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(50)) { }
            CK(hipEventRecord(h->ev[0], h->stream));
            { int rc = launch_scan(h, h->last_fA, nb, K, h->last_max_id, 1, h->stream); if (rc) return rc; }
            CK(hipEventRecord(h->ev[4], h->stream));
            CK(hipEventSynchronize(h->ev[4]));
            float ms = 0.0f;
            CK(hipEventElapsedTime(&ms, h->ev[0], h->ev[4]));
            if (i >= 3) t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        *avg_ms = t[t.size() / 2];   // median
        return GRAAL_OK;
    }
    if (!h || !avg_ms || reps < 1 || K < 1 || K > MAXK) return GRAAL_E_ARG;
    if (!h->scan_ready) return fail(h, GRAAL_E_STATE, "run graal_eval_candidates first (the replays reuse its tables)");
    CK(hipSetDevice(h->device));
    if (K != h->last_K) return fail(h, GRAAL_E_ARG, "K differs from the last evaluation");
    Neigh nb;
    for (int k = 0; k < MAXK; k++) nb.fB[k] = h->last_fB[k];
    for (int i = 0; i < 3; i++) { int rc = launch_scan(h, h->last_fA, nb, K, h->last_max_id, 1, h->stream); if (rc) return rc; }
    CK(hipEventRecord(h->ev[0], h->stream));
    for (int i = 0; i < reps; i++) { int rc = launch_scan(h, h->last_fA, nb, K, h->last_max_id, 1, h->stream); if (rc) return rc; }
    CK(hipEventRecord(h->ev[4], h->stream));
    CK(hipEventSynchronize(h->ev[4]));
    float ms = 0.0f;
    CK(hipEventElapsedTime(&ms, h->ev[0], h->ev[4]));
    *avg_ms = ms / (float)reps;
    return GRAAL_OK;
}

/* debug builds only (-DGRAAL_STAMPS): 32 wall-clock stamps (100 MHz) of the last step's kernels */
int graal_debug_stamps(graal_ctx* h, uint64_t out[32])
{
    if (!h || !out) return GRAAL_E_ARG;
#ifdef GRAAL_STAMPS
    CK(hipSetDevice(h->device));
    CK(hipDeviceSynchronize());
    CK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), 32 * sizeof(unsigned long long)));
    return GRAAL_OK;
#else
    return fail(h, GRAAL_E_UNSUPPORTED, "library built without GRAAL_STAMPS");
#endif
}

int graal_debug_s2eq(graal_ctx* h, uint64_t out[8], int reset)
{
    if (!h || !out) return GRAAL_E_ARG;
#ifdef GRAAL_STAMPS
    CK(hipSetDevice(h->device));
    CK(hipDeviceSynchronize());
    CK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_s2eq), 8 * sizeof(unsigned long long)));
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_s2eq), z, sizeof z)); }
    return GRAAL_OK;
#else
    return fail(h, GRAAL_E_UNSUPPORTED, "library built without GRAAL_STAMPS");
#endif
}

int graal_debug_hitstat(graal_ctx* h, uint64_t out[8], int reset)
{
    if (!h || !out) return GRAAL_E_ARG;
#ifdef GRAAL_STAMPS
    CK(hipSetDevice(h->device));
    CK(hipDeviceSynchronize());
    CK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hitstat), 8 * sizeof(unsigned long long)));
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_hitstat), z, sizeof z)); }
    return GRAAL_OK;
#else
    return fail(h, GRAAL_E_UNSUPPORTED, "library built without GRAAL_STAMPS");
#endif
}

int graal_debug_block_stamps(graal_ctx* h, uint64_t* out /* [4096][4] */)
{
    if (!h || !out) return GRAAL_E_ARG;
#ifdef GRAAL_STAMPS
    CK(hipSetDevice(h->device));
    CK(hipDeviceSynchronize());
    CK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_blk), 4096 * 4 * sizeof(unsigned long long)));
    return GRAAL_OK;
#else
    return fail(h, GRAAL_E_UNSUPPORTED, "library built without GRAAL_STAMPS");
#endif
}

int graal_last_counters(graal_ctx* h, int64_t out[4])
{
    if (!h || !out) return GRAAL_E_ARG;
    CK(hipSetDevice(h->device));
    long long res[3];
    CK(hipStreamSynchronize(h->stream));
    CK(hipMemcpy(res, h->d_scalars + 18, sizeof res, hipMemcpyDeviceToHost));
    out[0] = h->nnz; out[1] = res[0]; out[2] = res[2]; out[3] = res[1];
    return GRAAL_OK;
}

int graal_upload_own_obs(graal_ctx* h, const float* own, int32_t n_bins)
{
    if (!h || !own) return GRAAL_E_ARG;
    if (!h->have_sub || !h->have_contacts) return fail(h, GRAAL_E_STATE, "graal_upload_own_obs: upload sub-fragments and contacts first");
    if (n_bins != h->n_bins) return fail(h, GRAAL_E_ARG, "graal_upload_own_obs: one row of three counts per bin");
    if (h->single_sub || h->has_rep) return GRAAL_OK;      // (no own pixels to correct / not carried with repeats)
    CK(hipSetDevice(h->device));
    if (h->fstream) CK(hipStreamSynchronize(h->fstream));   // (a correction kernel may be reading the old table)
    if (!h->d_own_obs) CK(hipMalloc(&h->d_own_obs, sizeof(float) * 3 * (size_t)n_bins));
    CK(hipMemcpy(h->d_own_obs, own, sizeof(float) * 3 * (size_t)n_bins, hipMemcpyHostToDevice));
    h->own_complete = true;
    h->carry_q = 0; h->carry_bad = true;
    return GRAAL_OK;
}

int graal_take_carry_correction(graal_ctx* h, int64_t* q_out, int32_t* valid_out)
{
    if (!h) return GRAAL_E_ARG;
    if (!q_out || !valid_out) {   // discard: the caller evaluated the current layout in full -- the commits up to it are accounted for
        if (h->corr_inflight) h->corr_src = 2;
        else if (h->pending_commits > 0) h->corr_skip = true;
        h->carry_wanted = true;      // (a caller that anchors its total means to carry it)
    } else {
        *q_out = h->carry_bad ? 0 : h->carry_q;
        *valid_out = (h->carry_bad || !h->carry_wanted) ? 0 : 1;
        h->carry_wanted = true;      // (from the next commit on the corrections are computed: nobody pays for them who never asks)
    }
    h->carry_q = 0; h->carry_bad = false;
    return GRAAL_OK;
}

int graal_run_counters(graal_ctx* h, int64_t out[12])
{
    if (!h || !out) return GRAAL_E_ARG;
    out[0] = h->rc_evals; out[1] = h->rc_repeats; out[2] = h->spin_ok ? 1 : 0; out[3] = h->rc_gwait; out[4] = h->rc_gevent;
    out[5] = h->rc_flat; out[6] = h->rc_need_fin; out[7] = h->gave_up; out[8] = h->rc_list_grown; out[9] = (int64_t)h->slist_cap;
    out[10] = h->rc_carry_repairs; out[11] = 0;
    return GRAAL_OK;
}

#include "host_step.h"
#include "host_fit.h"

} // extern "C"
