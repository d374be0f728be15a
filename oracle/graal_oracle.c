/*
 * graal_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, scalar, single thread) of the reference's DENSE likelihood /
 * mutation algorithm: /root/reference/kernels3.cu.  Each function cites the kernel it follows.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product path (graal_amd/) never does.
 *
 * Parity pin: the reference is CUDA + PyCUDA (Python 2) and cannot be built or run in this
 * image without writing stand-ins for <curand_kernel.h> and the CUDA runtime, which is not
 * allowed; the reference ships no tests or golden vectors.  This restatement is pinned to the
 * worked known answers recorded from the reference's own kernels in SURVEY.md Appendix E
 * (tests/golden/appendix_e.json, tests/test_oracle_golden.py) and to the reference's implied
 * invariants (delta kernel == full-after - full-before; structural checks of
 * cuda_lib_gl.py:1530-1537).  Host libm (powf/expf/log) differs from CUDA's by a few ulp.
 *
 * Layout conventions (all int32 unless stated):
 *   frag SoA  : 14 arrays of n_frags, order = struct frag (kernels3.cu:9-24)
 *   sub_id    : [n_bins][4]  (x,y,z,w=n_sub)      kernels3.cu int4 id_sub_frags
 *   sub_len   : [n_bins][3]  float32, kb          float3 len_bp_sub_frags
 *   sub_accu  : [n_bins][3]                       int3 accu_sub_frags
 *   dispatcher: [n_bins][2]  [start,end) into collector_id
 *   obs       : dense float32 [width][width]
 *   param     : 8 float32 kuhn,lm,c1,slope,d,d_max,fact,v_inter (kernels3.cu:26-35)
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int *pos, *id_c, *start_bp, *len_bp, *circ, *id, *prev, *next, *l_cont, *l_cont_bp, *ori, *rep,
        *activ, *id_d;
} soa_t;

typedef struct {
    int pos, id_c, start_bp, len_bp, circ, prev, next, l_cont, l_cont_bp, ori, rep, activ, id_d;
} rec_t;

typedef struct {
    float kuhn, lm, c1, slope, d, d_max, fact, v_inter;
} param_t;

static soa_t mk_soa(int **p)
{
    soa_t s = {p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11], p[12], p[13]};
    return s;
}

static rec_t ld(const soa_t *s, int f)
{
    rec_t r = {s->pos[f],  s->id_c[f],   s->start_bp[f],  s->len_bp[f], s->circ[f],
               s->prev[f], s->next[f],   s->l_cont[f],    s->l_cont_bp[f],
               s->ori[f],  s->rep[f],    s->activ[f],     s->id_d[f]};
    return r;
}

static void st(const soa_t *s, int f, const rec_t *r)
{
    s->pos[f] = r->pos;
    s->id_c[f] = r->id_c;
    s->start_bp[f] = r->start_bp;
    s->len_bp[f] = r->len_bp;
    s->circ[f] = r->circ;
    s->id[f] = f; /* every kernel rewrites id[f] = f */
    s->prev[f] = r->prev;
    s->next[f] = r->next;
    s->l_cont[f] = r->l_cont;
    s->l_cont_bp[f] = r->l_cont_bp;
    s->ori[f] = r->ori;
    s->rep[f] = r->rep;
    s->activ[f] = r->activ;
    s->id_d[f] = r->id_d;
}

/* ------------------------------------------------------------------ copies */
/* simple_copy kernels3.cu:3755 ; copy_struct kernels3.cu:3720 (id_contigs may be NULL) */
void or_copy(int **dst_p, int **src_p, int *id_contigs, int n)
{
    soa_t d = mk_soa(dst_p), s = mk_soa(src_p);
    for (int f = 0; f < n; f++) {
        rec_t r = ld(&s, f);
        st(&d, f, &r);
        if (id_contigs) id_contigs[f] = r.id_c;
    }
}

/* flip_frag kernels3.cu:239 */
void or_flip(int **dst_p, int **src_p, int id_f_flip, int n)
{
    soa_t d = mk_soa(dst_p), s = mk_soa(src_p);
    for (int f = 0; f < n; f++) {
        rec_t r = ld(&s, f);
        if (f == id_f_flip) r.ori = r.ori * -1;
        st(&d, f, &r);
    }
}

/* swap_activity_frag kernels3.cu:283 */
void or_swap_activity(int **dst_p, int **src_p, int id_f, int max_id, int n)
{
    soa_t d = mk_soa(dst_p), s = mk_soa(src_p);
    for (int f = 0; f < n; f++) {
        rec_t r = ld(&s, f);
        if (f == id_f && r.rep == 1) {
            int a = r.activ;
            r.activ = 0 * (a == 1) + 1 * (a == 0);
            r.id_c = r.id_c * (a == 1) + (max_id + 1) * (a == 0);
        }
        st(&d, f, &r);
    }
}

/* pop_out_frag kernels3.cu:329 */
void or_pop_out(int **dst_p, int **src_p, int *pop_id_contigs, int id_f_pop, int max_id, int n)
{
    soa_t d = mk_soa(dst_p), s = mk_soa(src_p);
    const rec_t P = ld(&s, id_f_pop);
    for (int f = 0; f < n; f++) {
        rec_t r = ld(&s, f);
        if (P.l_cont > 2 && r.id_c == P.id_c) {
            if (r.pos < P.pos) {
                if (f == P.next && P.circ == 1) r.prev = P.prev;
                if (r.pos == P.pos - 1) r.next = P.next;
                r.l_cont -= 1;
                r.l_cont_bp -= P.len_bp;
            } else if (r.pos == P.pos) {
                r.pos = 0; r.id_c = max_id + 1; r.start_bp = 0; r.circ = 0; r.ori = 1;
                r.prev = -1; r.next = -1; r.l_cont = 1; r.l_cont_bp = r.len_bp;
            } else {
                int prev_fi = r.prev, next_fi = r.next;
                r.prev = (r.pos == P.pos + 1) ? P.prev : prev_fi;
                r.next = (f == P.prev && P.circ == 1) ? P.next : next_fi;
                r.pos -= 1;
                r.start_bp -= P.len_bp;
                r.l_cont -= 1;
                r.l_cont_bp -= P.len_bp;
            }
        } else if (P.l_cont == 2 && r.id_c == P.id_c) {
            if (r.pos < P.pos) {
                r.circ = 0; r.prev = -1; r.next = -1;
                r.l_cont -= 1; r.l_cont_bp -= P.len_bp;
            } else if (r.pos == P.pos) {
                r.pos = 0; r.id_c = max_id + 1; r.start_bp = 0; r.circ = 0; r.ori = 1;
                r.prev = -1; r.next = -1; r.l_cont = 1; r.l_cont_bp = r.len_bp;
            } else {
                r.pos -= 1; r.start_bp -= P.len_bp; r.circ = 0; r.prev = -1; r.next = -1;
                r.l_cont -= 1; r.l_cont_bp -= P.len_bp;
            }
        }
        st(&d, f, &r);
        if (pop_id_contigs) pop_id_contigs[f] = r.id_c;
    }
}

/* pop_in_frag_1..4 kernels3.cu:565 / 814 / 1081 / 1267.  which = 1..4 */
void or_pop_in(int which, int **dst_p, int **src_p, int id_f_pop, int id_f_ins, int max_id, int ori_f_pop,
               int n)
{
    soa_t d = mk_soa(dst_p), s = mk_soa(src_p);
    const rec_t P = ld(&s, id_f_pop), I = ld(&s, id_f_ins);
    if (!(I.activ == 1 && P.activ == 1)) {
        for (int f = 0; f < n; f++) { rec_t r = ld(&s, f); st(&d, f, &r); }
        return;
    }
    for (int f = 0; f < n; f++) {
        rec_t r = ld(&s, f);
        const int pos_fi = r.pos, start_fi = r.start_bp, prev_fi = r.prev, next_fi = r.next;
        if (which == 1) { /* split insert @ left */
            if (f == id_f_pop) {
                r.pos = 0; r.start_bp = 0; r.len_bp = P.len_bp; r.circ = 0; r.ori = ori_f_pop;
                r.prev = -1; r.next = id_f_ins;
                if (I.circ == 0) {
                    r.id_c = max_id + 1;
                    r.l_cont = I.l_cont - I.pos + 1;
                    r.l_cont_bp = I.l_cont_bp - I.start_bp + P.len_bp;
                } else {
                    r.id_c = I.id_c;
                    r.l_cont = I.l_cont + 1;
                    r.l_cont_bp = I.l_cont_bp + P.len_bp;
                }
            } else if (r.id_c == I.id_c) {
                if (I.circ == 0) {
                    if (pos_fi < I.pos) {
                        r.circ = 0;
                        r.next = (pos_fi == I.pos - 1) ? -1 : next_fi;
                        r.l_cont = I.pos; r.l_cont_bp = I.start_bp;
                    } else if (pos_fi == I.pos) {
                        r.pos = 1; r.id_c = max_id + 1; r.start_bp = P.len_bp; r.circ = 0;
                        r.ori = I.ori; r.prev = id_f_pop; r.next = I.next;
                        r.l_cont = I.l_cont - I.pos + 1;
                        r.l_cont_bp = I.l_cont_bp - I.start_bp + P.len_bp;
                    } else {
                        r.pos = pos_fi - I.pos + 1; r.id_c = max_id + 1;
                        r.start_bp = start_fi - I.start_bp + P.len_bp; r.circ = 0;
                        r.l_cont = I.l_cont - I.pos + 1;
                        r.l_cont_bp = I.l_cont_bp - I.start_bp + P.len_bp;
                    }
                } else { /* circular target contig is linearised at f_ins */
                    if (pos_fi < I.pos) {
                        r.pos = I.l_cont - I.pos + pos_fi + 1;
                        r.start_bp = I.l_cont_bp - I.start_bp + start_fi + P.len_bp;
                        r.circ = 0;
                        r.next = (pos_fi == I.pos - 1) ? -1 : next_fi;
                    } else if (pos_fi == I.pos) {
                        r.pos = 1; r.start_bp = P.len_bp; r.len_bp = I.len_bp; r.circ = 0;
                        r.ori = I.ori; r.prev = id_f_pop; r.next = I.next;
                    } else {
                        r.pos = pos_fi - I.pos + 1;
                        r.start_bp = start_fi - I.start_bp + P.len_bp; r.circ = 0;
                        r.next = (f == I.prev) ? -1 : next_fi;
                    }
                    r.id_c = I.id_c;
                    r.l_cont = I.l_cont + 1; r.l_cont_bp = I.l_cont_bp + P.len_bp;
                }
            }
        } else if (which == 2) { /* split insert @ right */
            if (f == id_f_pop) {
                r.id_c = I.id_c; r.len_bp = P.len_bp; r.circ = 0; r.ori = ori_f_pop;
                r.prev = id_f_ins; r.next = -1;
                if (I.circ == 0) {
                    r.pos = I.pos + 1; r.start_bp = I.start_bp + I.len_bp;
                    r.l_cont = I.pos + 2;
                    r.l_cont_bp = I.start_bp + I.len_bp + P.len_bp;
                } else {
                    r.pos = (I.l_cont - (I.pos + 1)) + I.pos + 1;
                    r.start_bp = (I.l_cont_bp - (I.start_bp + I.len_bp)) + I.start_bp + I.len_bp;
                    r.l_cont = I.l_cont + 1; r.l_cont_bp = I.l_cont_bp + P.len_bp;
                }
            } else if (r.id_c == I.id_c) {
                if (I.circ == 0) {
                    if (pos_fi < I.pos) {
                        r.circ = 0; r.l_cont = I.pos + 2;
                        r.l_cont_bp = I.start_bp + I.len_bp + P.len_bp;
                    } else if (pos_fi == I.pos) {
                        r.circ = 0; r.ori = I.ori; r.prev = I.prev; r.next = id_f_pop;
                        r.l_cont = I.pos + 2;
                        r.l_cont_bp = I.start_bp + I.len_bp + P.len_bp;
                    } else {
                        r.pos = pos_fi - (I.pos + 1); r.id_c = max_id + 1;
                        r.start_bp = start_fi - (I.start_bp + I.len_bp); r.circ = 0;
                        r.prev = (pos_fi == I.pos + 1) ? -1 : prev_fi;
                        r.l_cont = I.l_cont - (I.pos + 1);
                        r.l_cont_bp = I.l_cont_bp - (I.start_bp + I.len_bp);
                    }
                } else {
                    if (pos_fi < I.pos) {
                        r.pos = (I.l_cont - (I.pos + 1)) + pos_fi;
                        r.start_bp = (I.l_cont_bp - (I.start_bp + I.len_bp)) + start_fi;
                        r.circ = 0;
                        r.prev = (f == I.next) ? -1 : prev_fi;
                    } else if (pos_fi == I.pos) {
                        r.pos = (I.l_cont - (I.pos + 1)) + I.pos;
                        r.start_bp = (I.l_cont_bp - (I.start_bp + I.len_bp)) + I.start_bp;
                        r.len_bp = I.len_bp; r.circ = 0;
                        r.prev = I.prev; r.next = id_f_pop; /* ori keeps or_fi */
                    } else {
                        r.pos = pos_fi - (I.pos + 1);
                        r.start_bp = start_fi - (I.start_bp + I.len_bp); r.circ = 0;
                        r.prev = (pos_fi == I.pos + 1) ? -1 : prev_fi;
                    }
                    r.id_c = I.id_c;
                    r.l_cont = I.l_cont + 1; r.l_cont_bp = I.l_cont_bp + P.len_bp;
                }
            }
        } else if (which == 3) { /* insert @ right of f_ins */
            if (f == id_f_pop) {
                r.pos = I.pos + 1; r.id_c = I.id_c; r.start_bp = I.start_bp + I.len_bp;
                r.len_bp = P.len_bp; r.circ = I.circ; r.ori = ori_f_pop;
                r.prev = id_f_ins; r.next = I.next;
                r.l_cont = I.l_cont + 1; r.l_cont_bp = I.l_cont_bp + P.len_bp;
            } else if (r.id_c == I.id_c) {
                r.circ = I.circ;
                if (pos_fi < I.pos) {
                    r.prev = (f == I.next && I.circ == 1) ? id_f_pop : prev_fi;
                } else if (pos_fi == I.pos) {
                    r.ori = I.ori; r.next = id_f_pop;
                } else {
                    r.pos = pos_fi + 1; r.start_bp = start_fi + P.len_bp;
                    r.prev = (pos_fi == I.pos + 1) ? id_f_pop : prev_fi;
                }
                r.l_cont = I.l_cont + 1; r.l_cont_bp = I.l_cont_bp + P.len_bp;
            }
        } else { /* which == 4 : insert @ left of f_ins */
            if (f == id_f_pop) {
                r.pos = I.pos; r.id_c = I.id_c; r.start_bp = I.start_bp;
                r.len_bp = P.len_bp; r.circ = I.circ; r.ori = ori_f_pop;
                r.prev = I.prev; r.next = id_f_ins;
                r.l_cont = I.l_cont + 1; r.l_cont_bp = I.l_cont_bp + P.len_bp;
            } else if (r.id_c == I.id_c) {
                r.circ = I.circ;
                if (pos_fi < I.pos) {
                    r.next = (pos_fi == I.pos - 1) ? id_f_pop : next_fi;
                } else if (pos_fi == I.pos) {
                    r.pos = I.pos + 1; r.start_bp = I.start_bp + P.len_bp;
                    r.ori = I.ori; r.prev = id_f_pop; r.next = I.next;
                } else {
                    r.pos = pos_fi + 1; r.start_bp = start_fi + P.len_bp;
                }
                r.l_cont = I.l_cont + 1; r.l_cont_bp = I.l_cont_bp + P.len_bp;
            }
        }
        st(&d, f, &r);
    }
}

/* split_contig kernels3.cu:1451 */
void or_split(int **dst_p, int **src_p, int *split_id_contigs, int id_f_cut, int upstream, int max_id, int n)
{
    soa_t d = mk_soa(dst_p), s = mk_soa(src_p);
    const rec_t C = ld(&s, id_f_cut);
    for (int f = 0; f < n; f++) {
        rec_t r = ld(&s, f);
        const int pos_fi = r.pos, start_fi = r.start_bp, prev_fi = r.prev, next_fi = r.next;
        if (C.activ == 1 && C.l_cont > 1 && r.id_c == C.id_c) {
            if (C.circ == 0) {
                if (upstream == 1) {
                    if (pos_fi < C.pos) {
                        r.circ = 0;
                        r.next = (pos_fi == C.pos - 1) ? -1 : next_fi;
                        r.l_cont = C.pos; r.l_cont_bp = C.start_bp;
                    } else if (pos_fi == C.pos) {
                        r.pos = 0; r.id_c = max_id + 1; r.start_bp = 0; r.len_bp = C.len_bp;
                        r.circ = 0; r.prev = -1; r.next = C.next;
                        r.l_cont = C.l_cont - C.pos; r.l_cont_bp = C.l_cont_bp - C.start_bp;
                    } else {
                        r.pos = pos_fi - C.pos; r.id_c = max_id + 1;
                        r.start_bp = start_fi - C.start_bp; r.circ = 0;
                        r.l_cont = C.l_cont - C.pos; r.l_cont_bp = C.l_cont_bp - C.start_bp;
                    }
                } else {
                    if (pos_fi < C.pos) {
                        r.circ = 0;
                        r.l_cont = C.pos + 1; r.l_cont_bp = C.start_bp + C.len_bp;
                    } else if (pos_fi == C.pos) {
                        r.pos = C.pos; r.start_bp = C.start_bp; r.len_bp = C.len_bp; r.circ = 0;
                        r.prev = C.prev; r.next = -1;
                        r.l_cont = C.pos + 1; r.l_cont_bp = C.start_bp + C.len_bp;
                    } else {
                        r.pos = pos_fi - (C.pos + 1); r.id_c = max_id + 1;
                        r.start_bp = start_fi - (C.start_bp + C.len_bp); r.circ = 0;
                        r.prev = (pos_fi == C.pos + 1) ? -1 : prev_fi;
                        r.l_cont = C.l_cont - (C.pos + 1);
                        r.l_cont_bp = C.l_cont_bp - (C.start_bp + C.len_bp);
                    }
                }
            } else { /* circular contig: becomes linear, keeps its label and length */
                if (upstream == 1) {
                    if (pos_fi < C.pos) {
                        r.pos = C.l_cont - C.pos + pos_fi;
                        r.start_bp = C.l_cont_bp - C.start_bp + start_fi;
                        r.next = (pos_fi == C.pos - 1) ? -1 : next_fi;
                    } else if (pos_fi == C.pos) {
                        r.pos = 0; r.start_bp = 0; r.len_bp = C.len_bp;
                        r.prev = -1; r.next = C.next;
                    } else {
                        r.pos = pos_fi - C.pos; r.start_bp = start_fi - C.start_bp;
                        r.next = (f == C.prev) ? -1 : next_fi;
                    }
                } else {
                    if (pos_fi < C.pos) {
                        r.pos = (C.l_cont - (C.pos + 1)) + pos_fi;
                        r.start_bp = (C.l_cont_bp - (C.start_bp + C.len_bp)) + start_fi;
                        r.prev = (f == C.next) ? -1 : prev_fi;
                    } else if (pos_fi == C.pos) {
                        r.pos = (C.l_cont - (C.pos + 1)) + pos_fi;
                        r.start_bp = (C.l_cont_bp - (C.start_bp + C.len_bp)) + C.start_bp;
                        r.len_bp = C.len_bp; r.prev = C.prev; r.next = -1;
                    } else {
                        r.pos = pos_fi - (C.pos + 1);
                        r.start_bp = start_fi - (C.start_bp + C.len_bp);
                        r.prev = (pos_fi == C.pos + 1) ? -1 : prev_fi;
                    }
                }
                r.id_c = C.id_c; r.circ = 0;
                r.l_cont = C.l_cont; r.l_cont_bp = C.l_cont_bp;
            }
        }
        st(&d, f, &r);
        if (split_id_contigs) split_id_contigs[f] = r.id_c;
    }
}

/* paste_contigs kernels3.cu:1786.  Returns the number of frags NOT written (the reference's
 * "stale slot": same contig, neither end condition, kernels3.cu:1977-2033); those entries of dst
 * keep whatever they held, exactly as the reference does. */
int or_paste(int **dst_p, int **src_p, int id_fA, int id_fB, int max_id, int n)
{
    (void)max_id;
    soa_t d = mk_soa(dst_p), s = mk_soa(src_p);
    const rec_t A = ld(&s, id_fA), B = ld(&s, id_fB);
    int stale = 0;
    for (int f = 0; f < n; f++) {
        rec_t r = ld(&s, f);
        const int pos_fi = r.pos, start_fi = r.start_bp, prev_fi = r.prev, next_fi = r.next;
        if (A.activ == 1 && B.activ == 1) {
            if (A.id_c != B.id_c) {
                if (r.id_c == A.id_c) {
                    if (A.pos == 0) { /* contig A is reversed so that fA becomes its tail */
                        r.pos = A.l_cont - (pos_fi + 1);
                        r.start_bp = A.l_cont_bp - (start_fi + r.len_bp);
                        r.ori = r.ori * -1;
                        r.prev = (pos_fi == A.l_cont - 1) ? -1 : next_fi;
                        r.next = (pos_fi == A.pos) ? id_fB : prev_fi;
                    } else {
                        r.next = (pos_fi == A.pos) ? id_fB : next_fi;
                    }
                    r.id_c = A.id_c; r.circ = 0;
                    r.l_cont = A.l_cont + B.l_cont; r.l_cont_bp = A.l_cont_bp + B.l_cont_bp;
                } else if (r.id_c == B.id_c) {
                    if (B.pos == 0) {
                        r.pos = A.l_cont + pos_fi;
                        r.start_bp = A.l_cont_bp + start_fi;
                        r.prev = (pos_fi == B.pos) ? id_fA : prev_fi;
                    } else { /* contig B is reversed so that fB becomes its head */
                        r.pos = A.l_cont + (B.l_cont - (pos_fi + 1));
                        r.start_bp = A.l_cont_bp + (B.l_cont_bp - (start_fi + r.len_bp));
                        r.ori = r.ori * -1;
                        r.prev = (pos_fi == B.pos) ? id_fA : next_fi;
                        r.next = (pos_fi == 0) ? -1 : prev_fi;
                    }
                    r.id_c = A.id_c; r.circ = 0;
                    r.l_cont = A.l_cont + B.l_cont; r.l_cont_bp = A.l_cont_bp + B.l_cont_bp;
                }
            } else { /* same contig: circularise if fA / fB are its two ends */
                if (r.id_c == A.id_c) {
                    if (A.pos == 0 && B.pos == A.l_cont - 1) {
                        r.circ = 1;
                        r.prev = (pos_fi == A.pos) ? id_fB : prev_fi;
                        r.next = (pos_fi == A.l_cont - 1) ? id_fA : next_fi;
                        r.l_cont = A.l_cont; r.l_cont_bp = A.l_cont_bp;
                    } else if (A.pos == A.l_cont - 1 && B.pos == 0) {
                        r.circ = 1;
                        r.prev = (pos_fi == B.pos) ? id_fA : prev_fi;
                        r.next = (pos_fi == A.l_cont - 1) ? id_fB : next_fi;
                        r.l_cont = A.l_cont; r.l_cont_bp = A.l_cont_bp;
                    } else {
                        stale++;
                        continue; /* nothing written */
                    }
                }
            }
        }
        st(&d, f, &r);
    }
    return stale;
}

/* fill_sub_index_fA / fB kernels3.cu:3225 / 3238 (offset = 0 for A, l_cont_fA for B) */
void or_fill_sub_index(int **src_p, int *sub_index, int contig, int offset, int n)
{
    soa_t s = mk_soa(src_p);
    for (int f = 0; f < n; f++)
        if (s.id_c[f] == contig) sub_index[offset + s.pos[f]] = s.id_d[f];
}

/* contig relabel part of gl_update_pos kernels3.cu:3848-3851 */
void or_relabel(int **src_p, const int *old_2_new, int *id_contigs, int n)
{
    soa_t s = mk_soa(src_p);
    for (int f = 0; f < n; f++) {
        int c = old_2_new[s.id_c[f]];
        s.id_c[f] = c;
        if (id_contigs) id_contigs[f] = c;
    }
}

/* ------------------------------------------------------------- likelihood */
/* factorial kernels3.cu:80 (float) */
static float factorial_f(float n)
{
    float result = 1;
    n = floorf(n);
    if (n < 10) {
        for (int c = 1; c <= n; c++) result = result * c;
    } else {
        /* CUDA resolves exp(-n) on a float to the float overload; sqrtf takes the double
         * product 2*M_PI*n rounded to float */
        result = powf(n, n) * expf(-n) * sqrtf((float)(2 * M_PI * n));
    }
    return result;
}

/* rippe_contacts kernels3.cu:120 (all float: pow/exp resolve to float overloads) */
static float rippe_contacts(float s, const param_t *p)
{
    float result = 0.0f;
    if ((s > 0.0f) && (s < p->d_max)) {
        result = (p->c1 * powf(s, p->slope) * expf((p->d - 2) / (powf(s * p->lm / p->kuhn, 2.0f) + p->d))) *
                 p->fact;
    }
    return fmaxf(result, p->v_inter);
}

/* rippe_contacts_circ kernels3.cu:135 */
static float rippe_contacts_circ(float s, float s_tot, const param_t *p)
{
    float result = 0.0f;
    if ((s > 0.0f) && (s < p->d_max)) {
        float K = p->lm / p->kuhn;
        float n_dist = s, n_tot = s_tot;
        float nmax = K * 1;
        float n = K * n_dist * (n_tot - n_dist) / n_tot;
        float norm_lin = rippe_contacts(s, p);
        float norm_circ =
            (powf(p->kuhn, -3.0f) * powf(nmax, p->slope) * expf((p->d - 2.0f) / (powf(nmax, 2.0f) + p->d))) *
            p->fact;
        float val =
            (powf(p->kuhn, -3.0f) * powf(n, p->slope) * expf((p->d - 2.0f) / (powf(n, 2.0f) + p->d))) * p->fact;
        result = val * norm_lin / norm_circ;
    }
    return fmaxf(result, p->v_inter);
}

/* evaluate_likelihood_double kernels3.cu:191 */
static double lik_double(double ex, double ob)
{
    double res = 0;
    const double lim = 15;
    if (ex != 0) {
        if (ob >= lim) {
            res = ob * log(ex) - ex - (ob * log(ob) - ob + log(sqrt(ob * 2.0 * M_PI)));
        } else if ((ob > 0) && (ob < lim)) {
            res = ob * log(ex) - ex - log((double)factorial_f((float)ob));
        } else if (ob == 0) {
            res = -ex;
        }
    }
    return res;
}

typedef struct {
    const float *obs;
    int width;
    soa_t fr;
    const int *collector;
    const int *dispatcher; /* [n_bins][2] */
    const int *sub_id;     /* [n_bins][4] */
    const float *sub_len;  /* [n_bins][3] */
    const int *sub_accu;   /* [n_bins][3] */
    param_t p;
    float nfpb;
    int fix_trans_accu; /* 0 = reference behaviour (kernels3.cu:3155 / 3638 accu quirk kept) */
    float obs_store[3][3]; /* like the reference's per-thread local_storage_obs: never cleared */
} lctx_t;

/* walk order of the sub-frags of one bin copy (kernels3.cu:2997-3060) */
typedef struct {
    int n;         /* limit + 1 */
    float s[3];    /* centre coordinate, kb, walked order */
    int id[3];     /* sub-level row/col */
    int slot[3];   /* data slot (un-reversed sub index) */
    int accu[3];
} walk_t;

static void walk_cis(const lctx_t *c, int f, int id_data, walk_t *w)
{
    const int *sid = c->sub_id + 4 * id_data;
    const float *len = c->sub_len + 3 * id_data;
    const int *acc = c->sub_accu + 3 * id_data;
    const int limit = sid[3] - 1;
    const float start = (float)c->fr.start_bp[f];
    float run;
    w->n = limit + 1;
    if (c->fr.ori[f] == 1) {
        run = start / 1000.0f + len[0];
        w->s[0] = start / 1000.0f + len[0] / 2.0f;
        w->id[0] = sid[0]; w->slot[0] = 0; w->accu[0] = acc[0];
        for (int i = 1; i <= limit; i++) {
            w->s[i] = run + len[i] / 2.0f;
            run = run + len[i];
            w->id[i] = sid[i]; w->slot[i] = i; w->accu[i] = acc[i];
        }
    } else {
        run = start / 1000.0f + len[limit];
        w->s[0] = start / 1000.0f + len[limit] / 2.0f;
        w->id[0] = sid[limit]; w->slot[0] = limit; w->accu[0] = acc[limit];
        for (int i = 1; i <= limit; i++) {
            w->s[i] = run + len[limit - i] / 2.0f;
            run = run + len[limit - i];
            w->id[i] = sid[limit - i]; w->slot[i] = limit - i; w->accu[i] = acc[limit - i];
        }
    }
}

/* trans branch (kernels3.cu:3133-3182): `first` selects the fi side, which carries the
 * reference's accu quirk for reversed bins (accu_sub_fi[limit_fi] for every i >= 1). */
static void walk_trans(const lctx_t *c, int f, int id_data, int first, walk_t *w)
{
    const int *sid = c->sub_id + 4 * id_data;
    const int *acc = c->sub_accu + 3 * id_data;
    const int limit = sid[3] - 1;
    w->n = limit + 1;
    if (c->fr.ori[f] == 1) {
        for (int i = 0; i <= limit; i++) { w->id[i] = sid[i]; w->slot[i] = i; w->accu[i] = acc[i]; }
    } else {
        for (int i = 0; i <= limit; i++) {
            w->id[i] = sid[limit - i]; w->slot[i] = limit - i;
            w->accu[i] = (first && i >= 1 && !c->fix_trans_accu) ? acc[limit] : acc[limit - i];
        }
    }
}

/* One pixel = one pair of unique bins (or one bin's own upper triangle when on_diag).
 * Body of evaluate_likelihood (kernels3.cu:2895-3220) == body of sub_compute_likelihood
 * (kernels3.cu:3383-3697). */
static double pixel_lik(lctx_t *c, int bin_i, int bin_j, int on_diag)
{
    float ex[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    const int di0 = c->dispatcher[2 * bin_i], di1 = c->dispatcher[2 * bin_i + 1];
    const int dj0 = c->dispatcher[2 * bin_j], dj1 = c->dispatcher[2 * bin_j + 1];
    const int init_limit_fi = c->sub_id[4 * c->fr.id_d[c->collector[di0]] + 3] - 1;
    const int init_limit_fj = c->sub_id[4 * c->fr.id_d[c->collector[dj0]] + 3] - 1;
    int loop_id_i = 0, loop_id_j = 0;
    for (int ri = di0; ri < di1; ri++) {
        if (c->fr.activ[c->collector[ri]] != 1) continue;
        for (int rj = dj0; rj < dj1; rj++) {
            int fi = c->collector[ri], fj = c->collector[rj];
            if (c->fr.activ[fj] != 1) continue;
            int id_data_fi = c->fr.id_d[fi], id_data_fj = c->fr.id_d[fj];
            const int first_obs = (loop_id_i == 0) && (loop_id_j == 0);
            walk_t wi, wj;
            if (c->fr.id_c[fi] == c->fr.id_c[fj]) {
                int swap = 0;
                if (c->fr.pos[fi] > c->fr.pos[fj]) { /* fi is always the closest frag to the origin */
                    swap = 1;
                    int t = fi; fi = fj; fj = t;
                    t = id_data_fi; id_data_fi = id_data_fj; id_data_fj = t;
                }
                const float s_tot = (float)c->fr.l_cont_bp[fi] / 1000.0f;
                walk_cis(c, fi, id_data_fi, &wi);
                walk_cis(c, fj, id_data_fj, &wj);
                for (int i = 0; i < wi.n; i++)
                    for (int j = 0; j < wj.n; j++) {
                        float s = fabsf(wj.s[j] - wi.s[i]);
                        float norm = (float)(wi.accu[i] * wj.accu[j]) / c->nfpb;
                        float e = (c->fr.circ[fi] == 1) ? rippe_contacts_circ(s, s_tot, &c->p) * norm
                                                        : rippe_contacts(s, &c->p) * norm;
                        int a = swap ? wj.slot[j] : wi.slot[i];
                        int b = swap ? wi.slot[i] : wj.slot[j];
                        ex[a][b] = ex[a][b] + e;
                        if (first_obs) c->obs_store[a][b] = c->obs[(size_t)wi.id[i] * c->width + wj.id[j]];
                    }
            } else {
                walk_trans(c, fi, id_data_fi, 1, &wi);
                walk_trans(c, fj, id_data_fj, 0, &wj);
                for (int i = 0; i < wi.n; i++)
                    for (int j = 0; j < wj.n; j++) {
                        float norm = (float)(wi.accu[i] * wj.accu[j]) / c->nfpb;
                        float e = c->p.v_inter * norm;
                        int a = wi.slot[i], b = wj.slot[j];
                        ex[a][b] = ex[a][b] + e;
                        if (first_obs) c->obs_store[a][b] = c->obs[(size_t)wi.id[i] * c->width + wj.id[j]];
                    }
            }
            loop_id_j += 1; /* never reset per fi: obs are read for the first active pair only */
        }
        loop_id_i += 1;
    }
    double val = 0.0;
    for (int i = 0; i <= init_limit_fi; i++)
        for (int j = on_diag ? (i + 1) : 0; j <= init_limit_fj; j++)
            val = lik_double((double)ex[i][j], (double)c->obs_store[i][j]) + val;
    return val;
}

static void mk_ctx(lctx_t *c, const float *obs, int width, int **fr, const int *collector, const int *dispatcher,
                   const int *sub_id, const float *sub_len, const int *sub_accu, const float *param, float nfpb,
                   int fix_trans_accu)
{
    memset(c, 0, sizeof(*c));
    c->obs = obs; c->width = width; c->fr = mk_soa(fr);
    c->collector = collector; c->dispatcher = dispatcher;
    c->sub_id = sub_id; c->sub_len = sub_len; c->sub_accu = sub_accu;
    memcpy(&c->p, param, sizeof(param_t));
    c->nfpb = nfpb; c->fix_trans_accu = fix_trans_accu;
}

/* index of pixel (i<j) in the per-pixel vector: conv_plan_pos_2_lin kernels3.cu:226 */
static size_t pix_index(int i, int j) { return (size_t)j * (size_t)(j - 1) / 2 + (size_t)i; }

/* evaluate_likelihood kernels3.cu:2802.  likelihood has n_bins(n_bins-1)/2 + n_bins entries;
 * returns their sum (the reference's gpuarray.sum, cuda_lib_gl.py:1848). */
double or_evaluate_likelihood(const float *obs, int width, int **fr, const int *collector, const int *dispatcher,
                              const int *sub_id, const float *sub_len, const int *sub_accu, const float *param,
                              float nfpb, int n_bins, double *likelihood, int fix_trans_accu)
{
    lctx_t c;
    mk_ctx(&c, obs, width, fr, collector, dispatcher, sub_id, sub_len, sub_accu, param, nfpb, fix_trans_accu);
    const size_t n_up = (size_t)n_bins * (size_t)(n_bins - 1) / 2;
    double tot = 0.0;
    for (int j = 1; j < n_bins; j++)
        for (int i = 0; i < j; i++) {
            double v = pixel_lik(&c, i, j, 0);
            if (likelihood) likelihood[pix_index(i, j)] = v;
            tot += v;
        }
    for (int i = 0; i < n_bins; i++) {
        double v = pixel_lik(&c, i, i, 1);
        if (likelihood) likelihood[n_up + i] = v;
        tot += v;
    }
    return tot;
}

/* sub_compute_likelihood kernels3.cu:3259: sum over the 4 pixel ranges of (new - curr). */
double or_sub_compute_likelihood(const float *obs, int width, int **fr, const int *sub_index, int n_no_rep,
                                 const int *list_rep, int n_rep, const int *list_uniq, int n_uniq,
                                 const int *collector, const int *dispatcher, const int *sub_id,
                                 const float *sub_len, const int *sub_accu, const float *param, float nfpb,
                                 int n_bins, const double *curr_likelihood, int fix_trans_accu)
{
    lctx_t c;
    mk_ctx(&c, obs, width, fr, collector, dispatcher, sub_id, sub_len, sub_accu, param, nfpb, fix_trans_accu);
    const size_t n_up = (size_t)n_bins * (size_t)(n_bins - 1) / 2;
    double out = 0.0;
#define PAIR(a_, b_)                                                                           \
    do {                                                                                       \
        int lo = (a_) < (b_) ? (a_) : (b_), hi = (a_) < (b_) ? (b_) : (a_);                    \
        /* lo == hi cannot happen: list_uniq excludes duplicated bins (cuda_lib_gl.py:74) */   \
        out = out + pixel_lik(&c, lo, hi, 0) - curr_likelihood[pix_index(lo, hi)];             \
    } while (0)
    for (int y = 1; y < n_no_rep; y++)
        for (int x = 0; x < y; x++) PAIR(sub_index[x], sub_index[y]);
    for (int r = 0; r < n_rep; r++)
        for (int u = 0; u < n_uniq; u++) PAIR(list_rep[r], list_uniq[u]);
    for (int y = 1; y < n_rep; y++)
        for (int x = 0; x < y; x++) PAIR(list_rep[x], list_rep[y]);
    for (int r = 0; r < n_rep; r++)
        out = out + pixel_lik(&c, list_rep[r], list_rep[r], 1) - curr_likelihood[n_up + list_rep[r]];
#undef PAIR
    return out;
}

/* scalar helpers exported for unit tests */
float or_rippe(float s, const float *param) { return rippe_contacts(s, (const param_t *)param); }
float or_rippe_circ(float s, float s_tot, const float *param)
{
    return rippe_contacts_circ(s, s_tot, (const param_t *)param);
}
double or_lik(double ex, double ob) { return lik_double(ex, ob); }
