// host_step.h -- the per-step HOST logic of the sampler behind the C ABI (included by graal_hip.hip inside extern "C").
//
// What cuda_lib_gl.sampler.step_max_likelihood does on the host between its kernel launches (cuda_lib_gl.py:1873-1961:
// score post-processing and sampling; :2295-2331 return_neighbours), which graal_amd/sampler.py spells out in Python
// (select_move, legacy_choice_one, legacy_choice_without_replacement, return_neighbours) -- here in C++, bit for bit: the same
// float64 operations in the same order (numpy's pairwise summation included) and the same draws from the SAME generator: the
// caller passes the address of its numpy MT19937 state (`RandomState._bit_generator.ctypes.state_address`: uint32 key[624],
// int pos), which is advanced in place exactly as RandomState.random_sample would.  Anything unusual (probabilities that do not
// sum to one, no candidate, a temperature other than 1) is NOT handled here: the call returns GRAAL_STEP_FALLBACK before any
// draw and the Python path takes the step; a score vector the selection does not judge itself comes back as GRAAL_STEP_SELECT --
// neighbours drawn, scores valid, the selection (and its draw) left to the caller.  tests/test_host_logic.py compares values AND generator state with the Python
// path (and so, transitively, with numpy itself).

namespace {

struct MtState { uint32_t key[624]; int pos; };   // numpy/random/src/mt19937/mt19937.h: mt19937_state

static void mt_gen(MtState* s)
{
    const uint32_t MATRIX_A = 0x9908b0dfu, UPPER = 0x80000000u, LOWER = 0x7fffffffu;
    const int N = 624, M = 397;
    uint32_t y;
    int i;
    for (i = 0; i < N - M; i++) {
        y = (s->key[i] & UPPER) | (s->key[i + 1] & LOWER);
        s->key[i] = s->key[i + M] ^ (y >> 1) ^ ((0u - (y & 1u)) & MATRIX_A);
    }
    for (; i < N - 1; i++) {
        y = (s->key[i] & UPPER) | (s->key[i + 1] & LOWER);
        s->key[i] = s->key[i + (M - N)] ^ (y >> 1) ^ ((0u - (y & 1u)) & MATRIX_A);
    }
    y = (s->key[N - 1] & UPPER) | (s->key[0] & LOWER);
    s->key[N - 1] = s->key[M - 1] ^ (y >> 1) ^ ((0u - (y & 1u)) & MATRIX_A);
    s->pos = 0;
}

static inline uint32_t mt_next(MtState* s)
{
    if (s->pos == 624) mt_gen(s);
    uint32_t y = s->key[s->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// RandomState.random_sample: mt19937_next_double
static inline double mt_double(MtState* s)
{
    const int32_t a = (int32_t)(mt_next(s) >> 5), b = (int32_t)(mt_next(s) >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

// numpy's float64 add-reduce over a contiguous array: 0 + pairwise sum (8 accumulators below 128 elements)
static double np_pairwise_sum(const double* a, long n)
{
    if (n < 8) {
        double res = 0.0;
        for (long i = 0; i < n; i++) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        long i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    long n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}
static double np_sum(const double* a, long n) { return 0.0 + np_pairwise_sum(a, n); }

struct HostStep {
    bool ready = false;
    bool timing = getenv("GRAAL_STEP_TIMING") != nullptr;   // debug: mean host time of each phase of graal_step, printed at destroy
    double t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long t_n = 0;
    double t_mark = 0.0;
    int n_bins = 0, k = 0, n = 0;
    std::vector<int> xk;             // [n_bins][k]
    std::vector<double> pk;          // [n_bins][k] float32 values as doubles
    std::vector<int> nnz_row;        // entries != 0 per row
    std::vector<int> id_d, disp, coll;
    std::vector<unsigned char> is_dup, black;
    bool trace = getenv("GRAAL_STEP_TRACE") != nullptr;      // debug: one line per proposal / selection on stderr
    bool force_select = getenv("GRAAL_STEP_FORCE_SELECT") != nullptr;   // test hook: every selection is handed back (GRAAL_STEP_SELECT)
    // between begin and finish
    bool paused = false;             // graal_step returned GRAAL_STEP_PAUSED: graal_step_finish is due (and nothing else)
    int fA = -1, max_id = 0;
    std::vector<int> nb;
    long long stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

void hs_free(HostStep* p)
{
    if (p && p->timing && p->t_n)
        fprintf(stderr, "graal_step timing over %lld deferred steps (us): relabel launch %.1f | proposal %.1f | scoring launches + wait %.1f | statistics %.1f | "
                        "selection %.1f | commit launch %.1f\n", p->t_n, p->t_acc[0] / p->t_n, p->t_acc[1] / p->t_n, p->t_acc[2] / p->t_n, p->t_acc[3] / p->t_n,
                p->t_acc[4] / p->t_n, p->t_acc[5] / p->t_n);
    delete p;
}
inline double hs_now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

} // namespace

int graal_upload_proposal_tables(graal_ctx* h, const int32_t* xk, const float* pk, int32_t n_bins, int32_t k, const int32_t* id_d,
                                 int32_t n_frags, const int32_t* dispatcher, const int32_t* collector, int32_t n_collector,
                                 const uint8_t* dup_bin_flags, const uint8_t* black_frag_flags)
{
    if (!h || !xk || !pk || !id_d || !dispatcher || !collector || !dup_bin_flags || !black_frag_flags || n_bins < 1 || k < 1 || k > 64 ||
        n_frags < 1 || n_collector < 1)
        return GRAAL_E_ARG;
    if (!h->hs) h->hs = new HostStep();
    HostStep& S = *h->hs;
    S.n_bins = n_bins; S.k = k; S.n = n_frags;
    S.xk.assign(xk, xk + (size_t)n_bins * k);
    S.pk.resize((size_t)n_bins * k);
    S.nnz_row.assign(n_bins, 0);
    for (int b = 0; b < n_bins; b++)
        for (int j = 0; j < k; j++) {
            const float v = pk[(size_t)b * k + j];
            S.pk[(size_t)b * k + j] = (double)v;
            S.nnz_row[b] += v != 0.0f;
            if (xk[(size_t)b * k + j] < 0 || xk[(size_t)b * k + j] >= n_bins) return fail(h, GRAAL_E_ARG, "proposal tables: xk out of range");
        }
    S.id_d.assign(id_d, id_d + n_frags);
    S.disp.assign(dispatcher, dispatcher + 2 * (size_t)n_bins);
    S.coll.assign(collector, collector + n_collector);
    for (int b = 0; b < n_bins; b++)
        if (S.disp[2 * b] < 0 || S.disp[2 * b + 1] > n_collector || S.disp[2 * b + 1] < S.disp[2 * b]) return fail(h, GRAAL_E_ARG, "proposal tables: dispatcher out of range");
    for (int i = 0; i < n_collector; i++) if (S.coll[i] < 0 || S.coll[i] >= n_frags) return fail(h, GRAAL_E_ARG, "proposal tables: collector out of range");
    for (int f = 0; f < n_frags; f++) if (S.id_d[f] < 0 || S.id_d[f] >= n_bins) return fail(h, GRAAL_E_ARG, "proposal tables: id_d out of range");
    S.is_dup.assign(dup_bin_flags, dup_bin_flags + n_bins);
    S.black.assign(black_frag_flags, black_frag_flags + n_frags);
    S.ready = true;
    return GRAAL_OK;
}

// return_neighbours (cuda_lib_gl.py:2295-2331 as graal_amd/sampler.py has it).  false = leave it to the Python path (nothing drawn)
static bool hs_neighbours(HostStep& S, MtState* mt, int fA, int delta0, std::vector<int>& out)
{
    const int ori = S.id_d[fA], k = S.k;
    const int delta = std::min(10, delta0);                       // n_neighbors = 10 (cuda_lib_gl.py:444)
    const int size = std::min(delta, S.nnz_row[ori]);
    double pl[64];
    double tot0 = 0.0, mn = 0.0;
    int npos = 0;
    for (int j = 0; j < k; j++) { pl[j] = S.pk[(size_t)ori * k + j]; tot0 += pl[j]; mn = j ? std::min(mn, pl[j]) : pl[j]; npos += pl[j] > 0.0; }
    if (size < 1 || size > k || !(fabs(tot0 - 1.0) < 1e-4) || mn < 0.0 || npos < size) return false;
    // RandomState.choice(a, size, replace=False, p=p): numpy's legacy algorithm
    int found[64], n_found = 0;
    double cdf[64], x[64];
    while (n_found < size) {
        const int need = size - n_found;
        for (int i = 0; i < need; i++) x[i] = mt_double(mt);
        for (int i = 0; i < n_found; i++) pl[found[i]] = 0.0;
        double acc = 0.0;
        for (int j = 0; j < k; j++) { acc += pl[j]; cdf[j] = acc; }
        const double tot = cdf[k - 1];
        for (int j = 0; j < k; j++) cdf[j] = cdf[j] / tot;
        for (int i = 0; i < need; i++) {
            const int idx = (int)(std::upper_bound(cdf, cdf + k, x[i]) - cdf);   // bisect_right
            bool seen = false;
            for (int f = 0; f < n_found; f++) seen = seen || found[f] == idx;
            if (!seen && idx < k) found[n_found++] = idx;
        }
    }
    out.clear();
    if (S.is_dup[ori]) {   // the other copies of a repeated fragment (np.setdiff1d: sorted, unique)
        std::vector<int> c(S.coll.begin() + S.disp[2 * ori], S.coll.begin() + S.disp[2 * ori + 1]);
        std::sort(c.begin(), c.end());
        c.erase(std::unique(c.begin(), c.end()), c.end());
        for (int v : c) if (v != fA) out.push_back(v);
    }
    for (int i = 0; i < size; i++) {   // every copy of a proposed bin
        const int b = S.xk[(size_t)ori * k + found[i]];
        for (int r = S.disp[2 * b]; r < S.disp[2 * b + 1]; r++) out.push_back(S.coll[r]);
    }
    size_t w = 0;
    for (size_t i = 0; i < out.size(); i++) if (!S.black[out[i]]) out[w++] = out[i];
    out.resize(w);
    return true;
}

// select_move (cuda_lib_gl.py:1898-1947) for F_t == 1.  Returns the sampled index, or -1 = leave it to the Python path
// (nothing drawn).  score: n values.
static int hs_select(MtState* mt, const double* score, int n, int n_tmp)
{
    if (n < 1 || n > 128 * N_OPS) return -1;   // (graal_step takes up to 128 neighbours)
    int id_max = 0;
    bool any_nan = false;
    double mn = score[0];
    for (int i = 0; i < n; i++) {
        if (score[i] != score[i]) { if (!any_nan) id_max = i; any_nan = true; }
        if (!any_nan && score[i] > score[id_max]) id_max = i;
        if (score[i] < mn) mn = score[i];
    }
    if (any_nan) return id_max;            // (numpy: argmax = first NaN, and nothing left to sample from)
    std::vector<double> f(n);
    for (int i = 0; i < n; i++) f[i] = score[i] - mn;
    for (int i = n_tmp; i < n; i += n_tmp) f[i] = 0.0;            // remove extra pop
    for (int i = n_tmp + 1; i < n; i += n_tmp) f[i] = 0.0;        // remove extra flip
    double mx = f[0];
    for (int i = 1; i < n; i++) if (f[i] > mx) mx = f[i];
    const double shift = mx - 30.0;
    std::vector<int> ids;
    std::vector<double> sub;
    for (int i = 0; i < n; i++) {
        double v = f[i] - shift;
        if (v < 0.0) v = 0.0;
        if (v > 0.0) { ids.push_back(i); sub.push_back(v); }
    }
    if (ids.size() <= 1) return id_max;
    const long m = (long)sub.size();
    double s1 = np_sum(sub.data(), m);
    for (long i = 0; i < m; i++) sub[i] /= s1;
    // (np.power(sub, 1 / F_t) with F_t == 1: x ** 1.0 == x)
    double s2 = np_sum(sub.data(), m);
    for (long i = 0; i < m; i++) sub[i] /= s2;
    // RandomState.choice(m, 1, p=sub): one uniform, inverse CDF
    std::vector<double> cdf(m);
    double acc = 0.0;
    for (long i = 0; i < m; i++) { acc += sub[i]; cdf[i] = acc; }
    const double tot = cdf[m - 1];
    if (!(fabs(tot - 1.0) < 1e-9)) return -1;
    for (long i = 0; i < m; i++) cdf[i] /= tot;
    const MtState keep = *mt;
    const double u = mt_double(mt);
    long idx = (long)(std::upper_bound(cdf.begin(), cdf.end(), u) - cdf.begin());
    if (idx >= m) { *mt = keep; return -1; }   // (cannot happen -- cdf[m - 1] is 1 -- but -1 promises that nothing was drawn)
    return ids[(size_t)idx];
}

// second half of a step: score the candidates of the neighbours drawn by graal_step, sample, commit.  Returns GRAAL_STEP_DONE,
// GRAAL_STEP_SELECT or 16 + a GRAAL_E_* code (error codes 1 .. 3 must not be taken for PAUSED / FALLBACK / SELECT)
#define CK16(call) do { const hipError_t e16_ = (call); if (e16_ != hipSuccess) { h->err = hipGetErrorString(e16_); return 16 + GRAAL_E_HIP; } } while (0)
static int hs_finish(graal_ctx* h, MtState* mt, double likelihood_t, int want_dist, graal_step_out* out, bool deferred = false,
                     bool full_inside = false, bool carry_corr = false /* flag 16 */, bool total_is_full = false /* the caller's total is a full evaluation of this layout */)
{
    if (!h->hs) return 16 + fail(h, GRAAL_E_STATE, "graal_upload_proposal_tables first");
    HostStep& S = *h->hs;
    const int K = (int)S.nb.size();
    S.paused = false;                                  // (whatever happens below, the step is no longer waiting for graal_step_finish)
    if (K < 1 || K > 128) return 16 + fail(h, GRAAL_E_STATE, "graal_step_finish: no step in progress");
    struct Drop { HostStep& s; bool keep = false; ~Drop() { if (!keep) s.nb.clear(); } } drop{S};   // an error leaves no stale proposal behind
    if (full_inside) {
        // the step's total is due for a full re-evaluation (the reference re-evaluates every step, cuda_lib_gl.py:1828-1848): its kernels
        // go out on a stream of their own, behind the relabel, and run NEXT TO the scoring kernels -- which are latency, not throughput
        CK16(hipEventRecord(h->ev_full, h->stream));
        CK16(hipStreamWaitEvent(h->fstream, h->ev_full, 0));
        const int rc = full_launch(h, h->fstream);
        if (rc) return 16 + rc;
    }
    long long q[128 * N_OPS], qc[128 * N_OPS];
    for (int k0 = 0; k0 < K; k0 += MAXK) {
        const int kk = std::min(MAXK, K - k0);
        const int rc = eval_sync(h, S.fA, S.nb.data() + k0, kk, S.max_id, h->nccl_comm ? h->n_rank : (h->x_host ? h->x_rank : 0),
                                 h->nccl_comm ? h->n_world : (h->x_host ? h->x_world : 1), q + k0 * N_OPS, qc + k0 * N_OPS);
        if (rc) { if (full_inside) { int64_t dump[2]; (void)full_collect(h, h->fstream, dump); } return 16 + rc; }
    }
    if (full_inside) {
        int64_t fq[2];
        int rc = full_collect(h, h->fstream, fq);
        if (rc) return 16 + rc;
        if (h->x_host) { rc = full_exchange(h, fq); if (rc) return 16 + rc; }   // several ranks: the contacts' part is the sum over their shards
        // (float(q0 + q1) / Q_SCALE of the Python path: the integer sum is exact, one conversion, one division by a power of two)
        likelihood_t = fq[0] == Q_BAD ? (double)NAN : (double)(fq[0] + fq[1]) / Q_SCALE;
        out->full_likelihood = likelihood_t;
    }
    const double t2 = S.timing ? hs_now() : 0.0;
    if (deferred) {   // the statistics of the layout this step started from: published long ago, read now
        const int rc = begin_step_collect(h, out->stats, &out->max_id);
        if (rc) return 16 + rc;
        S.max_id = out->max_id;
    }
    {   // The total carried from the last step's accepted candidate lacks what that commit did to the pixels no delta contains -- the moved bins'
        // OWN pixels (k_apply: own_pixel_q) --, which arrived with this layout's statistics.  Flag 16: add it, i.e. start from the full
        // likelihood of this layout as the reference does (cuda_lib_gl.py:1828-1848) without evaluating it; a full evaluation (flag 8, or
        // the caller's after a pause) supersedes it; a correction that is unknown -- a layout not one commit away from the last, a term out
        // of range -- is replaced by the evaluation itself.
        int64_t cq = 0;
        int32_t cv = 0;
        // (a step that starts from a full evaluation -- inside the step, or the caller's behind a pause -- voids whatever has accumulated:
        // the commits of an explode_genome in front of the first step, the last step's)
        if (carry_corr && (full_inside || total_is_full)) (void)graal_take_carry_correction(h, nullptr, nullptr);
        else if (carry_corr) (void)graal_take_carry_correction(h, &cq, &cv);
        if (carry_corr && !full_inside && !total_is_full) {
            if (cv) likelihood_t += (double)cq / Q_SCALE;
            else {
                int64_t fq[2];
                h->rc_carry_repairs += 1;
                int rc = graal_eval_full_q(h, fq);
                if (rc) return 16 + rc;
                // (several ranks over the host exchange: every rank is here at the same step -- the correction is the same number on all of
                // them -- and the contacts' part is the sum over their shards, as for the in-step evaluation above)
                if (h->x_host) { rc = full_exchange(h, fq); if (rc) return 16 + rc; }
                likelihood_t = fq[0] == Q_BAD ? (double)NAN : (double)(fq[0] + fq[1]) / Q_SCALE;
                out->full_likelihood = likelihood_t;
            }
        }
    }
    for (int i = 0; i < K * N_OPS; i++) {
        const double d = q_value(q[i], qc[i]);
        out->scores[i] = d + likelihood_t;
    }
    const double t3 = S.timing ? hs_now() : 0.0;
    const int pos_before = mt->pos;
    const int pick = S.force_select ? -1 : hs_select(mt, out->scores, K * N_OPS, N_OPS);
    if (S.trace) fprintf(stderr, "[step] fA %d K %d deferred %d pick %d mt.pos %d -> %d max_id %d seq %lld spin_ok %d\n", S.fA, K, (int)deferred, pick, pos_before, mt->pos, S.max_id, h->seq, (int)h->spin_ok);
    const double t4 = S.timing ? hs_now() : 0.0;
    if (pick < 0) return GRAAL_STEP_SELECT;   // (the neighbours were drawn and the scores are in out->scores; nothing was drawn for the selection:
                                              // the caller selects and commits.  NOT GRAAL_STEP_FALLBACK, which promises that nothing was drawn at all)
    out->sample_out = pick;
    out->id_f_sampled = S.nb[pick / N_OPS];
    out->op_sampled = pick % N_OPS;
    out->o = out->scores[pick];
    int rc = graal_apply_move(h, S.fA, out->id_f_sampled, out->op_sampled, S.max_id, nullptr);
    if (rc) return 16 + rc;
    if (S.timing && deferred) { S.t_acc[2] += t2 - S.t_mark; S.t_acc[3] += t3 - t2; S.t_acc[4] += t4 - t3; S.t_acc[5] += hs_now() - t4; S.t_n += 1; }
    out->dist_half_units = 0;
    if (want_dist) { rc = graal_genome_distance(h, &out->dist_half_units); if (rc) return 16 + rc; }
    return GRAAL_STEP_DONE;   // (the proposal is dropped with `drop`)
}

/* One MCMC step of step_max_likelihood (cuda_lib_gl.py:1793-1980) for a fragment that is not blacklisted: relabel + statistics,
 * neighbour proposal, [pause], candidate scores, score post-processing + sampling, commit, [genome distance].
 * mt_state: address of the caller's numpy MT19937 state.  flags: bit 0 = pause after the proposal when circular contigs exist
 * now or existed at the previous step (`prev_circ`) -- the caller re-evaluates the full likelihood and continues with
 * graal_step_finish; bit 1 = pause always; bit 2 = genome distance after the commit.
 * Returns GRAAL_STEP_DONE (0), GRAAL_STEP_PAUSED (1: out->stats / neighbours valid, call graal_step_finish),
 * GRAAL_STEP_FALLBACK (2: nothing drawn, nothing committed -- take the step through the individual entry points), or
 * 16 + a GRAAL_E_* code. */
int graal_step(graal_ctx* h, void* mt_state, int32_t fA, int32_t delta, double likelihood_t, int32_t flags, int32_t prev_circ,
               graal_step_out* out)
{
    if (!h || !mt_state || !out) return 16 + GRAAL_E_ARG;
    if (!h->hs || !h->hs->ready) return 16 + fail(h, GRAAL_E_STATE, "graal_upload_proposal_tables first");
    HostStep& S = *h->hs;
    if (!h->have_frags) return 16 + fail(h, GRAAL_E_STATE, "no fragments uploaded");
    if (fA < 0 || fA >= S.n || S.n != h->n) return 16 + fail(h, GRAAL_E_ARG, "graal_step: fA out of range");
    MtState* mt = (MtState*)mt_state;
    if (mt->pos < 0 || mt->pos > 624) return 16 + fail(h, GRAAL_E_ARG, "graal_step: not an MT19937 state");
    S.paused = false;
    if (S.black[fA]) { S.nb.clear(); return GRAAL_STEP_FALLBACK; }   // nothing is proposed for a blacklisted fragment (cuda_lib_gl.py:1962-1978): the caller's branch
    const double t0 = S.timing ? hs_now() : 0.0;
    const bool defer = (flags & 3) == 0;
    int rc = h->begin_launched ? GRAAL_OK : begin_step_launch(h, defer);
    if (rc) return 16 + rc;
    const double t1 = S.timing ? hs_now() : 0.0;
    // the proposal is drawn while the relabel runs
    MtState keep = *mt;
    if (!hs_neighbours(S, mt, fA, delta, S.nb) || S.nb.empty() || S.nb.size() > 128) { *mt = keep; S.nb.clear(); return GRAAL_STEP_FALLBACK; }
    std::sort(S.nb.begin(), S.nb.end());
    if (S.trace) fprintf(stderr, "[step] fA %d proposal drawn: mt.pos %d -> %d, %d neighbours, first %d\n", fA, keep.pos, mt->pos, (int)S.nb.size(), S.nb[0]);
    S.fA = fA;
    out->n_neighbours = (int32_t)S.nb.size();
    for (size_t i = 0; i < S.nb.size(); i++) out->neighbours[i] = S.nb[i];
    if (defer) {
        // nobody needs the statistics before the scores: the scoring kernels go out right behind the relabel, the host waits once
        begin_step_assume(h);
        S.max_id = -1;
        if (S.timing) { S.t_mark = hs_now(); S.t_acc[0] += t1 - t0; S.t_acc[1] += S.t_mark - t1; }
        return hs_finish(h, mt, likelihood_t, (flags & 4) != 0, out, true, false, (flags & 16) != 0);   // (GRAAL_STEP_DONE / _FALLBACK, or 16 + an error code)
    }
    rc = graal_begin_step(h, out->stats, &out->max_id);
    if (rc) { *mt = keep; S.nb.clear(); return 16 + rc; }
    S.max_id = out->max_id;
    if ((flags & 2) || ((flags & 1) && (out->stats[6] != 0 || prev_circ != 0))) {
        // a full re-evaluation is due: inside the step (flag 8; with an exchange attached the ranks' contact parts are summed through
        // it, full_exchange) or by the caller
        if ((flags & 8) && !(h->nccl_comm && h->n_world > 1)) return hs_finish(h, mt, likelihood_t, (flags & 4) != 0, out, false, true, (flags & 16) != 0);
        S.paused = true;
        return GRAAL_STEP_PAUSED;
    }
    return hs_finish(h, mt, likelihood_t, (flags & 4) != 0, out, false, false, (flags & 16) != 0);
}

int graal_step_finish(graal_ctx* h, void* mt_state, double likelihood_t, int32_t flags, graal_step_out* out)
{
    if (!h || !mt_state || !out) return 16 + GRAAL_E_ARG;
    if (!h->hs || !h->hs->paused) return 16 + fail(h, GRAAL_E_STATE, "graal_step_finish: no paused step (graal_step did not return GRAAL_STEP_PAUSED)");
    MtState* mt = (MtState*)mt_state;
    if (mt->pos < 0 || mt->pos > 624) return 16 + fail(h, GRAAL_E_ARG, "graal_step_finish: not an MT19937 state");
    return hs_finish(h, mt, likelihood_t, (flags & 4) != 0, out, false, false, (flags & 16) != 0, true);   // (the caller re-evaluated behind the pause)
}

/* A run of MCMC steps in one call (the inner loop of start_EM, cuda_lib_gl.py:2196-2220: one step_max_likelihood per fragment of the
 * shuffled list): graal_step for ids[0], ids[1], ... with the carried total and the circular-contig count handed from step to step.
 * Stops BEHIND the first step that does not end GRAAL_STEP_DONE (its state is in `out`: the caller finishes it exactly as after
 * graal_step) or whose score is not finite (the caller re-evaluates).  rows[i][GRAAL_STEPS_ROW] of every step that ended DONE:
 * o, contigs, shortest, total bp / contigs, longest, op, fragment, genome-distance half units, circular contigs, stale pastes. */
int graal_steps(graal_ctx* h, void* mt_state, const int32_t* ids, int32_t n, int32_t delta, double likelihood_t, int32_t flags,
                int32_t prev_circ, double* rows, int32_t* n_done, graal_step_out* out)
{
    if (!h || !mt_state || !ids || !rows || !n_done || !out || n < 0) return 16 + GRAAL_E_ARG;
    *n_done = 0;
    int rc = GRAAL_STEP_DONE;
    for (int i = 0; i < n; i++) {
        rc = graal_step(h, mt_state, ids[i], delta, likelihood_t, flags, prev_circ, out);
        if (rc != GRAAL_STEP_DONE) return rc;
        double* r = rows + (size_t)GRAAL_STEPS_ROW * (size_t)i;
        r[0] = out->o; r[1] = (double)out->stats[0]; r[2] = (double)out->stats[5];
        r[3] = (double)out->stats[3] / (double)out->stats[2];     // (float(st[3]) / float(st[2]) of the Python path: both exact in float64)
        r[4] = (double)out->stats[4]; r[5] = (double)out->op_sampled; r[6] = (double)out->id_f_sampled;
        r[7] = (double)out->dist_half_units; r[8] = (double)out->stats[6]; r[9] = (double)out->stats[7];
        *n_done = i + 1;
        likelihood_t = out->o;
        prev_circ = (int32_t)out->stats[6];
        if (!std::isfinite(out->o)) break;
    }
    return rc;
}

/* explode_genome (cuda_lib_gl.py:1539-1556): every fragment 0 .. n-1 in index order is ejected from its contig -- the contig relabel in front of
 * each (modify_gl_cuda_buffer), then the commit of candidate 0 of the pair (fragment, 0) (test_copy_struct(i, 0, 0, max_id)) -- as ONE call: the
 * loop's two calls per fragment were 24 us of Python and ctypes each time (40,000 fragments: a second).  *n_stale = the unwritten-paste
 * fragments the relabels reported (graal_begin_step's stats[7], summed).  The last commit is left pending, as after the loop. */
int graal_explode(graal_ctx* h, int64_t* n_stale)
{
    if (!h) return GRAAL_E_ARG;
    if (!h->have_frags) return fail(h, GRAAL_E_STATE, "no fragments uploaded");
    long long stale = 0;
    for (int i = 0; i < h->n; i++) {
        int64_t st[8];
        int32_t max_id = 0;
        int rc = graal_begin_step(h, st, &max_id);
        if (rc) return rc;
        stale += st[7];
        rc = graal_apply_move(h, i, 0, 0, max_id, nullptr);
        if (rc) return rc;
    }
    if (n_stale) *n_stale = stale;
    return GRAAL_OK;
}

/* test hooks of the host logic (CPU-only: no device call): numpy's sum, the neighbour draw and the move sampling */
double graal_host_np_sum(const double* a, int64_t n) { return np_sum(a, (long)n); }
int graal_host_select_move(void* mt_state, const double* score, int32_t n, int32_t n_tmp) { return hs_select((MtState*)mt_state, score, n, n_tmp); }
int graal_host_neighbours(graal_ctx* h, void* mt_state, int32_t fA, int32_t delta, int32_t* out, int32_t cap)
{
    if (!h || !h->hs || !h->hs->ready || !mt_state || !out) return -1;
    std::vector<int> nb;
    if (!hs_neighbours(*h->hs, (MtState*)mt_state, fA, delta, nb)) return -2;
    if ((int)nb.size() > cap) return -3;
    for (size_t i = 0; i < nb.size(); i++) out[i] = nb[i];
    return (int)nb.size();
}
