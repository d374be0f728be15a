/*
 * graal_hip.h -- C ABI of libgraal_hip.so, the MI355X (gfx950) engine for GRAAL's per-move likelihood
 * scan.  Plain pointers and sizes only; loaded from Python with ctypes (graal_amd/lib.py).
 *
 * What it replaces in the reference: the 23 PyCUDA entry points that cuda_lib_gl.sampler looks up in
 * kernels3.cu (cuda_lib_gl.py:378-402) and the device-side state they operate on
 * (cuda_lib_gl.py:194-360).  Each function below cites the reference call sequence it stands for.
 *
 * Conventions
 *   - every function returns 0 on success, a GRAAL_E_* code otherwise; graal_last_error() gives text.
 *     Nothing aborts; nothing falls back to the CPU (no GPU => GRAAL_E_HIP at graal_create).
 *   - host pointers are borrowed for the duration of the call; "d_" pointers are DEVICE pointers
 *     (e.g. the data_ptr() of a torch tensor) and `stream` is a hipStream_t passed as void*.
 *   - a handle is not thread safe; one host thread per handle, one handle per GPU / process.
 *   - log-likelihood values cross the boundary either as double or as int64 fixed point
 *     ("Q" = value * 2^GRAAL_Q_BITS, rounded per term).  Q sums are order independent, hence
 *     bit-reproducible for any sharding of the contact list over GPUs.
 */
#ifndef GRAAL_HIP_H
#define GRAAL_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRAAL_ABI_VERSION 7
#define GRAAL_N_OPS 13        /* candidates per (fA, fB): cuda_lib_gl.py:112 n_tmp_struct */
#define GRAAL_MAX_NEIGHBOURS 10 /* neighbours scored by one scan pass (the reference proposes at most n_neighbors = 10, cuda_lib_gl.py:444) */
#define GRAAL_Q_BITS 30
#define GRAAL_N_FIELDS 14     /* struct frag, kernels3.cu:9-24 */

enum {
    GRAAL_OK = 0,
    GRAAL_E_ARG = 1,          /* bad argument / shape */
    GRAAL_E_HIP = 2,          /* HIP runtime error (incl. no device) */
    GRAAL_E_STATE = 3,        /* call order (e.g. eval before upload) */
    GRAAL_E_UNSUPPORTED = 4   /* e.g. more sub-fragments than the scan's LDS bitmap holds */
};

typedef struct graal_ctx graal_ctx;

/* device bring-up; replaces cuda_gl_init + loadProgram (main_gl.py:690-702, cuda_lib_gl.py:708-713) */
int graal_create(int device, graal_ctx** out);
void graal_destroy(graal_ctx* h);
const char* graal_last_error(const graal_ctx* h);
int graal_abi_version(void);

/* Rippe model parameters: 8 floats kuhn,lm,c1,slope,d,d_max,fact,v_inter (kernels3.cu:26-35);
 * replaces memcpy_htod(gpu_param_simu, ...) cuda_lib_gl.py:1294, 2073 */
int graal_set_params(graal_ctx* h, const float* param8);

/* sub-fragment tables of the n_bins unique bins (simulation_loader.py:673-704):
 * sub_id[n_bins][4] = (x,y,z,w=n_sub), sub_len_kb[n_bins][3], sub_accu[n_bins][3];
 * replaces the uploads of cuda_lib_gl.py:210-215 */
int graal_upload_subfrags(graal_ctx* h, const int32_t* sub_id, const float* sub_len_kb, const int32_t* sub_accu,
                          int32_t n_bins, int32_t n_sub_total, float n_frags_per_bins);

/* Repeated bins (allow_repeats; simulation_loader.py:182-280, kernels3.cu:2915-2930, 3356-3380).  Call after
 * graal_upload_subfrags and before graal_upload_contacts / graal_upload_frags (which may then hold more fragments than
 * bins).  dup_bins[n_dup]: the repeated unique bins; dispatcher[n_bins][2], collector[n_collector]: the fragment copies of
 * every bin (frag_dispatcher / collector_id_repeats); obs_rows[n_dup][3][n_sub_total]: the rows of the sub-level
 * observation matrix of the repeated bins' sub-fragments (row of slot a of dup_bins[i] at (i*3 + a)*n_sub_total; unused
 * slots ignored).  Every pixel of a repeated bin is priced densely from these rows, summed over the ACTIVE copies; the
 * contact list must then contain no contact of a repeated bin.  n_dup = 0 switches repeats off. */
int graal_upload_repeats(graal_ctx* h, const int32_t* dup_bins, int32_t n_dup, const int32_t* dispatcher, const int32_t* collector,
                         int32_t n_collector, const float* obs_rows);

/* this rank's shard of the sub-level contact list in COO form, row < col, any order (sorted by
 * (row, col) is fastest); counts are the observed contacts.  Replaces the dense S x S upload of
 * cuda_lib_gl.py:194 (which simulation_loader.py:81-82 densifies first). */
int graal_upload_contacts(graal_ctx* h, const int32_t* row, const int32_t* col, const int32_t* count, int64_t nnz);
/* same with float32 counts -- the type of the reference's observation matrix; needed when blacklisted bins carry the
 * non-integer fill value mean_value_trans (cuda_lib_gl.py:161-172) */
int graal_upload_contacts_f32(graal_ctx* h, const int32_t* row, const int32_t* col, const float* count, int64_t nnz);

/* fragment layout: 14 arrays of n int32 in the order of struct frag; replaces GPUStruct.copy_to_gpu
 * (cuda_lib_gl.py:264-265) / copy_from_gpu (gpustruct.py:175) */
int graal_upload_frags(graal_ctx* h, const int32_t* const soa[GRAAL_N_FIELDS], int32_t n);
int graal_download_frags(graal_ctx* h, int32_t* const soa[GRAAL_N_FIELDS]);

/* contig relabel: id_c <- rank of the contig when contigs are sorted by (l_cont, old label);
 * replaces modify_gl_cuda_buffer's compute half + gl_update_pos (cuda_lib_gl.py:1697-1722,
 * kernels3.cu:3848-3851).  *max_id = n_contigs - 1.  Also rebuilds the position index. */
int graal_relabel_contigs(graal_ctx* h, int32_t* max_id);
/* start of an MCMC step in ONE call and one synchronisation: the statistics of graal_layout_stats (stats[0..6], taken
 * before the relabel, which does not change them) + stats[7] = fragments that hit the unwritten paste branch in the
 * commits since the previous call, then the relabel of graal_relabel_contigs.  stats may be NULL. */
int graal_begin_step(graal_ctx* h, int64_t stats[8], int32_t* max_id);
/* The launching half of graal_begin_step alone: returns at once.  A caller with host work that does not need the statistics
 * (drawing the proposal) calls this, does that work, then calls graal_begin_step, which only waits.  Optional. */
int graal_begin_step_launch(graal_ctx* h);

/* layout statistics step_max_likelihood returns (cuda_lib_gl.py:1809-1816):
 * out[0]=n_contigs out[1]=sum(l_cont) out[2]=#(start_bp==0) out[3]=sum(l_cont_bp | start_bp==0)
 * out[4]=max(l_cont) out[5]=min(l_cont) out[6]=#(circ==1) out[7]=reserved(0) */
int graal_layout_stats(graal_ctx* h, int64_t out[8]);

/* full log-likelihood of the current layout over this rank's contact shard, as Q fixed point:
 *   q_out[0] = sum over contacts of ob*log(ex) - log-factorial term   (shard dependent)
 *   q_out[1] = - expected mass of all pixels                          (identical on every rank)
 * logL = (sum over ranks of q_out[0] + q_out[1]) / 2^GRAAL_Q_BITS.
 * Replaces evaluate_likelihood + gpuarray.sum (cuda_lib_gl.py:543-631, 1473-1509). */
int graal_eval_full_q(graal_ctx* h, int64_t q_out[2]);
/* The same behind new parameters and the relabel of the current layout, with ONE wait: graal_set_params(param8) (skipped when NULL; the
 * parameters stay in force) + graal_begin_step (stats[8], *max_id as there; either may be NULL) + graal_eval_full_q.  What a
 * nuisance-parameter step does between two MCMC steps: compute_likelihood_4_nuisance, cuda_lib_gl.py:1986-2017. */
int graal_eval_full_params(graal_ctx* h, const float* param8, int64_t stats[8], int32_t* max_id, int64_t q_out[2]);

/* delta log-likelihood of the 13 candidates of each of K (<= GRAAL_MAX_NEIGHBOURS) neighbours fB[k]
 * of fA, over this rank's contact shard; rank / world shard the expected-mass work.
 * Writes to the DEVICE buffer d_q_out -- 3 * GRAAL_MAX_NEIGHBOURS * 13 int64: the K*13 Q values at [0, K*13), their coarse
 * companions (terms of 2^31 log-likelihood units and more, rounded to whole units; zero but for a rare candidate) at
 * [GRAAL_MAX_NEIGHBOURS*13, ... + K*13) and at [2*GRAAL_MAX_NEIGHBOURS*13, ... + K*13) 1 for a candidate that met a term that was not
 * finite (its Q value is then 0), else 0 -- on `stream` (asynchronous; a NULL stream means the engine's own stream, NOT the HIP
 * default stream -- pass an explicit stream to order a collective after it): sum all three over ranks (one RCCL all-reduce of the
 * whole buffer); a candidate's value is NaN where the flags' sum is not 0, else coarse + Q / 2^GRAAL_Q_BITS.  A rank whose step FAILED (a kernel of
 * the step gave up waiting for another, a work list overflowed) adds 2^32 to the FIRST flag word: a summed first flag >= 2^32 means "no scores
 * this step, on any rank" -- every rank sees it, none may use the sums (graal_eval_candidates_x with graal_attach_rccl returns an error then).
 * max_id must be the value returned by graal_relabel_contigs for the current layout.
 * Replaces new_perform_modificationS + 13 x sub_compute_likelihood per neighbour
 * (cuda_lib_gl.py:2392-2546). */
int graal_eval_candidates_q(graal_ctx* h, int32_t fA, const int32_t* fB, int32_t K, int32_t max_id, int32_t rank,
                            int32_t world, int64_t* d_q_out, void* stream);
/* single-GPU convenience: same, synchronous, into a host buffer of K*13 doubles (the last block of the last kernel
 * publishes the sums to pinned host memory; the call spins on that instead of a copy + stream synchronise) */
int graal_eval_candidates(graal_ctx* h, int32_t fA, const int32_t* fB, int32_t K, int32_t max_id, double* delta);

/* Node-local exchange of the per-rank Q vectors through pinned HOST memory (one process per GPU, all on one node).
 * The reference has no distributed path (its only hint: "place where we want to spread the workload accross the network!",
 * cuda_lib_gl.py:1886).  The per-step message is K*13 int64 = 520 bytes: every rank's GPU already publishes its sums to
 * pinned host memory (graal_eval_candidates); with an exchange attached that memory is this rank's slot of a segment
 * shared by the ranks (e.g. an mmap of a /dev/shm file, page aligned, graal_exchange_bytes(world) bytes, zero filled,
 * which the library registers with HIP), and graal_eval_candidates_x returns the sum over all ranks' slots -- bit
 * identical on every rank and for every world size (int64 sums).  No collective kernel, no device->host copy; the
 * alternative is graal_eval_candidates_q + one RCCL all-reduce (graal_amd/sampler.py implements both).
 * graal_attach_exchange(segment = NULL) only reports the context's step sequence number in *seq_now; ranks attach with
 * seq_floor = the maximum over ranks so that all continue from the same number.  The segment must outlive the handle. */
int graal_exchange_bytes(int32_t world, int64_t* bytes);
int graal_attach_exchange(graal_ctx* h, void* segment, int64_t bytes, int32_t rank, int32_t world, int64_t seq_floor,
                          int64_t* seq_now);
/* Self-test of an attached exchange, collective: every rank calls phase 0 (its GPU writes `tag + rank` into a spare word of
 * its two slots), the ranks synchronise (any barrier), every rank calls phase 1 (its HOST must see every rank's tag; gives up
 * after 2 s with GRAAL_E_STATE).  graal_detach_exchange unregisters the segment (the caller then falls back to the RCCL
 * all-reduce of graal_eval_candidates_q). */
int graal_exchange_selftest(graal_ctx* h, int64_t tag, int32_t phase);
int graal_detach_exchange(graal_ctx* h);
/* The RCCL form of the exchange, driven by the library: graal_attach_rccl joins a communicator (rank 0 makes the 128-byte id with
 * graal_rccl_unique_id and hands it to the others -- torch.distributed's broadcast in graal_amd/sampler.py; librccl is dlopen'ed, no
 * link-time dependency); from then on every candidate evaluation of this handle -- graal_eval_candidates, _x, graal_step -- leaves the
 * rank's sums in a device buffer, sums them with ONE ncclAllReduce (3 * GRAAL_MAX_NEIGHBOURS * 13 int64) on the engine's stream and
 * publishes the total to pinned host memory behind it: collective and publication are on the GPU timeline, the host waits once.
 * Every rank must make the same calls.  Not together with graal_attach_exchange. */
int graal_rccl_unique_id(void* id128);
int graal_attach_rccl(graal_ctx* h, const void* id128, int32_t rank, int32_t world);
int graal_detach_rccl(graal_ctx* h);
/* synchronous, sharded: K*13 int64 Q sums over ALL ranks into host buffers (every rank must make the same call).  A candidate's value is
 * c_sum + q_sum / 2^30: q_sum holds the terms below 2^31 in fixed point (2^-30), c_sum -- zero but for a rare candidate -- the larger finite
 * ones rounded to whole log-likelihood units; q_sum == INT64_MIN exactly: a term was not finite (NaN). */
int graal_eval_candidates_x(graal_ctx* h, int32_t fA, const int32_t* fB, int32_t K, int32_t max_id, int64_t* q_sum, int64_t* c_sum);

/* Reference-arithmetic switches (default 0 = neither).
 * GRAAL_MODE_REF_TRANS_ACCU: trans pixels are priced with the reference's RF-count indexing of reversed bins
 *   (kernels3.cu:3155 / 3638: list_accu_data_i[i] = accu_sub_fi[limit_fi]) in the full evaluation and in strict deltas; only bins
 *   whose sub-fragments carry different RF counts see a difference.  The oracle's fix_trans_accu=False.
 * GRAAL_MODE_STRICT: graal_eval_candidates* price every candidate the way sub_compute_likelihood defines it
 *   (kernels3.cu:3259-3718): every pixel between two different bins of contig(fA) u contig(fB) again, from the float32 kb
 *   coordinates of the candidate layout -- including the pairs whose geometry the move does not change (the default path treats
 *   those as exactly unchanged; the reference's values for them move by float32 rounding noise, which on megabase contigs reaches
 *   ~1e-4 of logL).  O(m^2) per candidate: a validation mode.  Not available with repeated bins. */
#define GRAAL_MODE_REF_TRANS_ACCU 1
#define GRAAL_MODE_STRICT 2
int graal_set_mode(graal_ctx* h, int32_t flags);

/* In the synchronous single-GPU call, a step that leaves little work (short contigs) is finished by the last block of the
 * table kernel, which runs concurrently with the streaming kernel and waits for it; otherwise a third kernel finishes it.
 * Both give bit-identical sums.  enabled = 0 always uses the third kernel (the library does that by itself after noticing
 * that the two kernels do not run concurrently, e.g. under a profiler that serialises dispatches); 1 re-enables. */
int graal_set_finisher(graal_ctx* h, int32_t enabled);

/* commit candidate `op` of (fA, fB); replaces test_copy_struct (cuda_lib_gl.py:1156-1180).
 * *n_stale = fragments that hit the reference's unwritten paste branch (expected 0); NULL = do not wait for the
 * commit (the count is then reported by the next graal_begin_step).  The geometry index is stale until then.  The commit
 * kernel also publishes the statistics of the layout it writes, so the next graal_begin_step launches no statistics kernel. */
int graal_apply_move(graal_ctx* h, int32_t fA, int32_t fB, int32_t op, int32_t max_id, int32_t* n_stale);

/* Genome distance to the initial genome, dist_inter_genome (cuda_lib_gl.py:475-541: a Python loop over all fragments after
 * a full device->host copy, every step).  graal_upload_distance_ref takes the initial prev / next / ori (np_init_prev,
 * np_init_next, np_init_ori), the orientable flags and which fragments count (not repeated, not blacklisted), n entries each,
 * after graal_upload_frags; graal_genome_distance evaluates the CURRENT layout on the device and returns twice the sum of
 * the terms the reference subtracts (an integer): distance = (3 * n_counted - half_units / 2) / (3 * n_counted). */
int graal_upload_distance_ref(graal_ctx* h, const int32_t* init_prev, const int32_t* init_next, const int32_t* init_ori,
                              const int32_t* orientable, const uint8_t* counted, int32_t n);
int graal_genome_distance(graal_ctx* h, int64_t* half_units);

/* HIP event pairs around the streaming kernel of graal_eval_candidates*: enabled = n > 0 records a pair on every n-th
 * call (default 8; each pair costs a few microseconds of command-processor gaps on the step's critical path), 0 = off */
int graal_set_timing(graal_ctx* h, int32_t enabled);
/* duration (ms) of the streaming scan kernel of the last graal_eval_candidates* call: a pair of HIP events recorded
 * around it on the stream it ran on; out[1] = k_scan, the other entries are 0 (per-kernel times of k_prep / k_post come
 * from rocprofv3).  graal_scan_times returns the last n calls (a ring of 1024 event pairs). */
int graal_last_timing(graal_ctx* h, float out[4]);
int graal_scan_times(graal_ctx* h, int32_t n, float* out_ms);
/* the same for the tiled reference-arithmetic kernel (k_strict2) of the last n calls that launched it with an event pair (a ring of 256
 * pairs): the duration that prices its VALU roofline (bench.py: late_stage.roofline) */
int graal_strict_times(graal_ctx* h, int32_t n, float* out_ms);
/* reps < 0: MEDIAN duration of -reps ISOLATED replays (an event pair around each launch, the device idle in between: what a launch
 * costs on its own, without the other kernels of a step).  reps > 0:
 * average duration (ms) of the streaming scan kernel over `reps` back-to-back replays of the last call's scan between
 * two HIP events on the engine's stream (the replays count relevant contacts but queue nothing): the per-launch
 * event overhead of graal_last_timing is amortised away. */
int graal_time_scan(graal_ctx* h, int32_t K, int32_t reps, float* avg_ms);
/* counters of the last call: out[0]=contacts scanned out[1]=relevant (contact, neighbour) pairs
 * out[2]=queue slots (contacts with both ends in an affected contig; the few the third test rejects leave empty entries)
 * out[3]=mass work items (task x 64-fragment chunk) */
/* explode_genome (cuda_lib_gl.py:1539-1556) as one call: for every fragment i = 0 .. n-1 the contig relabel (graal_begin_step) and the commit of
 * candidate 0 of the pair (i, 0) (graal_apply_move(i, 0, 0, max_id): the fragment is ejected from its contig).  *n_stale (may be NULL) = the
 * unwritten-paste fragments the relabels reported, summed.  The last commit is left pending, as after the reference's loop. */
int graal_explode(graal_ctx* h, int64_t* n_stale);
int graal_last_counters(graal_ctx* h, int64_t out[4]);
/* What the commits since the last call did to the pixels NO candidate delta contains: with several sub-fragments per bin the reference's
 * sub_compute_likelihood revisits the pixels between DIFFERENT bins of contig(A) u contig(B) only (kernels3.cu:3356-3380), while a bin's
 * own pixel -- its sub-fragment pairs, evaluate_likelihood's on_diag pixels (:3213) -- moves with the bin's float32 coordinates and with its
 * contig's circular model.  A kernel next to the commit (k_own_corr, on a stream of its own) computes that difference for the bins the
 * commit moves (same terms, same roundings as the full evaluation), graal_begin_step collects it with the layout statistics, and *q_out
 * (Q: value x 2^30) is its sum over the graal_begin_step calls since the last take:  score of the accepted candidate + *q_out / 2^30 == the full likelihood of the committed layout (to the rounding of one term per
 * pair), without repeats.  *valid_out = 0: unknown -- a layout that is not one commit away from the last one, a term out of range,
 * repeats, a sharded list, or, with the trans-branch indexing (GRAAL_MODE_REF_TRANS_ACCU), a commit that mirrored a bin whose
 * sub-fragments carry different RF counts (that also changes the bin's trans pixels with every bin outside the two contigs) --: evaluate
 * instead.  q_out = NULL: DISCARD -- the caller
 * holds a full evaluation of the current layout, which accounts for every commit so far, the one whose statistics have not been collected
 * included.  The corrections are computed from the first take on (the first one reports "unknown"): a caller that never asks pays nothing.
 * No device access. */
int graal_take_carry_correction(graal_ctx* h, int64_t* q_out, int32_t* valid_out);
/* The observed counts of every bin's OWN sub-fragment pairs, own[3 * bin + {0, 1, 2}] = the contacts between the bin's sub-fragments in
 * data slots (0,1), (0,2), (1,2) (0 = none), for the correction above.  graal_upload_contacts builds this table from the list it is given;
 * a rank that holds a SHARD of the list hands in the table of the WHOLE list here (after its contacts): every rank then computes the same
 * correction, nothing is exchanged, and graal_step's flag 16 works over the host exchange as on one rank (an unknown correction is
 * replaced by a full evaluation whose contact part is summed over the ranks, like flag 8's).  Not with an RCCL communicator of several
 * ranks.  No-op at one sub-fragment per bin and with repeats. */
int graal_upload_own_obs(graal_ctx* h, const float* own, int32_t n_bins);
/* counters of the handle's whole life (no device access): out[0] = candidate evaluations started, out[1] = evaluations REPEATED behind
 * events because an in-kernel wait between two kernels of a step ran out (k_tm waiting for the scan's announcement, k_strict2 waiting for
 * k_gprep's completion word: the kernels were not resident together -- a profiler serialising dispatches; results are the same, the
 * engine stays with events from then on), out[2] = 1 while the in-kernel waits are still in use, out[3] / out[4] = launches of the tiled
 * reference-arithmetic kernel that followed k_gprep through the completion word / behind the event, out[5] = k_strict_flat launches,
 * out[6] = evaluations the table kernel handed to a finishing kernel through the host, out[7] = times its finisher gave up waiting for
 * the scan, out[8] = evaluations repeated with a longer unit list (the tiled kernel's list starts at a soft cap instead of its quadratic
 * worst case and grows when a step overflows it), out[9] = that list's capacity now (entries of 8 bytes), out[10] = graal_step calls with flag 16 whose own-pixel
 * correction was unknown and that evaluated the layout in full instead (graal_take_carry_correction), out[11] = 0 (reserved).
 * bench.py reports out[1] as `fallbacks`. */
int graal_run_counters(graal_ctx* h, int64_t out[12]);

/* ---- the sampler's per-step HOST logic behind the boundary (graal_amd/csrc/host_step.h) --------------------------------
 * What cuda_lib_gl.sampler.step_max_likelihood does on the host between its launches: return_neighbours
 * (cuda_lib_gl.py:2295-2331: RandomState.choice(xk, n, p=pk, replace=False), expansion to the copies of repeated bins,
 * blacklist filter), the score post-processing and the sampling of the move (cuda_lib_gl.py:1898-1947: duplicate eject / flip
 * entries zeroed, shift to max - 30, RandomState.choice(ids, 1, p=...)), then the commit (test_copy_struct, :1156).  Same
 * float64 operations in numpy's order, same draws from the CALLER'S generator: `mt_state` is the address of a numpy MT19937
 * state (uint32 key[624]; int pos -- RandomState._bit_generator.ctypes.state_address), advanced in place. */
typedef struct graal_step_out {
    int64_t stats[8];        /* graal_begin_step's statistics of the layout the step started from */
    int32_t max_id;
    int32_t n_neighbours;
    int32_t neighbours[128]; /* the proposal, sorted */
    int32_t sample_out;      /* index of the sampled candidate: neighbour * 13 + op */
    int32_t op_sampled, id_f_sampled;
    int32_t pad;
    double o;                /* its score (likelihood_t + delta) */
    int64_t dist_half_units; /* graal_genome_distance of the new layout (flag bit 2) */
    double scores[128 * GRAAL_N_OPS];
    double full_likelihood;  /* flag 8: the full likelihood of the layout the step started from, re-evaluated inside the step */
} graal_step_out;

enum { GRAAL_STEP_DONE = 0, GRAAL_STEP_PAUSED = 1, GRAAL_STEP_FALLBACK = 2, GRAAL_STEP_SELECT = 3 /* 16 + GRAAL_E_*: error */ };

/* setup_distri_frags' tables (cuda_lib_gl.py:2363-2390): xk[n_bins][k], pk[n_bins][k] (float32); id_d[n_frags] (fragment ->
 * bin), frag_dispatcher / collector_id_repeats, one flag per bin (repeated) and per fragment (blacklisted) */
int graal_upload_proposal_tables(graal_ctx* h, const int32_t* xk, const float* pk, int32_t n_bins, int32_t k, const int32_t* id_d,
                                 int32_t n_frags, const int32_t* dispatcher, const int32_t* collector, int32_t n_collector,
                                 const uint8_t* dup_bin_flags, const uint8_t* black_frag_flags);
/* One MCMC step for a fragment that is not blacklisted: relabel + statistics (graal_begin_step), proposal, candidate scores
 * (graal_eval_candidates / _x), sampling, commit (graal_apply_move), optionally the genome distance.
 * flags: 1 = pause after the proposal if circular contigs exist now or did at the previous step (`prev_circ`); 2 = pause
 * always; 4 = genome distance; 8 = instead of pausing, re-evaluate the full likelihood INSIDE the step, next to the scoring
 * kernels (out->full_likelihood replaces `likelihood_t`).  One rank, or several over the HOST exchange (graal_attach_exchange): every rank
 * evaluates its shard, the contacts' parts are summed through spare words of the exchange slots -- which requires that EVERY rank
 * passes flag 8 on the SAME steps (they do when they run the same sampler: the condition is a function of the layout and the step count).
 * With an RCCL communicator of several ranks attached the step PAUSES as with flag 1 / 2 and the caller supplies the total.  Returns GRAAL_STEP_DONE, GRAAL_STEP_PAUSED (the caller refreshes its total and calls
 * graal_step_finish -- valid only then), GRAAL_STEP_FALLBACK (a blacklisted fragment, or an unusual proposal numpy itself has to judge:
 * NOTHING drawn from the generator, nothing committed: take the step through the individual entry points), GRAAL_STEP_SELECT (the
 * neighbours are drawn and out->neighbours / out->scores / out->stats / out->max_id [/ out->full_likelihood] valid, but the selection
 * is left to the caller -- nothing drawn for it, nothing committed: select, then graal_apply_move) or 16 + an error code.
 * flag 16 = `likelihood_t` is the score of the LAST step's accepted candidate, carried over: add the own-pixel correction of that commit
 * (graal_take_carry_correction) so that the step starts from the full likelihood of its layout, as the reference's per-step
 * evaluate_likelihood does (cuda_lib_gl.py:1828-1848), without evaluating it (a full evaluation -- flag 8, or the caller's behind a pause --
 * supersedes the correction; an unknown correction is replaced by the evaluation itself, out->full_likelihood). */
int graal_step(graal_ctx* h, void* mt_state, int32_t fA, int32_t delta, double likelihood_t, int32_t flags, int32_t prev_circ,
               graal_step_out* out);
int graal_step_finish(graal_ctx* h, void* mt_state, double likelihood_t, int32_t flags, graal_step_out* out);
/* A run of steps in one call -- the inner loop of start_EM (cuda_lib_gl.py:2196-2220): graal_step for ids[0 .. n), the carried total
 * and the circular-contig count handed from step to step.  Stops behind the first step that does not end GRAAL_STEP_DONE (`out` holds
 * its state: finish it as after graal_step) or whose score is not finite.  *n_done = steps that ended DONE; rows[i][GRAAL_STEPS_ROW] =
 * o, contigs, shortest contig, total bp / contigs, longest contig, op, fragment, genome-distance half units, circular contigs,
 * stale pastes of step i.  Returns the last step's graal_step code. */
#define GRAAL_STEPS_ROW 10
int graal_steps(graal_ctx* h, void* mt_state, const int32_t* ids, int32_t n, int32_t delta, double likelihood_t, int32_t flags,
                int32_t prev_circ, double* rows, int32_t* n_done, graal_step_out* out);
/* test hooks of that logic; no device needed (a handle whose graal_create failed for lack of a GPU will do) */
double graal_host_np_sum(const double* a, int64_t n);
int graal_host_select_move(void* mt_state, const double* score, int32_t n, int32_t n_tmp);
int graal_host_neighbours(graal_ctx* h, void* mt_state, int32_t fA, int32_t delta, int32_t* out, int32_t cap);
/* estimate_max_dist_intra (optim_rippe_curve_update.py:117-135), which step_nuisance_parameters calls after three of its four
 * perturbations (cuda_lib_gl.py:2053, 2060, 2071): the distance at which the Rippe curve of p5 = (kuhn, lm, slope, d, A) meets val_inter --
 * scipy's fsolve = MINPACK hybrd from x0 = 500, restated for one unknown with the residual in C (graal_amd/csrc/host_fit.h; the case in
 * which MINPACK gives up and hands the start value back, SURVEY H6, included).  f32 != 0: the parameters are numpy float32 scalars, as the
 * nuisance step passes them.  *info = MINPACK's termination code.  No device needed. */
int graal_host_max_dist_intra(const double* p5, double val_inter, int32_t f32, double* x_out, int32_t* info_out);

#ifdef __cplusplus
}
#endif
#endif
