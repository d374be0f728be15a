"""TEST INFRASTRUCTURE -- the CPU oracle for the GRAAL hot path.  Not product code.

Two layers:

* :class:`DenseOracle` -- ctypes front-end to ``libgraal_oracle.so`` (``graal_oracle.c``), the C
  restatement of the reference's dense kernels (``kernels3.cu``).
* :class:`OracleSampler` -- a line-by-line Python-3 restatement of the host logic of
  ``cuda_lib_gl.sampler`` for the ``start_EM`` path (``cuda_lib_gl.py:448-541,841-954,1045-1048,
  1156-1180,1539-1556,1695-1722,1793-1980,2295-2331,2363-2390,2392-2546``) with every PyCUDA launch
  replaced by the matching :class:`DenseOracle` call.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.

Parity pin: see the header of ``graal_oracle.c``.  Decisions taken where the reference is
under-specified: every ``argsort`` is ``kind='stable'``; the numpy legacy
``RandomState`` is passed in explicitly (the reference uses the never-seeded global one).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgraal_oracle.so")

FIELDS = ("pos", "id_c", "start_bp", "len_bp", "circ", "id", "prev", "next", "l_cont", "l_cont_bp", "ori",
          "rep", "activ", "id_d")  # order of struct frag, kernels3.cu:9-24
N_TMP_STRUCT = 13  # cuda_lib_gl.py:112

_i32p = ctypes.POINTER(ctypes.c_int32)
_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_pp = ctypes.POINTER(_i32p)


def build(force=False):
    """Compile graal_oracle.c with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "graal_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgraal_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.or_evaluate_likelihood.restype = ctypes.c_double
        L.or_sub_compute_likelihood.restype = ctypes.c_double
        L.or_paste.restype = ctypes.c_int
        L.or_rippe.restype = ctypes.c_float
        L.or_rippe_circ.restype = ctypes.c_float
        L.or_lik.restype = ctypes.c_double
        _lib = L
    return _lib


def new_state(n):
    """Zero-initialised fragment SoA like the reference's collector slots (cuda_lib_gl.py:271-284)."""
    s = {k: np.zeros(n, dtype=np.int32) for k in FIELDS}
    s["ori"][:] = 1
    s["activ"][:] = 1
    return s


def copy_state(s):
    return {k: np.array(s[k], dtype=np.int32, copy=True) for k in FIELDS}


def _ptrs(s):
    arr = (_i32p * 14)()
    for i, k in enumerate(FIELDS):
        a = s[k]
        assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
        arr[i] = a.ctypes.data_as(_i32p)
    return arr


def _ip(a):
    return a.ctypes.data_as(_i32p)


class DenseOracle:
    """Dense data + the reference's kernels, one call per kernel launch."""

    def __init__(self, obs, sub_id, sub_len, sub_accu, dispatcher, collector, n_bins, nfpb, param,
                 fix_trans_accu=False):
        self.obs = np.ascontiguousarray(obs, dtype=np.float32)
        assert self.obs.ndim == 2 and self.obs.shape[0] == self.obs.shape[1]
        self.width = int(self.obs.shape[0])
        self.sub_id = np.ascontiguousarray(sub_id, dtype=np.int32).reshape(-1, 4)
        self.sub_len = np.ascontiguousarray(sub_len, dtype=np.float32).reshape(-1, 3)
        self.sub_accu = np.ascontiguousarray(sub_accu, dtype=np.int32).reshape(-1, 3)
        self.dispatcher = np.ascontiguousarray(dispatcher, dtype=np.int32).reshape(-1, 2)
        self.collector = np.ascontiguousarray(collector, dtype=np.int32)
        self.n_bins = int(n_bins)
        self.nfpb = np.float32(nfpb)
        self.set_param(param)
        self.fix_trans_accu = int(bool(fix_trans_accu))
        self.n_pix = self.n_bins * (self.n_bins - 1) // 2 + self.n_bins

    def set_param(self, param):
        self.param = np.ascontiguousarray(np.asarray(param, dtype=np.float32).reshape(8))

    # -- mutation kernels -------------------------------------------------------------------
    @staticmethod
    def copy(dst, src, id_contigs=None):
        lib().or_copy(_ptrs(dst), _ptrs(src), _ip(id_contigs) if id_contigs is not None else None,
                      ctypes.c_int(len(src["pos"])))

    @staticmethod
    def flip(dst, src, f):
        lib().or_flip(_ptrs(dst), _ptrs(src), ctypes.c_int(int(f)), ctypes.c_int(len(src["pos"])))

    @staticmethod
    def swap_activity(dst, src, f, max_id):
        lib().or_swap_activity(_ptrs(dst), _ptrs(src), ctypes.c_int(int(f)), ctypes.c_int(int(max_id)),
                               ctypes.c_int(len(src["pos"])))

    @staticmethod
    def pop_out(dst, src, id_contigs, f, max_id):
        lib().or_pop_out(_ptrs(dst), _ptrs(src), _ip(id_contigs), ctypes.c_int(int(f)), ctypes.c_int(int(max_id)),
                         ctypes.c_int(len(src["pos"])))

    @staticmethod
    def pop_in(which, dst, src, f_pop, f_ins, max_id, ori):
        lib().or_pop_in(ctypes.c_int(which), _ptrs(dst), _ptrs(src), ctypes.c_int(int(f_pop)),
                        ctypes.c_int(int(f_ins)), ctypes.c_int(int(max_id)), ctypes.c_int(int(ori)),
                        ctypes.c_int(len(src["pos"])))

    @staticmethod
    def split(dst, src, id_contigs, f_cut, upstream, max_id):
        lib().or_split(_ptrs(dst), _ptrs(src), _ip(id_contigs), ctypes.c_int(int(f_cut)),
                       ctypes.c_int(int(upstream)), ctypes.c_int(int(max_id)), ctypes.c_int(len(src["pos"])))

    @staticmethod
    def paste(dst, src, fA, fB, max_id):
        return lib().or_paste(_ptrs(dst), _ptrs(src), ctypes.c_int(int(fA)), ctypes.c_int(int(fB)),
                              ctypes.c_int(int(max_id)), ctypes.c_int(len(src["pos"])))

    @staticmethod
    def fill_sub_index(src, sub_index, contig, offset):
        lib().or_fill_sub_index(_ptrs(src), _ip(sub_index), ctypes.c_int(int(contig)), ctypes.c_int(int(offset)),
                                ctypes.c_int(len(src["pos"])))

    @staticmethod
    def relabel(state, old_2_new, id_contigs=None):
        o2n = np.ascontiguousarray(old_2_new, dtype=np.int32)
        lib().or_relabel(_ptrs(state), _ip(o2n), _ip(id_contigs) if id_contigs is not None else None,
                         ctypes.c_int(len(state["pos"])))

    # -- likelihood kernels -----------------------------------------------------------------
    def evaluate(self, state, per_pixel=None):
        """evaluate_likelihood + gpuarray.sum.  Returns total; fills per_pixel (float64[n_pix])."""
        pp = per_pixel.ctypes.data_as(_f64p) if per_pixel is not None else None
        return lib().or_evaluate_likelihood(
            self.obs.ctypes.data_as(_f32p), ctypes.c_int(self.width), _ptrs(state), _ip(self.collector),
            _ip(self.dispatcher), _ip(self.sub_id), self.sub_len.ctypes.data_as(_f32p), _ip(self.sub_accu),
            self.param.ctypes.data_as(_f32p), ctypes.c_float(self.nfpb), ctypes.c_int(self.n_bins), pp,
            ctypes.c_int(self.fix_trans_accu))

    def sub_compute(self, state, sub_index_no_rep, list_rep, list_uniq, curr_likelihood):
        a = np.ascontiguousarray(sub_index_no_rep, dtype=np.int32)
        r = np.ascontiguousarray(list_rep, dtype=np.int32)
        u = np.ascontiguousarray(list_uniq, dtype=np.int32)
        return lib().or_sub_compute_likelihood(
            self.obs.ctypes.data_as(_f32p), ctypes.c_int(self.width), _ptrs(state), _ip(a), ctypes.c_int(len(a)),
            _ip(r), ctypes.c_int(len(r)), _ip(u), ctypes.c_int(len(u)), _ip(self.collector), _ip(self.dispatcher),
            _ip(self.sub_id), self.sub_len.ctypes.data_as(_f32p), _ip(self.sub_accu),
            self.param.ctypes.data_as(_f32p), ctypes.c_float(self.nfpb), ctypes.c_int(self.n_bins),
            curr_likelihood.ctypes.data_as(_f64p), ctypes.c_int(self.fix_trans_accu))


def rippe(s, param):
    p = np.ascontiguousarray(param, dtype=np.float32)
    return float(lib().or_rippe(ctypes.c_float(s), p.ctypes.data_as(_f32p)))


def rippe_circ(s, s_tot, param):
    p = np.ascontiguousarray(param, dtype=np.float32)
    return float(lib().or_rippe_circ(ctypes.c_float(s), ctypes.c_float(s_tot), p.ctypes.data_as(_f32p)))


def lik(ex, ob):
    return float(lib().or_lik(ctypes.c_double(ex), ctypes.c_double(ob)))


class OracleSampler:
    """Host logic of cuda_lib_gl.sampler (start_EM path), restated literally over DenseOracle.

    ``problem`` holds what simulation_loader hands to the sampler ctor (cuda_lib_gl.py:33-42):
    S_o_A_frags, collector_id_repeats, frag_dispatcher, id_frag_duplicated, id_frags_blacklisted,
    n_frags (unique bins), n_new_frags (bins incl. repeats), hic_matrix_sub_sampled (bin-level dense),
    np_sub_frags_len_bp / _id / _accu, mean_squared_frags_per_bin, hic_matrix (sub-level dense),
    mean_value_trans, plus param_simu (8 floats; SURVEY H6: the fit is an input of the hot path).
    """

    def __init__(self, problem, rng, fix_trans_accu=False):
        p = problem
        self.rng = rng
        self.o = 0
        self.id_frags_blacklisted = list(p.get("id_frags_blacklisted", []))
        self.id_frag_duplicated = list(p.get("id_frag_duplicated", []))
        self.np_id_frag_duplicated = np.int32(self.id_frag_duplicated)
        self.n_frags = np.int32(p["n_frags"])
        self.n_new_frags = np.int32(p["n_new_frags"])
        self.uniq_frags = np.int32(np.setdiff1d(np.arange(0, self.n_frags, dtype=np.int32),
                                                self.np_id_frag_duplicated))  # cuda_lib_gl.py:74
        self.n_frags_uniq = np.int32(len(self.uniq_frags))
        self.n_tmp_struct = N_TMP_STRUCT
        self.collector_id_repeats = np.ascontiguousarray(p["collector_id_repeats"], dtype=np.int32)
        self.frag_dispatcher = np.ascontiguousarray(p["frag_dispatcher"], dtype=np.int32).reshape(-1, 2)
        self.np_sub_frags_id = np.ascontiguousarray(p["np_sub_frags_id"], dtype=np.int32).reshape(-1, 4)
        self.mean_value_trans = p["mean_value_trans"]
        S = p["S_o_A_frags"]
        n = int(self.n_new_frags)
        # cuda_lib_gl.py:153-172 : float32 copies, zero diagonals, blacklist fill
        self.hic_matrix = np.copy(np.float32(p["hic_matrix"]))
        self.hic_matrix[np.diag_indices_from(self.hic_matrix)] = 0
        self.hic_matrix_sub_sampled = np.copy(np.float32(p["hic_matrix_sub_sampled"]))
        self.hic_matrix_sub_sampled[np.diag_indices_from(self.hic_matrix_sub_sampled)] = 0
        for id_f in self.id_frags_blacklisted:
            real_id = S["id_d"][id_f]
            self.hic_matrix_sub_sampled[real_id, :] = 0
            self.hic_matrix_sub_sampled[:, real_id] = 0
            da = self.np_sub_frags_id[real_id]
            for i in range(0, da[3]):
                self.hic_matrix[da[i], :] = self.mean_value_trans
                self.hic_matrix[:, da[i]] = self.mean_value_trans
        self.dev = DenseOracle(self.hic_matrix, p["np_sub_frags_id"], p["np_sub_frags_len_bp"],
                               p["np_sub_frags_accu"], self.frag_dispatcher, self.collector_id_repeats,
                               int(self.n_frags), p["mean_squared_frags_per_bin"], p["param_simu"],
                               fix_trans_accu=fix_trans_accu)
        self.param_simu_rippe = np.dtype([('kuhn', np.float32), ('lm', np.float32), ('c1', np.float32),
                                          ('slope', np.float32), ('d', np.float32), ('l_max', np.float32),
                                          ('fact', np.float32), ('v_inter', np.float32)], align=True)  # cuda_lib_gl.py:136
        self.param_simu = np.array([tuple(np.asarray(p["param_simu"], dtype=np.float32))], dtype=self.param_simu_rippe)
        # initial state, cuda_lib_gl.py:226-262
        self.np_init_prev = np.copy(np.int32(S["prev"]))
        self.np_init_next = np.copy(np.int32(S["next"]))
        self.np_init_orientable = np.array(
            [self.np_sub_frags_id[S["id_d"][idf]][3] > 1 for idf in range(n)], dtype=np.int32)
        self.np_init_ori = np.ones((n,), dtype=np.int32)
        self.gpu_vect_frags = {k: np.array(S[k], dtype=np.int32, copy=True) for k in FIELDS if k != "ori"}
        self.gpu_vect_frags["ori"] = np.ones((n,), dtype=np.int32)
        self.gpu_id_contigs = np.copy(np.int32(S["id_c"]))
        self.collector_gpu_vect_frags = [new_state(n) for _ in range(self.n_tmp_struct)]
        self.pop_gpu_vect_frags = new_state(n)
        self.pop_gpu_id_contigs = np.copy(self.gpu_id_contigs)
        self.trans1_gpu_vect_frags = new_state(n)
        self.trans1_gpu_id_contigs = np.copy(self.gpu_id_contigs)
        self.trans2_gpu_vect_frags = new_state(n)
        self.trans2_gpu_id_contigs = np.copy(self.gpu_id_contigs)
        self.gpu_sub_index = np.zeros((n,), dtype=np.int32)
        self.curr_likelihood = np.zeros((self.dev.n_pix,), dtype=np.float64)
        self.n_neighbors = 10  # cuda_lib_gl.py:444
        self.bins = np.asarray(p.get("bins", np.zeros(0)), dtype=np.float64)  # cuda_lib_gl.py:1236 (set by estimate_parameters)
        self.n_stale_paste = 0  # how often the stale-slot branch of paste_contigs was hit
        self.setup_distri_frags()
        self.define_repeats()

    # cuda_lib_gl.py:448
    def init_likelihood(self):
        self.likelihood_t = self.dev.evaluate(self.gpu_vect_frags, self.curr_likelihood)
        return self.likelihood_t

    # cuda_lib_gl.py:452
    def define_repeats(self):
        c = self.gpu_vect_frags
        tmp_repeated = c["id"] != c["id_d"]
        id_repeat = np.unique(c["id_d"][tmp_repeated])
        self.is_repeat = []
        self.n_frags_duplicated = 0
        tmp = []
        tmp.extend(self.id_frags_blacklisted)
        for id_f in range(0, self.n_new_frags):
            if c["id_d"][id_f] in id_repeat:
                self.is_repeat.append(True)
                self.n_frags_duplicated += 1
                tmp.append(id_f)
            else:
                self.is_repeat.append(False)
        self.n_frags_4_dist = len(np.unique(tmp))

    # cuda_lib_gl.py:475
    def dist_inter_genome(self, g1):
        d = 3.0 * (self.n_new_frags - self.n_frags_4_dist)
        norm_distance = 3.0 * (self.n_new_frags - self.n_frags_4_dist)
        for id_f in range(0, self.n_new_frags):
            if id_f not in self.id_frags_blacklisted and not self.is_repeat[id_f]:
                prev_t0 = self.np_init_prev[id_f]
                tmp_prev_t1 = g1["prev"][id_f]
                prev_t1 = g1["id_d"][tmp_prev_t1] if tmp_prev_t1 != -1 else tmp_prev_t1
                next_t0 = self.np_init_next[id_f]
                tmp_next_t1 = g1["next"][id_f]
                next_t1 = g1["id_d"][tmp_next_t1] if tmp_next_t1 != -1 else tmp_next_t1
                ori_t0 = self.np_init_ori[id_f]
                ori_t1 = g1["ori"][id_f]
                swap = 1
                if ((prev_t1 == prev_t0) and (next_t1 == next_t0)) or ((prev_t1 == next_t0) and (next_t1 == prev_t0)):
                    d -= 1
                if self.np_init_orientable[id_f]:
                    if ori_t0 != ori_t1:
                        prev_t1, next_t1 = next_t1, prev_t1
                        swap = -1
                    if prev_t0 == prev_t1:
                        if prev_t0 == -1:
                            d -= 1
                        elif not (self.np_init_orientable[prev_t1]):
                            d -= 1
                        else:
                            d -= 0.5
                            if self.np_init_ori[prev_t0] == swap * g1["ori"][prev_t1]:
                                d -= 0.5
                    if next_t0 == next_t1:
                        if next_t0 == -1:
                            d -= 1
                        elif not (self.np_init_orientable[next_t1]):
                            d -= 1
                        else:
                            d -= 0.5
                            if self.np_init_ori[next_t0] == swap * g1["ori"][next_t1]:
                                d -= 0.5
                else:
                    if (prev_t1 == prev_t0) or (prev_t1 == next_t0):
                        d -= 1
                    if (next_t1 == next_t0) or (next_t1 == prev_t0):
                        d -= 1
        return d / norm_distance

    # cuda_lib_gl.py:841
    def pop_out_pop_in(self, id_f_pop, id_f_ins, mode, max_id):
        D = self.dev
        D.pop_out(self.pop_gpu_vect_frags, self.gpu_vect_frags, self.pop_gpu_id_contigs, id_f_pop, max_id)
        max_id2 = np.int32(self.pop_gpu_id_contigs.max())
        out = self.collector_gpu_vect_frags[mode]
        if mode == 0:
            D.copy(out, self.pop_gpu_vect_frags)
        elif mode == 1:
            D.flip(out, self.gpu_vect_frags, id_f_pop)
        elif mode in (2, 3):
            D.pop_in(1, out, self.pop_gpu_vect_frags, id_f_pop, id_f_ins, max_id2, 1 if mode == 2 else -1)
        elif mode in (4, 5):
            D.pop_in(2, out, self.pop_gpu_vect_frags, id_f_pop, id_f_ins, max_id2, 1 if mode == 4 else -1)
        elif mode in (6, 7):
            D.pop_in(3, out, self.pop_gpu_vect_frags, id_f_pop, id_f_ins, max_id2, 1 if mode == 6 else -1)
        elif mode == 8:
            D.swap_activity(out, self.pop_gpu_vect_frags, id_f_pop, max_id2)

    # cuda_lib_gl.py:916
    def transloc(self, id_fA, id_fB, max_id):
        D = self.dev
        mode = 0
        for upstreamfA in range(0, 2):
            D.split(self.trans1_gpu_vect_frags, self.gpu_vect_frags, self.trans1_gpu_id_contigs, id_fA, upstreamfA,
                    max_id)
            for upstreamfB in range(0, 2):
                max_id1 = np.int32(self.trans1_gpu_id_contigs.max())
                D.split(self.trans2_gpu_vect_frags, self.trans1_gpu_vect_frags, self.trans2_gpu_id_contigs, id_fB,
                        upstreamfB, max_id1)
                max_id2 = np.int32(self.trans2_gpu_id_contigs.max())
                self.n_stale_paste += int(D.paste(self.collector_gpu_vect_frags[9 + mode],
                                                  self.trans2_gpu_vect_frags, id_fA, id_fB, max_id2) > 0)
                mode += 1

    # cuda_lib_gl.py:1045
    def new_perform_modificationS(self, id_fA, id_fB, max_id, is_first):
        for mode in range(0, 9):
            self.pop_out_pop_in(id_fA, id_fB, mode, max_id)
        self.transloc(id_fA, id_fB, max_id)

    # cuda_lib_gl.py:1156
    def test_copy_struct(self, id_fA, id_f_sampled, mode, max_id):
        if mode < 9:
            self.pop_out_pop_in(id_fA, id_f_sampled, mode, max_id)
        elif mode < 13:
            self.transloc(id_fA, id_f_sampled, max_id)
        self.dev.copy(self.gpu_vect_frags, self.collector_gpu_vect_frags[mode], self.gpu_id_contigs)

    # cuda_lib_gl.py:1539
    def explode_genome(self, dt=0):
        for i in range(0, self.n_new_frags):
            self.modify_gl_cuda_buffer(i, dt)
            max_id = self.gpu_vect_frags["id_c"].max()
            self.test_copy_struct(i, 0, 0, max_id)

    # cuda_lib_gl.py:1695-1722 + kernels3.cu:3848-3851 (display half dropped)
    def modify_gl_cuda_buffer(self, id_fi, dt=0):
        l_cont = np.copy(self.gpu_vect_frags["l_cont"])
        self.id_contigs = np.copy(self.gpu_vect_frags["id_c"])
        idc_un, idx_un = np.unique(self.id_contigs, return_index=True)
        n_new_contigs = len(idc_un)
        list_len_contigs = l_cont[idx_un]
        ord_length = np.argsort(list_len_contigs, kind="stable")
        old_2_new_indexes = np.zeros((idc_un.max() + 1,), dtype=np.int32)
        old_2_new_indexes[idc_un[ord_length]] = np.arange(0, n_new_contigs, 1, dtype=np.int32)
        self.dev.relabel(self.gpu_vect_frags, old_2_new_indexes, self.gpu_id_contigs)
        return np.int32(np.float32(n_new_contigs - 1))

    def temperature(self, t, n_step):
        return 1.0  # cuda_lib_gl.py:2602

    # cuda_lib_gl.py:1793
    def step_max_likelihood(self, id_fA, delta, size_block=512, dt=0, t=0, n_step=1):
        g = self.gpu_vect_frags
        if id_fA not in self.id_frags_blacklisted:
            id_start = np.nonzero(g["start_bp"] == 0)[0]
            max_id = self.modify_gl_cuda_buffer(id_fA, dt)
            n_contigs = len(np.unique(g["id_c"]))
            mean_len = g["l_cont"].mean()
            mean_len_bp = g["l_cont_bp"][id_start].mean()
            max_len = g["l_cont"].max()
            min_len = g["l_cont"].min()
            self.curr_likelihood.fill(np.float64(0))
            likelihood_t = self.dev.evaluate(g, self.curr_likelihood)
            self.likelihood_t = likelihood_t
            len_contig_A = g["l_cont"][id_fA]
            contig_A = g["id_c"][id_fA]
            max_id = np.int32(self.gpu_id_contigs.max())
            self.dev.fill_sub_index(g, self.gpu_sub_index, contig_A, 0)
            id_neighbours = self.return_neighbours(id_fA, delta)
            n_neighbours = len(id_neighbours)
            self.score = np.zeros((n_neighbours * self.n_tmp_struct,), dtype=np.float64)
            id_neighbours.sort()
            self.last_neighbours = list(id_neighbours)
            for id_x in range(0, n_neighbours):
                id_fB = id_neighbours[id_x]
                self.stream_likelihood(id_fA, contig_A, len_contig_A, id_fB, id_x, likelihood_t, max_id)
            scores_2_remove = []
            scores_2_remove.extend(range(self.n_tmp_struct, len(self.score), self.n_tmp_struct))
            scores_2_remove.extend(range(self.n_tmp_struct + 1, len(self.score), self.n_tmp_struct))
            id_max = self.score.argmax()
            or_score = np.copy(self.score)
            filtered_score = self.score - self.score.min()
            filtered_score[scores_2_remove] = 0
            max_score = filtered_score.max()
            thresh_overflow = 30
            filtered_score = filtered_score - (max_score - thresh_overflow)
            filtered_score[filtered_score < 0] = 0
            id_ok_4_sampling = np.ix_(filtered_score > 0)
            self.sub_score = filtered_score[id_ok_4_sampling]
            F_t = self.temperature(t, n_step)
            self.sub_score = self.sub_score / self.sub_score.sum()
            self.sub_score[self.sub_score > 0] = np.power(self.sub_score[self.sub_score > 0], 1. / F_t)
            self.sub_score = self.sub_score / self.sub_score.sum()
            if len(id_ok_4_sampling[0]) == 1 or len(id_ok_4_sampling[0]) == 0:
                sample_out = id_max
            else:
                sample_out = self.rng.choice(id_ok_4_sampling[0], 1, p=self.sub_score)[0]
            id_f_sampled = id_neighbours[sample_out // self.n_tmp_struct]
            op_sampled = sample_out % self.n_tmp_struct
            self.test_copy_struct(id_fA, id_f_sampled, op_sampled, max_id)
            o = or_score[sample_out]
            self.o = o
        else:
            o = self.o
            id_start = np.nonzero(g["start_bp"] == 0)[0]
            max_id = self.modify_gl_cuda_buffer(id_fA, dt)
            n_contigs = len(np.unique(g["id_c"]))
            mean_len = g["l_cont"].mean()
            mean_len_bp = g["l_cont_bp"][id_start].mean()
            max_len = g["l_cont"].max()
            min_len = g["l_cont"].min()
            op_sampled = -1
            id_f_sampled = id_fA
            F_t = self.temperature(t, n_step)
        dist = self.dist_inter_genome(g)
        self.likelihood_t = o
        return o, n_contigs, min_len, mean_len_bp, max_len, op_sampled, id_f_sampled, dist, F_t

    # cuda_lib_gl.py:1986-2017 : full evaluation with the TEST parameters
    def compute_likelihood_4_nuisance(self):
        keep = np.copy(self.dev.param)
        self.dev.set_param(self._flat(self.param_simu_test))
        out = self.dev.evaluate(self.gpu_vect_frags, None)
        self.dev.set_param(keep)
        return out

    @staticmethod
    def _flat(p):
        return np.array([p[0][k] for k in p.dtype.names], dtype=np.float32)

    # cuda_lib_gl.py:2022-2107 (optim_rippe_curve_update.peval / estimate_max_dist_intra restated in oracle/optim_ref.py)
    def step_nuisance_parameters(self, dt, t, n_step):
        from oracle import optim_ref as opti
        curr_param = np.copy(self.param_simu)
        kuhn, lm, c1, slope, d, d_max, fact, d_nuc = curr_param[0]
        self.sigma_fact = 10 ** (np.log10(fact) - 2)
        self.sigma_slope = 0.05
        self.sigma_d_max = 100
        self.sigma_d_nuc = 0.5
        self.sigma_d = 10
        id_modif = self.rng.choice(4)
        if id_modif == 0:  # scale factor
            new_fact = fact + self.rng.normal(loc=0.0, scale=self.sigma_fact)
            test_param = [kuhn, lm, slope, d, new_fact]
            new_d_max = opti.estimate_max_dist_intra(test_param, d_nuc)
            c1 = np.float32((0.53 * np.power(lm / kuhn, slope)) * np.power(kuhn, -3))
            out_test_param = [(kuhn, lm, c1, slope, d, new_d_max, new_fact, d_nuc)]
        elif id_modif == 1:  # slope
            new_slope = slope + self.rng.normal(loc=0.0, scale=self.sigma_slope)
            test_param = [kuhn, lm, new_slope, d, fact]
            new_d_max = opti.estimate_max_dist_intra(test_param, d_nuc)
            c1 = np.float32((0.53 * np.power(lm / kuhn, new_slope)) * np.power(kuhn, -3))
            out_test_param = [(kuhn, lm, c1, new_slope, d, new_d_max, fact, d_nuc)]
        elif id_modif == 2:  # max distance intra
            new_d_max = d_max + self.rng.normal(loc=0.0, scale=self.sigma_d_max)
            test_param = [kuhn, lm, slope, d, fact]
            new_d_nuc = opti.peval(new_d_max, test_param)  # 5-list: param[3] = d is used as the amplitude (H3)
            c1 = np.float32((0.53 * np.power(lm / kuhn, slope)) * np.power(kuhn, -3))
            out_test_param = [(kuhn, lm, c1, slope, d, new_d_max, fact, new_d_nuc)]
        elif id_modif == 3:  # val trans
            new_d_nuc = d_nuc + self.rng.normal(loc=0.0, scale=self.sigma_d_nuc)
            test_param = [kuhn, lm, slope, d, fact]
            new_d_max = opti.estimate_max_dist_intra(test_param, new_d_nuc)
            c1 = np.float32((0.53 * np.power(lm / kuhn, slope)) * np.power(kuhn, -3))
            out_test_param = [(kuhn, lm, c1, slope, d, new_d_max, fact, new_d_nuc)]
        else:  # d -- unreachable: choice(4) never returns 4 (H3)
            new_d = d + self.rng.normal(loc=0.0, scale=self.sigma_d)
            test_param = [kuhn, lm, slope, new_d, fact]
            new_d_max = opti.estimate_max_dist_intra(test_param, d_nuc)
            c1 = np.float32((0.53 * np.power(lm / kuhn, slope)) * np.power(kuhn, -3))
            out_test_param = [(kuhn, lm, c1, slope, new_d, new_d_max, fact, d_nuc)]
        out_test_param = np.array(out_test_param, dtype=self.param_simu_rippe)
        self.param_simu_test = out_test_param
        test_likelihood = self.compute_likelihood_4_nuisance()
        F_t = self.temperature(t, n_step)
        ratio = np.exp((test_likelihood - self.likelihood_t) / F_t)
        u = self.rng.rand()
        success = 0
        if ratio >= u:
            success = 1
            self.dev.set_param(self._flat(out_test_param))
            self.param_simu = out_test_param
            self.likelihood_t = test_likelihood
        kuhn, lm, c1, slope, d, d_max, fact, d_nuc = self.param_simu[0]
        p0 = [kuhn, lm, slope, d, fact]
        y_rippe = opti.peval(self.bins, p0)
        return fact, d, d_max, d_nuc, slope, self.likelihood_t, success, y_rippe

    # cuda_lib_gl.py:2295
    def return_neighbours(self, id_fA, delta0):
        ori_id = self.gpu_vect_frags["id_d"][id_fA]
        delta = min(self.n_neighbors, delta0)
        distri = self.distri_frags[ori_id]["pk"]
        n_max_candidates = min(delta, np.nonzero(distri != 0)[0].shape[0])
        init_id = self.rng.choice(self.distri_frags[ori_id]["xk"], n_max_candidates, p=distri, replace=False)
        out = []
        if ori_id in self.id_frag_duplicated:
            d = self.frag_dispatcher[ori_id]
            l = self.collector_id_repeats[d[0]: d[1]]
            dup = np.setdiff1d(l, id_fA)
            out.extend(dup)
        for id_fB in init_id:
            d = self.frag_dispatcher[id_fB]
            out.extend(self.collector_id_repeats[d[0]: d[1]])
        real_out = []
        for ele in out:
            if ele not in self.id_frags_blacklisted:
                real_out.append(ele)
        return real_out

    # cuda_lib_gl.py:2363
    def setup_distri_frags(self):
        self.distri_frags = dict()
        fact = 3
        for i in range(0, self.n_frags):
            v = np.float32(self.hic_matrix_sub_sampled[i, :])
            vtmp = np.copy(v)
            id_sort = np.argsort(vtmp, kind="stable")
            id_sort_l = list(id_sort)
            id_sort_l.reverse()
            id_sort_l = np.array(id_sort_l, dtype=np.int32)
            xk = id_sort_l[: self.n_neighbors]
            dat = vtmp[xk] ** fact
            if dat.sum() > 0:
                pk = dat / dat.sum()
            else:
                tmp = np.ones_like(dat, dtype=np.float32)
                pk = tmp / tmp.sum()
            self.distri_frags[i] = dict()
            self.distri_frags[i]["xk"] = xk
            self.distri_frags[i]["pk"] = pk

    # cuda_lib_gl.py:2392
    def stream_likelihood(self, id_fA, contig_A, len_contig_A, id_fB, id_x, likelihood_t, max_id):
        g = self.gpu_vect_frags
        len_contig_B = g["l_cont"][id_fB]
        contig_B = g["id_c"][id_fB]
        self.new_perform_modificationS(id_fA, id_fB, max_id, id_x == 0)
        if contig_B != contig_A:
            self.dev.fill_sub_index(g, self.gpu_sub_index, contig_B, len_contig_A)
            size_sub_index = len_contig_A + len_contig_B
        else:
            size_sub_index = len_contig_A
        init_sub_index = self.gpu_sub_index[:size_sub_index]
        sub_index_no_repeats = np.setdiff1d(init_sub_index, self.np_id_frag_duplicated)
        sub_index_repeats = np.intersect1d(init_sub_index, self.np_id_frag_duplicated)
        for j in range(0, self.n_tmp_struct):
            delta_j = self.dev.sub_compute(self.collector_gpu_vect_frags[j], sub_index_no_repeats, sub_index_repeats,
                                           self.uniq_frags, self.curr_likelihood)
            self.score[id_x * self.n_tmp_struct + j] = delta_j + likelihood_t
