"""Headless MCMC outer loop: ``main_gl.window.start_EM`` (``main_gl.py:210-283``) and its trace writer
(``save_behaviour_to_txt``, ``main_gl.py:321-342``) without GLUT / wx.

``run_em`` drives any object with the reference sampler's surface (``init_likelihood``, ``modify_gl_cuda_buffer``,
``explode_genome``, ``step_max_likelihood``, optionally ``step_nuisance_parameters``), so the same loop runs the
MI355X sampler and -- in tests -- the oracle's restatement, with the reference's RNG call order (SURVEY.md
Appendix C: one ``shuffle`` per cycle, then the per-step draws inside the sampler).
"""
import os

import numpy as np


class Trace(object):
    """The per-step series ``start_EM`` collects (``main_gl.py:250-278``)."""

    def __init__(self):
        self.likelihood, self.full_likelihood = [], []
        self.n_contigs, self.mean_len, self.op_sampled = [], [], []
        self.id_fA, self.id_fB, self.dist = [], [], []
        self.fact, self.d, self.d_max, self.d_nuc, self.slope = [], [], [], [], []
        self.likelihood_nuisance, self.success = [], []

    def mutations(self):
        """(id_fA, id_fB, id_mutation) rows -- the accepted-move trace of ``list_mutations.txt``."""
        return np.array([self.id_fA, self.id_fB, self.op_sampled], dtype=np.int64).T.reshape(-1, 3)


def run_em(sampler, n_cycles, n_neighbours, rng=None, sample_param=False, scrambled=True, dt=0, on_step=None, matrix_files=None):
    """``matrix_files = (before, after)``: the two images start_EM writes through display_current_matrix (``main_gl.py:213, 283``:
    pre_simu.tiff before the first step -- of the layout as loaded --, post_em.tiff behind the last cycle); None: no images."""
    if rng is None and getattr(getattr(sampler, "group", None), "world", 1) > 1:
        raise ValueError("run_em over a sharded sampler needs the sampler's (identically seeded) rng: every rank shuffles itself")
    rng = np.random if rng is None else rng
    trace = Trace()
    if matrix_files and matrix_files[0]:
        sampler.display_current_matrix(matrix_files[0])
    sampler.init_likelihood()
    sampler.modify_gl_cuda_buffer(0, dt)
    if scrambled:
        sampler.explode_genome(dt)
    list_frags = np.arange(0, sampler.n_new_frags, dtype=np.int32)
    n_iter = np.float32(n_cycles)
    for j in range(0, n_cycles):
        rng.shuffle(list_frags)
        if not sample_param and on_step is None and hasattr(sampler, "steps_max_likelihood"):
            # the cycle's steps in runs behind the C ABI (sampler.steps_max_likelihood); same values, same order
            res = sampler.steps_max_likelihood(list_frags, n_neighbours, 512, dt, np.float32(j), n_iter)
            kuhn, lm, c1, slope, d, d_max, fact, d_nuc = [sampler.param_simu[0][k] for k in sampler.param_simu.dtype.names]
            for i, (o, n_contigs, min_len, mean_len, max_len, op_sampled, id_f_sampled, dist, temp) in zip(list_frags, res):
                trace.full_likelihood.append(o)
                trace.likelihood.append(o)
                trace.n_contigs.append(n_contigs)
                trace.mean_len.append(mean_len)
                trace.op_sampled.append(op_sampled)
                trace.id_fB.append(id_f_sampled)
                trace.id_fA.append(i)
                trace.dist.append(dist)
                trace.fact.append(fact)
                trace.d.append(d)
                trace.d_max.append(d_max)
                trace.d_nuc.append(d_nuc)
                trace.slope.append(slope)
                trace.likelihood_nuisance.append(o)
                trace.success.append(1)
            continue
        for i in list_frags:
            o, n_contigs, min_len, mean_len, max_len, op_sampled, id_f_sampled, dist, temp = \
                sampler.step_max_likelihood(i, n_neighbours, 512, dt, np.float32(j), n_iter)
            trace.full_likelihood.append(sampler.likelihood_t)
            trace.likelihood.append(o)
            trace.n_contigs.append(n_contigs)
            trace.mean_len.append(mean_len)
            trace.op_sampled.append(op_sampled)
            trace.id_fB.append(id_f_sampled)
            trace.id_fA.append(i)
            trace.dist.append(dist)
            if sample_param:
                fact, d, d_max, d_nuc, slope, likeli, success, y_eval = sampler.step_nuisance_parameters(
                    dt, np.float32(j), n_iter)
            else:
                success = 1
                kuhn, lm, c1, slope, d, d_max, fact, d_nuc = [sampler.param_simu[0][k] for k in
                                                               sampler.param_simu.dtype.names]
                likeli = o
            trace.fact.append(fact)
            trace.d.append(d)
            trace.d_max.append(d_max)
            trace.d_nuc.append(d_nuc)
            trace.slope.append(slope)
            trace.likelihood_nuisance.append(likeli)
            trace.success.append(success)
            if on_step is not None:
                on_step(j, i, trace)
    if matrix_files and len(matrix_files) > 1 and matrix_files[1]:
        sampler.display_current_matrix(matrix_files[1])
    return trace


def save_behaviour_to_txt(trace, output_folder, id_exp=""):
    """The text files of ``main_gl.py:126-138,321-342`` (same names, one value per line; tab-separated mutations)."""
    if not os.path.isdir(output_folder):
        os.makedirs(output_folder)
    series = [("list_mean_len.txt", trace.mean_len), ("list_n_contigs.txt", trace.n_contigs),
              ("list_dist_init_genome.txt", trace.dist), ("list_likelihood.txt", trace.likelihood),
              ("list_fact.txt", trace.fact), ("list_slope.txt", trace.slope), ("list_d_max.txt", trace.d_max),
              ("list_d_nuc.txt", trace.d_nuc), ("list_success.txt", trace.success)]
    for name, data in series:
        with open(os.path.join(output_folder, str(id_exp) + name), "w") as h:
            for item in data:
                h.write("%s\n" % item)
    with open(os.path.join(output_folder, str(id_exp) + "list_mutations.txt"), "w") as f:
        f.write("%s\t%s\t%s\n" % ("id_fA", "id_fB", "id_mutation"))
        for a, b, m in zip(trace.id_fA, trace.id_fB, trace.op_sampled):
            f.write("%s\t%s\t%s\n" % (a, b, m))


def load_mutations(path):
    """Read a ``list_mutations.txt`` back (replay input, ``main_gl.py:140-207``)."""
    rows = []
    with open(path) as f:
        next(f)
        for line in f:
            if line.strip():
                rows.append([int(x) for x in line.split()])
    return np.array(rows, dtype=np.int64).reshape(-1, 3)


def replay(sampler, mutations, dt=0):
    """``replay_simu`` / ``apply_replay_simu`` (``main_gl.py:140-207``, ``cuda_lib_gl.py:1559-1578``): re-apply an
    accepted-move trace; steps with id_mutation == -1 (blacklisted fragment) are skipped."""
    for id_fA, id_fB, op in mutations:
        max_id = sampler.modify_gl_cuda_buffer(int(id_fA), dt)
        if op >= 0:
            sampler.test_copy_struct(int(id_fA), int(id_fB), int(op), max_id)
