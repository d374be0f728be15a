"""Manual staged GPU bring-up script (not a test): prints after every engine call."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
t0 = time.time()
def log(*a):
    print("[%.1fs]" % (time.time() - t0), *a, flush=True)
from graal_amd import synth
from graal_amd.lib import Engine
log("imports done")
par = synth.make_param_simu(fact=300.0, v_inter=0.03)
P = synth.make_problem(n_bins=60, nnz=900, n_sub=1, seed=7, contig_weights=(5, 3, 2), mean_len_bp=1500.0, param=par)
log("problem")
e = Engine(0); log("engine created")
e.upload_subfrags(P["np_sub_frags_id"], P["np_sub_frags_len_bp"], P["np_sub_frags_accu"], P["init_n_sub_frags"], P["mean_squared_frags_per_bin"]); log("subfrags")
e.upload_contacts(P["coo_row"], P["coo_col"], P["coo_val"]); log("contacts")
e.set_params(P["param_simu"]); log("params")
e.upload_frags(P["S_o_A_frags"]); log("frags")
log("stats", e.layout_stats())
m = e.relabel_contigs(); log("relabel", m)
log("full", e.eval_full())
d = e.eval_candidates(7, [8, 30, 55], m); log("cands", d)
log("timing", e.last_timing(), "counters", e.last_counters())
log("apply", e.apply_move(7, 8, 6, m))
log("done")
