"""GPU box: the union-set kernels (k_gprep + k_strict2) against the O(m^2) validation kernel and against round 3's per-neighbour
kernels on tests/strict_cases.py, case by case (int64 sums must be EQUAL).  Usage: python tools/s2_check.py [cases, e.g. 0,1,4]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import strict_cases  # noqa: E402

only = sys.argv[1] if len(sys.argv) > 1 else ",".join(str(i) for i in range(len(strict_cases.CASES)))
tmp = tempfile.mkdtemp()


def run(tag, **env):
    out = os.path.join(tmp, tag + ".npz")
    e = dict(os.environ)
    e.update(env)
    subprocess.check_call([sys.executable, "-m", "tests.strict_cases", out, only], cwd=ROOT, env=e, timeout=1500)
    return np.load(out)


dense = run("dense", GRAAL_STRICT_DENSE="1")
variants = {"v2 tiled (no flat)": dict(GRAAL_NO_FLAT="1"), "default (flat + v2)": {}, "v1 tiled (no flat)": dict(GRAAL_NO_FLAT="1", GRAAL_STRICT_V1="1")}
ok = True
for tag, env in variants.items():
    got = run(tag.split()[0] + str(len(env)), **env)
    for key in dense.files:
        w, g = dense[key], got[key]
        bad = np.argwhere(w != g)
        name = strict_cases.CASES[int(key[4:])][0]
        print("%-22s %-6s %s: %d of %d sums differ%s" % (tag, key, name[:60], len(bad), w.size,
              "" if len(bad) == 0 else "  first %s: want %d got %d" % (tuple(bad[0]), w[tuple(bad[0])], g[tuple(bad[0])])), flush=True)
        if len(bad):
            ok = False
            # which candidates (op) and neighbours differ
            ops = np.unique(bad[:, -1]); ks = np.unique(bad[:, -2])
            print("    ops", ops.tolist(), "neighbours", ks.tolist(), "max |diff| (Q30 units)", int(np.abs(w - g).max()))
sys.exit(0 if ok else 1)
