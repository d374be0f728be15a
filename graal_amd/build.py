"""Build the engine's shared libraries in-tree (they travel to the GPU box with the snapshot).

* ``libgraal_hip.so``  -- the product: HIP kernels + C ABI, ``hipcc --offload-arch=gfx950``.
* ``libgraal_hostcheck.so`` -- TEST ONLY: ``frag_ops.h`` (host/device shared layout algebra) compiled
  for the host with g++ so that the CPU test-suite can check it against the oracle without a GPU.
  It contains no likelihood code and is never loaded by the product path.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HIP_LIB = os.environ.get("GRAAL_HIP_LIB") or os.path.join(HERE, "libgraal_hip.so")   # (override: experiments with variant builds)
HOSTCHECK_LIB = os.path.join(HERE, "libgraal_hostcheck.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def hipcc_path():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build_hip(force=False, verbose=False):
    srcs = [os.path.join(CSRC, "graal_hip.hip"), os.path.join(CSRC, "frag_ops.h"), os.path.join(CSRC, "host_step.h"),
            os.path.join(CSRC, "model_math.h"), os.path.join(CSRC, "strict_sets.h"), os.path.join(CSRC, "strict2.h"), os.path.join(ROOT, "include", "graal_hip.h")]
    if force or _newer(HIP_LIB, srcs):
        cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off",  # float32 model arithmetic exactly as written (parity with the oracle)
               "-Wall", "-Wno-unused-function", "-o", HIP_LIB, srcs[0]]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        subprocess.check_call(cmd)
    return HIP_LIB


def build_hostcheck(force=False):
    srcs = [os.path.join(CSRC, "host_check.cpp"), os.path.join(CSRC, "frag_ops.h"), os.path.join(CSRC, "strict_sets.h")]
    if force or _newer(HOSTCHECK_LIB, srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", HOSTCHECK_LIB, srcs[0]])
    return HOSTCHECK_LIB


if __name__ == "__main__":
    build_hip(force="--force" in sys.argv, verbose="-v" in sys.argv)
    build_hostcheck(force="--force" in sys.argv)
    print("built", HIP_LIB, HOSTCHECK_LIB)
