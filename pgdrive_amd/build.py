"""Builds libpgdrive_hip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "pgd_engine.hip")
DEPS = [SRC] + [os.path.join(HERE, "csrc", h) for h in ("pgd_device.h", "pgd_vehicle.h", "pgd_localize.h", "pgd_idm.h",
                                                         "pgd_dynamics.h", "pgd_observe.h", "pgd_gather.h", "pgd_topdown.h", "pgd_policy.h")] + \
    [os.path.join(HERE, "..", "include", "pgdrive_hip.h"), os.path.join(HERE, "..", "include", "pgd_state_layout.h"),
     os.path.abspath(__file__)]  # the flags below are part of the build: a change of this file rebuilds
LIB = os.path.join(HERE, "libpgdrive_hip.so")


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP engine cannot be built")


FAST_FP = ["-fno-hip-fp32-correctly-rounded-divide-sqrt", "-fgpu-flush-denormals-to-zero", "-funsafe-math-optimizations"]
# Optimisation level, measured on the step kernel (profiles/r03_notes.md): -O3 23.65 us, -O2 23.16, -O1 23.56, -Oz 24.16,
# -Os 22.92, -Os without SLP vectorisation (packed f32 VALU is slower than two scalar ops on this hardware) 22.72.  -Os hoists and
# batches less: 105 instead of 126 VGPRs, 34.6 instead of 36.6 KB of code for the benchmark kernel; every other kernel is as fast
# or faster with it (C5 +1 %, 32768 envs +1 %, dense traffic +1 %, top-down unchanged).
# Round 5, after the records went into piece planes and the image mask out of the one-env kernels: -O2 is ahead again -- metric 17.12 ->
# 17.08 us, respawn traffic 24.5 -> 24.1, 8 agents 22.1 -> 21.9, 32768 envs 77.1 -> 75.3, ego-only 1024 envs 10.7 -> 9.4 (the general
# kernels lose 2.7 %; k_observe_env keeps size-optimised code by attribute): profiles/r05_notes.md.
# SimplifyCFG folds `a && b` into one branch when b costs at most this many speculated instructions (default 1): the kernel spends a
# fifth of its instructions on exec-mask bookkeeping of divergent branches (221 s_and_saveexec sites -> 195): another 0.9 %.
OPT = ["-O2", "-fno-slp-vectorize", "-mllvm", "-bonus-inst-threshold=4"]


def source_sha():
    """What a binary of the engine is identified by: sha256 over the sources it is compiled from (csrc/*, include/*, and this file,
    whose flags are part of the build), first 16 hex digits.  build() compiles it into the library (pgd_source_sha), the profile
    passes of tools/ stamp their summaries with it, and bench.py only quotes a committed counter pass whose stamp equals the loaded
    library's (a kernel change without a re-profile must not ship a line whose counters describe another binary)."""
    import hashlib
    h = hashlib.sha256()
    for d in sorted(os.path.abspath(x) for x in DEPS):
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False, extra=()):
    if not force and not needs_build():
        return LIB
    # the kernel is issue-bound and the parity tolerances are 1e-5 and looser: fp32 `/` compiles to rcp * x and sqrtf to
    # the rsq sequence (2.5 ulp) instead of the correctly rounded, denormal-safe expansions (about ten VALU instructions
    # each).  -funsafe-math-optimizations adds re-association (no finite-math assumption: the NaN test of the action input
    # stays): another 2 % (22.33 -> 21.89 us); lane_local and wrap_to_pi, whose bits a checkpoint round trip relies on, switch
    # it off for themselves.  The flag also implies -fapprox-func (hipcc links oclc_unsafe_math_on): expf -- the energy term of
    # after_step, its only user -- compiles to the 9-instruction v_exp_f32 form (1 ulp of 2^x plus the input scaling) instead of
    # the 26-instruction library version; sinf / cosf / atan2f compile to the same code either way.  SF_ENERGY is held to
    # 2e-6 + 2e-5 rel against the fp64 oracle after every teacher-forced step (tests/util.py), measured 0.56 of that.
    cmd = [hipcc(), "--offload-arch=gfx950", *OPT, "-std=c++17", *FAST_FP, "-shared", "-fPIC",
           '-DPGD_SOURCE_SHA="%s"' % source_sha(), "-o", LIB, SRC] + list(extra)
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    import sys
    build(force=True, verbose=True, extra=sys.argv[1:])
