"""Build libatlasfit.so (HIP C++ for gfx950) in-tree with hipcc.  Usage: python build.py [--force]"""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libatlasfit.so")
UNITS = ["mlp.hip", "mlpbf.hip", "mlphf.hip", "mlp16.hip", "dw.hip", "elem.hip", "host.hip"]
HEADERS = ["af_dev.h", "elem.h", "mlp_common.h", "bfsplit.h", "dw_slots.h", os.path.join("..", "..", "include", "atlasfit.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage"] + os.environ.get("AF_HIPCC_EXTRA", "").split()     # AF_HIPCC_EXTRA: -D switches of the kernel experiments (tools/experiments/README.md); use with --force


def _check_no_scratch(name, stderr):
    """Every kernel of the library must compile without scratch memory (register spills): the chains run at 256 VGPR + ~240 AGPR by
    design, and a spill is silent — it shows up as extra HBM traffic only (round 3: a possibly-empty layer loop cost the training
    forward 86 spilled registers per chain, +12 % HBM writes).  The compiler's kernel-resource-usage remarks are parsed here and a
    non-zero ScratchSize fails the build loudly."""
    import re
    bad, cur = [], None
    for line in stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and int(m.group(1)) != 0:
            bad.append((cur, int(m.group(1))))
    if bad and not os.environ.get("AF_ALLOW_SCRATCH"):
        raise RuntimeError("%s: kernels with scratch (spilled registers): %s" % (name, bad))


def _isa_check():
    import importlib.util
    spec = importlib.util.spec_from_file_location("af_isa_check", os.path.join(HERE, "isa_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True):
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(SRC, h) for h in HEADERS]
    jobs = []
    for u in UNITS:
        src = os.path.join(SRC, u)
        obj = os.path.join(objdir, u.replace(".hip", ".o"))
        if force or _stale(obj, [src] + hdrs):
            jobs.append(([_hipcc()] + FLAGS + ["-c", src, "-o", obj], u))

    def run(job):
        cmd, name = job
        if verbose:
            print("[build] hipcc", name, flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (name, r.stderr))
        _check_no_scratch(name, r.stderr)
        _isa_check().check_unit(name, cmd[-1], verbose=verbose)      # ISA invariants of the hand-scheduled kernels (isa_check.py): fail the build, not the run
        if verbose and ("warning:" in r.stderr or "error:" in r.stderr):      # the resource-usage remarks alone are not worth printing
            print("\n".join(l for l in r.stderr.splitlines() if "kernel-resource-usage" not in l))

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(objdir, u.replace(".hip", ".o")) for u in UNITS]
    if force or jobs or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr)
        if verbose:
            print("[build] linked", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
