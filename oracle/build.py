"""Build recipe for the CPU oracle (test infrastructure, not product code).

  python oracle/build.py            # liboracle.so (+ oracle/_ref/* when /root/reference exists)

Outputs
  oracle/_build/liboracle.so   the repo's own C restatement (oracle/*.c), always built.
  oracle/_ref/libref_cpu.so    the reference's OWN CPU functions, compiled from where they lie
  oracle/_ref/libref_knn.so    under /root/reference.  Built only in the authoring container
  oracle/_ref/selection_sort   (the GPU box has no /root/reference and uses the prebuilt files).

oracle/_ref/ is git-ignored: no reference source text is written into the repository.  The
TF-op files (tf_approxmatch.cpp, tf_interpolate.cpp, tf_nndistance.cpp) include TensorFlow
headers that do not exist here, so they are NOT compiled as files and no stand-in headers are
written: the self-contained CPU function bodies (no TF symbol inside) are streamed by line
range straight into the compiler's stdin (SURVEY.md Appendix C recipe); nothing but the
resulting .so is stored.  knn_.cxx (+ vendored nanoflann.hpp) and selection_sort.cpp compile
whole, unmodified.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("DISPU_REFERENCE", "/root/reference")
BUILD = os.path.join(HERE, "_build")
REFOUT = os.path.join(HERE, "_ref")


def _has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            return " fma " in f.read()
    except OSError:
        return False


def _run(cmd, **kw):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, **kw)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout.decode(errors="replace")))
    return r


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def build_oracle(force=False):
    os.makedirs(BUILD, exist_ok=True)
    srcs = sorted(os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".c"))
    out = os.path.join(BUILD, "liboracle.so")
    if not force and _newer(out, srcs + [os.path.abspath(__file__)]):
        return out
    flags = ["-O2", "-std=gnu11", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off",
             "-fvisibility=hidden", "-fno-math-errno"]
    if _has_fma():
        flags.append("-mfma")  # explicit fmaf() -> one vfmadd; nothing else may fuse (contract=off)
    _run(["gcc"] + flags + srcs + ["-o", out + ".tmp", "-lm"])
    os.replace(out + ".tmp", out)
    return out


# (file, first line, last line, sed-style transform) of self-contained reference CPU functions
_REF_RANGES = [
    ("tf_ops/approxmatch/tf_approxmatch.cpp", 23, 140, None),   # approxmatch_cpu, matchcost_cpu, matchcostgrad_cpu
    ("tf_ops/interpolation/tf_interpolate.cpp", 57, 153, None),  # threenn_cpu, threeinterpolate(_grad)_cpu
    ("tf_ops/nn_distance/tf_nndistance.cpp", 21, 43, "unstatic"),  # nnsearch
    ("tf_ops/grouping/query_ball_point.cpp", 17, 85, None),     # query_ball_point_cpu, group_point(_grad)_cpu
]


def build_reference(force=False):
    """Compile the reference's own CPU functions.  Returns False when /root/reference is absent."""
    if not os.path.isdir(REF):
        return False
    os.makedirs(REFOUT, exist_ok=True)
    knn_src = os.path.join(REF, "libs/nearest_neighbors/knn_.cxx")
    knn_out = os.path.join(REFOUT, "libref_knn.so")
    if force or not os.path.exists(knn_out):
        _run(["g++", "-O2", "-std=c++11", "-fopenmp", "-fPIC", "-shared", "-w",
              "-I" + os.path.dirname(knn_src), knn_src, "-o", knn_out])
    cpu_out = os.path.join(REFOUT, "libref_cpu.so")
    if force or not os.path.exists(cpu_out):
        tu = ["#include <algorithm>\n#include <vector>\n#include <math.h>\n#include <cmath>\n"
              "#include <cstring>\n#include <cstdio>\nusing namespace std;\nextern \"C\" {\n"]
        for rel, lo, hi, tr in _REF_RANGES:
            with open(os.path.join(REF, rel)) as f:
                lines = f.readlines()[lo - 1:hi]
            if tr == "unstatic":
                lines = [l[len("static "):] if l.startswith("static ") else l for l in lines]
            tu += lines
        tu.append("}\n")
        # same flags the reference's own build uses for host code (compile_ops.sh: g++ -O2), plus an
        # explicit contract=off (x86-64 g++ -O2 without -mfma cannot fuse anyway).
        _run(["g++", "-O2", "-std=c++11", "-fPIC", "-shared", "-w", "-ffp-contract=off", "-x", "c++", "-",
              "-o", cpu_out], input="".join(tu).encode())
    ss_out = os.path.join(REFOUT, "selection_sort")
    if force or not os.path.exists(ss_out):
        _run(["g++", "-O2", "-w", os.path.join(REF, "tf_ops/grouping/selection_sort.cpp"), "-o", ss_out])
    return True


if __name__ == "__main__":
    force = "--force" in sys.argv
    print("oracle:", build_oracle(force))
    print("reference build:", "done" if build_reference(force) else "skipped (no /root/reference)")
