"""Build libdispu_hip.so (hand-written HIP for gfx950) in-tree: dis-pu_amd/lib/libdispu_hip.so.

hipcc cross-compiles without a GPU.  Flags: -ffp-contract=off pins the arithmetic (every fused
multiply-add in csrc/ is an explicit __builtin_fmaf); gfx950 only, no other offload arch.
"""
import concurrent.futures as cf
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(OUT_DIR, "libdispu_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function"]


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))


def source_hash():
    """sha1 over the kernel sources (csrc/*.hip, csrc/*.h, this file's flags): names the build a profile was taken from -- the GPU box
    has no .git, so counter summaries are stamped with this and bench.py checks it against the tree it runs from."""
    import hashlib
    h = hashlib.sha1(" ".join(FLAGS).encode())
    for f in _sources() + _headers():
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OUT_DIR, os.path.basename(src)[:-4] + ".o")
    if _stale(obj, [src, os.path.abspath(__file__)] + _headers()):
        r = subprocess.run([HIPCC] + FLAGS + ["-c", src, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stdout.decode(errors="replace")))
        out = r.stdout.decode(errors="replace").strip()
        if out:
            print(out)
    return obj


def build(force=False, verbose=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    srcs = _sources()
    if force:
        for f in os.listdir(OUT_DIR):
            if f.endswith(".o") or f.endswith(".so"):
                os.remove(os.path.join(OUT_DIR, f))
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    if _stale(LIB, objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB + ".tmp"],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stdout.decode(errors="replace"))
        os.replace(LIB + ".tmp", LIB)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    import sys
    build(force="--force" in sys.argv, verbose=True)
