"""Build pigs_amd/libpigs_amd.so (the C-ABI library of include/pigs_amd.h) with hipcc for gfx950.

    python -m pigs_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU, so this also runs in GPU-less containers.  The library is
built in-tree (git-ignored) so that it travels with the source tree to the GPU box.
"""
import glob
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpigs_amd.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=fast",
         "-Wall", "-Wno-unused-function"]
# per-file additions.  plan.hip: hipcc's SLP vectoriser packs the per-pair arithmetic into v_pk_*_f32
# with v_mov shuffles around them; a packed f32 op costs two plain ones on gfx950, so the shuffles are
# pure loss (MI355X_MICROARCH.md, 'packed f32 VALU')
FILE_FLAGS = {"plan.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [
        os.path.join(HERE, "..", "include", "pigs_amd.h")]


STAMP = LIB + ".srchash"


def source_hash():
    """Hash of every source the library is built from plus the flags (content, not mtimes: the
    tree is copied between machines and copies do not keep timestamps)."""
    h = hashlib.sha256((" ".join(FLAGS) + repr(sorted(FILE_FLAGS.items()))).encode())
    for p in _deps():
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    return open(STAMP).read().strip() != source_hash()


def build(force=False, verbose=False):
    """Compile every HIP source into one shared library.  Returns the library path."""
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "hipcc")
    tmp = f"{LIB}.{os.getpid()}.tmp"      # several ranks may build at once: no shared temp file
    objdir = os.path.join(HERE, "build", f"obj.{os.getpid()}")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in sources():                 # one hipcc per source, all at once
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [hipcc] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    objs, log, failed = [], "", False
    for obj, proc in procs:
        out, _ = proc.communicate()
        log += out
        failed = failed or proc.returncode != 0
        objs.append(obj)
    if failed:
        raise RuntimeError("hipcc failed:\n" + log)
    proc = subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", tmp] + objs,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + log + proc.stdout)
    if verbose and (log + proc.stdout).strip():
        print(log + proc.stdout)
    for obj in objs:
        os.remove(obj)
    os.rmdir(objdir)
    os.replace(tmp, LIB)
    with open(f"{STAMP}.{os.getpid()}.tmp", "w") as f:
        f.write(source_hash())
    os.replace(f"{STAMP}.{os.getpid()}.tmp", STAMP)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
