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
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={ARCH}", "-ffp-contract=fast",
         "-Wall", "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [
        os.path.join(HERE, "..", "include", "pigs_amd.h")]


STAMP = LIB + ".srchash"


def source_hash():
    """Hash of every source the library is built from plus the flags (content, not mtimes: the
    tree is copied between machines and copies do not keep timestamps)."""
    h = hashlib.sha256(" ".join(FLAGS).encode())
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
    cmd = [hipcc] + FLAGS + ["-o", tmp] + sources()
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + proc.stdout)
    if verbose and proc.stdout.strip():
        print(proc.stdout)
    os.replace(tmp, LIB)
    with open(f"{STAMP}.{os.getpid()}.tmp", "w") as f:
        f.write(source_hash())
    os.replace(f"{STAMP}.{os.getpid()}.tmp", STAMP)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
