"""Build pigs_amd/libpigs_amd.so (the C-ABI library of include/pigs_amd.h) with hipcc for gfx950,
and pigs_amd/_pigs_host.so (the native host side of GaussianSampler: a torch C++ extension over
that C ABI, csrc_host/pigs_host.cpp) with g++.

    python -m pigs_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU, so this also runs in GPU-less containers.  Both libraries are
built in-tree (git-ignored) so that they travel with the source tree to the GPU box.
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


# ---- the native host extension (torch C++ extension; no device code: plain g++) -----------------
HOST_SRC = os.path.join(HERE, "csrc_host", "pigs_host.cpp")
HOST_LIB = os.path.join(HERE, "_pigs_host.so")
HOST_STAMP = HOST_LIB + ".srchash"
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
              "-DTORCH_EXTENSION_NAME=_pigs_host", "-DTORCH_API_INCLUDE_EXTENSION_H", "-w"]


def host_source_hash():
    import torch
    h = hashlib.sha256((" ".join(HOST_FLAGS) + torch.__version__).encode())
    for p in (HOST_SRC, os.path.join(HERE, "..", "include", "pigs_amd.h")):
        h.update(open(p, "rb").read())
    return h.hexdigest()


def host_needs_build():
    if not os.path.exists(HOST_LIB) or not os.path.exists(HOST_STAMP):
        return True
    return open(HOST_STAMP).read().strip() != host_source_hash()


def build_host(force=False, verbose=False):
    """Compile the native host extension against this interpreter's torch.  Needs libpigs_amd.so
    (it links against the C ABI).  Returns the extension path."""
    build(force=False, verbose=verbose)
    if not force and not host_needs_build():
        return HOST_LIB
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    tdir = os.path.dirname(torch.__file__)
    rocm = os.environ.get("ROCM_HOME") or ce.ROCM_HOME or "/opt/rocm"
    inc = [f"-I{p}" for p in ce.include_paths()] + [f"-I{rocm}/include", f"-I{sysconfig.get_paths()['include']}"]
    abi = f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"
    tmp = f"{HOST_LIB}.{os.getpid()}.tmp"
    cmd = ([os.environ.get("CXX", "g++")] + HOST_FLAGS + [abi] + inc + [HOST_SRC, "-o", tmp,
           f"-L{tdir}/lib", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip", "-ltorch_python",
           f"-L{rocm}/lib", "-lamdhip64", f"-L{HERE}", "-lpigs_amd", "-Wl,-rpath,$ORIGIN"])
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("host extension build failed:\n" + proc.stdout)
    os.replace(tmp, HOST_LIB)
    with open(f"{HOST_STAMP}.{os.getpid()}.tmp", "w") as f:
        f.write(host_source_hash())
    os.replace(f"{HOST_STAMP}.{os.getpid()}.tmp", HOST_STAMP)
    return HOST_LIB


def build_all(force=False, verbose=False):
    build(force=force, verbose=verbose)
    build_host(force=force, verbose=verbose)
    return LIB


def under_profiler():
    """True inside a rocprofv3 run: the profiler's preloaded tool library initialises the GPU before the
    program starts, and everything the process forks inherits the preload."""
    env = os.environ
    return ("rocprof" in env.get("LD_PRELOAD", "") or "ROCP_TOOL_LIBRARIES" in env
            or "ROCPROFILER_REGISTER_FORCE_LOAD" in env or any(k.startswith("ROCPROF_") for k in env))


def ensure_built(wait_for_rank0=False):
    """What an entry point (bench.py, tools/prof_*.py) calls as its FIRST statement, before anything touches
    the GPU: building forks hipcc / g++, whose wrappers exec the compiler proper, and an exec chain behind a
    GPU-initialised (or profiled) process is what the GPU pool forbids.  Inside a rocprofv3 run nothing is
    ever built: stale libraries end the run with the command that fixes it.  ``wait_for_rank0``: a rank other
    than local rank 0 waits for rank 0's build instead of racing it."""
    stale = needs_build() or host_needs_build()
    if not stale:
        return LIB
    if under_profiler():
        sys.exit("pigs_amd: the in-tree libraries are stale and this is a profiler run -- run "
                 "`python -m pigs_amd.build` first (nothing is compiled behind rocprofv3)")
    if wait_for_rank0:
        import time
        t0 = time.time()
        while needs_build() or host_needs_build():
            if time.time() - t0 > 900:
                sys.exit("pigs_amd: gave up waiting for local rank 0 to build the libraries")
            time.sleep(0.5)
        return LIB
    return build_all()


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
    print(LIB)
    print(HOST_LIB)
