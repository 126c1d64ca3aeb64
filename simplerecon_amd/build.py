"""Builds simplerecon_amd/libsimplerecon_hip.so for gfx950 with hipcc (in-tree).

    python -m simplerecon_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU.  One object per .hip translation unit (parallel),
linked into a single C-ABI shared library (include/simplerecon_hip.h)."""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
BUILD = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(HERE, "libsimplerecon_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC,
         "-Wall", "-Wno-unused-function", "-fno-fast-math"] + os.environ.get("SR_EXTRA_HIPCC_FLAGS", "").split()


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return hdrs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, verbose):
    obj = os.path.join(BUILD, os.path.basename(src) + ".o")
    cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr)
    return obj


def build(force=False, verbose=False):
    os.makedirs(BUILD, exist_ok=True)
    srcs = sources()
    hdrs = _deps()
    todo = [s for s in srcs if force or _stale(os.path.join(BUILD, os.path.basename(s) + ".o"), [s] + hdrs)]
    if todo:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(todo))) as ex:
            list(ex.map(lambda s: _compile(s, verbose), todo))
    objs = [os.path.join(BUILD, os.path.basename(s) + ".o") for s in srcs]
    if force or todo or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-L/opt/rocm/lib", "-lhipblaslt"]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
