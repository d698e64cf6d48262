"""Build libivid_hip.so (gfx950 only) in-tree with hipcc.

`python -m ivid_amd.build` or `ivid_amd.build.build_all()`.  Objects are cached by mtime under
ivid_amd/csrc/build/; the shared library lands in ivid_amd/lib/ so it travels with the tree.
hipcc cross-compiles for gfx950 without a GPU present.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libivid_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"] + os.environ.get("IVID_EXTRA_HIPCC_FLAGS", "").split()


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "ivid_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, verbose):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    spath = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(spath), _headers_mtime()):
        return obj
    cmd = [HIPCC] + FLAGS + ["-c", spath, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build_all(verbose=True, force=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, verbose), srcs))
    if (not os.path.exists(LIB)) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    _build_host_example(verbose)
    return LIB


HOST_SRC = os.path.join(os.path.dirname(HERE), "examples", "unet_engine_host.c")
HOST_BIN = os.path.join(LIBDIR, "unet_engine_host")
LOOP_SRC = os.path.join(os.path.dirname(HERE), "examples", "sample_loop_host.c")
LOOP_BIN = os.path.join(LIBDIR, "sample_loop_host")


def _build_host_example(verbose):
    """examples/unet_engine_host.c, examples/sample_loop_host.c: C hosts that run a planned forward / a whole sampling loop from
    engine files through the C ABI alone (plain C compiler, linked against libivid_hip.so and the HIP runtime)."""
    for src, binary in ((HOST_SRC, HOST_BIN), (LOOP_SRC, LOOP_BIN)):
        if not os.path.exists(src):
            continue
        if os.path.exists(binary) and os.path.getmtime(binary) > max(os.path.getmtime(src), os.path.getmtime(LIB)):
            continue
        rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
        cmd = ["gcc", "-O2", "-std=c11", "-I" + os.path.join(rocm, "include"), src, "-o", binary, "-L" + LIBDIR, "-livid_hip",
               "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(rocm, "lib")]
        if verbose:
            print(" ".join(cmd), flush=True)
        try:
            r = subprocess.run(cmd, capture_output=True, text=True)
            err = None if r.returncode == 0 else f"{r.stdout}\n{r.stderr}"
        except OSError as e:                      # no gcc on this machine
            err = str(e)
        if err is not None:
            # an optional example must not turn a library that linked into a failed build: say so and carry on (the tests that run
            # the C hosts, tests/test_engine_gpu.py and tests/test_sample_loop_gpu.py, report a missing binary themselves)
            import warnings
            warnings.warn(f"{os.path.relpath(src, os.path.dirname(HERE))} was not built (the library itself is fine):\n{err}")


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv))
