"""Build libemu_hip.so (hand-written gfx950 HIP kernels + C ABI) in-tree with hipcc.

    python -m emu_amd.build            # incremental
    python -m emu_amd.build --force

The shared object lands next to the sources (emu_amd/csrc/libemu_hip.so) so it travels with the tree; it is
git-ignored.  hipcc cross-compiles for gfx950 without a GPU present.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libemu_hip.so")
SOURCES = ["gemv.hip", "gemv_merge.hip", "gemv_thin.hip", "decode_layer.hip", "decode_engine.hip", "gemm.hip", "gemm256.hip", "gemm_w4.hip", "attention.hip", "beam.hip", "elementwise.hip", "unet.hip", "p2p.hip", "engine.hip", "unet_engine.hip"]
HEADERS = ["common.h", "kernels.h", "gemm_tile.h", os.path.join("..", "..", "include", "emu_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, trace: bool = False) -> str:
    """trace=True: the tools-only twin libemu_hip_trace.so (-DEMU_TRACE: GEMM kernels write per-workgroup timelines,
    tools/gemm_trace.py), objects in csrc/trace_obj/; never loaded by the product (emu_amd/_lib.py)."""
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, jobs = [], []
    odir = os.path.join(CSRC, "trace_obj") if trace else CSRC
    os.makedirs(odir, exist_ok=True)
    lib_path = os.path.join(CSRC, "libemu_hip_trace.so") if trace else LIB
    flags = FLAGS + (["-DEMU_TRACE"] if trace else [])
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(odir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc, *flags, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r.stderr

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        for warn in ex.map(run, jobs):
            if warn and verbose:
                print(warn, file=sys.stderr)
    if force or jobs or _stale(lib_path, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib_path,
             "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"])
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, trace="--trace" in sys.argv))
