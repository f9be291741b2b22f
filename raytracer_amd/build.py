"""In-tree build of librt_dp.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m raytracer_amd.build [--force]

Flags that matter for bit-exactness: -ffp-contract=off (the RTL has separate
multiplier and adder units, never a fused multiply-add), no fast-math, and f32
denormals left enabled (the hipcc default for gfx9).
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
LIB = os.path.join(LIBDIR, "librt_dp.so")
ROOT = os.path.dirname(_HERE)

ARCH = "gfx950"
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
          "-fno-gpu-flush-denormals-to-zero", "-Wall", "-Wno-unused-function",
          "-I", os.path.join(ROOT, "include"), "-I", CSRC]
SOURCES = ["kernels.hip", "knn.hip", "stream.hip", "stream_fused.hip", "trace.hip", "compact.hip", "pairs.hip", "build_device.hip", "build_kd.hip", "api.cpp", "build_index.cpp", "shard.cpp"]
HOST_TOOLS = {"rt_dp_replay": ["rt_dp_replay.cpp"]}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _compile(job):
    cmd, verbose = job
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build(force=False, verbose=False):
    """One object per source (compiled in parallel, only when the source or a header is newer), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))] + [os.path.join(ROOT, "include", "rt_dp.h"), __file__]
    jobs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [os.path.join(CSRC, src)] + headers):
            jobs.append(([_hipcc(), "--offload-arch=" + ARCH, "-x", "hip"] + COMMON + ["-c", os.path.join(CSRC, src), "-o", obj], verbose))
    if jobs:
        workers = max(1, min(len(jobs), int(os.environ.get("MAX_JOBS", "0")) or (os.cpu_count() or 4), 8))
        with ThreadPoolExecutor(workers) as pool:
            list(pool.map(_compile, jobs))
    if force or jobs or _stale(LIB, objs):
        _compile(([_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs, verbose))
    for tool, srcs in HOST_TOOLS.items():
        paths = [os.path.join(CSRC, s) for s in srcs]
        if not all(os.path.exists(p) for p in paths):
            continue
        out = os.path.join(LIBDIR, tool)
        if force or _stale(out, paths + [LIB]):
            cmd = ["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", out] + paths + \
                  ["-L", LIBDIR, "-lrt_dp", "-pthread", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    for name in ("kat_from_c", "multi_gpu"):   # plain C: the header and the ABI without any C++
        ex_src, ex = os.path.join(ROOT, "examples", name + ".c"), os.path.join(LIBDIR, name)
        if os.path.exists(ex_src) and (force or _stale(ex, [ex_src, LIB, os.path.join(ROOT, "include", "rt_dp.h")])):
            cmd = ["gcc", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), ex_src, "-o", ex, "-L", LIBDIR, "-lrt_dp", "-lm", "-pthread",
                   "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    mb_src, mb = os.path.join(CSRC, "membench.hip"), os.path.join(LIBDIR, "membench")
    if os.path.exists(mb_src) and (force or _stale(mb, [mb_src])):
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", mb_src, "-o", mb]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
