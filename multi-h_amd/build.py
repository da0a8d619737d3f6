"""In-tree build of the gfx950 engine: multi-h_amd/libmultih_hip.so (C ABI,
include/multih_hip.h) and the C++ host layer multi-h_amd/libmultih_host.so
(class MultiH over the C ABI) plus its harness binary.

hipcc cross-compiles without a GPU; the .so files are git-ignored but travel to
the GPU box with the gpurun snapshot.  -ffp-contract=off is part of the
numerical contract (every FP64 operation rounds once, like the reference's
scalar C++ built by MSVC /fp:precise or g++ -ffp-contract=off)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
BUILD = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libmultih_hip.so")
HOST_LIB = os.path.join(HERE, "libmultih_host.so")
HARNESS = os.path.join(HERE, "multih_harness")
RCCL_LIB = os.path.join(HERE, "libmultih_rccl.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
HIP_FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
             "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
KERNEL_SOURCES = ["residual.hip", "dlt4.hip", "datacost.hip", "reestimate.hip", "expand.hip",
                  "knn.hip", "graph.hip", "fund.hip", "meanshift.hip", "refine.hip", "select.hip", "score32.hip", "compat.hip",
                  "capi.hip", "capi_front.hip", "capi_score.hip", "capi_select.hip", "capi_label.hip"]


def _newer(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _run(cmd: list[str]) -> None:
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + "\n")
        raise RuntimeError("build failed: " + os.path.basename(cmd[-1]))


def build_engine(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(BUILD, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".hpp")]
    headers.append(os.path.join(ROOT, "include", "multih_hip.h"))
    jobs = []
    objs = []
    for src in KERNEL_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(BUILD, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(o, [s] + headers):
            jobs.append([HIPCC] + HIP_FLAGS + ["-c", s, "-o", o])
    if jobs:
        if verbose:
            print(f"[build] compiling {len(jobs)} HIP translation unit(s) for {ARCH}")
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(_run, jobs))
    if force or jobs or _newer(LIB, objs):
        _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


def build_host(force: bool = False, verbose: bool = False) -> str:
    """C++ host mirror of the reference class (no HIP in these files; g++)."""
    srcs = [os.path.join(HOST, f) for f in ("MultiH.cpp", "merge_step.cpp", "approx_neighbours.cpp")]
    hdrs = [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".h")]
    hdrs.append(os.path.join(ROOT, "include", "multih_hip.h"))
    hdrs.append(os.path.join(ROOT, "include", "multih_rccl.h"))
    cxx = os.environ.get("CXX", "g++")
    flags = ["-O2", "-std=c++17", "-fPIC", "-pthread", "-ffp-contract=off", "-Wall", "-I" + os.path.join(ROOT, "include"),
             "-I" + HOST]
    if force or _newer(HOST_LIB, srcs + hdrs + [LIB]):
        if verbose:
            print("[build] host layer (class MultiH over the C ABI)")
        _run([cxx] + flags + ["-shared", "-o", HOST_LIB] + srcs +
             ["-L" + HERE, "-lmultih_hip", "-Wl,-rpath,$ORIGIN"])
    # the native multi-GPU transport (RCCL's ncclAllGather on the engine's stream): a library of its own, so that neither
    # the engine nor the host class depends on librccl
    rccl_src = os.path.join(HOST, "rccl_transport.cpp")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    if os.path.exists(rccl_src) and (force or _newer(RCCL_LIB, [rccl_src] + hdrs)):
        if verbose:
            print("[build] RCCL transport (libmultih_rccl.so)")
        _run([cxx, "-O2", "-std=c++17", "-fPIC", "-pthread", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I" + rocm + "/include", "-I" + HOST, "-I" + os.path.join(ROOT, "include"),
              "-shared", "-o", RCCL_LIB, rccl_src, "-L" + rocm + "/lib", "-lrccl", "-lamdhip64", "-Wl,-rpath," + rocm + "/lib"])
    main = os.path.join(HOST, "main.cpp")
    if os.path.exists(main) and (force or _newer(HARNESS, [main, HOST_LIB, RCCL_LIB] + hdrs)):
        _run([cxx] + flags + ["-o", HARNESS, main, "-L" + HERE, "-lmultih_host", "-lmultih_hip", "-ldl",
                              "-Wl,-rpath,$ORIGIN"])
    return HOST_LIB


def build_tuning(force: bool = False, verbose: bool = False) -> str:
    """The measurement library for tools/kernel_sweep.py: the same sources with -DMH_TUNING, which adds the residual /
    score kernel variants (other tilings, nt stores, compiler division, store-only calibration, fused multiply-adds)
    that the product library does not carry.  Load it with MH_LIB=multi-h_amd/libmultih_hip_tuning.so."""
    out_dir = os.path.join(HERE, "_build_tuning")
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(HERE, "libmultih_hip_tuning.so")
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".hpp")]
    headers.append(os.path.join(ROOT, "include", "multih_hip.h"))
    jobs, objs = [], []
    for src in KERNEL_SOURCES:
        s_, o = os.path.join(CSRC, src), os.path.join(out_dir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(o, [s_] + headers):
            jobs.append([HIPCC] + HIP_FLAGS + ["-DMH_TUNING", "-c", s_, "-o", o])
    if jobs:
        if verbose:
            print(f"[build] tuning library: compiling {len(jobs)} translation unit(s) with -DMH_TUNING")
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(_run, jobs))
    if force or jobs or _newer(lib, objs):
        _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs)
    return lib


def build_all(force: bool = False, verbose: bool = False) -> None:
    build_engine(force, verbose)
    if os.path.exists(os.path.join(HOST, "MultiH.cpp")):
        build_host(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
    print("ok:", LIB)
    if "--tuning" in sys.argv:
        print("ok:", build_tuning(force="--force" in sys.argv, verbose=True))
