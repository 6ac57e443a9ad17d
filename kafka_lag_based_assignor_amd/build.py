"""Builds the native code in-tree.

    python -m kafka_lag_based_assignor_amd.build [--force]

* liblagassign.so  -- the HIP kernels + C ABI of include/lagassign.h (gfx950 only, hipcc).
                      No Python or torch dependency: it is what a JNI shim or ctypes loads.
* _host.*.so       -- the C++ host mirror of the reference's plugin class, bound with pybind11.
                      It links liblagassign.so and uses nothing but its C ABI.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "liblagassign.so")
HOST = os.path.join(HERE, "_host" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
SOURCES = ["la_api.hip", "la_lag.hip", "la_wave_tile.hip", "la_wave_tile_l8.hip", "la_wave_tile_l16.hip",
           "la_wave_tile_l32.hip", "la_wave_tile_l64.hip", "la_large.hip", "la_block.hip", "la_wire.hip"]
HOST_SOURCES = ["host/lag_based_partition_assignor.cpp", "host/pybind_host.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
         "-Wall", "-Wextra", "-Wno-unused-parameter"] + os.environ.get("LA_EXTRA_HIPCC_FLAGS", "").split()


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build liblagassign.so")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers():
    hs = []
    for root, _, files in os.walk(CSRC):
        if os.path.basename(root) == "build":
            continue
        hs += [os.path.join(root, f) for f in files if f.endswith((".h", ".hpp"))]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "lagassign.h"))
    return hs


def build(force: bool = False, verbose: bool = False, host: bool = True) -> str:
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    headers = _headers()
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc, *FLAGS, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, 8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--no-undefined", "-o", LIB, *objs, "-ldl"])
    if host:
        build_host(force=force, verbose=verbose)
    return LIB


def build_host(force: bool = False, verbose: bool = False) -> str:
    import pybind11
    srcs = [os.path.join(CSRC, s) for s in HOST_SOURCES]
    if force or _stale(HOST, srcs + _headers() + [LIB]):
        cxx = shutil.which("g++") or "g++"
        cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall",
               "-I" + sysconfig.get_paths()["include"], "-I" + pybind11.get_include(),
               *srcs, "-o", HOST, "-L" + HERE, "-llagassign", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return HOST


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
