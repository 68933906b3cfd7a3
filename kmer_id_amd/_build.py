"""Build the in-tree native artefacts with hipcc (gfx950 only).

`libkmer_id_amd.so` is a plain C-ABI shared library (include/kmer_id_amd.h); it is
compiled in-tree so that it travels to the GPU box with the repository snapshot.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
LIB = os.path.join(HERE, "libkmer_id_amd.so")
BIN_DIR = os.path.join(HERE, "bin")
NK10 = os.path.join(BIN_DIR, "nk10")
# front-end -> its main file; everything else under host/ is shared
FRONT_ENDS = {"nk10": "nk10_main.cpp", "kmer_read_vf6": "vf6_main.cpp", "kmer_read_m3": "m3_main.cpp"}

HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-value", "-Wno-unused-result"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X library cannot be built")
    return exe


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def lib_sources():
    return [os.path.join(CSRC, f) for f in ("kid_api.hip", "kid_kernels.hip.h", "kid_common.h")] + [
        os.path.join(ROOT, "include", "kmer_id_amd.h"), os.path.join(ROOT, "include", "kmer_id_amd_bench.h")]


def build_library(force=False, verbose=False):
    srcs = lib_sources()
    if force or _newer(LIB, srcs):
        cmd = [_hipcc()] + HIPCC_FLAGS + ["-shared", "-o", LIB, os.path.join(CSRC, "kid_api.hip")]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


def host_sources():
    if not os.path.isdir(HOST):
        return []
    return sorted(os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith((".cpp", ".h")))


def build_cli(force=False, verbose=False):
    """nk10: the reference-compatible command line program (host C++ over the C ABI)."""
    srcs = host_sources()
    mains = set(FRONT_ENDS.values())
    shared = [s for s in srcs if s.endswith(".cpp") and os.path.basename(s) not in mains]
    if not shared:
        return None
    build_library(force=force, verbose=verbose)
    os.makedirs(BIN_DIR, exist_ok=True)
    cxx = shutil.which("g++") or "g++"
    for name, main in FRONT_ENDS.items():
        main_path = os.path.join(HOST, main)
        if not os.path.exists(main_path):
            continue
        out = os.path.join(BIN_DIR, name)
        if force or _newer(out, srcs + [LIB]):
            cmd = [cxx, "-O2", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", out, main_path] + shared + [
                "-L", HERE, "-lkmer_id_amd", "-lz", "-lpthread", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib"]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    return NK10


def cli_path(name):
    return os.path.join(BIN_DIR, name)


def build_tools(force=False, verbose=False):
    """kid_synth_files: the synthetic workload written as the files the reference's programs read (a full-scale
    probes10.txt.gz, FASTQ.gz pairs); plain C++ + zlib, no GPU."""
    src = os.path.join(ROOT, "tools", "kid_synth_files.cpp")
    out = os.path.join(BIN_DIR, "kid_synth_files")
    if not os.path.exists(src):
        return None
    os.makedirs(BIN_DIR, exist_ok=True)
    if force or _newer(out, [src, os.path.join(CSRC, "kid_common.h")]):
        cmd = [shutil.which("g++") or "g++", "-O2", "-std=c++17", "-Wall", "-o", out, src, "-lz", "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    # kid_gzcat: the host code's gzip reader beside zlib's (tests/test_host_inflate.py)
    gsrc = [os.path.join(ROOT, "tools", "kid_gzcat.cpp"), os.path.join(HOST, "kid_inflate.cpp"), os.path.join(HOST, "kid_pargz.cpp"),
            os.path.join(HOST, "kid_textio.cpp")]
    gout = os.path.join(BIN_DIR, "kid_gzcat")
    if all(os.path.exists(s) for s in gsrc) and (force or _newer(gout, gsrc + [os.path.join(HOST, h) for h in ("kid_inflate.h", "kid_inflate_internal.h", "kid_pargz.h", "kid_textio.h")])):
        cmd = [shutil.which("g++") or "g++", "-O2", "-std=c++17", "-Wall", "-o", gout] + gsrc + ["-lz", "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return out


def build_all(force=False, verbose=False):
    build_library(force=force, verbose=verbose)
    build_cli(force=force, verbose=verbose)
    build_tools(force=force, verbose=verbose)
