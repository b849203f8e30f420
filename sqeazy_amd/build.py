"""Builds sqeazy_amd/lib/libsqeazy_amd.so (HIP kernels + C-ABI) for gfx950 with hipcc.

In-tree on purpose: the built .so travels to the GPU box with the snapshot.  hipcc cross-compiles
without a GPU.  `python -m sqeazy_amd.build` or `sqeazy_amd.build.build()`.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsqeazy_amd.so")
SOURCES = ["sqy_kernels.hip", "sqy_pipeline.cpp", "sqy_capi.cpp", "sqy_rccl.cpp"]
HEADERS = ["sqy_kernels.h", "sqy_pipeline.hpp", os.path.join("..", "..", "include", "sqeazy_amd.h")]
ARCH = "gfx950"
# the one and only configuration of libsqeazy_amd.so; kernel experiments live in tools/ and build their own binaries
# (tried: -mllvm -amdgpu-sched-strategy=max-ilp -- the LZ4 parse kernels alone 3 % faster, the bench with three calls in flight 1.5 % slower)
# SQY_EXTRA_FLAGS: experiment builds of tools/ only (-DSQY_EXP_STATS for tools/exp_stats.py); the product is built without it
FLAGS = ["-O3", "-fPIC", "-std=c++17", "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-DSQY_PRODUCT_BUILD"] + os.environ.get("SQY_EXTRA_FLAGS", "").split()
BINDIR = os.path.join(HERE, "bin")
CLI = os.path.join(BINDIR, "sqy")                 # command line front end over the C-ABI (csrc/sqy_cli.cpp)


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(CLI):
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(CLI))
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS + ["sqy_cli.cpp", "sqy_h5_filter.c"]] + [os.path.abspath(__file__)]   # (FLAGS live here)
    deps += [os.path.join(HERE, "..", "tools", f) for f in ("slabs_c_test.c", "h5_roundtrip.c")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    common = list(FLAGS)
    for src in SOURCES:
        obj = os.path.join(LIBDIR, os.path.splitext(src)[0] + ".o")
        cmd = [_hipcc(), "--offload-arch=" + ARCH] + common + ["-c", os.path.join(CSRC, src), "-o", obj]
        if src.endswith(".cpp"):
            cmd.insert(1, "-x")
            cmd.insert(2, "hip")
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]     # (RCCL is dlopen'ed at first use: sqy_rccl.cpp)
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    # the `sqy` tool: plain host C++ over the exported C symbols only, finds the library next to itself
    os.makedirs(BINDIR, exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", os.path.join(CSRC, "sqy_cli.cpp"), "-o", CLI, "-L" + LIBDIR, "-lsqeazy_amd",
           "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath-link," + os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    # a plain C caller of the throughput entry point (tools/slabs_c_test.c): C compiler, HIP runtime API for device memory only
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["gcc", "-O2", "-fopenmp", "-Wall", "-I" + os.path.join(rocm, "include"), os.path.join(HERE, "..", "tools", "slabs_c_test.c"), "-o",
           os.path.join(BINDIR, "slabs_c_test"), "-L" + LIBDIR, "-lsqeazy_amd", "-L" + os.path.join(rocm, "lib"), "-lamdhip64",
           "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath," + os.path.join(rocm, "lib")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    build_h5_plugin(verbose)
    return LIB


HDF5_ROOT = os.environ.get("SQY_HDF5_ROOT", "/opt/conda")       # this image ships HDF5 1.10.6 (C library + headers) there
H5_PLUGIN = os.path.join(LIBDIR, "libh5sqy_amd.so")
H5_TOOL = os.path.join(BINDIR, "h5_roundtrip")


def build_h5_plugin(verbose=False):
    """HDF5 filter plugin (csrc/sqy_h5_filter.c) and its test client, when an HDF5 C library is present.

    libhdf5 is linked by file and found at run time through a symlink next to our library: adding its directory to a
    search path would also put that directory's (older) libstdc++ in front of the system's."""
    inc = os.path.join(HDF5_ROOT, "include")
    so = None
    for name in ("libhdf5.so.103", "libhdf5.so.200", "libhdf5.so"):
        if os.path.exists(os.path.join(HDF5_ROOT, "lib", name)):
            so = os.path.join(HDF5_ROOT, "lib", name)
            break
    if so is None or not os.path.exists(os.path.join(inc, "hdf5.h")):
        return None
    link = os.path.join(LIBDIR, os.path.basename(so))
    if os.path.lexists(link):
        os.remove(link)
    os.symlink(so, link)
    common = ["gcc", "-O2", "-Wall", "-I" + inc]
    cmds = [common + ["-shared", "-fPIC", "-fvisibility=hidden", os.path.join(CSRC, "sqy_h5_filter.c"), "-o", H5_PLUGIN,
                      "-L" + LIBDIR, "-lsqeazy_amd", so, "-Wl,-rpath,$ORIGIN",
                      "-Wl,-rpath-link," + os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")],
            common + [os.path.join(HERE, "..", "tools", "h5_roundtrip.c"), "-o", H5_TOOL, "-L" + LIBDIR, "-lsqeazy_amd", so,
                      "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath-link," + os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")]]
    for cmd in cmds:
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return H5_PLUGIN


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
