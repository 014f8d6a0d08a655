"""In-tree build of libgenomicsdb_amd.so (HIP kernels for gfx950 + host layer + C ABI) with hipcc.

hipcc cross-compiles gfx950 without a GPU; the .so lands next to this file so that it travels with the
repo snapshot to the GPU box.  Also builds the test oracle (oracle/Makefile) and the hostsim harness.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(ROOT, "build", "obj")
LIB = os.path.join(PKG, "libgenomicsdb_amd.so")
TOOL = os.path.join(PKG, "gt_mpi_gather")
TOOLS = {"gt_mpi_gather": TOOL, "vcf2tiledb": os.path.join(PKG, "vcf2tiledb")}

SOURCES = [
    "kernels/gdb_pipeline.hip",
    "kernels/gdb_bgzf.hip",
    "host/vid_mapper.cc",
    "host/variant_query_config.cc",
    "host/combine_plan.cc",
    "host/fragment.cc",
    "host/reference_genome.cc",
    "host/vcf_importer.cc",
    "host/vcf_index.cc",
    "api/genomicsdb_bcf_generator.cc",
    "api/genomicsdb_operators.cc",
    "api/capi.cc",
]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-gline-tables-only", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def _headers():
    out = []
    for d, _, fs in os.walk(CSRC):
        out += [os.path.join(d, f) for f in fs if f.endswith((".h", ".hpp", ".inc"))]
    out.append(os.path.join(ROOT, "include", "genomicsdb_amd.h"))
    return out


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hdrs = _headers()
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace("/", "_") + ".o")
        objs.append(obj)
        if _newer(obj, [src] + hdrs):
            cmd = [HIPCC] + FLAGS + (["-x", "hip"] if s.endswith(".hip") else []) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lz"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    # command lines of the reference's query tool (--produce-Broad-GVCF mode) and import tool, linked against the library
    for name, exe in TOOLS.items():
        tool_src = os.path.join(CSRC, "tools", name + ".cc")
        if _newer(exe, [tool_src, LIB] + hdrs):
            cmd = [HIPCC, "-O2", "-std=c++17", "-Wall", "-Wno-unused-function", tool_src, "-o", exe, "-L" + PKG, "-lgenomicsdb_amd", "-Wl,-rpath,$ORIGIN", "-lz"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    build_jni(verbose)
    return LIB


def build_jni(verbose=False):
    """libtiledbgenomicsdb.so (the name the reference's Java loader asks for) = the GenomicsDBQueryStream natives and the
    one-time initialiser over the C ABI.  Compiled against a JDK's jni.h when $JAVA_HOME has one, else against the minimal
    stand-in csrc/jni/stub/jni.h (specification function-table indices only), so the glue is built and symbol-checked always."""
    jh = os.environ.get("JAVA_HOME", "")
    inc = os.path.join(jh, "include") if jh else ""
    if inc and os.path.exists(os.path.join(inc, "jni.h")):
        incs = ["-I" + inc, "-I" + os.path.join(inc, "linux")]
    else:
        incs = ["-I" + os.path.join(CSRC, "jni", "stub")]
    out = os.path.join(PKG, "libtiledbgenomicsdb.so")
    src = os.path.join(CSRC, "jni", "jni_query_stream.cc")
    if _newer(out, [src, LIB, os.path.join(CSRC, "jni", "stub", "jni.h"), os.path.join(ROOT, "include", "genomicsdb_amd.h")]):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall"] + incs + [src, "-o", out, "-L" + PKG, "-lgenomicsdb_amd", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return out


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return os.path.join(ROOT, "oracle", "liboracle_gvcf.so")


def build_hostsim():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "hostsim")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "jni_harness")])   # JVM-less driver of the JNI glue (tests only)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "compat")])        # reference-shaped C++ caller (tests only)


if __name__ == "__main__":
    print(build_native(verbose=True))
    print(build_oracle())
