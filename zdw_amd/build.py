"""In-tree build of the native code (hipcc --offload-arch=gfx950; g++ for the host shim)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.environ.get("ZDW_LIBDIR") or os.path.join(HERE, "lib")      # (ZDW_LIBDIR: the profiling tools' own build, tools/build_experiments.sh)
LIB_HIP = os.path.join(LIBDIR, "libzdw_hip.so")
LIB_CLI = os.path.join(LIBDIR, "libzdw_cli.so")
CLI_EXE = os.path.join(LIBDIR, "convertDWfile")
LIB_SYNTH = os.path.join(LIBDIR, "libzdw_synth.so")
DECODER_EXE = os.path.join(LIBDIR, "unconvertDWfile")
TRIM_EXE = os.path.join(LIBDIR, "trim_spaces")


def build(jobs=4, quiet=True):
    """(Re)build libzdw_hip.so, libzdw_cli.so and convertDWfile.  make decides what is stale."""
    if os.environ.get("ZDW_LIBDIR"):
        # A library directory of the tools' own (tools/build_experiments.sh, build_variant.sh) was compiled with flags this function
        # does not know: rebuilding stale objects in it without them would mix two builds in one library.  Use it as it is.
        for p in (LIB_HIP, LIB_SYNTH):
            if not os.path.exists(p):
                raise RuntimeError("ZDW_LIBDIR is set but holds no " + p + " (build it with the tool that owns the directory)")
        return LIB_HIP
    cmd = ["make", "-C", CSRC, "-j%d" % jobs, "all", "OUT=" + LIBDIR]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL if quiet else None)
    for p in (LIB_HIP, LIB_CLI, CLI_EXE, LIB_SYNTH, DECODER_EXE, TRIM_EXE):
        if not os.path.exists(p):
            raise RuntimeError("native build did not produce " + p)
    return LIB_HIP


def have_sources():
    return os.path.exists(os.path.join(CSRC, "Makefile"))
