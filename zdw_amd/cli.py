"""The convertDWfile command line, in-process (libzdw_cli.so exports the CLI's main as
zdw_convertDWfile_main) or as a child process (zdw_amd/lib/convertDWfile)."""
import ctypes as C
import os
import subprocess

from . import build as _build

_cli = None


def _lib():
    global _cli
    if _cli is None:
        if not os.path.exists(_build.LIB_CLI):
            raise RuntimeError("libzdw_cli.so is not built: " + _build.LIB_CLI)
        C.CDLL(_build.LIB_HIP, mode=C.RTLD_GLOBAL)
        _cli = C.CDLL(_build.LIB_CLI)
        _cli.zdw_convertDWfile_main.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
        _cli.zdw_convertDWfile_main.restype = C.c_int
    return _cli


def main_inprocess(args, exe_name=None):
    """Runs the command line inside this process; returns its exit code."""
    argv = [(exe_name or _build.CLI_EXE).encode()] + [a.encode() if isinstance(a, str) else a for a in args]
    arr = (C.c_char_p * (len(argv) + 1))(*argv, None)
    return _lib().zdw_convertDWfile_main(len(argv), arr)


def run(args, env=None, **kw):
    """Runs zdw_amd/lib/convertDWfile as a child process."""
    return subprocess.run([_build.CLI_EXE] + list(args), env=env, **kw)
