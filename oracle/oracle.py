"""ctypes face of oracle/zdw_oracle.c and a runner for the compiled reference.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under zdw_amd/ may import this module.
"""
import ctypes as C
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libzdw_oracle.so")
REF_BIN = os.path.join(HERE, "_ref", "convertDWfile")
REF_SHIM = os.path.join(HERE, "_ref", "shim")


class Info(C.Structure):
    _fields_ = [
        ("nblocks", C.c_uint32), ("rows_total", C.c_uint32), ("bad_row", C.c_uint32),
        ("longest_line", C.c_uint32), ("dict_entries", C.c_uint64), ("dict_bytes", C.c_uint64),
        ("used_cols", C.c_uint32), ("reserved", C.c_uint32), ("header_bytes", C.c_uint64),
    ]


_lib = None


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(
            os.path.join(HERE, "zdw_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", HERE, os.path.join(HERE, "libzdw_oracle.so")])


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.zdwo_encode_desc.restype = C.c_int
        _lib.zdwo_encode_desc.argtypes = [
            C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t,
            C.POINTER(C.c_char_p), C.c_uint32, C.c_int,
            C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32,
            C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(Info)]
        _lib.zdwo_free.argtypes = [C.c_void_p]
        _lib.zdwo_set_mem_limit.argtypes = [C.c_double]
        _lib.zdwo_set_mem_limit.restype = None
    return _lib


def encode(desc: bytes, tsv, metadata=None, trim=False, split_rows=(), split_cols=None, mem_limit_mb=0.0):
    """Returns (rc, zdw_bytes_or_None, Info).  `tsv` is bytes or a C-contiguous uint8 numpy array.
    mem_limit_mb: the reference's --mem-limit as the deterministic model of zdw_oracle.c (0 = blocks end only
    where split_rows says)."""
    L = lib()
    L.zdwo_set_mem_limit(float(mem_limit_mb))
    meta = list((metadata or {}).items())
    arr = (C.c_char_p * (2 * len(meta) or 1))()
    for i, (k, v) in enumerate(meta):
        arr[2 * i] = k if isinstance(k, bytes) else k.encode()
        arr[2 * i + 1] = v if isinstance(v, bytes) else v.encode()
    ns = len(split_rows)
    sr = (C.c_uint32 * (ns or 1))(*split_rows)
    sc = (C.c_uint32 * (ns or 1))(*(split_cols if split_cols is not None else [0] * ns))
    out = C.c_void_p()
    out_len = C.c_size_t()
    info = Info()
    if isinstance(tsv, (bytes, bytearray)):
        tsv_buf = (C.c_char * len(tsv)).from_buffer_copy(tsv) if len(tsv) else None
        tsv_ptr, n = C.cast(tsv_buf, C.c_void_p) if tsv_buf is not None else None, len(tsv)
    else:  # numpy
        tsv_ptr, n = C.c_void_p(tsv.ctypes.data), tsv.size
    rc = L.zdwo_encode_desc(desc, len(desc), tsv_ptr, n, arr, len(meta), int(bool(trim)),
                            sr, sc if split_cols is not None else None, ns,
                            C.byref(out), C.byref(out_len), C.byref(info))
    data = None
    if rc == 0:
        data = C.string_at(out, out_len.value)
        L.zdwo_free(out)
    return rc, data, info


# ---------------------------------------------------------------------------------------------
# the compiled reference (oracle/_ref/convertDWfile), when it has been built
# ---------------------------------------------------------------------------------------------
def have_ref():
    return os.path.exists(REF_BIN) and os.path.exists(os.path.join(REF_SHIM, "gzip"))


def run_ref(desc: bytes, tsv: bytes, args=(), metadata_file: bytes = None, stem="t", timeout=600):
    """Run the reference CLI on (desc, tsv) in a scratch dir with the compressor replaced by `cat`.

    Returns (exit_code, raw_zdw_bytes_or_None, stdout, stderr).
    """
    d = tempfile.mkdtemp(prefix="zdwref_")
    try:
        with open(os.path.join(d, stem + ".sql"), "wb") as f:
            f.write(tsv)
        with open(os.path.join(d, stem + ".desc.sql"), "wb") as f:
            f.write(desc)
        if metadata_file is not None:
            with open(os.path.join(d, stem + ".metadata"), "wb") as f:
                f.write(metadata_file)
        env = dict(os.environ)
        env["PATH"] = REF_SHIM + os.pathsep + env.get("PATH", "")
        p = subprocess.run([REF_BIN, *args, os.path.join(d, stem + ".sql")], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
        outp = os.path.join(d, stem + ".zdw.gz")
        data = open(outp, "rb").read() if os.path.exists(outp) else None
        return p.returncode, data, p.stdout, p.stderr
    finally:
        shutil.rmtree(d, ignore_errors=True)
