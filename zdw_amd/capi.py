"""ctypes binding of include/zdw_hip.h.  Fails loudly when libzdw_hip.so is missing: there is no
CPU fallback for the encode path."""
import ctypes as C
import os

from . import build as _build

OK, OUT_OF_MEMORY, WRONG_NUM_OF_COLUMNS_ON_A_ROW, UNKNOWN_ERROR = 0, 7, 15, 20
NUM_STAGES = 12
STAGE_NAMES = ["scan_rows", "pass1", "dict_compact", "dict_sort", "dict_emit", "pass2_size", "pass2_emit", "h2d", "d2h"]


class Opts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("trim_trailing_spaces", C.c_uint32),
                ("first_row_capacity", C.c_uint32), ("dict_slots", C.c_uint64), ("max_dict_bytes", C.c_uint64), ("max_device_bytes", C.c_uint64), ("mem_limit_mb", C.c_uint64), ("track_first_occurrence", C.c_uint64), ("defer_bad_row", C.c_uint64)]


class SplitHint(C.Structure):
    _fields_ = [("max_rows", C.c_uint32), ("partial_fields", C.c_uint32)]


class Pass1Result(C.Structure):
    _fields_ = [("num_rows", C.c_uint32), ("longest_line", C.c_uint32), ("is_last_block", C.c_uint32), ("bad_row", C.c_uint32),
                ("bytes_consumed", C.c_uint64), ("dict_entries", C.c_uint64), ("table_slots", C.c_uint64)]


class StageTimes(C.Structure):
    _fields_ = [("ms", C.c_float * NUM_STAGES), ("launches", C.c_uint32 * NUM_STAGES)]


class ModelState(C.Structure):
    _fields_ = [("arenas", C.c_uint64), ("entries", C.c_uint64), ("arena_free", C.c_uint64)]


class MultiBlock(C.Structure):
    _fields_ = [("num_rows", C.c_uint32), ("longest_line", C.c_uint32), ("is_last_block", C.c_uint32), ("bad_row", C.c_uint32), ("dict_bytes", C.c_uint64)]


WRITE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_uint64)
READ_FN = C.CFUNCTYPE(C.c_int64, C.c_void_p, C.POINTER(C.c_uint8), C.c_uint64)


class DecBlock(C.Structure):
    _fields_ = [("num_rows", C.c_uint32), ("longest_line", C.c_uint32), ("is_last_block", C.c_uint32), ("reserved", C.c_uint32),
                ("text_bytes", C.c_uint64), ("zdw_bytes", C.c_uint64), ("dict_bytes", C.c_uint64), ("walk_passes", C.c_uint64)]


class ZdwError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("zdw error %d: %s" % (code, text))
        self.code = code
        self.text = text


_lib = None


def model_upper_bound_mb(entries, nbytes, max_len):
    """the most the reference's process can have grown to with a dictionary of that many entries / bytes (the model)"""
    return float(lib().zdw_model_upper_bound_mb(entries, nbytes, max_len))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_build.LIB_HIP):
            raise RuntimeError("libzdw_hip.so is not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`"
                               % _build.LIB_HIP)
        L = C.CDLL(_build.LIB_HIP, mode=C.RTLD_GLOBAL)
        vp, u8p, u64p, u32p = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
        L.zdw_abi_version.restype = C.c_uint32
        L.zdw_ctx_create.argtypes = [u8p, C.c_uint32, C.POINTER(Opts), C.POINTER(vp)]
        L.zdw_ctx_destroy.argtypes = [vp]
        L.zdw_ctx_destroy.restype = None
        L.zdw_last_error.argtypes = [vp]
        L.zdw_last_error.restype = C.c_char_p
        L.zdw_ctx_enable_timing.argtypes = [vp, C.c_int]
        L.zdw_ctx_stage_times.argtypes = [vp, C.POINTER(StageTimes)]
        L.zdw_set_input_host.argtypes = [vp, vp, C.c_uint64]
        L.zdw_set_input_device.argtypes = [vp, vp, C.c_uint64]
        L.zdw_set_input_stream.argtypes = [vp, READ_FN, vp, C.c_uint64]
        L.zdw_set_input_host_async.argtypes = [vp, C.c_char_p, C.c_uint64]
        L.zdw_block_pass1.argtypes = [vp, C.POINTER(SplitHint), C.POINTER(Pass1Result)]
        L.zdw_ctx_pass1_launches.argtypes = [vp]
        L.zdw_ctx_pass1_launches.restype = C.c_uint64
        L.zdw_block_partial_fields.argtypes = [vp]
        L.zdw_block_partial_fields.restype = C.c_uint32
        L.zdw_block_dict.argtypes = [vp, C.POINTER(vp), u64p, u8p]
        L.zdw_block_dict_device.argtypes = [vp, C.POINTER(vp), u64p, u8p]
        L.zdw_block_stats.argtypes = [vp, u8p, u64p, u32p]
        L.zdw_block_pass2.argtypes = [vp, C.POINTER(vp), u64p]
        L.zdw_block_pass2_device.argtypes = [vp, C.POINTER(vp), u64p]
        L.zdw_block_next.argtypes = [vp]
        L.zdw_shard_export_ranges.argtypes = [vp, u64p, u64p, u8p]
        L.zdw_shard_import_ranges.argtypes = [vp, u64p, u64p, u8p]
        L.zdw_shard_export_uniques.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), u64p, u64p]
        L.zdw_shard_import_uniques.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint64]
        L.zdw_shard_sort_uniques.argtypes = [vp, u64p]
        L.zdw_shard_sample_keys.argtypes = [vp, C.c_uint32, u64p]
        L.zdw_shard_partition.argtypes = [vp, C.c_uint32, u64p, u64p, u64p, C.POINTER(vp), C.POINTER(vp)]
        L.zdw_owner_build.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint64, u64p]
        L.zdw_owner_offsets.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
        L.zdw_shard_set_offsets.argtypes = [vp, vp, C.c_uint64, C.c_uint64, C.c_int]
        L.zdw_owner_section.argtypes = [vp, C.c_uint64, C.c_int]
        L.zdw_model_upper_bound_mb.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
        L.zdw_model_upper_bound_mb.restype = C.c_double
        L.zdw_shard_model_counts.argtypes = [vp, u64p, u64p, u32p]
        L.zdw_shard_positions.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
        L.zdw_owner_import.argtypes = [vp, vp, vp, vp, C.c_uint64, C.c_uint64]
        L.zdw_owner_first_flags.argtypes = [vp, C.POINTER(vp)]
        L.zdw_owner_finish.argtypes = [vp, C.c_uint64, u64p]
        L.zdw_owner_size.argtypes = [vp, u64p]
        L.zdw_shard_rows_early.argtypes = [vp, C.c_uint64]
        L.zdw_shard_model_walk.argtypes = [vp, vp, C.c_uint64, C.c_double, C.POINTER(ModelState), u32p, u32p, u32p, u64p, u64p]
        L.zdw_shard_trim.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(Pass1Result)]
        L.zdw_multi_next.argtypes = [vp]
        L.zdw_shard_last_row.argtypes = [vp, u64p]
        L.zdw_shard_set_prev_row.argtypes = [vp, u64p]
        L.zdw_shard_line_caps.argtypes = [vp, u32p]
        L.zdw_multi_create.argtypes = [u8p, C.c_uint32, C.POINTER(Opts), C.POINTER(C.c_int32), C.c_uint32, C.POINTER(vp)]
        L.zdw_multi_destroy.argtypes = [vp]
        L.zdw_multi_destroy.restype = None
        L.zdw_multi_last_error.argtypes = [vp]
        L.zdw_multi_last_error.restype = C.c_char_p
        L.zdw_multi_set_input_host.argtypes = [vp, vp, C.c_uint64]
        L.zdw_multi_set_input_device.argtypes = [vp, C.POINTER(vp), u64p]
        L.zdw_multi_encode.argtypes = [vp, C.POINTER(MultiBlock)]
        L.zdw_multi_stats.argtypes = [vp, u8p, u64p, u32p]
        L.zdw_dec_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.zdw_dec_destroy.argtypes = [vp]
        L.zdw_dec_destroy.restype = None
        L.zdw_dec_last_error.argtypes = [vp]
        L.zdw_dec_last_error.restype = C.c_char_p
        L.zdw_dec_set_input_host.argtypes = [vp, C.c_char_p, C.c_uint64]
        L.zdw_dec_header.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.zdw_dec_column.argtypes = [vp, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint8), C.POINTER(C.c_uint16)]
        L.zdw_dec_next_block.argtypes = [vp, C.POINTER(DecBlock)]
        L.zdw_dec_text_device.argtypes = [vp, C.POINTER(vp), u64p]
        L.zdw_dec_text_write.argtypes = [vp, WRITE_FN, vp, u64p]
        L.zdw_dec_stage_ms.argtypes = [vp, C.POINTER(C.c_float * 5)]
        L.zdw_multi_dict_write.argtypes = [vp, WRITE_FN, vp, u64p]
        L.zdw_multi_rows_write.argtypes = [vp, WRITE_FN, vp, u64p]
        _lib = L
    return _lib


class MultiEncoder:
    """zdw_multi_*: the blocks of an input, each row-sharded over several devices inside the library (RCCL between distinct devices).
    mem_limit_mb: where the reference's --mem-limit ends the blocks (0 = the whole input is one block)."""

    def __init__(self, col_types, devices, trim=False, mem_limit_mb=0):
        self.L = lib()
        self.ncols = len(col_types)
        types = (C.c_uint8 * max(1, self.ncols))(*col_types)
        o = Opts()
        o.struct_size = C.sizeof(Opts)
        o.trim_trailing_spaces = 1 if trim else 0
        o.mem_limit_mb = mem_limit_mb
        devs = (C.c_int32 * len(devices))(*devices)
        h = C.c_void_p()
        rc = self.L.zdw_multi_create(types, self.ncols, C.byref(o), devs, len(devices), C.byref(h))
        if rc:
            raise ZdwError(rc, self.L.zdw_multi_last_error(None).decode())
        self.h = h
        self._keep = None

    def close(self):
        if getattr(self, "h", None):
            self.L.zdw_multi_destroy(self.h)
            self.h = None

    __del__ = close

    def error(self):
        return self.L.zdw_multi_last_error(self.h).decode()

    def set_input(self, data):
        """bytes / bytearray / C-contiguous uint8 numpy array (kept alive until the next input)"""
        if isinstance(data, (bytes, bytearray)):
            buf = (C.c_char * len(data)).from_buffer_copy(data) if len(data) else None
            self._keep = buf
            rc = self.L.zdw_multi_set_input_host(self.h, C.cast(buf, C.c_void_p) if buf is not None else None, len(data))
        else:
            self._keep = data
            rc = self.L.zdw_multi_set_input_host(self.h, C.c_void_p(data.ctypes.data), data.size)
        if rc:
            raise ZdwError(rc, self.error())

    def set_input_device(self, pieces, keepalive=None):
        """pieces: [(device pointer, bytes)] -- piece i lies on device i (whole rows, 16-byte aligned), in row order; nothing is copied"""
        n = len(pieces)
        ptrs = (C.c_void_p * n)(*[p for p, _ in pieces])
        lens = (C.c_uint64 * n)(*[l for _, l in pieces])
        self._keep = keepalive
        rc = self.L.zdw_multi_set_input_device(self.h, ptrs, lens)
        if rc:
            raise ZdwError(rc, self.error())

    def next_block(self, sink=None):
        """Encodes the block behind the last one.  Returns (rc, block bytes as the file holds them -- block header, dictionary, stats,
        rows -- or None, MultiBlock); with `sink` (a callable taking bytes) the pieces go there and the second value is their length."""
        import struct
        blk = MultiBlock()
        rc = self.L.zdw_multi_encode(self.h, C.byref(blk))
        if rc or blk.num_rows == 0:
            return rc, None, blk
        out = bytearray()
        total = [0]

        def put(b):
            total[0] += len(b)
            if sink is None:
                out.extend(b)
            else:
                sink(b)

        def cb_fn(_user, p, n):
            put(C.string_at(p, n))
            return 0
        put(struct.pack("<IIB", blk.num_rows, blk.longest_line, 1 if blk.is_last_block else 0))
        cb = WRITE_FN(cb_fn)
        rc = self.L.zdw_multi_dict_write(self.h, cb, None, None)
        cs = (C.c_uint8 * max(1, self.ncols))()
        cbs = (C.c_uint64 * max(1, self.ncols))()
        used = C.c_uint32()
        if rc == 0:
            rc = self.L.zdw_multi_stats(self.h, cs, cbs, C.byref(used))
        if rc == 0:
            put(bytes(cs)[:self.ncols])
            put(struct.pack("<%dQ" % used.value, *list(cbs)[:used.value]))
            rc = self.L.zdw_multi_rows_write(self.h, cb, None, None)
        if rc == 0:
            rc = self.L.zdw_multi_next(self.h)
        if rc:
            raise ZdwError(rc, self.error())
        return 0, (bytes(out) if sink is None else total[0]), blk

    def encode(self, data):
        """The input's first block (all of it without a memory limit): (rc, block bytes or None, MultiBlock)."""
        self.set_input(data)
        return self.next_block()

    def encode_all(self, data, sink=None):
        """All blocks of `data`: (rc, concatenated block bytes (or their length with `sink`), [MultiBlock])."""
        self.set_input(data)
        out, blocks, total = bytearray(), [], 0
        while True:
            rc, b, blk = self.next_block(sink)
            if rc:
                return rc, (bytes(out) if sink is None else total), blocks + [blk]
            if b is None:
                break
            if sink is None:
                out.extend(b)
            else:
                total += b
            blocks.append(blk)
            if blk.is_last_block:
                break
        return 0, (bytes(out) if sink is None else total), blocks


class Decoder:
    """zdw_dec_*: a raw ZDW stream back to its TSV rows, block by block, on the device (UnconvertFromZDW.cpp:758-1000, 1270-1462)."""

    def __init__(self, device=0):
        self.L = lib()
        h = C.c_void_p()
        rc = self.L.zdw_dec_create(device, C.byref(h))
        if rc:
            raise ZdwError(rc, self.L.zdw_dec_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.zdw_dec_destroy(self.h)
            self.h = None

    def error(self):
        return self.L.zdw_dec_last_error(self.h).decode()

    def set_input(self, data):
        """returns the reference decoder's ERR_CODE (0 = OK)"""
        self._keep = data
        return self.L.zdw_dec_set_input_host(self.h, data, len(data))

    def columns(self):
        v, n = C.c_uint32(), C.c_uint32()
        self.L.zdw_dec_header(self.h, C.byref(v), C.byref(n))
        cols = []
        for i in range(n.value):
            name, t, cs = C.c_char_p(), C.c_uint8(), C.c_uint16()
            self.L.zdw_dec_column(self.h, i, C.byref(name), C.byref(t), C.byref(cs))
            cols.append((name.value, t.value, cs.value))
        return v.value, cols

    def next_block(self):
        blk = DecBlock()
        rc = self.L.zdw_dec_next_block(self.h, C.byref(blk))
        return rc, blk

    def text_device(self):
        p, n = C.c_void_p(), C.c_uint64()
        rc = self.L.zdw_dec_text_device(self.h, C.byref(p), C.byref(n))
        if rc:
            raise ZdwError(rc, self.error())
        return p.value, n.value

    def text(self):
        parts = []

        def sink(_user, p, n):
            parts.append(C.string_at(p, n))
            return 0
        cb = WRITE_FN(sink)
        rc = self.L.zdw_dec_text_write(self.h, cb, None, None)
        if rc:
            raise ZdwError(rc, self.error())
        return b"".join(parts)

    def stage_ms(self):
        ms = (C.c_float * 5)()
        self.L.zdw_dec_stage_ms(self.h, C.byref(ms))
        return dict(zip(("row_chain", "expand", "repeat_fill", "text_lengths", "text"), [round(float(x), 3) for x in ms]))

    def decode_all(self, data):
        """(rc, text of all blocks, [DecBlock])"""
        rc = self.set_input(data)
        if rc:
            return rc, b"", []
        out, blocks = [], []
        while True:
            rc, blk = self.next_block()
            if rc:
                return rc, b"".join(out), blocks
            if blk.num_rows:
                out.append(self.text())
                blocks.append(blk)
            if blk.is_last_block:
                return 0, b"".join(out), blocks


class Block:
    """Host copies of one encoded block's sections (what the reference writes between two block headers)."""

    def __init__(self, num_rows, longest_line, is_last, dict_section, col_size, col_base, rows, p1):
        self.num_rows, self.longest_line, self.is_last = num_rows, longest_line, is_last
        self.dict_section, self.col_size, self.col_base, self.rows, self.p1 = dict_section, col_size, col_base, rows, p1

    def tobytes(self):
        import struct
        return (struct.pack("<IIB", self.num_rows, self.longest_line, 1 if self.is_last else 0) + self.dict_section
                + self.col_size + struct.pack("<%dQ" % len(self.col_base), *self.col_base) + self.rows)


class Encoder:
    """One zdw_ctx: the per-block seam of ConvertToZDW::processFile (cplusplus/ConvertToZDW.cpp:805-872)."""

    def __init__(self, col_types, trim=False, device=-1, dict_slots=0, first_row_capacity=0, max_dict_bytes=0, max_device_bytes=0, mem_limit_mb=0,
                 track_first_occurrence=False, defer_bad_row=False):
        self.L = lib()
        self.ncols = len(col_types)
        types = (C.c_uint8 * max(1, self.ncols))(*col_types)
        o = Opts()
        o.struct_size = C.sizeof(Opts)
        o.device = device
        o.trim_trailing_spaces = 1 if trim else 0
        o.first_row_capacity = first_row_capacity
        o.dict_slots = dict_slots
        o.max_dict_bytes = max_dict_bytes
        o.max_device_bytes = max_device_bytes
        o.mem_limit_mb = mem_limit_mb
        o.track_first_occurrence = 1 if track_first_occurrence else 0
        o.defer_bad_row = 1 if defer_bad_row else 0
        h = C.c_void_p()
        rc = self.L.zdw_ctx_create(types, self.ncols, C.byref(o), C.byref(h))
        if rc:
            raise ZdwError(rc, self.L.zdw_last_error(None).decode())
        self.h = h
        self._keep = None

    def close(self):
        if getattr(self, "h", None):
            self.L.zdw_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def _chk(self, rc):
        if rc:
            raise ZdwError(rc, self.L.zdw_last_error(self.h).decode())

    def enable_timing(self, on=True):
        self._chk(self.L.zdw_ctx_enable_timing(self.h, 1 if on else 0))

    def stage_times(self):
        t = StageTimes()
        self._chk(self.L.zdw_ctx_stage_times(self.h, C.byref(t)))
        return {STAGE_NAMES[i]: (t.ms[i], t.launches[i]) for i in range(len(STAGE_NAMES))}

    def set_input_host(self, data):
        """bytes / bytearray / C-contiguous uint8 numpy array"""
        if isinstance(data, (bytes, bytearray)):
            buf = (C.c_char * len(data)).from_buffer_copy(data) if len(data) else None
            self._keep = buf
            self._chk(self.L.zdw_set_input_host(self.h, C.cast(buf, C.c_void_p) if buf is not None else None, len(data)))
        else:
            self._keep = data
            self._chk(self.L.zdw_set_input_host(self.h, C.c_void_p(data.ctypes.data), data.size))

    def set_input_device(self, ptr, nbytes, keepalive=None):
        self._keep = keepalive
        self._chk(self.L.zdw_set_input_device(self.h, C.c_void_p(ptr), nbytes))

    def set_input_host_async(self, data):
        """bytes that stay where they are: the library copies them to the device piece by piece while blocks are encoded"""
        self._keep = data
        self._chk(self.L.zdw_set_input_host_async(self.h, data, len(data)))

    def set_input_stream(self, reader, size_hint=0):
        """reader(cap) -> bytes (at most cap of them; b"" = end of input); called from a thread of the library"""
        def cb(_user, buf, cap):
            try:
                data = reader(int(cap))
            except Exception:  # noqa: BLE001
                return -1
            if len(data) > cap:
                return -1                     # a reader that hands over more than it was asked for: a failed read, never an overflow
            if data:
                C.memmove(buf, data, len(data))
            return len(data)
        reader_cb = READ_FN(cb)
        self._keep = None
        # (the thunk of the input before stays alive until the library has joined the thread that may still be calling it)
        self._chk(self.L.zdw_set_input_stream(self.h, reader_cb, None, size_hint))
        self._reader = reader_cb

    def pass1(self, max_rows=None, partial_fields=0):
        res = Pass1Result()
        hint = None
        if max_rows is not None:
            hint = C.byref(SplitHint(max_rows, partial_fields))
        rc = self.L.zdw_block_pass1(self.h, hint, C.byref(res))
        return rc, res

    def pass1_launches(self):
        return int(self.L.zdw_ctx_pass1_launches(self.h))

    def partial_fields(self):
        return int(self.L.zdw_block_partial_fields(self.h))

    def dict_section(self):
        p, n, w = C.c_void_p(), C.c_uint64(), C.c_uint8()
        self._chk(self.L.zdw_block_dict(self.h, C.byref(p), C.byref(n), C.byref(w)))
        return C.string_at(p, n.value), w.value

    def dict_section_device(self):
        p, n, w = C.c_void_p(), C.c_uint64(), C.c_uint8()
        self._chk(self.L.zdw_block_dict_device(self.h, C.byref(p), C.byref(n), C.byref(w)))
        return p.value, n.value, w.value

    def stats(self):
        cs = (C.c_uint8 * max(1, self.ncols))()
        cb = (C.c_uint64 * max(1, self.ncols))()
        used = C.c_uint32()
        self._chk(self.L.zdw_block_stats(self.h, cs, cb, C.byref(used)))
        return bytes(cs)[:self.ncols], list(cb)[:used.value]

    def pass2(self):
        p, n = C.c_void_p(), C.c_uint64()
        self._chk(self.L.zdw_block_pass2(self.h, C.byref(p), C.byref(n)))
        return C.string_at(p, n.value) if n.value else b""

    def pass2_device(self):
        p, n = C.c_void_p(), C.c_uint64()
        self._chk(self.L.zdw_block_pass2_device(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def next_block(self):
        self._chk(self.L.zdw_block_next(self.h))

    def encode_block(self, max_rows=None, partial_fields=0):
        """pass 1 .. pass 2 for the next block; returns (rc, Block or None, Pass1Result)."""
        rc, p1 = self.pass1(max_rows, partial_fields)
        if rc or p1.num_rows == 0:
            return rc, None, p1
        sec, _ = self.dict_section()
        cs, cb = self.stats()
        rows = self.pass2()
        return 0, Block(p1.num_rows, p1.longest_line, bool(p1.is_last_block), sec, cs, cb, rows, p1), p1

    # ---- shard exchange (multi-GPU) ----
    def export_ranges(self):
        n = max(1, self.ncols)
        mn, mx, seen = (C.c_uint64 * n)(), (C.c_uint64 * n)(), (C.c_uint8 * n)()
        self._chk(self.L.zdw_shard_export_ranges(self.h, mn, mx, seen))
        import numpy as np
        k = self.ncols
        return (np.frombuffer(mn, dtype=np.uint64, count=k).tolist(), np.frombuffer(mx, dtype=np.uint64, count=k).tolist(),
                np.frombuffer(seen, dtype=np.uint8, count=k).tolist())

    def import_ranges(self, mn, mx, seen):
        n = max(1, self.ncols)
        self._chk(self.L.zdw_shard_import_ranges(self.h, (C.c_uint64 * n)(*mn), (C.c_uint64 * n)(*mx), (C.c_uint8 * n)(*seen)))

    def export_uniques(self):
        lens, byts, cnt, nb = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64()
        self._chk(self.L.zdw_shard_export_uniques(self.h, C.byref(lens), C.byref(byts), C.byref(cnt), C.byref(nb)))
        return lens.value, byts.value, cnt.value, nb.value

    def import_uniques(self, d_lens, d_bytes, count, nbytes):
        self._chk(self.L.zdw_shard_import_uniques(self.h, C.c_void_p(d_lens), C.c_void_p(d_bytes), count, nbytes))

    # ---- range-partitioned dictionary (see include/zdw_hip.h) ----
    def sort_uniques(self):
        n = C.c_uint64()
        self._chk(self.L.zdw_shard_sort_uniques(self.h, C.byref(n)))
        return n.value

    def sample_keys(self, nsamples):
        """[(k0, k1)] * nsamples: 16-byte prefix keys (two big-endian words) at evenly spaced sorted positions"""
        buf = (C.c_uint64 * max(1, 2 * nsamples))()
        self._chk(self.L.zdw_shard_sample_keys(self.h, nsamples, buf))
        return [(buf[2 * i], buf[2 * i + 1]) for i in range(nsamples)]

    def partition(self, splitters):
        """splitters: sorted [(k0, k1)] * (nparts - 1).  Returns (counts[nparts], nbytes[nparts], d_lens, d_bytes)."""
        nparts = len(splitters) + 1
        flat = (C.c_uint64 * max(1, 2 * len(splitters)))(*[w for k in splitters for w in k])
        counts, nb = (C.c_uint64 * nparts)(), (C.c_uint64 * nparts)()
        lens, byts = C.c_void_p(), C.c_void_p()
        self._chk(self.L.zdw_shard_partition(self.h, nparts, flat, counts, nb, C.byref(lens), C.byref(byts)))
        return list(counts), list(nb), lens.value, byts.value

    def owner_build(self, d_lens, d_bytes, count, nbytes):
        out = C.c_uint64()
        self._chk(self.L.zdw_owner_build(self.h, C.c_void_p(d_lens), C.c_void_p(d_bytes), count, nbytes, C.byref(out)))
        return out.value

    def owner_import(self, d_lens, d_bytes, d_pos, count, nbytes):
        self._chk(self.L.zdw_owner_import(self.h, C.c_void_p(d_lens), C.c_void_p(d_bytes), C.c_void_p(d_pos) if d_pos else None, count, nbytes))

    def owner_first_flags(self):
        p = C.c_void_p()
        self._chk(self.L.zdw_owner_first_flags(self.h, C.byref(p)))
        return p.value

    def owner_finish(self, last_pos=None):
        out = C.c_uint64()
        self._chk(self.L.zdw_owner_finish(self.h, (1 << 64) - 1 if last_pos is None else last_pos, C.byref(out)))
        return out.value

    def owner_size(self):
        """(owner, after owner_import) the bytes this range will take in the dictionary: known before the range is sorted"""
        out = C.c_uint64()
        self._chk(self.L.zdw_owner_size(self.h, C.byref(out)))
        return out.value

    def rows_early(self, dict_size):
        """(requester) the dictionary's size is known: the row lengths start on the context's second stream, beside the owners' sort"""
        self._chk(self.L.zdw_shard_rows_early(self.h, dict_size))

    def owner_section(self, dict_size, first_range):
        self._chk(self.L.zdw_owner_section(self.h, dict_size, 1 if first_range else 0))

    def model_counts(self):
        """(entries, bytes, longest string) of this shard's dictionary: see model_upper_bound_mb"""
        e, b, ml = C.c_uint64(), C.c_uint64(), C.c_uint32()
        self._chk(self.L.zdw_shard_model_counts(self.h, C.byref(e), C.byref(b), C.byref(ml)))
        return e.value, b.value, ml.value

    def positions(self, base):
        p = C.c_void_p()
        self._chk(self.L.zdw_shard_positions(self.h, base, C.byref(p)))
        return p.value

    def model_walk(self, d_flags, count, limit_mb, state):
        """state: (arenas, entries, arena_free) from the shard before.  Returns (state, None) or (state, (row, fields, row_start, trip_pos))."""
        st = ModelState(*state)
        found, row, fields = C.c_uint32(), C.c_uint32(), C.c_uint32()
        rs, tp = C.c_uint64(), C.c_uint64()
        self._chk(self.L.zdw_shard_model_walk(self.h, C.c_void_p(d_flags) if d_flags else None, count, float(limit_mb), C.byref(st), C.byref(found),
                                              C.byref(row), C.byref(fields), C.byref(rs), C.byref(tp)))
        new = (st.arenas, st.entries, st.arena_free)
        return new, ((row.value, fields.value, rs.value, tp.value) if found.value else None)

    def trim(self, row, fields, row_start):
        res = Pass1Result()
        self._chk(self.L.zdw_shard_trim(self.h, row, fields, row_start, C.byref(res)))
        return res

    def owner_offsets(self, base):
        p = C.c_void_p()
        self._chk(self.L.zdw_owner_offsets(self.h, base, C.byref(p)))
        return p.value

    def set_offsets(self, d_offsets, count, dict_size, first_range):
        self._chk(self.L.zdw_shard_set_offsets(self.h, C.c_void_p(d_offsets), count, dict_size, 1 if first_range else 0))

    def last_row(self):
        out = (C.c_uint64 * max(1, self.ncols))()
        self._chk(self.L.zdw_shard_last_row(self.h, out))
        import numpy as np
        return np.frombuffer(out, dtype=np.uint64, count=self.ncols).tolist()

    def line_caps(self):
        """[cap after this shard's rows when the line buffer held 16384 << k before them, k = 0..17]"""
        out = (C.c_uint32 * 18)()
        self._chk(self.L.zdw_shard_line_caps(self.h, out))
        return list(out)

    def set_prev_row(self, stored):
        if stored is None:
            self._chk(self.L.zdw_shard_set_prev_row(self.h, None))
        else:
            self._chk(self.L.zdw_shard_set_prev_row(self.h, (C.c_uint64 * max(1, self.ncols))(*stored)))
