"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol
include/zdw_hip.h declares, and fails loudly (no CPU fallback) when no device is present."""
import ctypes
import os
import re
import subprocess

import pytest

import __graft_entry__ as entry
from zdw_amd import build, capi


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(build.LIB_HIP) or not os.path.exists(build.LIB_CLI):
        build.build()


def test_every_declared_symbol_is_exported():
    lib = capi.lib()
    names = entry.declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert lib.zdw_abi_version() == 1


def test_header_compiles_as_c():
    hdr = os.path.join(entry.ROOT, "include", "zdw_hip.h")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", hdr], check=True)


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.ZdwError) as e:
        capi.Encoder([0, 9])
    assert e.value.code == capi.UNKNOWN_ERROR and "device" in e.value.text


def test_product_never_touches_the_oracle():
    """Nothing under zdw_amd/ may import, link or load anything under oracle/."""
    bad = []
    for root, _, files in os.walk(os.path.join(entry.ROOT, "zdw_amd")):
        if os.sep + "lib" in root:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", "Makefile")):
                txt = open(os.path.join(root, f), errors="replace").read()
                if re.search(r"\boracle\b", txt) and f not in ():
                    bad.append(os.path.join(root, f))
    assert not bad, bad


def test_cli_argument_errors_match_reference(capfd):
    """Exit codes of argument errors (cplusplus/convertDWfile.cpp:98-205) need no GPU."""
    from zdw_amd import cli
    assert cli.main_inprocess([]) == 1                         # NO_ARGS
    assert cli.main_inprocess(["-q"]) == 18                    # NO_INPUT_FILES
    assert cli.main_inprocess(["-d"]) == 11                    # MISSING_ARGUMENT
    assert cli.main_inprocess(["--bogus", "x.sql"]) == 16      # BAD_PARAMETER
    assert cli.main_inprocess(["-x", "x.sql"]) == 16
    assert cli.main_inprocess(["--metadata:novalue", "x.sql"]) == 16
    assert cli.main_inprocess(["--mem-limit=0", "x.sql"]) == 16
    assert cli.main_inprocess(["--version"]) == 0
    out = capfd.readouterr()
    assert "ConvertToZDW, Version 11b" in out.out
    assert cli.main_inprocess(["/nonexistent/file.txt"]) == 2  # MISSING_SQL_FILE collapses to CONVERSION_FAILED
    err = capfd.readouterr().err
    assert "code=5 (MISSING_SQL_FILE)" in err
    assert cli.main_inprocess(["/nonexistent/file.sql"]) == 2
    assert "code=4 (MISSING_DESC_FILE)" in capfd.readouterr().err


def test_the_row_form_of_pass_1_keeps_its_loads_in_flight_to_itself():
    """csrc/k_rows.hip keeps loads in flight across unrolled steps in accumulation registers a[0:47] that only its own asm statements
    name (global_load ... a[..] to fill them, v_accvgpr_read to take them); the register allocator must use none of them for anything
    else, and the kernel must not spill (scratch) -- checked on the ISA hipcc writes for gfx950, which is what ships."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zdw_amd", "csrc", "k_rows.hip")
    asm = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", "-", src],
                         check=True, capture_output=True, text=True).stdout
    kernels = re.findall(r"^(_ZN3zdw12k_pass1_rows\w+):[^\n]*\n(.*?)s_endpgm", asm, flags=re.S | re.M)
    assert len(kernels) == 2                      # <TRACK> x 2 (the look-up-only variants exist in the experiments build alone)
    for name, body in kernels:
        assert "scratch_" not in body, name
        stray = []
        for line in body.split("\n"):
            t = line.strip()
            if not t or t.startswith(";") or t.startswith("."):
                continue
            for m in re.finditer(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]", t):
                lo = int(m.group(1) if m.group(1) else m.group(2))
                if lo < 48 and not (t.startswith("global_load_dwordx4 a[") or t.startswith("v_accvgpr_read_b32")):
                    stray.append(t)
        assert not stray, (name, stray[:5])
