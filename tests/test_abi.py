"""CPU-only: the C-ABI library loads, exports exactly what include/adlhip.h declares, and fails loudly
(no CPU fallback) when no GPU can be opened."""
import ctypes
import os
import re
import subprocess

import pytest

from oclradixsort_amd import _lib

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
HEADER = os.path.join(ROOT, "include", "adlhip.h")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(adlhip_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    decl = declared_functions()
    assert len(decl) >= 30
    assert sorted(_lib.SIGNATURES) == decl


def test_every_declared_symbol_is_exported(lib):
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (adlhip_[a-z0-9_]+)", out))
    for name in declared_functions():
        assert name in exported, name
        getattr(lib, name)


def test_library_contains_gfx950_code_object():
    data = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in data
    # built for gfx950 only: no other offload arch is bundled
    archs = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", data))
    assert archs == {b"gfx950"}, archs


def test_no_float_expanded_integer_division_in_the_device_code(tmp_path):
    """The compiler expands `a / b` and `a % b` of integers it knows to fit 24 bits through float (v_rcp_iflag_f32, v_trunc_f32, one
    upward correction).  For some operands the quotient comes out one too high and the remainder wraps -- that sent the key
    probe's loads 64 MiB past its array (tools/stress.py).  The 32-bit expansion (v_mul_hi_u32 corrections)
    is exact and is allowed; the 24-bit one is recognisable by its v_trunc_f32, which nothing else in this library uses."""
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    data = open(_lib.LIB_PATH, "rb").read()
    at = [m.start() for m in re.finditer(b"\x7fELF", data) if int.from_bytes(data[m.start() + 18:m.start() + 20], "little") == 224]
    assert at, "no AMDGPU code object inside the library"
    co = tmp_path / "device.co"
    co.write_bytes(data[at[0]:])
    out = subprocess.run([objdump, "-d", "--mcpu=gfx950", str(co)], capture_output=True, text=True).stdout
    assert out.count("s_endpgm") >= 50, "disassembly failed"
    kernel, hits = None, []
    for line in out.splitlines():
        if line.endswith(">:"):
            kernel = line.split("<", 1)[1][:-2]
        elif "v_trunc_f32" in line:
            hits.append(kernel)
    assert not hits, sorted(set(hits))


def test_header_compiles_as_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "adlhip.h"\nint main(void){ adlhip_info i; (void)i; return ADLHIP_SUCCESS; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                        "-c", str(src), "-o", str(tmp_path / "t.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_no_gpu_means_loud_failure_not_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert lib.adlhip_device_count() == 0
    h = ctypes.c_void_p()
    rc = lib.adlhip_device_create(0, ctypes.byref(h))
    assert rc != 0 and not h.value
    assert b"no HIP device" in lib.adlhip_last_error()
    from oclradixsort_amd import AdlHipError, DeviceUtils
    with pytest.raises(AdlHipError):
        DeviceUtils.allocate()


def test_null_handle_is_rejected(lib):
    assert lib.adlhip_sync(None) != 0
    assert lib.adlhip_radix_sort_u32(None, None, None, None, 0, 0, 32) != 0
    sz = ctypes.c_size_t()
    assert lib.adlhip_radix_sort_scratch_bytes(None, 0, 1024, ctypes.byref(sz), ctypes.byref(sz)) != 0
    assert lib.adlhip_version().startswith(b"adlhip")


def test_product_never_touches_the_oracle():
    """The product package, the C ABI, the facade and the demo harness must not reference oracle/ at all."""
    bad = []
    for base in ("oclradixsort_amd", "include", "src", os.path.join("tests", "demo")):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".inl", ".c")) or f == "Makefile":
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"import oracle|from oracle|liboracle|oracle/|libref|_ref/", text):
                        # the demo harness may SAY it does not depend on oracle/; nothing else may mention it
                        if f == "demo_main.cpp" and "depends on nothing under oracle/" in text:
                            continue
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad
