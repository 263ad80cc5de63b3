"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every
symbol include/arrowhip.h declares; the oracle is never reachable from the product."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from arrow_go_amd import _native as N
    declared = N.declared_symbols()
    assert len(declared) >= 40
    lib = ctypes.CDLL(N.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.ah_version.restype = ctypes.c_char_p
    assert b"arrowhip" in lib.ah_version()


def test_exported_symbols_are_exactly_the_header():
    from arrow_go_amd import _native as N
    out = subprocess.check_output(["nm", "-D", "--defined-only", N.LIB_PATH], text=True)
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l)
    assert exported == N.declared_symbols()


def _c_prototypes(header):
    """function names a C header declares (comments stripped so that prose such as "foo(bar)" does not count)"""
    text = re.sub(r"/\*.*?\*/", "", open(header).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(ahc?_[a-z0-9_]+)\s*\(", text)))


def test_compute_library_exports_exactly_its_header():
    """libarrowhip_compute.so (the array-level C API: sessions, datums, ahc_call, IPC) == include/arrowhip_compute.h"""
    lib_path = os.path.join(ROOT, "arrow_go_amd", "libarrowhip_compute.so")
    header = os.path.join(ROOT, "include", "arrowhip_compute.h")
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib_path], text=True)
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l and not l.split()[-1].startswith("_Z"))
    declared = _c_prototypes(header)
    assert len(declared) >= 30
    assert exported == declared
    # ... and no C++ symbol leaks out of the library (everything else is hidden)
    assert not [l for l in out.splitlines() if " T _Z" in l]


def test_headers_are_plain_c():
    """both boundary headers compile as C11 with nothing but the C library (what cgo does with them)"""
    for h in ("arrowhip.h", "arrowhip_compute.h"):
        src = '#include "%s"\nint main(void) { return 0; }\n' % h
        subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-x", "c", "-"],
                       input=src, text=True, check=True)


def test_product_never_references_the_oracle():
    """A product path that routes through the oracle voids every parity claim."""
    pkg = os.path.join(ROOT, "arrow_go_amd")
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc", ".cpp", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"liboracle|oracle_lib|orc_[a-z]", text), os.path.join(dirpath, f)
    out = subprocess.check_output(["ldd", os.path.join(pkg, "libarrowhip.so")], text=True)
    assert "oracle" not in out


def test_no_gpu_gives_a_loud_error_not_a_fallback():
    import arrow_go_amd as ah
    if ah.device_count() > 0:
        return  # on the GPU box this test has nothing to say
    try:
        ah.Context(0)
    except ah.ErrHip:
        return
    raise AssertionError("Context(0) must fail without a GPU")
