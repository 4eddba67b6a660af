"""bindings/go/ cannot be compiled here (no Go toolchain).  What can be checked without one: every C function the
Go files call is declared in include/xflate_amd.h and exported by the built library, every C constant and struct
field they name exists in the header, and every //export callback has its extern declaration in the cgo preamble."""
import ctypes
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "xflate_amd.h")).read()
GO_FILES = sorted(glob.glob(os.path.join(ROOT, "bindings", "go", "**", "*.go"), recursive=True))


def test_files_present():
    assert len(GO_FILES) >= 4


def test_functions_declared_and_exported():
    lib = ctypes.CDLL(os.path.join(ROOT, "compress_amd", "lib", "libxflate_amd.so"))
    called = set()
    for f in GO_FILES:
        called |= set(re.findall(r"\bC\.(xfg_[a-z0-9_]+)\(", open(f).read()))
    called = {c for c in called if not c.endswith("_fn")}      # C.xfg_read_fn(...) is a conversion to the callback type
    assert len(called) >= 15
    for name in sorted(called):
        assert re.search(r"\b%s\s*\(" % name, HEADER), name + " is not declared in the header"
        assert hasattr(lib, name), name + " is not exported"


def test_constants_types_and_fields():
    consts, types, fields = set(), set(), set()
    for f in GO_FILES:
        s = open(f).read()
        consts |= set(re.findall(r"\bC\.(XFG_[A-Z0-9_]+)\b", s))
        types |= set(re.findall(r"\bC\.(xfg_[a-z0-9_]+)\b(?!\()", s))
        fields |= set(re.findall(r"\.(comp_off|comp_size|raw_off|raw_size|flags|status|in_consumed|out_produced)\b", s))
    for c in sorted(consts):
        assert re.search(r"#define\s+%s\b" % c, HEADER), c
    for t in sorted(types):
        assert re.search(r"\b%s\b" % t, HEADER), t
    for fld in sorted(fields):
        assert re.search(r"\b%s\s*;" % fld, HEADER), fld


def test_exported_callbacks_are_declared():
    for f in GO_FILES:
        s = open(f).read()
        for name in re.findall(r"^//export (\w+)", s, re.M):
            assert re.search(r"extern [^;]*\b%s\(" % name, s), name
            assert re.search(r"^func %s\(" % name, s, re.M), name
