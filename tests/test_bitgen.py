"""BitGen decoder vs. the reference's own vectors (internal/testutil/bitgen_test.go:14-237)."""
import pytest

from conftest import load_golden
from oracle.bitgen import decode_bitgen

VECS = load_golden("bitgen.json")["vectors"]


@pytest.mark.parametrize("i", range(len(VECS)))
def test_bitgen_vector(i):
    v = VECS[i]
    try:
        out = decode_bitgen(v["input"])
        ok = True
    except Exception:
        out, ok = b"", False
    assert ok == v["valid"]
    if ok:
        assert out.hex() == v["output"]


def test_bitgen_doc_example():
    # worked example in the format description, internal/testutil/bitgen.go:93-115
    text = """<<< # DEFLATE uses LE bit-packing order
        ( < 0 00 0*5 < H16:0004 H16:fffb X:deadcafe )*2
        ( < 1 10 < D5:1 D5:0 D4:15 < 000*3 001 000*13 001 000 > 0*256 1*2 > 0 > 1 0 )
    """
    want = ("000400fbffdeadcafe000400fbffdeadcafe0de0010400000000100000000000"
            "0000000000000000000000000000000000000000000000000000002c")
    assert decode_bitgen(text).hex() == want
