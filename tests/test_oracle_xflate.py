"""Pin the Python xflate/meta oracle to the reference's vectors.

  meta/reader_test.go:433-452   TestReader loop      -> test_meta_vector
  meta/reader_test.go:584-609   TestReaderReset loop -> test_meta_streams
  xflate/reader_test.go:277-301 TestReader loop      -> test_xflate_vector
  xflate/reader_test.go:304-365 TestReaderReset      -> test_xflate_reset
"""
import pytest

from conftest import load_golden
from oracle import xflate_oracle as xo

META = load_golden("meta_reader.json")
XFL = load_golden("xflate_reader.json")


def _errf_ok(errf, err):
    if errf == "":
        return err is None
    if errf == "IsEOF":
        return isinstance(err, xo.EOF)
    if errf == "IsUnexpectedEOF":
        return isinstance(err, xo.UnexpectedEOF)
    if errf == "IsCorrupted":
        return xo.is_corrupted(err)
    raise AssertionError(errf)


@pytest.mark.parametrize("i", range(len(META["vectors"])))
def test_meta_vector(i):
    v = META["vectors"][i]
    inp = bytes.fromhex(v["input"])
    mr = xo.MetaReader(inp)
    err = mr.decode_block()
    assert mr.buf.hex() == v["output"], v["desc"]
    if err is None:
        assert mr.input_offset == len(inp)
    assert mr.final == v["final"]
    assert _errf_ok(v["errf"], err), (v["desc"], err)


def test_meta_streams():
    data = bytes.fromhex(META["streams_input"])
    pos = 0
    for v in META["streams"]:
        mr = xo.MetaReader(data, pos)
        buf, err = mr.read_all()
        assert err is None
        assert buf.hex() == v["data"]
        assert mr.close() is None
        assert mr.input_offset == v["inOff"]
        assert mr.output_offset == v["outOff"]
        assert mr.num_blocks == v["numBlks"]
        assert mr.final_mode == v["final"]
        pos += mr.input_offset


def test_meta_reset_misc():
    # meta/reader_test.go:470-477
    mr = xo.MetaReader(b"corrupt")
    _, err = mr.read_all()
    assert xo.is_corrupted(err)
    assert xo.is_corrupted(mr.close())


def test_reverse_search():
    assert xo.reverse_search(b"") == -1
    footer = bytes.fromhex("0d008705000048c82a51e8ff37dbf1")
    assert xo.reverse_search(footer) == 0
    assert xo.reverse_search(b"garbage" + footer) == 7


@pytest.mark.parametrize("i", range(len(XFL["vectors"])))
def test_xflate_vector(i):
    v = XFL["vectors"][i]
    xr = xo.XflateReader(bytes.fromhex(v["input"]))
    err = xr.err
    buf = b""
    if err is None:
        buf, err = xr.read_all()
    if v["errf"]:
        assert xo.is_corrupted(err), (v["desc"], err)
    else:
        assert err is None, (v["desc"], err)
        assert buf.hex() == v["output"], v["desc"]


def test_xflate_reset():
    r = {k: bytes.fromhex(h) for k, h in XFL["reset"].items()}
    xr = xo.XflateReader(r["empty"])
    assert xr.err is None
    buf, err = xr.read_all()
    assert err is None and buf == b""
    assert xr.close() is None
    assert xr.close() is None
    _, err = xr.read_all()
    assert err is xo.ERR_CLOSED
    # Reset on garbage data: sticky Corrupted for Seek and Close
    err = xr.reset(b"garbage" + r["empty"])
    assert xo.is_corrupted(err)
    assert xo.is_corrupted(xr.seek(0, xo.SEEK_START)[1])
    assert xo.is_corrupted(xr.close())
    # corruption found in the discard phase
    for key in ("badData", "badSize"):
        assert xr.reset(r[key]) is None
        assert xr.seek(-1, xo.SEEK_END)[1] is None
        _, err = xr.read_all()
        assert xo.is_corrupted(err), key
