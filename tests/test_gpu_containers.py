"""XFLATE inside gzip and zip containers (xflate/example_test.go:29-248),
and the CRC-32 of the decoded bytes computed on the GPU for the gzip trailer."""
import io
import os
import struct
import zipfile
import zlib
import gzip

import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

# the reference's documented output for both examples (example_test.go:127-128, 246-247)
TWAIN_MIDDLE = b"ver, white with foam, the driving spray of spume-flakes, the dim\noutlines of the"


@pytest.fixture(scope="module")
def ctx():
    import torch
    torch.cuda.init()                   # torch's HIP runtime first, the library's second (the order bench.py uses)
    from compress_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def _xflate(data, chunk=1 << 16, level=1):
    from compress_amd import xflate
    out = bytearray()
    xw = xflate.NewWriter(out, xflate.WriterConfig(Level=level, ChunkSize=chunk))
    xw.Write(data)
    xw.Close()
    return bytes(out)


def _gpu_crc32(ctx, stream, raw_len):
    """Decode every record of `stream` on the device, CRC each chunk there, fold on the host."""
    import torch
    from compress_amd import capi
    import ctypes
    err, chunks = capi.index_parse(stream)
    assert err == capi.ERR_NONE
    n = len(chunks)
    dev = torch.device("cuda", 0)
    d_comp = torch.frombuffer(bytearray(stream) + bytearray(64), dtype=torch.uint8).to(dev)
    arr = (capi.Chunk * n)(*chunks)
    d_chunks = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    d_raw = torch.zeros(raw_len + 64, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(n * ctypes.sizeof(capi.Result), dtype=torch.uint8, device=dev)
    d_crc = torch.zeros(n, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    ctx.decode_device(d_comp.data_ptr(), len(stream), d_chunks.data_ptr(), n, d_raw.data_ptr(), raw_len, d_res.data_ptr(), s)
    ctx.crc32_device(d_raw.data_ptr(), d_raw.numel(), d_chunks.data_ptr(), n, d_res.data_ptr(), d_crc.data_ptr(), s)
    torch.cuda.synchronize(dev)
    res = d_res.cpu().numpy().view("i4,u4,i8,i8")
    assert (res["f0"] == 0).all()
    crcs = d_crc.cpu().numpy().view("u4")
    total = 0
    for c, r, v in zip(chunks, res, crcs):
        total = capi.crc32_combine(total, int(v), int(r["f3"]))
    return total, bytes(d_raw[:raw_len].cpu().numpy())


def test_gzip_file(ctx):
    """Example_gzipFile (xflate/example_test.go:146-248): gzip header + XFLATE stream + CRC-32/ISIZE trailer.
    A standard gzip reader must accept it; the XFLATE reader seeks into the middle; the trailer's CRC is
    checked against the value computed on the GPU from the decoded chunks."""
    from compress_amd import xflate
    twain = open(os.path.join(GOLDEN, "twain.txt"), "rb").read()
    header = b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff"
    body = _xflate(twain)
    blob = header + body + struct.pack("<II", zlib.crc32(twain), len(twain) & 0xffffffff)
    assert gzip.decompress(blob) == twain                       # RFC 1952 compliant
    section = blob[10:len(blob) - 8]                            # io.NewSectionReader(rd, 10, size - 18)
    xr, err = xflate.NewReader(io.BytesIO(section), ctx=ctx)
    assert err is None
    pos = len(twain) // 2
    xr.Seek(pos, xflate.SeekStart)
    got = bytearray()
    while len(got) < 80:
        got += xr.Read(80 - len(got))
    xr.Close()
    assert bytes(got) == TWAIN_MIDDLE == twain[pos:pos + 80]
    crc, raw = _gpu_crc32(ctx, section, len(twain))
    want_crc, want_size = struct.unpack("<II", blob[-8:])
    assert raw == twain and (crc, len(raw)) == (want_crc, want_size)


def test_zip_file(ctx):
    """Example_zipFile (xflate/example_test.go:29-137): each member is an XFLATE stream stored as method 8
    (DEFLATE); a standard zip reader inflates it, the XFLATE reader seeks inside the member's data."""
    from compress_amd import xflate, synth
    files = {"twain.txt": open(os.path.join(GOLDEN, "twain.txt"), "rb").read(),
             "digits.txt": synth.gen_digits(100003, 1), "huffman.txt": synth.gen_huffman(262144, 2)}
    buf = io.BytesIO()
    central = []
    for name, body in files.items():
        comp = _xflate(body)
        off = buf.tell()
        hdr = struct.pack("<IHHHHHIIIHH", 0x04034b50, 20, 0, 8, 0, 0x21, zlib.crc32(body), len(comp), len(body), len(name), 0)
        buf.write(hdr + name.encode() + comp)
        central.append(struct.pack("<IHHHHHHIIIHHHHHII", 0x02014b50, 20, 20, 0, 8, 0, 0x21, zlib.crc32(body), len(comp),
                                   len(body), len(name), 0, 0, 0, 0, 0, off) + name.encode())
    cd_off = buf.tell()
    for c in central:
        buf.write(c)
    buf.write(struct.pack("<IHHHHIIH", 0x06054b50, 0, 0, len(central), len(central), buf.tell() - cd_off, cd_off, 0))
    blob = buf.getvalue()
    with zipfile.ZipFile(io.BytesIO(blob)) as zf:               # backwards compatible with a standard inflater
        for name, body in files.items():
            assert zf.read(name) == body
        infos = {i.filename: i for i in zf.infolist()}
    for name, body in files.items():
        info = infos[name]
        data_off = info.header_offset + 30 + len(name)          # f.DataOffset()
        section = blob[data_off:data_off + info.compress_size]
        xr, err = xflate.NewReader(io.BytesIO(section), ctx=ctx)
        assert err is None
        pos = info.file_size // 2
        xr.Seek(pos, xflate.SeekStart)
        got = bytearray()
        while len(got) < 80:
            got += xr.Read(80 - len(got))
        xr.Close()
        assert bytes(got) == body[pos:pos + 80]
        if name == "twain.txt":
            assert bytes(got) == TWAIN_MIDDLE
        crc, raw = _gpu_crc32(ctx, section, len(body))
        assert raw == body and crc == info.CRC


def test_crc32_ragged_chunks(ctx):
    """xfg_crc32_device on chunks of every awkward size and alignment against C zlib."""
    import ctypes
    import numpy as np
    import torch
    from compress_amd import capi
    rng = np.random.Generator(np.random.PCG64(9))
    sizes = [0, 1, 2, 3, 15, 16, 17, 31, 63, 64, 65, 1000, 1023, 1024, 1025, 4096, 65535, 65536, 65537,
             (1 << 20) + 3, 300001]
    data = rng.integers(0, 256, sum(sizes) + 3 * len(sizes) + 64, dtype=np.uint8)
    descs, off = [], 5
    for sz in sizes:
        descs.append(capi.Chunk(0, 0, off, sz, 0, 0))
        off += sz + 3                                           # ragged starts
    n = len(descs)
    dev = torch.device("cuda", 0)
    d_raw = torch.from_numpy(data).to(dev)
    d_chunks = torch.frombuffer(bytearray(bytes((capi.Chunk * n)(*descs))), dtype=torch.uint8).to(dev)
    d_crc = torch.zeros(n, dtype=torch.int32, device=dev)
    ctx.crc32_device(d_raw.data_ptr(), d_raw.numel(), d_chunks.data_ptr(), n, 0, d_crc.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    got = d_crc.cpu().numpy().view("u4")
    host = data.tobytes()
    for d, v in zip(descs, got):
        assert int(v) == zlib.crc32(host[d.raw_off:d.raw_off + d.raw_size]), d.raw_size
