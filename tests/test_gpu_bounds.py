"""The device-resident entry points do not trust their chunk table (include/xflate_amd.h, xfg_decode_device):
out-of-range and oversize descriptors are refused per chunk, nothing is written for them, the good chunks of
the same launch decode as usual; and the calls leave the caller's current HIP device alone."""
import ctypes
import zlib

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from compress_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def _sync_chunk(raw):
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    return c.compress(raw) + c.flush(zlib.Z_SYNC_FLUSH)


def test_decode_device_refuses_bad_descriptors(ctx):
    import torch
    from compress_amd import capi
    dev = torch.device("cuda", 0)
    raws = [bytes([65 + i]) * 3000 + b"tail %d" % i for i in range(4)]
    comps = [_sync_chunk(r) for r in raws]
    comp = bytearray()
    offs = []
    for c in comps:
        offs.append(len(comp))
        comp += c
    comp_len = len(comp)
    slot = 4096
    raw_len = 6 * slot
    F = capi.CHUNK_XFLATE_DEFLATE
    good = lambda i, k: capi.Chunk(offs[i], len(comps[i]), k * slot, len(raws[i]), F, 0)
    descs = [
        good(0, 0),
        capi.Chunk(offs[1], len(comps[1]), raw_len - 100, len(raws[1]), F, 0),          # output range leaves the buffer
        capi.Chunk(comp_len - 10, len(comps[2]), 2 * slot, len(raws[2]), F, 0),         # input range leaves the buffer
        capi.Chunk(offs[3], capi.MAX_CHUNK_COMP + 1, 3 * slot, len(raws[3]), F, 0),     # compressed size over the limit
        capi.Chunk(offs[3], len(comps[3]), 3 * slot, capi.MAX_CHUNK_RAW + 1, F, 0),     # raw size over the limit
        capi.Chunk(1 << 62, len(comps[3]), 3 * slot, len(raws[3]), F, 0),               # offset overflow
        capi.Chunk(offs[3], len(comps[3]), (1 << 64) - 8, len(raws[3]), F, 0),          # offset + size wraps
        capi.Chunk(offs[3], len(comps[3]), 3 * slot, len(raws[3]), F, 7),               # reserved != 0
        good(1, 4),
        good(3, 5),
    ]
    n = len(descs)
    arr = (capi.Chunk * n)(*descs)
    guard = 256
    d_comp = torch.frombuffer(bytearray(comp) + bytearray(64), dtype=torch.uint8).to(dev)
    d_chunks = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    d_raw = torch.full((raw_len + guard,), 0xEE, dtype=torch.uint8, device=dev)
    d_res = torch.full((n * ctypes.sizeof(capi.Result),), 0xFF, dtype=torch.uint8, device=dev)
    ctx.decode_device(d_comp.data_ptr(), comp_len, d_chunks.data_ptr(), n, d_raw.data_ptr(), raw_len, d_res.data_ptr(), None)
    ctx.sync()
    res = (capi.Result * n).from_buffer_copy(d_res.cpu().numpy().tobytes())
    out = d_raw.cpu().numpy().tobytes()
    for i in (0, 8, 9):
        d = descs[i]
        assert (res[i].status, res[i].out_produced) == (capi.ST_OK, d.raw_size), i
    assert out[0:len(raws[0])] == raws[0]
    assert out[4 * slot:4 * slot + len(raws[1])] == raws[1]
    assert out[5 * slot:5 * slot + len(raws[3])] == raws[3]
    for i in range(1, 8):
        assert (res[i].status, res[i].in_consumed, res[i].out_produced) == (capi.ST_INTERNAL, 0, 0), i
    # nothing was written for the refused chunks, inside the buffer or behind it
    assert out[raw_len:] == b"\xEE" * guard
    for k in (1, 2, 3):
        assert out[k * slot:(k + 1) * slot] == b"\xEE" * slot, k
    assert out[raw_len - 100:raw_len] == b"\xEE" * 100


def test_crc32_and_deflate_device_refuse_bad_descriptors(ctx):
    import torch
    from compress_amd import capi
    dev = torch.device("cuda", 0)
    raw = bytes(range(256)) * 16
    raw_len = len(raw)
    d_raw = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
    descs = [capi.Chunk(0, 0, 0, raw_len, 0, 0), capi.Chunk(0, 0, raw_len - 10, 100, 0, 0), capi.Chunk(0, 0, 1 << 40, 16, 0, 0)]
    arr = (capi.Chunk * 3)(*descs)
    d_chunks = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    d_crc = torch.full((3,), 0x55555555, dtype=torch.int32, device=dev)
    ctx.crc32_device(d_raw.data_ptr(), raw_len, d_chunks.data_ptr(), 3, 0, d_crc.data_ptr(), None)
    ctx.sync()
    crc = [int(x) & 0xffffffff for x in d_crc.cpu().tolist()]
    assert crc == [zlib.crc32(raw), 0, 0]

    bound = capi.lib().xfg_deflate_bound(raw_len)
    comp_len = 2 * bound
    descs = [capi.Chunk(0, bound, 0, raw_len, 0, 0),
             capi.Chunk(bound, bound, 100, raw_len, 0, 0),           # input range leaves the buffer
             capi.Chunk(comp_len - 8, bound, 0, raw_len, 0, 0)]      # output slot leaves the buffer
    arr = (capi.Chunk * 3)(*descs)
    d_chunks = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    d_comp = torch.full((comp_len + 64,), 0xEE, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(3 * ctypes.sizeof(capi.Result), dtype=torch.uint8, device=dev)
    ctx.deflate_device(d_raw.data_ptr(), raw_len, d_chunks.data_ptr(), 3, d_comp.data_ptr(), comp_len, d_res.data_ptr(), None)
    ctx.sync()
    res = (capi.Result * 3).from_buffer_copy(d_res.cpu().numpy().tobytes())
    out = d_comp.cpu().numpy().tobytes()
    assert res[0].status == capi.ST_OK
    assert zlib.decompressobj(-15).decompress(out[:res[0].out_produced]) == raw
    assert [res[1].status, res[2].status] == [capi.ST_INTERNAL, capi.ST_INTERNAL]
    assert out[bound:] == b"\xEE" * (len(out) - bound)


def test_current_device_is_restored(ctx):
    """an entry point must not leave the calling thread on another device.  With one GPU on the box the
    observable is that the current device is unchanged across the calls."""
    import torch
    before = torch.cuda.current_device()
    hip = ctypes.CDLL("libamdhip64.so")
    dev = ctypes.c_int(-1)
    assert hip.hipGetDevice(ctypes.byref(dev)) == 0
    d0 = dev.value
    ctx.decode(b"\x03\x00", [(0, 2, 0, 16, 0)], 16)
    ctx.sync()
    assert hip.hipGetDevice(ctypes.byref(dev)) == 0 and dev.value == d0
    assert torch.cuda.current_device() == before


def test_host_meta_shortcut_equals_inflating(ctx):
    """XFG_CHUNK_HOST_META (index / footer records the host decoded while parsing the index) must report exactly what
    inflating those records on the device reports: every valid stream of the reference's xflate vectors and a
    synthetic stream with several indexes, decoded with the flag and with the flag stripped."""
    from conftest import load_golden
    from compress_amd import capi, synth
    streams = [bytes.fromhex(v["input"]) for v in load_golden("xflate_reader.json")["vectors"] if not v["errf"]]
    streams.append(synth.build_mix_stream(23, 5000, index_size=4)[0])
    seen = 0
    for stream in streams:
        err, chunks = capi.index_parse(stream)
        assert err == capi.ERR_NONE
        raw_len = sum(c.raw_size for c in chunks)
        with_flag = [(c.comp_off, c.comp_size, c.raw_off, c.raw_size, c.flags) for c in chunks]
        without = [(c.comp_off, c.comp_size, c.raw_off, c.raw_size, c.flags & ~capi.CHUNK_HOST_META) for c in chunks]
        seen += sum(1 for c in chunks if c.flags & capi.CHUNK_HOST_META)
        raw_a, res_a = ctx.decode(stream, with_flag, raw_len)
        raw_b, res_b = ctx.decode(stream, without, raw_len)
        assert [(r.status, r.in_consumed, r.out_produced) for r in res_a] == [(r.status, r.in_consumed, r.out_produced) for r in res_b]
        assert all(r.status == capi.ST_OK for r in res_a)
        assert raw_a.raw[:raw_len] == raw_b.raw[:raw_len]
    assert seen >= len(streams)                                  # every stream has a footer record at least
