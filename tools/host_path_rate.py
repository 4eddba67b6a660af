#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point xfg_decode (never bench.py's `value`):
compressed bytes go host -> HBM, the kernel runs, raw bytes and results come back."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from compress_amd import capi  # noqa: E402

NCHUNKS = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
bench.WORKLOADS["host"] = (NCHUNKS, 65536, ("huffman", "repeats", "digits"), 6, 768)
shard = bench.build_shard("host", 0)
ctx = capi.Context(0)
n = shard["n"]
raw = ctypes.create_string_buffer(shard["raw_len"])
res = (capi.Result * n)()
comp = shard["comp"]
L = capi.lib()
for it in range(8):
    t0 = time.perf_counter()
    rc = L.xfg_decode(ctx._h, comp, len(comp), shard["descs"], n, raw, shard["raw_len"], res)
    dt = time.perf_counter() - t0
    assert rc == 0 and all(res[i].status == 0 for i in range(0, n, 97))
    print("xfg_decode host path: %.1f MiB raw in %.2f ms = %.2f GB/s decompressed" % (shard["raw_len"] / 2**20, dt * 1e3, shard["raw_len"] / dt / 1e9))
ctx.close()

# ---- the same chunks as one XFLATE stream, read sequentially through xfg_reader (xflate.Reader.Read): the source
# callback copies from a resident buffer, reads are 64 MiB
import io
from compress_amd import synth, xflate  # noqa: E402

stream, _ = synth.build_mix_stream(NCHUNKS, 65536, n_distinct=768)
total = NCHUNKS * 65536
rd, e = xflate.NewReader(io.BytesIO(stream))
assert e is None, e
buf = ctypes.create_string_buffer(64 << 20)
for it in range(5):
    if it:
        rd.Reset(io.BytesIO(stream))                # the Reader keeps its window buffers
    err = ctypes.c_int()
    got = 0
    t0 = time.perf_counter()
    while True:
        k = L.xfg_reader_read(rd._h, buf, len(buf), ctypes.byref(err))
        if k <= 0:
            break
        got += k
    dt = time.perf_counter() - t0
    assert got == total and err.value == capi.ERR_EOF, (got, err.value)
    print("xfg_reader sequential Read: %.1f MiB in %.2f ms = %.2f GB/s (%d bytes read from the source)" % (got / 2**20, dt * 1e3, got / dt / 1e9, rd.bytes_read()))
rd.Close()

ctx2 = capi.Context(0)
# ---- the same stream read where it lies (xfg_reader_new_mem: no callbacks), 64 MiB reads
h = ctypes.c_void_p()
assert L.xfg_reader_new_mem(ctx2._h, stream, len(stream), ctypes.byref(h)) == 0
for it in range(5):
    if it:
        assert L.xfg_reader_reset_mem(h, stream, len(stream)) == 0
    got = 0
    t0 = time.perf_counter()
    while True:
        k = L.xfg_reader_read(h, buf, len(buf), ctypes.byref(err))
        if k <= 0:
            break
        got += k
    dt = time.perf_counter() - t0
    assert got == total and err.value == capi.ERR_EOF, (got, err.value)
    print("xfg_reader over memory, sequential Read, 64 MiB requests: %.1f MiB in %.2f ms = %.2f GB/s" % (got / 2**20, dt * 1e3, got / dt / 1e9))
big = ctypes.create_string_buffer(total)
for it in range(5):
    assert L.xfg_reader_reset_mem(h, stream, len(stream)) == 0
    got = 0
    t0 = time.perf_counter()
    while True:
        k = L.xfg_reader_read(h, big, total, ctypes.byref(err))
        if k <= 0:
            break
        got += k
    dt = time.perf_counter() - t0
    assert got == total and err.value == capi.ERR_EOF, (got, err.value)
    print("xfg_reader over memory, one request for everything: %.1f MiB in %.2f ms = %.2f GB/s" % (got / 2**20, dt * 1e3, got / dt / 1e9))
L.xfg_reader_free(h)
ctx2.close()
