#!/usr/bin/env python3
"""Rate of xfg_crc32_device (HBM-read-bound) on the decoded output of a bench workload."""
import ctypes
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from compress_amd import capi  # noqa: E402

for w in sys.argv[1:] or ["c2"]:
    torch.cuda.init()
    shard = bench.build_shard(w, 0)
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0)
    n = shard["n"]
    d_comp = torch.frombuffer(bytearray(shard["comp"]) + bytearray(64), dtype=torch.uint8).to(dev)
    d_chunks = torch.frombuffer(bytearray(bytes(shard["descs"])), dtype=torch.uint8).to(dev)
    d_raw = torch.zeros(shard["raw_len"] + 64, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(n * ctypes.sizeof(capi.Result), dtype=torch.uint8, device=dev)
    d_crc = torch.zeros(n, dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream(dev)
    s = stream.cuda_stream
    ctx.decode_device(d_comp.data_ptr(), len(shard["comp"]), d_chunks.data_ptr(), n, d_raw.data_ptr(), shard["raw_len"], d_res.data_ptr(), s)
    for _ in range(2):
        ctx.crc32_device(d_raw.data_ptr(), d_raw.numel(), d_chunks.data_ptr(), n, d_res.data_ptr(), d_crc.data_ptr(), s)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(5):
        ctx.crc32_device(d_raw.data_ptr(), d_raw.numel(), d_chunks.data_ptr(), n, d_res.data_ptr(), d_crc.data_ptr(), s)
    b.record(stream)
    torch.cuda.synchronize(dev)
    ms = a.elapsed_time(b) / 5
    got = d_crc.cpu().numpy().view("u4")
    nd = shard["nd"]
    ok = all(int(got[i]) == zlib.crc32(shard["made"][(shard["first"] + i) % nd][0]) for i in range(0, n, 37))
    print("%s: crc32 of %.2f GiB in %.3f ms = %.1f GB/s read (%.1f %% of 8 TB/s), values %s" %
          (w, shard["raw_len"] / 2**30, ms, shard["raw_len"] / ms / 1e6, shard["raw_len"] / ms / 1e6 / 80, "ok" if ok else "WRONG"))
    ctx.close()
