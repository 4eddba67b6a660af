#!/usr/bin/env python3
"""Throughput of the GPU compressor (xfg_deflate_device) on the bench workloads' raw data, device-resident:
raw GB/s in, compression ratio, and a GPU decode of the result compared with the input.
usage: tools/deflate_rate.py [--dynamic] [workload ...]   (--dynamic: XFG_DEFLATE_DYNAMIC, codes built per chunk)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from compress_amd import capi  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0)
    L = capi.lib()
    args = [a for a in sys.argv[1:] if a != "--dynamic"]
    flag = capi.DEFLATE_DYNAMIC if "--dynamic" in sys.argv[1:] else 0
    for w in args or ["c2"]:
        shard = bench.build_shard(w, 0)
        n, cs, nd = shard["n"], shard["chunk_size"], shard["nd"]
        raw = b"".join(shard["made"][i % nd][0] for i in range(n))
        bound = L.xfg_deflate_bound(cs)
        descs = (capi.Chunk * n)()
        for i in range(n):
            descs[i] = capi.Chunk(i * bound, bound, i * cs, cs, flag, 0)
        d_raw = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        d_chunks = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
        d_comp = torch.zeros(n * bound + 64, dtype=torch.uint8, device=dev)
        d_res = torch.zeros(n * ctypes.sizeof(capi.Result), dtype=torch.uint8, device=dev)
        stream = torch.cuda.Stream(dev)

        def step():
            ctx.deflate_device(d_raw.data_ptr(), len(raw), d_chunks.data_ptr(), n, d_comp.data_ptr(), n * bound,
                               d_res.data_ptr(), stream.cuda_stream)
        step()
        torch.cuda.synchronize(dev)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for a, b in evs:
            a.record(stream); step(); b.record(stream)
        torch.cuda.synchronize(dev)
        ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
        res = d_res.cpu().numpy().view("i4,u4,i8,i8")
        assert (res["f0"] == 0).all()
        csum = int(res["f3"].sum())
        # decode what was written and compare with the input
        dd = (capi.Chunk * n)()
        for i in range(n):
            dd[i] = capi.Chunk(i * bound, int(res["f3"][i]), i * cs, cs, capi.CHUNK_XFLATE_DEFLATE, 0)
        d_dd = torch.frombuffer(bytearray(bytes(dd)), dtype=torch.uint8).to(dev)
        d_back = torch.zeros(len(raw) + 64, dtype=torch.uint8, device=dev)
        d_res2 = torch.zeros_like(d_res)
        torch.cuda.synchronize(dev)                             # the fills above run on torch's own stream
        ctx.decode_device(d_comp.data_ptr(), n * bound, d_dd.data_ptr(), n, d_back.data_ptr(), len(raw), d_res2.data_ptr(),
                          stream.cuda_stream)
        torch.cuda.synchronize(dev)
        ok = bool((d_res2.cpu().numpy().view("i4,u4,i8,i8")["f0"] == 0).all()) and bool(torch.equal(d_back[:len(raw)], d_raw))
        if not ok:                                              # which side: let zlib decode the chunks that differ
            import zlib
            st2 = d_res2.cpu().numpy().view("i4,u4,i8,i8")
            neq = (d_back[:len(raw)].view(n, cs) != d_raw.view(n, cs)).any(dim=1).cpu().numpy()
            comp_h = d_comp.cpu().numpy()
            shown = 0
            for i in range(n):
                if neq[i] or st2["f0"][i] != 0:
                    c = bytes(comp_h[i * bound:i * bound + int(res["f3"][i])])
                    try:
                        z = zlib.decompressobj(-15).decompress(c)
                        zs = "zlib ok" if z == raw[i * cs:(i + 1) * cs] else "zlib differs (%d bytes)" % len(z)
                    except zlib.error as e:
                        zs = "zlib error %s" % e
                    print("  chunk %d (distinct %d): decode status %d produced %d; %s; comp %d bytes"
                          % (i, i % nd, st2["f0"][i], st2["f3"][i], zs, len(c)), flush=True)
                    shown += 1
                    if shown >= 6:
                        break
            print("  chunks that differ: %d" % int(neq.sum()))
        print("%s%s: %d chunks x %d KiB  compress %.2f ms = %.1f GB/s raw in, ratio %.3f (zlib level 6: %.3f), round trip %s"
              % (w, " (dynamic codes)" if flag else "", n, cs >> 10, ms, len(raw) / ms / 1e6, len(raw) / csum, len(raw) / len(shard["comp"]),
                 "bit-exact" if ok else "MISMATCH"), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
