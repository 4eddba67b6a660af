#!/usr/bin/env python3
"""bench.py -- XFLATE decode throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (xfg_decode_device: the inflate kernels) over this
rank's shard of a synthetic XFLATE stream that is already resident in HBM.  The
workload is BASELINE.json configs[1]: 1 GiB of raw data per GPU in 64 KiB chunks,
round-robin huffman / repeats / digits generators, zlib level 6, one sync flush per
chunk, 4096 records per index.  With N ranks the stream is N GiB and rank k owns the
index records [k*16384, (k+1)*16384) (index-range sharding; no data-path collective),
so per-GPU work is fixed: "scaling": "weak".

Rank 0 prints ONE JSON line with the metric plus
  roofline     achieved = (sum csize + sum rsize) per launch / mean kernel time (HIP events
               on the launch stream) against the 8 TB/s HBM peak
  cpu_baseline the CPU oracle (oracle/inflate_oracle.c, a port of the reference's flate
               reader) decoding the same chunks on this box's host cores, threads stated.
"""
import argparse
import ctypes
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def measured_traffic(workload):
    """HBM bytes per launch from the committed rocprofv3 counter passes (profiles/r1/traffic.json), or None.
    FETCH_SIZE is doubled (gfx950 tallies 128-byte read requests at 64 bytes), WRITE_SIZE is exact; both in KB."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r1", "traffic.json")))[workload]
        return int((2 * t["FETCH_SIZE_KB"] + t["WRITE_SIZE_KB"]) * 1024)
    except (OSError, KeyError, ValueError):
        return None

WORKLOADS = {
    # name: (chunks per GPU, chunk size, kinds, zlib level, distinct chunks generated)
    "c2": (16384, 65536, ("huffman", "repeats", "digits"), 6, 768),
    "c3": (16384, 262144, ("huffman", "repeats", "digits"), 6, 384),     # 4 GiB per GPU, 256 KiB chunks
    "small": (768, 65536, ("huffman", "repeats", "digits"), 6, 96),
    "c2x4": (65536, 65536, ("huffman", "repeats", "digits"), 6, 768),    # diagnostic: how much of c2's time is the grid's tail

    # single-kind diagnostics (not bench lines): where does the time go per data shape
    "huffman": (16384, 65536, ("huffman",), 6, 256),
    "repeats": (16384, 65536, ("repeats",), 6, 256),
    "digits": (16384, 65536, ("digits",), 6, 256),
    "text": (16384, 65536, ("text",), 6, 64),           # English text (twain.txt), dynamic blocks
    "zeros": (16384, 65536, ("zeros",), 6, 4),
    "random": (16384, 65536, ("random",), 6, 256),
    # C4: match-copy-bound, 1 MiB chunks written token by token (258-byte matches at 1-32 KiB distances; dist-1 runs)
    "c4": (1024, 1 << 20, ("match258", "match258", "match258", "run1"), 6, 16),
    # C5 corners: stored / fixed / dynamic at 16 KiB and 4 MiB chunks
    "c5_stored_16k": (65536, 16384, ("random",), 0, 256),
    "c5_stored_4m": (256, 4 << 20, ("random",), 0, 8),
    "c5_fixed_64k": (16384, 65536, ("text_fixed",), 6, 64),
    "c5_dynamic_4m": (256, 4 << 20, ("text",), 6, 4),
}


def build_shard(name, rank):
    """-> dict(stream bytes of this rank's chunks (deflate records only), descriptors, distinct raws)."""
    import zlib
    from compress_amd import capi, synth
    n_chunks, chunk_size, kinds, level, nd = WORKLOADS[name]
    first = rank * n_chunks                       # global index of this rank's first record

    def make(j):
        kind = kinds[j % len(kinds)]
        seed = 1 + 1000003 * (j % len(kinds)) + j
        if kind == "match258":
            return synth.match_chunk(chunk_size, seed)
        if kind == "run1":
            return synth.run_chunk(chunk_size, seed & 255)
        if kind in ("text", "text_fixed"):
            twain = open(os.path.join(ROOT, "tests", "golden", "twain.txt"), "rb").read()
            rep = twain * (chunk_size // len(twain) + 2)
            raw = rep[(seed * 977) % len(twain):][:chunk_size]
            return raw, synth.deflate_chunk(raw, level, zlib.Z_FIXED if kind == "text_fixed" else zlib.Z_DEFAULT_STRATEGY)
        raw = synth.GENERATORS[kind](chunk_size, seed)
        return raw, synth.deflate_chunk(raw, level, zlib.Z_DEFAULT_STRATEGY)

    with ThreadPoolExecutor(16) as ex:
        made = list(ex.map(make, range(nd)))
    comp = bytearray()
    descs = (capi.Chunk * n_chunks)()
    for i in range(n_chunks):
        raw, c = made[(first + i) % nd]
        descs[i] = capi.Chunk(len(comp), len(c), i * chunk_size, len(raw), capi.CHUNK_XFLATE_DEFLATE, 0)
        comp += c
    return {"comp": bytes(comp), "descs": descs, "n": n_chunks, "chunk_size": chunk_size, "made": made,
            "first": first, "nd": nd, "raw_len": n_chunks * chunk_size}


def cpu_baseline(shard, threads):
    """The oracle (a C port of the reference's inflater) over the same descriptors on host threads."""
    import oracle
    lib = oracle.lib()
    n = shard["n"]
    comp = shard["comp"]
    raw = ctypes.create_string_buffer(shard["raw_len"])
    res = (oracle.OrcResult * n)()
    descs = ctypes.cast(shard["descs"], ctypes.c_void_p)
    max_c = max(shard["descs"][i].comp_size for i in range(n))
    per = (n + threads - 1) // threads

    def work(t):
        tmp = ctypes.create_string_buffer(max_c + 16)
        lo = t * per
        cnt = max(0, min(per, n - lo))
        if cnt:
            lib.oracle_inflate_batch(comp, descs, lo, cnt, raw, res, tmp)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(work, range(threads)))
    dt = time.perf_counter() - t0
    bad = sum(1 for i in range(n) if res[i].status != 0 or res[i].out_produced != shard["descs"][i].raw_size)
    return dt, bad, raw


def zlib_baseline(shard, threads):
    """System C zlib (raw inflate) over the same chunks on host threads: the second CPU figure SURVEY 8(d) asks for."""
    import zlib
    n = shard["n"]
    comp = memoryview(shard["comp"])
    descs = shard["descs"]
    per = (n + threads - 1) // threads
    end_block = b"\x01\x00\x00\xff\xff"

    def work(t):
        bad = 0
        for i in range(t * per, min(n, (t + 1) * per)):
            d = descs[i]
            out = zlib.decompressobj(-15).decompress(bytes(comp[d.comp_off:d.comp_off + d.comp_size]) + end_block)
            bad += len(out) != d.raw_size
        return bad

    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        bad = sum(ex.map(work, range(threads)))
    return time.perf_counter() - t0, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-verify", action="store_true")
    args = ap.parse_args()

    import torch
    from compress_amd import capi

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # Rehearsal on a box with fewer GPUs than ranks (never a bench line): BENCH_ONE_DEVICE=1 puts every
    # rank on device 0 and rendezvous goes over gloo, since RCCL refuses two ranks on one device.
    rehearsal = os.environ.get("BENCH_ONE_DEVICE") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    shard = build_shard(args.workload, rank)
    n = shard["n"]
    comp_len, raw_len = len(shard["comp"]), shard["raw_len"]

    ctx = capi.Context(local_rank)
    d_comp = torch.frombuffer(bytearray(shard["comp"]) + bytearray(64), dtype=torch.uint8).to(dev)
    d_chunks = torch.frombuffer(bytearray(bytes(shard["descs"])), dtype=torch.uint8).to(dev)
    d_raw = torch.zeros(raw_len + 64, dtype=torch.uint8, device=dev)
    d_res = torch.zeros(n * ctypes.sizeof(capi.Result), dtype=torch.uint8, device=dev)
    # A non-default torch stream: its handle is what the kernels are launched on, so the
    # torch.cuda.Event pairs below time exactly that stream.
    stream = torch.cuda.Stream(dev)
    assert stream.cuda_stream != 0

    def step():
        ctx.decode_device(d_comp.data_ptr(), comp_len, d_chunks.data_ptr(), n,
                          d_raw.data_ptr(), raw_len, d_res.data_ptr(), stream.cuda_stream)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    torch.cuda.synchronize(dev)
    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize(dev)

    # parity gate (untimed): every chunk ok, sizes as promised, bytes equal to the generator's raw data
    verified = None
    if not args.no_verify:
        res = d_res.cpu().numpy().view("i4,u4,i8,i8")
        ok = bool((res["f0"] == 0).all()) and bool((res["f3"] == shard["chunk_size"]).all())
        nd, cs = shard["nd"], shard["chunk_size"]
        distinct = torch.frombuffer(bytearray(b"".join(m[0] for m in shard["made"])), dtype=torch.uint8).to(dev).view(nd, cs)
        idx = (torch.arange(n, device=dev) + shard["first"]) % nd
        got = d_raw[:raw_len].view(n, cs)
        eq = True
        for lo in range(0, n, 2048):                          # bounded temporaries
            eq = eq and bool(torch.equal(got[lo:lo + 2048], distinct[idx[lo:lo + 2048]]))
        verified = ok and eq
        del distinct
        if not verified:
            print(json.dumps({"error": "parity check failed on rank %d" % rank, "status_ok": ok, "bytes_equal": eq}))
            sys.exit(1)

    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for a, b in evs:
        a.record(stream)
        step()
        b.record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = sum(a.elapsed_time(b) for a, b in evs) / max(args.steps, 1)

    if dist is not None:
        t = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kern_ms = float(t[0]), float(t[1])

    out = None
    if rank == 0:
        total_raw = raw_len * world * args.steps
        value = total_raw / elapsed / 1e9
        algo_bytes = comp_len + raw_len                      # per launch, this rank (every rank has the same shape)
        achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
        n_chunks, chunk_size, kinds, level, nd = WORKLOADS[args.workload]
        out = {
            "metric": "decompressed GB/s (node), xflate decode",
            "value": round(value, 3),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": ("REHEARSAL, all ranks on one device; " if rehearsal else "") + "synthetic (%d distinct chunks per kind-mix tiled to %d chunks per GPU; seeds in compress_amd/synth.py)" % (nd, n_chunks),
            "config": {"workload": "%s: %d x %d KiB chunks per GPU (%s), zlib level %d, sync flush per chunk; "
                                   "index-range sharded over %d GPU(s)" % (args.workload, n_chunks, chunk_size >> 10,
                                                                          "/".join(kinds), level, world),
                       "raw_bytes_per_gpu": raw_len, "comp_bytes_per_gpu": comp_len,
                       "ratio": round(raw_len / comp_len, 3), "verified_bit_exact": verified},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": measured_traffic(args.workload),
                         "kernel": "inflate_chunks_kernel", "kernel_ms": round(kern_ms, 4),
                         "algorithmic_bytes_per_launch": algo_bytes},
        }
        if not args.no_cpu and world == 1:                  # the CPU leg runs at N=1 only
            import oracle
            oracle.build()
            threads = min(16, os.cpu_count() or 1)
            # about 20 core-seconds of CPU work: the whole shard, several times over; the mean rate is reported
            reps, total_dt, bad = 0, 0.0, 0
            while reps < 3 or (total_dt * threads < 20.0 and reps < 64):
                dt, b, _ = cpu_baseline(shard, threads)
                total_dt += dt
                bad += b
                reps += 1
            out["cpu_baseline"] = {"value": round(raw_len * reps / total_dt / 1e9, 3), "unit": "GB/s", "cores": threads,
                                   "kind": "port",
                                   "sample": "this rank's whole shard %d times (%d chunks, %.2f GiB raw each), "
                                             "oracle/inflate_oracle.c over %d host threads, %.1f s wall; %d bad chunks"
                                             % (reps, n, raw_len / 2**30, threads, total_dt, bad)}
            zdt, zbad = zlib_baseline(shard, threads)
            out["cpu_baseline"]["zlib"] = {"value": round(raw_len / zdt / 1e9, 3), "unit": "GB/s", "cores": threads,
                                           "sample": "the same shard once, system zlib %s raw inflate through Python's zlib module "
                                                     "over %d host threads, %.1f s wall; %d bad chunks"
                                                     % (__import__("zlib").ZLIB_RUNTIME_VERSION, threads, zdt, zbad)}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
