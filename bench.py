#!/usr/bin/env python3
"""bench.py -- XFLATE decode throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (xfg_decode_device: the inflate kernels) over this
rank's shard of a synthetic XFLATE stream that is already resident in HBM.

  N = 1   BASELINE.json configs[1] ("c2"): 1 GiB of raw data in 16 384 chunks of 64 KiB,
          round-robin huffman / repeats / digits generators, zlib level 6, one sync flush
          per chunk.
  N > 1   BASELINE.json configs[2] ("c3_stream"): ONE 32 GiB stream of 131 072 chunks of
          256 KiB with its 32 indexes (4096 records each) and footer.  Every rank parses the
          index with xfg_index_parse, cuts the record table with xfg_shard_plan and decodes
          the records [first[k], first[k+1]) of its rank -- index and footer records included,
          as xflate.Reader does -- from its own device buffers: index-range sharding, no
          data-path collective.  The total work is fixed: "scaling": "strong".  The stream's
          chunk bytes are 384 distinct chunks tiled (chunks are independent, so a tiling with
          a fresh index is a valid stream); only the indexes and the footer exist on the host
          (a sparse file), each rank's compressed range is put together on its device.

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
    """Fabric-side bytes per launch from the committed rocprofv3 counter passes (profiles/r3/traffic.json, collected by
    tools/traffic_pass.sh with this same bench.py command per workload), or None.
    FETCH_SIZE counts the narrow match-source loads 1 : 1 (64 bytes per scattered load) and the wide coalesced reads of the
    compressed bytes at half (calibrated: profiles/r2/calibration_narrow_loads.txt), WRITE_SIZE is exact; both in KB."""
    for rnd in ("r3", "r2"):
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", rnd, "traffic.json")))[workload]
            return int((t["FETCH_SIZE_KB"] + t["WRITE_SIZE_KB"]) * 1024 + t["comp_bytes"] // 2)
        except (OSError, KeyError, ValueError):
            continue
    return None


def stream_traffic(algo_bytes_per_gpu):
    """The 32 GiB stream's launches (c3_stream: one GPU's record range) have no counter pass of their own -- the driver
    runs them on N GPUs; their chunks are the c3 workload's (the same 384 distinct 256 KiB chunks, tiled), so the
    traffic of c3's launch is scaled by the algorithmic bytes of this launch.  Stated in roofline.traffic_note."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r3", "traffic.json")))["c3"]
        b = json.load(open(os.path.join(ROOT, "profiles", "r3", "traffic_c3_bench.json")))
        ref_algo = b["roofline"]["algorithmic_bytes_per_launch"]
        ref = (t["FETCH_SIZE_KB"] + t["WRITE_SIZE_KB"]) * 1024 + t["comp_bytes"] // 2
        return int(ref * (algo_bytes_per_gpu / ref_algo))
    except (OSError, KeyError, ValueError):
        return None

WORKLOADS = {
    # name: (chunks per GPU, chunk size, kinds, zlib level, distinct chunks generated)
    "c2": (16384, 65536, ("huffman", "repeats", "digits"), 6, 768),
    "c3": (16384, 262144, ("huffman", "repeats", "digits"), 6, 384),     # 4 GiB per GPU, 256 KiB chunks
    "small": (768, 65536, ("huffman", "repeats", "digits"), 6, 96),
    # BASELINE configs[2]: the whole 32 GiB stream (chunk count is the STREAM's, shared out over the ranks by xfg_shard_plan)
    "c3_stream": (131072, 262144, ("huffman", "repeats", "digits"), 6, 384),
    "c3_stream_small": (8192, 262144, ("huffman", "repeats", "digits"), 6, 96),   # the same path at 2 GiB (rehearsals, tests)
    "c2x4": (65536, 65536, ("huffman", "repeats", "digits"), 6, 768),    # diagnostic: how much of c2's time is the grid's tail
    # diagnostics for the workgroup-per-chunk mode: few chunks, one data shape each
    "mid1k": (1024, 65536, ("huffman", "repeats", "digits"), 6, 192),
    "mid2k": (2048, 65536, ("huffman", "repeats", "digits"), 6, 192),
    "mid3k": (3072, 65536, ("huffman", "repeats", "digits"), 6, 192),
    "mid4k": (4096, 65536, ("huffman", "repeats", "digits"), 6, 192),
    "mid6k": (6144, 65536, ("huffman", "repeats", "digits"), 6, 192),
    "mid8k": (8192, 65536, ("huffman", "repeats", "digits"), 6, 192),
    "small_huffman": (768, 65536, ("huffman",), 6, 96),
    "small_repeats": (768, 65536, ("repeats",), 6, 96),
    "small_digits": (768, 65536, ("digits",), 6, 96),
    "small_text": (768, 65536, ("text",), 6, 64),
    "one_text_64k": (1, 65536, ("text",), 6, 1),
    "one_text_4m": (1, 4 << 20, ("text",), 6, 1),
    "few_text_256k": (64, 262144, ("text",), 6, 16),

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


def make_distinct(name):
    """The workload's distinct chunks: [(raw, compressed)...]."""
    import zlib
    from compress_amd import synth
    n_chunks, chunk_size, kinds, level, nd = WORKLOADS[name]

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
        return list(ex.map(make, range(nd)))


def build_stream_shard(name, rank, world, dev):
    """BASELINE configs[2].  The stream exists as a sparse file (indexes and footer at their offsets, holes where
    the tiled chunks go); xfg_index_parse reads it through an mmap, xfg_shard_plan cuts the record table, and this
    rank's compressed byte range is put together on the device from the distinct chunks.
    -> dict like build_shard's, with the device tensors already made."""
    import mmap
    import tempfile
    import numpy as np
    import torch
    from compress_amd import capi, synth
    n_chunks, chunk_size, kinds, level, nd = WORKLOADS[name]
    made = make_distinct(name)
    csz = [len(m[1]) for m in made]
    total_len, blobs = synth.assemble_sparse([csz[i % nd] for i in range(n_chunks)], [chunk_size] * n_chunks)
    with tempfile.TemporaryFile(dir=os.environ.get("TMPDIR") or "/tmp") as f:
        f.truncate(total_len)
        for off, b in blobs:
            f.seek(off)
            f.write(b)
        f.flush()
        mm = mmap.mmap(f.fileno(), total_len, prot=mmap.PROT_READ)
        view = np.frombuffer(mm, dtype=np.uint8)
        ptr = ctypes.c_void_p(view.ctypes.data)
        L = capi.lib()
        nrec = ctypes.c_size_t()
        err = L.xfg_index_parse(ptr, total_len, None, 0, ctypes.byref(nrec))
        if err != capi.ERR_NONE:
            raise SystemExit("xfg_index_parse failed: class %d" % err)
        table = (capi.Chunk * nrec.value)()
        err = L.xfg_index_parse(ptr, total_len, table, nrec.value, ctypes.byref(nrec))
        if err != capi.ERR_NONE:
            raise SystemExit("xfg_index_parse failed: class %d" % err)
        del view
        mm.close()
    first = (ctypes.c_size_t * (world + 1))()
    capi.check(L.xfg_shard_plan(table, nrec.value, world, first), "xfg_shard_plan")
    lo, hi = first[rank], first[rank + 1]
    n = hi - lo
    comp_lo = table[lo].comp_off if n else 0
    comp_hi = table[hi - 1].comp_off + table[hi - 1].comp_size if n else 0
    raw_lo = table[lo].raw_off if n else 0
    raw_hi = table[hi - 1].raw_off + table[hi - 1].raw_size if n else 0
    comp_len, raw_len = comp_hi - comp_lo, raw_hi - raw_lo
    # this rank's descriptors, relative to its own buffers
    descs = (capi.Chunk * max(n, 1))()
    for i in range(n):
        c = table[lo + i]
        descs[i] = capi.Chunk(c.comp_off - comp_lo, c.comp_size, c.raw_off - raw_lo, c.raw_size, c.flags, 0)
    # the compressed range on the device: distinct chunks tiled, index / footer blobs in between
    d_distinct = torch.frombuffer(bytearray(b"".join(m[1] for m in made)), dtype=torch.uint8).to(dev)
    dstart = np.concatenate([[0], np.cumsum(csz)])
    d_comp = torch.zeros(comp_len + 64, dtype=torch.uint8, device=dev)
    blob_at = dict(blobs)
    n_deflate = 0
    deflate_first = None                                        # stream-wide ordinal of this rank's first deflate record
    ordinal = sum(1 for i in range(lo) if table[i].flags & capi.CHUNK_SYNC)
    for i in range(n):
        c = table[lo + i]
        dst = c.comp_off - comp_lo
        if c.flags & capi.CHUNK_SYNC:                           # a deflate record: the tiled chunk
            if deflate_first is None:
                deflate_first = ordinal
            k = (ordinal + n_deflate) % nd
            assert c.comp_size == csz[k]
            d_comp[dst:dst + c.comp_size] = d_distinct[int(dstart[k]):int(dstart[k + 1])]
            n_deflate += 1
        else:                                                   # an index or the footer: its own bytes
            b = blob_at[c.comp_off]
            assert len(b) == c.comp_size
            d_comp[dst:dst + len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    del d_distinct
    return {"descs": descs, "n": n, "chunk_size": chunk_size, "made": made, "first": deflate_first or 0, "nd": nd,
            "raw_len": raw_len, "comp_len": comp_len, "d_comp": d_comp, "n_deflate": n_deflate,
            "stream": {"records": nrec.value, "bytes": total_len, "raw_bytes": n_chunks * chunk_size,
                       "indexes": len(blobs) - 1, "rank_records": [int(first[k + 1] - first[k]) for k in range(world)]}}


def build_shard(name, rank):
    """-> dict(stream bytes of this rank's chunks (deflate records only), descriptors, distinct raws)."""
    from compress_amd import capi
    n_chunks, chunk_size, kinds, level, nd = WORKLOADS[name]
    first = rank * n_chunks                       # global index of this rank's first record
    made = make_distinct(name)
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


def go_baseline_note():
    """the reference Go reader, goroutine-parallel, is the baseline north_star names; it needs a Go
    toolchain and the reference module, neither of which exists on the box.  Say so, never substitute silently."""
    import shutil
    import subprocess
    exe = shutil.which("go")
    if not exe:
        return "unavailable (no toolchain: `go` is not on PATH)"
    try:
        ver = subprocess.run([exe, "version"], capture_output=True, text=True, timeout=10).stdout.strip()
    except Exception as e:                                     # noqa: BLE001
        return "unavailable (`go version` failed: %s)" % e
    return "toolchain present (%s) but the reference module is not on this box: not run" % ver


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: c2 on one GPU (BASELINE configs[1]), c3_stream on several (configs[2])")
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

    if args.workload is None:
        args.workload = "c2" if world == 1 else "c3_stream"
    is_stream = args.workload.startswith("c3_stream")
    if is_stream:
        shard = build_stream_shard(args.workload, rank, world, dev)
        n = shard["n"]
        comp_len, raw_len = shard["comp_len"], shard["raw_len"]
        d_comp = shard["d_comp"]
    else:
        shard = build_shard(args.workload, rank)
        n = shard["n"]
        comp_len, raw_len = len(shard["comp"]), shard["raw_len"]
        d_comp = torch.frombuffer(bytearray(shard["comp"]) + bytearray(64), dtype=torch.uint8).to(dev)

    ctx = capi.Context(local_rank)
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

    # parity gate (untimed; before the timed loop and once more after it, on what the LAST timed step left in HBM):
    # every chunk ok, sizes as promised, bytes equal to the generator's raw data
    def verify(when):
        import numpy as np
        res = d_res.cpu().numpy().view("i4,u4,i8,i8")
        want_out = np.array([shard["descs"][i].raw_size for i in range(n)], dtype=np.int64)
        ok = bool((res["f0"] == 0).all()) and bool((res["f3"] == want_out).all())
        nd, cs = shard["nd"], shard["chunk_size"]
        distinct = torch.frombuffer(bytearray(b"".join(m[0] for m in shard["made"])), dtype=torch.uint8).to(dev).view(nd, cs)
        nv = shard.get("n_deflate", n)                        # records that produce output (index and footer records produce none)
        ok = ok and nv * cs == raw_len
        idx = (torch.arange(nv, device=dev) + shard["first"]) % nd
        got = d_raw[:raw_len].view(nv, cs)
        eq = True
        for lo in range(0, nv, 2048):                         # bounded temporaries
            eq = eq and bool(torch.equal(got[lo:lo + 2048], distinct[idx[lo:lo + 2048]]))
        del distinct
        if not (ok and eq):
            print(json.dumps({"error": "parity check failed on rank %d (%s the timed loop)" % (rank, when), "status_ok": ok, "bytes_equal": eq}))
            sys.exit(1)
        return True

    verified = None
    if not args.no_verify:
        verified = verify("before")
        d_raw.zero_()                                         # the timed steps have to write every byte again
        d_res.zero_()
        torch.cuda.synchronize(dev)

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
    if not args.no_verify and args.steps > 0:
        verified = verify("after")

    sum_raw, sum_comp = float(raw_len), float(comp_len)
    if dist is not None:
        t = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kern_ms = float(t[0]), float(t[1])
        t = torch.tensor([sum_raw, sum_comp], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        sum_raw, sum_comp = float(t[0]), float(t[1])

    out = None
    if rank == 0:
        total_raw = sum_raw * args.steps                     # what all ranks decoded
        value = total_raw / elapsed / 1e9
        algo_bytes = int(sum_comp + sum_raw)                 # per launch, all ranks: compressed bytes read + raw bytes written
        # the dominant kernel's roofline, per GPU: this node's algorithmic bytes over the slowest rank's mean launch time,
        # against n_gpus x 8 TB/s
        achieved = algo_bytes / (kern_ms * 1e-3) / 1e9 / world
        n_chunks, chunk_size, kinds, level, nd = WORKLOADS[args.workload]
        # one wavefront per chunk, or a workgroup per chunk when the launch has few chunks (xfg_set_decode_waves)
        waves_per_chunk = int(capi.lib().xfg_decode_waves_for(n, torch.cuda.get_device_properties(dev).multi_processor_count))
        out = {
            "metric": "decompressed GB/s (node), xflate decode",
            "value": round(value, 3),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if is_stream else "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": ("REHEARSAL, all ranks on one device; " if rehearsal else "") + "synthetic (%d distinct chunks per kind-mix tiled to %d chunks per %s; seeds in compress_amd/synth.py)" % (nd, n_chunks, "stream" if is_stream else "GPU"),
            "config": ({"workload": "%s: ONE %.0f GiB XFLATE stream, %d x %d KiB chunks (%s), zlib level %d, sync flush per chunk, "
                                    "%d indexes + footer parsed by xfg_index_parse, records cut by xfg_shard_plan over %d GPU(s)"
                                    % (args.workload, shard["stream"]["raw_bytes"] / 2**30, n_chunks, chunk_size >> 10, "/".join(kinds), level,
                                       shard["stream"]["indexes"], world),
                        "stream_records": shard["stream"]["records"], "stream_bytes": shard["stream"]["bytes"],
                        "records_per_rank": shard["stream"]["rank_records"],
                        "raw_bytes_total": int(sum_raw), "comp_bytes_total": int(sum_comp),
                        "ratio": round(sum_raw / sum_comp, 3), "verified_bit_exact": verified}
                       if is_stream else
                       {"workload": "%s: %d x %d KiB chunks per GPU (%s), zlib level %d, sync flush per chunk; "
                                    "index-range sharded over %d GPU(s)" % (args.workload, n_chunks, chunk_size >> 10,
                                                                           "/".join(kinds), level, world),
                        "raw_bytes_per_gpu": raw_len, "comp_bytes_per_gpu": comp_len,
                        "ratio": round(raw_len / comp_len, 3), "verified_bit_exact": verified}),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": stream_traffic(algo_bytes / world) if is_stream else measured_traffic(args.workload),
                         "kernel": "inflate_chunks_kernel" if waves_per_chunk == 1 else "inflate_chunks_wg_kernel",
                         "waves_per_chunk": waves_per_chunk, "kernel_ms": round(kern_ms, 4),
                         "algorithmic_bytes_per_launch": algo_bytes if world == 1 else int(algo_bytes / world),
                         "per": "GPU (achieved and peak are per GPU; launches of all ranks run side by side)",
                         "traffic_note": ("per GPU: the counter passes of the c3 workload (the same 256 KiB chunks, 16 384 per launch; profiles/r3/traffic.json) "
                                          "scaled by this launch's algorithmic bytes" if is_stream else
                                          "per launch, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench command (profiles/r3/traffic.json, tools/traffic_pass.sh)")},
        }
        if not args.no_cpu and world == 1:                  # the CPU leg runs at N=1 only
            import oracle
            oracle.build()
            # host threads: the CPU share this process has (the GPU box grants 16 CPUs per GPU and exports it as
            # OMP_NUM_THREADS, while os.cpu_count() shows the whole machine); all of it is used and stated
            visible = os.cpu_count() or 1
            try:
                share = int(os.environ.get("OMP_NUM_THREADS", "0"))
            except ValueError:
                share = 0
            threads = max(1, min(visible, share)) if share > 0 else min(visible, len(os.sched_getaffinity(0)))
            # about 20 core-seconds of CPU work: the whole shard, several times over; the mean rate is reported
            reps, total_dt, bad = 0, 0.0, 0
            while reps < 3 or (total_dt * threads < 20.0 and reps < 64):
                dt, b, _ = cpu_baseline(shard, threads)
                total_dt += dt
                bad += b
                reps += 1
            out["cpu_baseline"] = {"value": round(raw_len * reps / total_dt / 1e9, 3), "unit": "GB/s", "cores": threads,
                                   "host_cpus_visible": visible,
                                   "kind": "port",
                                   "go": go_baseline_note(),
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
