"""The N>1 path on CPU: two processes over gloo, each parsing the same XFLATE stream, taking the
record range xfg_shard_plan gives its rank and inflating it (with the CPU oracle standing in for
the GPU, which this container does not have); rank 0 gathers the pieces and checks that they
tile the raw data exactly.  No data-path collective is part of the product: the gather here is
the test's own check."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from compress_amd import capi, shard, synth
        import oracle
        stream, raw = synth.build_mix_stream(37, 5000, index_size=6)      # same bytes on every rank (fixed seeds)
        err, sh = shard.my_shard(stream, rank, world)
        assert err == capi.ERR_NONE
        comp = stream[sh.comp_lo:sh.comp_hi]
        out = bytearray(sh.raw_len)
        for (co, cs, ro, rs, fl) in sh.descs:
            st, data, _, n = oracle.inflate_chunk(comp[co:co + cs], rs + 1)
            assert st == 0 and n == rs
            out[ro:ro + rs] = data
        # every rank learns every range; they must tile [0, n_records) and [0, len(raw))
        mine = torch.tensor([sh.first, sh.last, sh.raw_lo, sh.raw_hi, sh.comp_lo, sh.comp_hi], dtype=torch.int64)
        allr = [torch.zeros(6, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = [t.tolist() for t in allr]
        assert allr[0][0] == 0
        for a, b in zip(allr, allr[1:]):
            assert a[1] == b[0]
            if b[1] > b[0] and a[1] > a[0]:
                assert a[3] == b[2]
        # gather the pieces on rank 0 (padded to the largest)
        m = max(r[3] - r[2] for r in allr)
        piece = torch.zeros(m, dtype=torch.uint8)
        piece[:len(out)] = torch.frombuffer(out, dtype=torch.uint8) if out else piece[:0]
        pieces = [torch.zeros(m, dtype=torch.uint8) for _ in range(world)] if rank == 0 else None
        dist.gather(piece, pieces, dst=0)
        if rank == 0:
            got = b"".join(bytes(p[:r[3] - r[2]].numpy().tobytes()) for p, r in zip(pieces, allr))
            q.put(("ok", got == raw, allr))
    except Exception as e:  # noqa: BLE001
        q.put(("err", "rank %d: %r" % (rank, e), None))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_index_range_sharding_over_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    kind, ok, allr = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
    assert kind == "ok", ok
    assert ok, "sharded pieces do not tile the raw data"
    # balance: no shard carries more than twice its share of the work
    work = [r[3] - r[2] + r[5] - r[4] for r in allr]
    assert max(work) <= 2 * sum(work) / world + 10000


def test_shard_plan_properties():
    from compress_amd import capi
    import random
    rng = random.Random(5)
    for n in (0, 1, 2, 7, 64, 1000):
        chunks = []
        co = ro = 0
        for _ in range(n):
            cs, rs = rng.randrange(5, 70000), rng.randrange(0, 70000)
            chunks.append(capi.Chunk(co, cs, ro, rs, capi.CHUNK_XFLATE_DEFLATE, 0))
            co += cs
            ro += rs
        for world in (1, 2, 3, 8):
            first = capi.shard_plan(chunks, world)
            assert len(first) == world + 1 and first[0] == 0 and first[-1] == n
            assert all(a <= b for a, b in zip(first, first[1:]))
            if n >= 64:
                tot = co + ro
                for k in range(world):
                    w = sum(c.comp_size + c.raw_size for c in chunks[first[k]:first[k + 1]])
                    assert abs(w - tot / world) <= 140000 + 1
