"""Index-range sharding on the HIP path (xflate/index.go:27-31: the index holds every chunk's input
and output range, so contiguous record ranges are independent work).  One multi-index stream is parsed once, cut with
xfg_shard_plan for 2, 3 and 8 ranks, and every rank's range is decoded with xfg_decode_device -- descriptors rebased
to the rank's own device buffers, exactly as bench.py --gpus N and compress_amd/shard.py do it -- one rank after the
other on the one GPU.  The pieces must tile the generator's bytes and every record's result must equal the CPU
oracle's.  The plan's shards start and end on index records too (a small index size puts one every few chunks)."""
import ctypes

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def stream_and_oracle():
    """The stream, its raw bytes, its record table and the oracle's (status, consumed, produced, bytes) per record."""
    import oracle
    from compress_amd import capi, synth
    oracle.lib()
    stream, raw = synth.build_mix_stream(61, 9000, index_size=4)        # 61 chunks, an index after every 4: 16 indexes + footer
    err, chunks = capi.index_parse(stream)
    assert err == capi.ERR_NONE
    want = []
    for c in chunks:
        comp = stream[c.comp_off:c.comp_off + c.comp_size]
        if c.flags & capi.CHUNK_HOST_META:
            # index and footer records: decoded by the host when the index was parsed; consumed = all of it (+ endBlock)
            want.append((capi.ST_OK, c.comp_size + (0 if c.flags & capi.CHUNK_FOOTER else 5), 0, b""))
            continue
        st, data, o_in, o_n = oracle.inflate_chunk(comp, c.raw_size + 1)
        assert st == oracle.OK
        want.append((capi.ST_OK, o_in, o_n, data))
    return stream, raw, chunks, want


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("waves", [0, 1])
def test_shards_decode_on_the_gpu(stream_and_oracle, world, waves):
    import torch
    from compress_amd import capi, shard
    stream, raw, chunks, want = stream_and_oracle
    dev = torch.device("cuda", 0)
    capi.set_decode_waves(waves)
    ctx = capi.Context(0)
    try:
        shards = shard.plan(chunks, world)
        # the plan tiles the record table
        assert shards[0].first == 0 and shards[-1].last == len(chunks)
        assert all(a.last == b.first for a, b in zip(shards, shards[1:]))
        got = bytearray()
        starts_on_meta = ends_on_meta = 0
        for sh in shards:
            n = sh.last - sh.first
            if n == 0:
                continue
            starts_on_meta += bool(chunks[sh.first].flags & capi.CHUNK_HOST_META)
            ends_on_meta += bool(chunks[sh.last - 1].flags & capi.CHUNK_HOST_META)
            # this rank's compressed bytes and nothing else, at a 16-byte aligned device address
            d_comp = torch.frombuffer(bytearray(stream[sh.comp_lo:sh.comp_hi]) + bytearray(64), dtype=torch.uint8).to(dev)
            descs = (capi.Chunk * n)(*[capi.Chunk(co, cs, ro, rs, fl, 0) for (co, cs, ro, rs, fl) in sh.descs])
            d_chunks = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
            d_raw = torch.full((sh.raw_len + 64,), 0xA5, dtype=torch.uint8, device=dev)
            d_res = torch.zeros(n * ctypes.sizeof(capi.Result), dtype=torch.uint8, device=dev)
            ctx.decode_device(d_comp.data_ptr(), sh.comp_len, d_chunks.data_ptr(), n, d_raw.data_ptr(), sh.raw_len, d_res.data_ptr(), None)
            ctx.sync()
            res = (capi.Result * n).from_buffer_copy(d_res.cpu().numpy().tobytes())
            out = d_raw.cpu().numpy().tobytes()
            assert out[sh.raw_len:] == b"\xa5" * 64                # nothing written past the rank's range
            for i in range(n):
                st, o_in, o_n, data = want[sh.first + i]
                r = res[i]
                assert (r.status, r.in_consumed, r.out_produced) == (st, o_in, o_n), (world, sh.rank, i)
                ro = sh.descs[i][2]
                assert out[ro:ro + o_n] == data
            got += out[:sh.raw_len]
        assert bytes(got) == raw
        if world == 8:
            # with an index after every fourth chunk, some of the eight cuts fall on index records
            assert starts_on_meta + ends_on_meta > 0
    finally:
        ctx.close()
        capi.set_decode_waves(0)


def test_shard_boundaries_on_index_records(stream_and_oracle):
    """Cuts forced onto index records: ranges that begin with, end with, or consist only of index / footer records."""
    import torch
    from compress_amd import capi, shard
    stream, raw, chunks, want = stream_and_oracle
    dev = torch.device("cuda", 0)
    meta = [i for i, c in enumerate(chunks) if c.flags & capi.CHUNK_HOST_META]
    assert len(meta) >= 3
    cuts = [0, meta[0], meta[0] + 1, meta[1] + 1, meta[2], len(chunks) - 1, len(chunks)]     # [.., index) [index] (.. index] .. [footer]
    ctx = capi.Context(0)
    try:
        got = bytearray()
        for k, (a, b) in enumerate(zip(cuts, cuts[1:])):
            sh = shard.Shard(k, len(cuts) - 1, a, b, chunks)
            n = b - a
            d_comp = torch.frombuffer(bytearray(stream[sh.comp_lo:sh.comp_hi]) + bytearray(64), dtype=torch.uint8).to(dev)
            descs = (capi.Chunk * n)(*[capi.Chunk(co, cs, ro, rs, fl, 0) for (co, cs, ro, rs, fl) in sh.descs])
            d_chunks = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
            d_raw = torch.zeros(sh.raw_len + 64, dtype=torch.uint8, device=dev)
            d_res = torch.zeros(n * ctypes.sizeof(capi.Result), dtype=torch.uint8, device=dev)
            ctx.decode_device(d_comp.data_ptr(), sh.comp_len, d_chunks.data_ptr(), n, d_raw.data_ptr(), sh.raw_len, d_res.data_ptr(), None)
            ctx.sync()
            res = (capi.Result * n).from_buffer_copy(d_res.cpu().numpy().tobytes())
            for i in range(n):
                st, o_in, o_n, _ = want[a + i]
                assert (res[i].status, res[i].in_consumed, res[i].out_produced) == (st, o_in, o_n), (k, i)
            got += d_raw.cpu().numpy().tobytes()[:sh.raw_len]
        assert bytes(got) == raw
    finally:
        ctx.close()
