"""xfg_decode_device_noalloc (include/xflate_amd.h): the allocation-free entry point, through the C ABI, against
xfg_decode_device on the same inputs -- bit-exact output and identical per-chunk results -- on two calls in a row,
the second with a smaller input (the counters in the __device__ array keep the last call's contents)."""
import ctypes

import pytest

pytestmark = pytest.mark.gpu


def _table(capi, stream):
    err, chunks = capi.index_parse(stream)
    assert err == capi.ERR_NONE
    return chunks


def _run_both(capi, ctx, dev, stream, chunks, raw_len):
    import torch
    n = len(chunks)
    arr = (capi.Chunk * n)(*chunks)
    d_comp = torch.frombuffer(bytearray(stream) + bytearray(64), dtype=torch.uint8).to(dev)
    d_chunks = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    outs = []
    for which in ("ctx", "noalloc"):
        d_raw = torch.full((raw_len + 64,), 0xEE, dtype=torch.uint8, device=dev)
        d_res = torch.full((n * ctypes.sizeof(capi.Result),), 0xFF, dtype=torch.uint8, device=dev)
        stream_h = torch.cuda.current_stream(dev).cuda_stream
        if which == "ctx":
            ctx.decode_device(d_comp.data_ptr(), len(stream), d_chunks.data_ptr(), n, d_raw.data_ptr(), raw_len,
                              d_res.data_ptr(), stream_h)
        else:
            rc = capi.lib().xfg_decode_device_noalloc(0, d_comp.data_ptr(), len(stream), d_chunks.data_ptr(), n,
                                                      d_raw.data_ptr(), raw_len, d_res.data_ptr(), stream_h or None)
            assert rc == capi.XFG_OK, capi.lib().xfg_last_error()
        torch.cuda.synchronize(dev)
        res = (capi.Result * n).from_buffer_copy(d_res.cpu().numpy().tobytes())
        outs.append((d_raw.cpu().numpy().tobytes(), [(r.status, r.in_consumed, r.out_produced) for r in res]))
    return outs


def test_noalloc_matches_decode_device_twice():
    import torch
    from compress_amd import capi, synth
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0)
    try:
        L = capi.lib()
        assert L.xfg_decode_device_noalloc(0, None, 0, None, 0, None, 0, None, None) == capi.XFG_OK    # set-up only
        big, raw_big = synth.build_mix_stream(300, 40000, index_size=64)        # more chunks than table slices
        small, raw_small = synth.build_mix_stream(7, 9000, index_size=3)
        for stream, raw in ((big, raw_big), (small, raw_small)):
            chunks = _table(capi, stream)
            (out_a, res_a), (out_b, res_b) = _run_both(capi, ctx, dev, stream, chunks, len(raw))
            assert res_a == res_b
            assert all(st == capi.ST_OK for st, _i, _o in res_b)
            assert out_b[:len(raw)] == raw
            assert out_a == out_b                                   # guard bytes behind the output included
        # refused sizes and arguments
        assert L.xfg_decode_device_noalloc(0, 16, 0, 16, (1 << 30) + 1, 16, 0, 16, None) == capi.XFG_E_LIMIT
        assert L.xfg_decode_device_noalloc(-1, None, 0, None, 0, None, 0, None, None) == capi.XFG_E_ARG
        assert L.xfg_decode_device_noalloc(0, None, 0, 16, 1, 16, 0, 16, None) == capi.XFG_E_ARG
    finally:
        ctx.close()
