"""The token-level DEFLATE writer behind BASELINE config C4 (compress_amd/synth.py), against C zlib
and the CPU oracle.  Host logic only."""
import random
import zlib

import pytest

from compress_amd import synth


def _inflate(comp):
    return zlib.decompressobj(-15).decompress(comp)


@pytest.mark.parametrize("dynamic", [True, False])
def test_small_token_lists(dynamic, oracle_lib):
    rng = random.Random(3)
    for trial in range(40):
        toks, n = [], 0
        for _ in range(rng.randrange(0, 200)):
            if n == 0 or rng.random() < 0.5:
                toks.append(rng.randrange(256))
                n += 1
            else:
                length = rng.choice([3, 4, 10, 11, 12, 130, 131, 257, 258, rng.randrange(3, 259)])
                dist = min(n, rng.choice([1, 2, 3, 4, 5, 24577, 32768, rng.randrange(1, 32769)]))
                toks.append((length, dist))
                n += length
        raw = synth.apply_tokens(toks)
        comp = synth.deflate_tokens(toks, dynamic)
        assert comp[-4:] == b"\x00\x00\xff\xff"
        assert _inflate(comp) == raw
        st, out, in_off, n_out = oracle_lib.inflate_chunk(comp, len(raw) + 1)
        assert (st, out, n_out) == (oracle_lib.OK, raw, len(raw))


@pytest.mark.parametrize("dynamic", [True, False])
def test_match_chunk_shape(dynamic, oracle_lib):
    """C4: 258-byte matches at 1-32 KiB distances after a 32 KiB random seed; and the dist-1 run."""
    raw, comp = synth.match_chunk(1 << 18, 11, dynamic=dynamic)
    assert len(raw) == 1 << 18 and _inflate(comp) == raw
    st, out, _, n_out = oracle_lib.inflate_chunk(comp, len(raw) + 1)
    assert (st, n_out) == (oracle_lib.OK, len(raw)) and out == raw
    raw, comp = synth.run_chunk(70000, 0, dynamic)
    assert raw == bytes(70000) and _inflate(comp) == raw
