"""bench.py's configs[2] stream is never materialised: synth.assemble_sparse gives its length and the bytes of its
indexes and footer.  It must describe exactly the stream synth.assemble_stream builds from the same chunks, and
xfg_index_parse + xfg_shard_plan over it must cover every record once."""
import os

from compress_amd import capi, synth


def test_sparse_layout_equals_full_assembly():
    raws = [os.urandom(50) * (i + 1) for i in range(7)]
    comps = [synth.deflate_chunk(r) for r in raws]
    k = 5
    full = synth.assemble_stream(comps * k, [len(r) for r in raws] * k, index_size=6)
    n, blobs = synth.assemble_sparse([len(c) for c in comps] * k, [len(r) for r in raws] * k, index_size=6)
    assert n == len(full)
    for off, b in blobs:
        assert full[off:off + len(b)] == b
    # holes + blobs tile the stream: the holes are exactly the chunks
    err, chunks = capi.index_parse(full)
    assert err == capi.ERR_NONE
    blob_at = dict(blobs)
    for c in chunks:
        if c.flags & capi.CHUNK_SYNC:
            assert c.comp_off not in blob_at
        else:
            assert len(blob_at[c.comp_off]) == c.comp_size
    # parsing the sparse image (zeros where the chunks go) gives the same table: the parser reads indexes and footer only
    img = bytearray(n)
    for off, b in blobs:
        img[off:off + len(b)] = b
    err2, chunks2 = capi.index_parse(bytes(img))
    assert err2 == capi.ERR_NONE
    assert [(c.comp_off, c.comp_size, c.raw_off, c.raw_size, c.flags) for c in chunks2] == \
           [(c.comp_off, c.comp_size, c.raw_off, c.raw_size, c.flags) for c in chunks]
    for world in (1, 2, 3, 8):
        first = capi.shard_plan(chunks2, world)
        assert first[0] == 0 and first[-1] == len(chunks2) and all(a <= b for a, b in zip(first, first[1:]))
