"""Index-range sharding of an XFLATE stream over the GPUs of a node.

One process per GPU.  Every rank parses the same index (host work, KiB-scale), computes the
same plan with xfg_shard_plan and takes the contiguous record range of its rank: it needs only
the compressed bytes [comp_lo, comp_hi) of the stream and produces the raw bytes
[raw_lo, raw_hi).  Chunks are independent (xflate/index.go:27-31: the index holds every
chunk's input and output range), so no data-path collective is involved.
"""
from . import capi


class Shard(object):
    """This rank's part of the stream: descriptors rebased to the shard's own buffers."""

    def __init__(self, rank, world, first, last, chunks):
        self.rank, self.world = rank, world
        self.first, self.last = first, last                  # record range [first, last)
        mine = chunks[first:last]
        self.comp_lo = mine[0].comp_off if mine else 0
        self.comp_hi = mine[-1].comp_off + mine[-1].comp_size if mine else 0
        self.raw_lo = mine[0].raw_off if mine else 0
        self.raw_hi = mine[-1].raw_off + mine[-1].raw_size if mine else 0
        # descriptors relative to comp[comp_lo:comp_hi] and raw[raw_lo:raw_hi]
        self.descs = [(c.comp_off - self.comp_lo, c.comp_size, c.raw_off - self.raw_lo, c.raw_size, c.flags)
                      for c in mine]

    @property
    def comp_len(self):
        return self.comp_hi - self.comp_lo

    @property
    def raw_len(self):
        return self.raw_hi - self.raw_lo


def plan(chunks, world):
    """-> [Shard...] for all ranks (every rank computes the same list)."""
    first = capi.shard_plan(chunks, world)
    return [Shard(k, world, first[k], first[k + 1], chunks) for k in range(world)]


def my_shard(stream, rank, world):
    """Parse the index of an in-memory stream and return (err class, this rank's Shard)."""
    err, chunks = capi.index_parse(stream)
    if err != capi.ERR_NONE:
        return err, None
    return err, plan(chunks, world)[rank]
