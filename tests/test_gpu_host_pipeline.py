"""xfg_decode's pipelined host path (slices, launch units on two streams, pinned slots, helper threads: xfg_api.cpp
decode_pipelined) must give
exactly what the plain path gives: same bytes, same per-chunk results, untouched bytes where a chunk failed."""
import ctypes
import os
import subprocess
import sys
import zlib

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import ctypes, hashlib, os, sys, zlib
sys.path.insert(0, %r)
from compress_amd import capi, synth
import numpy as np
rng = np.random.Generator(np.random.PCG64(5))
n, cs = 1400, 65536 + 4096                     # ~93 MiB of output: six slices in four launch units
kinds = [synth.gen_huffman, synth.gen_repeats, synth.gen_digits]
distinct = []
for j in range(24):
    raw = kinds[j %% 3](cs - (j * 37) %% 999, 900 + j)
    distinct.append((raw, synth.deflate_chunk(raw)))
comp = bytearray(); descs = []; ro = 0; want = []
for i in range(n):
    raw, c = distinct[i %% len(distinct)]
    c = bytearray(c)
    status_bad = False
    if i in (3, 700, 701, 1399):                # damaged chunks: partial output, then an error
        c[len(c) // 2] ^= 0x5a
        status_bad = True
    descs.append((len(comp), len(c), ro, len(raw), capi.CHUNK_XFLATE_DEFLATE))
    comp += c + b"\x00" * (i %% 3)              # gaps between chunks are allowed
    ro += len(raw) + (i %% 5)
    want.append((raw, status_bad))
ctx = capi.Context(0)
arr = (capi.Chunk * n)(*[capi.Chunk(*d, 0) for d in descs])
res = (capi.Result * n)()
out = ctypes.create_string_buffer(b"\xEE" * (ro + 64))
rc = capi.lib().xfg_decode(ctx.handle, bytes(comp), len(comp), arr, n, out, ro, res)
assert rc == 0, rc
h = hashlib.sha256(out.raw).hexdigest()
r = hashlib.sha256(bytes(res)).hexdigest()
ok = sum(1 for i in range(n) if res[i].status == 0)
mv = memoryview(out).cast("B")
for i in range(n):
    raw, bad = want[i]
    if not bad:
        assert res[i].status == 0 and mv[descs[i][2]:descs[i][2] + len(raw)] == raw, i
print(h, r, ok)
'''


def _run(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    p = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    return p.stdout.strip().split()


def test_pipelined_equals_plain():
    a = _run({})
    b = _run({"XFG_NO_PIPELINE": "1"})
    assert a == b
    assert int(a[2]) >= 1390
