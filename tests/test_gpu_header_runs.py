"""Dynamic block headers whose code lengths are written with every mix of explicit lengths and the 16 / 17 / 18 repeat
codes (flate/prefix.go:125-165), at every place in the 64-bit-position windows the header reader works in: runs that cross
a window, runs that cross from the literal/length into the distance lengths, a 16 straight after a zero run -- where Go's
standard library (what xflate.Reader inflates with: the previous array entry, zero) and the in-repo reader (clenLast, the
last explicit length) read different headers (XFG_CHUNK_RFC_REP16) -- and runs that overshoot HLIT + HDIST (corrupted).
The oracle (a restatement of the reference's reader, both rules) decides what every header means; the GPU must agree on
status, InputOffset, OutputOffset and bytes."""
import random

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from compress_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def orc():
    import oracle
    oracle.lib()
    return oracle


def _target_lengths(rng):
    """Complete literal/length and distance codes with stretches of equal lengths and of unused symbols."""
    from compress_amd import synth
    hlit = rng.choice([257, 258, 270, 286, rng.randrange(257, 287)])
    hdist = rng.choice([1, 2, 30, rng.randrange(1, 31)])

    def freqs(n, must):
        f = [0] * n
        i = 0
        while i < n:
            run = rng.choice([1, 1, 2, 3, 4, 7, 12, 20, 70, 140])
            kind = rng.random()
            val = 0 if kind < 0.35 else rng.choice([1, 1, 1, 2, 3, 5, 8, 40, 300, 5000])
            for j in range(i, min(n, i + run)):
                f[j] = val if kind < 0.8 else rng.choice([0, 1, 2, 50])
            i += run
        for m in must:
            if m < n and f[m] == 0:
                f[m] = 1
        return f

    lf = freqs(hlit, [256, rng.randrange(0, 256)])
    df = freqs(hdist, [0])
    ll = synth._limited_lengths(lf + [0] * (286 - hlit), 15)[:hlit]
    dl = synth._limited_lengths(df + [0] * (30 - hdist), 15)[:hdist]
    if sum(1 for x in dl if x) == 1:
        dl = [1 if x else 0 for x in dl]
    if sum(1 for x in ll if x) == 1:
        ll = [1 if x else 0 for x in ll]
    return hlit, hdist, ll, dl


def _encode_runs(seq, rng, rfc):
    """seq -> [(code-length symbol, extra value)], choosing at random among the ways to write each stretch."""
    out = []
    i, n = 0, len(seq)
    arr_prev, last_explicit = None, 0
    while i < n:
        v = seq[i]
        run = 1
        while i + run < n and seq[i + run] == v:
            run += 1
        opts = ["explicit"]
        if v == 0 and run >= 3:
            opts += ["17"] * 2
        if v == 0 and run >= 11:
            opts += ["18"] * 3
        rep_val = arr_prev if rfc else last_explicit
        if i > 0 and rep_val == v and run >= 3:
            opts += ["16"] * 3
        o = rng.choice(opts)
        if o == "explicit":
            out.append((v, 0))
            last_explicit = v
            i += 1
        elif o == "17":
            k = rng.randrange(3, min(run, 10) + 1)
            out.append((17, k - 3))
            i += k
        elif o == "18":
            k = rng.randrange(11, min(run, 138) + 1)
            out.append((18, k - 11))
            i += k
        else:
            k = rng.randrange(3, min(run, 6) + 1)
            out.append((16, k - 3))
            i += k
        arr_prev = v
    return out


def _chunk(rng, rfc, overshoot):
    from compress_amd import synth
    hlit, hdist, ll, dl = _target_lengths(rng)
    codes = _encode_runs(ll + dl, rng, rfc)
    if overshoot and codes[-1][0] >= 16:
        s, x = codes[-1]
        codes[-1] = (s, min(x + rng.randrange(1, 4), {16: 3, 17: 7, 18: 127}[s]))
    cf = [0] * 19
    for s, _ in codes:
        cf[s] += 1
    cl = synth._limited_lengths(cf, 7)
    if sum(1 for x in cl if x) == 1:
        cl = [1 if x else 0 for x in cl]
    hclen = max(4, max(i for i, s in enumerate(synth._CLEN_ORDER) if cl[s]) + 1)
    w = synth._BitWriter()
    w.bits(1, 1)
    w.bits(2, 2)
    w.bits(hlit - 257, 5)
    w.bits(hdist - 1, 5)
    w.bits(hclen - 4, 4)
    for i in range(hclen):
        w.bits(cl[synth._CLEN_ORDER[i]], 3)
    cc = synth._canonical(cl)
    for s, x in codes:
        w.code(cc[s], cl[s])
        if s >= 16:
            w.bits(x, {16: 2, 17: 3, 18: 7}[s])
    # the body: every literal that has a code, then matches through every length / distance code that can be used yet
    lc, dc = synth._canonical(ll), synth._canonical(dl)
    produced = 0
    lits = [s for s in range(min(256, hlit)) if ll[s]]
    for rep in range(4):
        for s in lits:
            w.code(lc[s], ll[s])
            produced += 1
    if produced:
        for ls in range(257, hlit):
            if not ll[ls]:
                continue
            cand = [d for d in range(hdist) if dl[d] and synth._DBASE[d] <= produced]
            if not cand:
                break
            d = rng.choice(cand)
            w.code(lc[ls], ll[ls])
            w.bits(rng.randrange(1 << synth._LEXT[ls - 257]) if synth._LEXT[ls - 257] else 0, synth._LEXT[ls - 257])
            w.code(dc[d], dl[d])
            ext = synth._DEXT[d]
            dx = rng.randrange(1 << ext) if ext else 0
            if synth._DBASE[d] + dx > produced:
                dx = 0
            w.bits(dx, ext)
            produced += 3                                   # at least
    w.code(lc[256], ll[256])
    w.align()
    return bytes(w.out)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_repeat_codes_everywhere(ctx, orc, seed):
    from compress_amd import capi
    rng = random.Random(9000 + seed)
    comp = bytearray()
    chunks = []
    cap = 8192
    for t in range(300):
        rfc = bool(t & 1)
        c = _chunk(rng, rfc, overshoot=rng.random() < 0.08)
        # decoded under the rule it was written for, and (every fourth one) under the other rule as well
        for flag_rfc in ([rfc, not rfc] if t % 4 == 0 else [rfc]):
            chunks.append((len(comp), len(c), len(chunks) * cap, cap, capi.CHUNK_RFC_REP16 if flag_rfc else 0))
        comp += c
        comp += bytes(rng.randrange(0, 4))
    out, res = ctx.decode(bytes(comp), chunks, cap * len(chunks))
    n_ok = 0
    for c, r in zip(chunks, res):
        data = bytes(comp[c[0]:c[0] + c[1]])
        st, o_out, o_in, o_on = orc.inflate(data, cap, rfc_rep16=bool(c[4] & capi.CHUNK_RFC_REP16))
        want = {orc.OK: capi.ST_OK, orc.CORRUPTED: capi.ST_CORRUPTED, orc.UNEXPECTED_EOF: capi.ST_UNEXPECTED_EOF,
                orc.OUT_FULL: capi.ST_OUT_FULL}[st]
        assert r.status == want, (capi.ST_NAMES[r.status], capi.ST_NAMES[want], c)
        assert r.out_produced == o_on, c
        assert out.raw[c[2]:c[2] + o_on] == o_out, c
        if st != orc.OUT_FULL:
            assert r.in_consumed == o_in, c
        n_ok += st == orc.OK
    assert n_ok >= 200                                      # most headers are valid ones: the test is about them
