"""Huffman-table and symbol statistics of the bench's chunk kinds (design input for the table layout; CPU only)."""
import sys, zlib, collections
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from compress_amd import synth

CLEN_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
LEN_BASE = [3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LEN_XB = [0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DIST_XB = [0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]

class BR:
    def __init__(s, data): s.d = data; s.p = 0
    def bits(s, n):
        v = 0
        for i in range(n):
            v |= ((s.d[s.p >> 3] >> (s.p & 7)) & 1) << i; s.p += 1
        return v

def canon(lens):
    cnt = collections.Counter(l for l in lens if l)
    code = 0; nxt = {}
    for l in range(1, 16):
        code = (code + cnt.get(l - 1, 0)) << 1 if l > 1 else 0
        nxt[l] = code
    # recompute properly
    code = 0; nxt = {}
    for l in range(1, 16):
        code = (code + cnt.get(l - 1, 0)) << 1
        nxt[l] = code
    m = {}
    for s, l in enumerate(lens):
        if l:
            m[(l, nxt[l])] = s; nxt[l] += 1
    return m

def dec(br, m):
    c = 0
    for l in range(1, 16):
        c = (c << 1) | br.bits(1)
        if (l, c) in m: return m[(l, c)], l
    raise ValueError("bad code")

def sub_entries(lens, root):
    """second-level entries, sized per root prefix by the longest code under it (zlib inftrees style)"""
    m = canon(lens)
    per = {}
    for (l, c), s in m.items():
        if l > root:
            pre = c >> (l - root)
            per[pre] = max(per.get(pre, 0), l - root)
    return sum(1 << v for v in per.values()), len(per)

def analyse(comp, st):
    br = BR(comp + b"\0" * 8)
    while True:
        final = br.bits(1); typ = br.bits(2)
        if typ == 0:
            br.p = (br.p + 7) & ~7
            n = br.bits(16); br.bits(16); br.p += 8 * n
        elif typ == 2:
            h0 = br.p
            hlit = br.bits(5) + 257; hdist = br.bits(5) + 1; hclen = br.bits(4) + 4
            cl = [0] * 19
            for i in range(hclen): cl[CLEN_ORDER[i]] = br.bits(3)
            cm = canon(cl)
            lens = []
            while len(lens) < hlit + hdist:
                s, _ = dec(br, cm)
                if s < 16: lens.append(s)
                elif s == 16: lens += [lens[-1]] * (3 + br.bits(2))
                elif s == 17: lens += [0] * (3 + br.bits(3))
                else: lens += [0] * (11 + br.bits(7))
            ll, dl = lens[:hlit], lens[hlit:hlit + hdist]
            st["blocks"] += 1
            st["hdr_bits"] += br.p - h0
            st["lit_max"][max(ll)] += 1
            st["dist_max"][max(dl) if any(dl) else 0] += 1
            for r in (9, 10):
                e, n = sub_entries(ll, r); st["lit_sub%d" % r].append(e)
            for r in (6, 7, 8):
                e, n = sub_entries(dl, r); st["dist_sub%d" % r].append(e)
            lm, dm = canon(ll), canon(dl)
            b0 = br.p
            while True:
                s, l = dec(br, lm)
                st["lit_syms"] += 1
                if l > 9: st["lit_gt9"] += 1
                if l > 10: st["lit_gt10"] += 1
                if s < 256: st["lits"] += 1; st["out"] += 1
                elif s == 256: break
                else:
                    ln = LEN_BASE[s - 257] + br.bits(LEN_XB[s - 257])
                    d, l2 = dec(br, dm)
                    st["dist_syms"] += 1
                    if l2 > 8: st["dist_gt8"] += 1
                    if l2 > 6: st["dist_gt6"] += 1
                    br.bits(DIST_XB[d])
                    st["matches"] += 1; st["out"] += ln
            st["body_bits"] += br.p - b0
        elif typ == 1:
            raise ValueError("fixed block: not analysed")
        if final: break
        if br.p >= 8 * len(comp): break

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    kinds = {"huffman": synth.gen_huffman, "repeats": synth.gen_repeats, "digits": synth.gen_digits}
    import os
    tw = open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "twain.txt"), "rb").read()
    for kind in ("huffman", "repeats", "digits", "text"):
        st = collections.defaultdict(int)
        for k in ("lit_max", "dist_max"): st[k] = collections.Counter()
        for k in ("lit_sub9", "lit_sub10", "dist_sub6", "dist_sub7", "dist_sub8"): st[k] = []
        for i in range(n):
            raw = tw[i * 65536:(i + 1) * 65536] if kind == "text" else kinds[kind](65536, 100 + i)
            c = zlib.compressobj(6, zlib.DEFLATED, -15)
            comp = c.compress(raw) + c.flush(zlib.Z_SYNC_FLUSH)
            analyse(comp, st)
        toks = st["lits"] + st["matches"]
        print("== %s: %d chunks, %d blocks, header %.0f bits/block, body %.0f bits/block" % (kind, n, st["blocks"], st["hdr_bits"] / st["blocks"], st["body_bits"] / st["blocks"]))
        print("   tokens %d (%.1f %% matches), %.2f bits/token, %.2f out bytes/token, %.1f tokens per 208 bits" % (
            toks, 100.0 * st["matches"] / toks, st["body_bits"] / toks, st["out"] / toks, 208.0 * toks / st["body_bits"]))
        print("   lit maxbits %s  dist maxbits %s" % (dict(sorted(st["lit_max"].items())), dict(sorted(st["dist_max"].items()))))
        for k in ("lit_sub9", "lit_sub10", "dist_sub6", "dist_sub7", "dist_sub8"):
            print("   %-10s sub entries: max %d mean %.0f" % (k, max(st[k]), sum(st[k]) / len(st[k])))
        print("   lit/len symbols with code > 9 bits: %.3f %%, > 10 bits: %.3f %%; dist codes > 8 bits: %.3f %%, > 6: %.3f %%" % (
            100.0 * st["lit_gt9"] / st["lit_syms"], 100.0 * st["lit_gt10"] / st["lit_syms"],
            100.0 * st["dist_gt8"] / max(1, st["dist_syms"]), 100.0 * st["dist_gt6"] / max(1, st["dist_syms"])))

main()
