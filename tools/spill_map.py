#!/usr/bin/env python3
"""usage: spill_map.py <listing.s built with -gline-tables-only> [function substring]
Where the register spills of the inflate kernels land.  For every function of the compiler's listing, every loop
(label .. backward branch) is listed with its size, the source lines (.loc) most of its instructions come from, and the
spill instructions it holds itself (not in a loop nested inside it): scratch_load / scratch_store (VGPR spills) and
v_writelane / v_readlane (SGPR spills into lanes of a VGPR; the readlanes of the kernel's own wave-uniform walks are
counted too, so a loop that uses readlane on purpose shows up with them)."""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().splitlines()
want = sys.argv[2] if len(sys.argv) > 2 else ""
funcs = []
cur = None
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):\s*(;.*)?$", l)
    if m and "scratchE" not in l:
        cur = (m.group(1), i)
    if l.startswith(".Lfunc_end") and cur:
        funcs.append((cur[0], cur[1], i))
        cur = None

def is_inst(l):
    s = l.strip()
    return bool(s) and not s.startswith((".", ";", "//")) and not s.endswith(":")

def kind(s):
    if s.startswith("scratch_load"): return "vgpr_reload"
    if s.startswith("scratch_store"): return "vgpr_spill"
    if s.startswith("v_writelane_b32"): return "sgpr_spill(writelane)"
    if s.startswith("v_readlane_b32"): return "readlane"
    return None

for name, a, b in funcs:
    if want not in name:
        continue
    label_at = {}
    for i in range(a, b):
        m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
        if m:
            label_at[m.group(1)] = i
    loops = set()
    for i in range(a, b):
        m = re.match(r"\s*s_c?branch\w*\s+(\.LBB\d+_\d+)", lines[i])
        if m and m.group(1) in label_at and label_at[m.group(1)] < i:
            loops.add((label_at[m.group(1)], i))
    loops = sorted(loops)
    ninst = sum(1 for i in range(a, b) if is_inst(lines[i]))
    tot = Counter(k for k in (kind(lines[i].strip()) for i in range(a, b)) if k)
    print("== %s: %d instructions, %d loops; %s" % (name[:70], ninst, len(loops), dict(tot)))
    owner = {}
    for i in range(a, b):
        inner = None
        for (s, e) in loops:
            if s <= i <= e and (inner is None or e - s < inner[1] - inner[0]):
                inner = (s, e)
        owner[i] = inner
    for lp in [None] + loops:
        c = Counter()
        src = Counter()
        n = 0
        for i in range(a, b):
            if owner[i] != lp:
                continue
            s = lines[i].strip()
            k = kind(s)
            if k: c[k] += 1
            if is_inst(lines[i]): n += 1
            m = re.match(r"\.loc\s+\d+\s+(\d+)", s)
            if m: src[int(m.group(1))] += 1
        is_walk = False
        if lp is not None:
            whole_ = sum(1 for i in range(lp[0], lp[1] + 1) if is_inst(lines[i]))
            text_ = "\n".join(lines[lp[0]:lp[1] + 1])
            is_walk = whole_ < 400 and "ds_read_b64" in text_ and "v_bitop3" in text_ and "v_lshrrev_b64" in text_
        if not c and not is_walk:
            continue
        top = ", ".join("%d" % l for l, _ in src.most_common(6))
        if lp is None:
            print("   outside any loop (%d own instructions): %s" % (n, dict(c)))
        else:
            whole = sum(1 for i in range(lp[0], lp[1] + 1) if is_inst(lines[i]))
            text = "\n".join(lines[lp[0]:lp[1] + 1])
            walk = " [a walk loop: bit-buffer top-up + table look-up]" if whole < 400 and "ds_read_b64" in text and "v_bitop3" in text and "v_lshrrev_b64" in text else ""
            depth = sum(1 for (s, e) in loops if s <= lp[0] and lp[1] <= e)
            print("   loop asm %d-%d depth %d, %d instructions (%d its own), source lines mostly %s%s: %s" % (lp[0] + 1, lp[1] + 1, depth, whole, n, top, walk, dict(c)))
