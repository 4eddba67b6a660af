#!/usr/bin/env python3
"""Where does a chunk's time go?  Runs one decode of a bench workload on the diagnostic build
(make -C compress_amd/csrc prof; XFG_LIB=compress_amd/lib/libxflate_amd_prof.so) and prints the
in-kernel phase counters (wave-cycles summed over all waves, and event counts)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("XFG_LIB", os.path.join(ROOT, "compress_amd", "lib", "libxflate_amd_prof.so"))
import torch  # noqa: E402
import bench  # noqa: E402
from compress_amd import capi  # noqa: E402

NAMES = ["header", "window", "count_passes", "emit_pass", "resolve", "stored", "total",
         "n_batch", "n_pass", "n_live_lanes", "n_rounds", "n_blocks", "flush", "n_tokens",
         "h_clen_table", "h_lens_loop", "h_lit_table", "h_dist_table", "r_sync", "r_sweep", "p_spec", "p_fix", "p_count", "n_fix",
         "c_fix_lost", "c_flag_cuts", "c_flag_lost", "c_cap_cuts", "c_cap_lost", "t_sort", "t_root", "t_sub",
         "n_wfull", "n_wf_tok", "n_wf_out", "n_long", "n_long_g", "t_long", "n_indep", "n_long_rdy",
         "w_chain", "w_cnt", "w_out", "w_water", "w_groups", "w_abort", "w_alone", "w_refix",
         "n_dirty1", "n_dirty2", "n_dirty3", "n_dirty4"]


def main():
    for w in sys.argv[1:] or ["c2"]:
        shard = bench.build_shard(w, 0)
        dev = torch.device("cuda", 0)
        ctx = capi.Context(0)
        n = shard["n"]
        d_comp = torch.frombuffer(bytearray(shard["comp"]) + bytearray(64), dtype=torch.uint8).to(dev)
        d_chunks = torch.frombuffer(bytearray(bytes(shard["descs"])), dtype=torch.uint8).to(dev)
        d_raw = torch.zeros(shard["raw_len"] + 64, dtype=torch.uint8, device=dev)
        d_res = torch.zeros(n * ctypes.sizeof(capi.Result), dtype=torch.uint8, device=dev)
        L = capi.lib()
        buf = (ctypes.c_ulonglong * 56)()
        for it in range(2):
            L.xfg_debug_profile(buf, 56)          # clear
            ctx.decode_device(d_comp.data_ptr(), len(shard["comp"]), d_chunks.data_ptr(), n,
                              d_raw.data_ptr(), shard["raw_len"], d_res.data_ptr(), None)
            ctx.sync()
        L.xfg_debug_profile(buf, 56)
        v = {k: buf[i] for i, k in enumerate(NAMES)}
        tot = max(v["total"], 1)
        print("== %s: %d chunks" % (w, n))
        for k in NAMES[:7]:
            print("  %-13s %14d wave-cycles  %5.1f %%" % (k, v[k], 100.0 * v[k] / tot))
        blk = max(v["n_blocks"], 1)
        print("  header per block: clen table %.0f, lengths loop %.0f, lit table %.0f, dist table %.0f" %
              (v["h_clen_table"] / blk, v["h_lens_loop"] / blk, v["h_lit_table"] / blk, v["h_dist_table"] / blk))
        print("  literal/length table per block: histogram + canonical order %.0f, root table %.0f, second-level tables %.0f" %
              (v["t_sort"] / blk, v["t_root"] / blk, v["t_sub"] / blk))
        nb = max(v["n_batch"], 1)
        print("  walks per batch: speculative %.0f, re-synchronisation %.0f (%.2f walks), count %.0f cycles" %
              (v["p_spec"] / nb, v["p_fix"] / nb, v["n_fix"] / nb, v["p_count"] / nb))
        print("  lanes that walk per batch: re-synchronisation walk 1: %.1f, walk 2: %.1f, walk 3: %.2f, walks 4 and later: %.2f" %
              (v["n_dirty1"] / nb, v["n_dirty2"] / nb, v["n_dirty3"] / nb, v["n_dirty4"] / nb))
        print("  resolve per batch: wait for earlier stores %.0f, pre-window sweep %.0f" % (v["r_sync"] / nb, v["r_sweep"] / nb))
        print("  batches %d  passes/batch %.2f  live lanes/batch %.1f  rounds/batch %.1f  blocks %d" %
              (v["n_batch"], v["n_pass"] / nb, v["n_live_lanes"] / nb, v["n_rounds"] / nb, v["n_blocks"]))
        print("  lanes not kept per batch: chain not settled %.1f; cut after a lane that ended the block %.1f (%.2f cuts); after a lane at its output cap %.1f (%.2f cuts)" %
              (v["c_fix_lost"] / nb, v["c_flag_lost"] / nb, v["c_flag_cuts"] / nb, v["c_cap_lost"] / nb, v["c_cap_cuts"] / nb))
        print("  window-fulls per batch %.2f (cut by the token list %.2f, by the output window %.2f); match tokens per batch %.0f" %
              (v["n_wfull"] / nb, v["n_wf_tok"] / nb, v["n_wf_out"] / nb, v["n_tokens"] / nb))
        print("  long matches per batch: %.1f copied one at a time (%.1f with source bytes before the window), %.0f cycles each; %.1f independent ones, four at a time; other long matches ready when one is copied: %.2f" %
              (v["n_long"] / nb, v["n_long_g"] / nb, v["t_long"] / max(v["n_long"], 1), v["n_indep"] / nb, v["n_long_rdy"] / max(v["n_long"], 1)))
        print("  cycles/batch: count %.0f emit %.0f resolve %.0f flush %.0f window %.0f, all %.0f; header cycles/block %.0f" %
              (v["count_passes"] / nb, v["emit_pass"] / nb, v["resolve"] / nb, v["flush"] / nb, v["window"] / nb,
               v["total"] / nb, v["header"] / max(v["n_blocks"], 1)))
        if v["w_groups"]:
            ng = v["w_groups"]          # counted by every wave of the workgroup
            print("  workgroup mode (waves per chunk: %d): wave-cycles waiting for the wave before: chain %.0f, count %.0f; for the group's outcome %.0f; "
                  "for the water mark %.0f per batch; group calls %d, aborted %d, stretches wave 0 decoded alone %d, chains repaired after the wave before settled %d" %
                  (L.xfg_decode_waves_for(n, 256), v["w_chain"] / nb, v["w_cnt"] / nb, v["w_out"] / nb, v["w_water"] / nb, ng, v["w_abort"], v["w_alone"], v["w_refix"]))
        ctx.close()


if __name__ == "__main__":
    main()
