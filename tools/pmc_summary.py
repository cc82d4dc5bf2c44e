#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; collected separately as MI355X_MICROARCH.md
prescribes) of `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline` into per-kernel HBM bytes per launch.

    python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>

Units: FETCH_SIZE / WRITE_SIZE are KB per dispatch.  gfx950 correction: FETCH_SIZE reports half the bytes of a
wide (16 B/lane) coalesced streaming read, so it is doubled; WRITE_SIZE is exact for streaming stores."""
import collections
import csv
import json
import sys

NAMES = ["walk_tails_kernel", "fused_pass2_tall_kernel", "fused_pass2_kernel", "fused_tails_kernel", "mfma_tails_kernel", "stream_tails_kernel", "carry_block_kernel", "carry_pair_kernel", "xscan_rows_kernel",
         "strided_pass_kernel", "generic_pass_kernel", "untiled_scan_kernel", "mx_pass1s_kernel", "mx_pass2s_kernel", "mx_pass2p_kernel", "mx_chain_kernel", "mx_apply_kernel"]


def key(row):
    for n in NAMES:
        if n in row["Kernel_Name"]:
            return n
    return None


def agg(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = key(r)
        if k:
            d[k].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in d.items()}


def main():
    fetch, write, out = sys.argv[1:4]
    fa, wa = agg(fetch), agg(write)
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    # kernel_sources_sha16: digest of recfilter_amd/csrc at collection time -- bench.py reports these bytes only while
    # the running sources have the same digest; git_head is stamped afterwards, where .git exists (tools/stamp_head.py)
    res = {"note": __doc__.strip().splitlines()[0], "kernel_sources_sha16": bench.kernel_sources_sha16(), "git_head": None,
           "kernels": {}}
    for k in sorted(fa):
        fs, n = fa[k]
        ws, _ = wa.get(k, (0.0, 0))
        res["kernels"][k] = {"launches": n, "FETCH_SIZE_KB": round(fs, 1), "WRITE_SIZE_KB": round(ws, 1),
                             "fetch_bytes_corrected": int(2 * fs * 1024), "write_bytes": int(ws * 1024)}
        print(f"{k:24s} n={n:3d} fetch x2 = {2 * fs * 1024 / 1e6:9.1f} MB   write = {ws * 1024 / 1e6:9.1f} MB")
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
