#!/usr/bin/env python3
"""Stamp a PMC summary copied from the GPU box with the commit whose kernel sources it was collected on.

    python tools/stamp_head.py profiles/r3/pmc_traffic.json

The GPU box has no .git, so tools/pmc_summary.py records the digest of recfilter_amd/csrc instead
(bench.kernel_sources_sha16).  Here, where .git exists: if the working tree's sources have that digest the summary
is stamped with HEAD (plus "+dirty" when the sources differ from HEAD's); otherwise it is left unstamped."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    path = sys.argv[1]
    doc = json.load(open(path))
    running = bench.kernel_sources_sha16()
    if doc.get("kernel_sources_sha16") != running:
        print(f"{path}: collected on sources {doc.get('kernel_sources_sha16')}, tree has {running}: not stamped")
        return 1
    head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], text=True).strip()
    dirty = subprocess.run(["git", "-C", ROOT, "diff", "--quiet", "HEAD", "--", "recfilter_amd/csrc"]).returncode != 0
    doc["git_head"] = head + ("+dirty" if dirty else "")
    json.dump(doc, open(path, "w"), indent=1)
    print(f"{path}: git_head = {doc['git_head']}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
