#!/usr/bin/env python3
"""Copy a counter summary from gpurun_out/ to profiles/ and stamp it with the commit it was taken at.

    tools/stamp_profile.py gpurun_out/<tag>/pmc_summary.json profiles/<name>.json [--latest]

Run in the authoring container (the GPU box has no .git): `_meta.git_head` = `git rev-parse HEAD`, `_meta.git_dirty` = whether the
kernel sources differ from that commit; `_meta.csrc_sha1` (written on the GPU box by tools/pmc_summary.py) already names the exact
sources.  --latest also refreshes profiles/pmc_traffic_latest.json, the file bench.py reads `roofline.traffic` from."""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, dst = sys.argv[1], sys.argv[2]
    with open(src) as f:
        d = json.load(f)
    meta = d.setdefault("_meta", {})
    meta["git_head"] = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, stdout=subprocess.PIPE).stdout.decode().strip()
    meta["git_dirty"] = bool(subprocess.run(["git", "status", "--porcelain", "dis-pu_amd/csrc", "dis-pu_amd/build.py"], cwd=ROOT,
                                            stdout=subprocess.PIPE).stdout.strip())
    with open(dst, "w") as f:
        json.dump(d, f, indent=1)
    if "--latest" in sys.argv:
        shutil.copyfile(dst, os.path.join(ROOT, "profiles", "pmc_traffic_latest.json"))
    print(dst, meta)


if __name__ == "__main__":
    main()
