#!/usr/bin/env python3
"""Table of VGPRs / spills / scratch / LDS per kernel from the compiler remarks build_hip.py keeps
next to each object (csrc/build*/NAME.remarks).  `python tools/kernel_resources.py [build_dir] [filter]`."""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    d = sys.argv[1] if len(sys.argv) > 1 and os.path.isdir(sys.argv[1]) else os.path.join(ROOT, "zest-nerf_amd", "csrc", "build")
    flt = [a for a in sys.argv[1:] if not os.path.isdir(a)]
    rows = []
    for f in sorted(glob.glob(os.path.join(d, "*.remarks"))):
        name = os.path.basename(f)[:-8]
        if flt and not any(x in name for x in flt):
            continue
        cur = None
        for line in open(f):
            m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
            if not m:
                continue
            t = m.group(1).strip()
            if t.startswith("Function Name:"):
                cur = {"obj": name, "fn": t.split(":", 1)[1].strip()}
                rows.append(cur)
            elif cur is not None and ":" in t:
                k, v = t.split(":", 1)
                cur[k.strip()] = v.strip()
    print("%-18s %-44s %5s %5s %6s %6s %7s %8s" % ("object", "kernel", "VGPR", "AGPR", "vspill", "sspill", "scratch", "LDS"))
    for r in rows:
        fn = r["fn"]
        fn = fn if len(fn) <= 44 else fn[:20] + ".." + fn[-22:]
        print("%-18s %-44s %5s %5s %6s %6s %7s %8s" % (r["obj"], fn, r.get("VGPRs", "?"), r.get("AGPRs", "?"),
                                                     r.get("VGPRs Spill", "?"), r.get("SGPRs Spill", "?"),
                                                     r.get("ScratchSize [bytes/lane]", "?"),
                                                     r.get("LDS Size [bytes/block]", "?")))


if __name__ == "__main__":
    main()
