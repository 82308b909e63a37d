#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM bytes per launch.

    python tools/pmc_traffic.py WORKLOAD[@MODE] fetch_counter_collection.csv write_counter_collection.csv [out.json]

Appends/updates profiles/pmc_traffic.json (or out.json).  Follows MI355X_MICROARCH.md (HBM section):
FETCH_SIZE and WRITE_SIZE are in KiB and come from separate passes (they do not fit one);
on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, so it is doubled;
WRITE_SIZE is exact for 16-byte-per-lane stores.
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mean_counter(path, counter, kernel="fused_blocks_kernel"):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
         if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]]
    return sum(v) / len(v), len(v)


def main():
    wl, fpath, wpath = sys.argv[1:4]
    wl = wl[:-5] if wl.endswith("@bf16") else wl          # the bf16 line keeps the bare workload name
    fetch, nf = mean_counter(fpath, "FETCH_SIZE")
    write, nw = mean_counter(wpath, "WRITE_SIZE")
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "profiles", "pmc_traffic.json")
    data = json.load(open(out)) if os.path.exists(out) else {}
    data[wl] = {"fetch_size_kib": fetch, "write_size_kib": write, "launches": [nf, nw],
                "hbm_bytes_per_launch": int((2 * fetch + write) * 1024),
                "sources": [os.path.basename(fpath), os.path.basename(wpath)]}
    json.dump(data, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(data[wl]))


if __name__ == "__main__":
    main()
