#!/usr/bin/env python
"""Summarises `rocprofv3 --pmc ... --output-format csv` passes (one directory per pass, as collected by
the command line recorded at the top of the output) into per-kernel averages.
usage: pmc_summary.py out.json dir1 dir2 ...   (also prints a text table)"""
import collections
import csv
import glob
import json
import sys


def main(out, dirs):
    table = collections.defaultdict(dict)
    for d in dirs:
        for f in glob.glob(d + "/*/*_counter_collection.csv"):
            acc = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                acc[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))  # templated kernels: "void k_x<8>" -> "k_x"
            for k, cs in acc.items():
                for c, v in cs.items():
                    table[k][c] = dict(avg=sum(v) / len(v), launches=len(v))
    res = {}
    print("%-28s %-24s %10s %16s" % ("kernel", "counter", "launches", "avg per launch"))
    for k in sorted(table, key=lambda k: -table[k].get("FETCH_SIZE", {}).get("avg", 0)):
        if not k.startswith("k_"):
            continue
        res[k] = {c: v["avg"] for c, v in table[k].items()}
        for c, v in sorted(table[k].items()):
            print("%-28s %-24s %10d %16.1f" % (k, c, v["launches"], v["avg"]))
        f, w = res[k].get("FETCH_SIZE"), res[k].get("WRITE_SIZE")
        if f is not None and w is not None:
            # MI355X_MICROARCH.md "HBM": on gfx950 FETCH_SIZE (KiB) tallies 128-B fabric read requests as 64 B
            # -> double it; WRITE_SIZE (KiB) calibrated ~1.0 on k_primal (known 3n*8 bytes written)
            res[k]["traffic_bytes_corrected"] = (2.0 * f + w) * 1024.0
            print("%-28s %-24s %10s %16.1f" % (k, "traffic MB (2*F+W)", "", res[k]["traffic_bytes_corrected"] / 1e6))
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
