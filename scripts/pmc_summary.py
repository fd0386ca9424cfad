#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs for the sweep kernel: mean counter value per
launch, with the gfx950 corrections (FETCH_SIZE counts 64 B per 128-B request
on wide coalesced reads, so it is doubled; both counters are in KiB)."""
import csv
import glob
import json
import sys


def mean_counter(pattern, counter, kernel_substr):
    vals = []
    for path in glob.glob(pattern, recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == counter and kernel_substr in row["Kernel_Name"]:
                    vals.append(float(row["Counter_Value"]))
    return vals


def main():
    fetch_dir, write_dir, kernel = sys.argv[1], sys.argv[2], sys.argv[3]
    f = mean_counter(f"{fetch_dir}/**/*counter_collection.csv", "FETCH_SIZE", kernel)
    w = mean_counter(f"{write_dir}/**/*counter_collection.csv", "WRITE_SIZE", kernel)
    f, w = f[1:], w[1:]        # the first launch is the tiptoe (aliased inputs)
    fm, wm = sum(f) / len(f), sum(w) / len(w)
    out = {"kernel": kernel, "launches": len(f),
           "FETCH_SIZE_KiB_raw": fm, "WRITE_SIZE_KiB_raw": wm,
           "fetch_bytes_corrected": fm * 1024 * 2, "write_bytes": wm * 1024,
           "bytes_per_launch": fm * 1024 * 2 + wm * 1024}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
