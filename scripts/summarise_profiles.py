#!/usr/bin/env python3
"""Condense what scripts/profile_r03.sh (or profile_r02.sh) left under gpurun_out/<dir> into profiles/<round>/:
per run the kernel's rocprofv3 statistics (calls, average / min / max duration), the HBM bytes
per launch from the FETCH_SIZE and WRITE_SIZE passes (KiB units; FETCH_SIZE doubled: on gfx950
it tallies 128-byte requests of wide coalesced reads at 64 bytes), the bench line printed
under the profiler, and the ratio of counted to algorithmic bytes.  Also rewrites
profiles/hbm_traffic.json (what bench.py reports as roofline.traffic)."""
import csv
import glob
import json
import os
import shutil
import sys

KERNEL = {"triple": "ring_step_kernel", "double": "fused2_step_kernel", "fused": "fused_step_kernel", "sweeps": "sweep_pipe_kernel"}


def counter(dirname, name, kernel):
    vals = []
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == name and kernel in row["Kernel_Name"]:
                    vals.append(float(row["Counter_Value"]))
    return vals[1:]            # the first launch of a run differs (tiptoe / spare arrays being made)


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    traffic = {"_note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes; "
                        "FETCH_SIZE doubled per the gfx950 note; KiB units) of `python3 bench.py --no-cpu "
                        "--no-two-sweep --repeats 1 ...`; scripts/profile_r03.sh (or the earlier profile_r02.sh), scripts/summarise_profiles.py"}
    summary = {}
    for log in sorted(glob.glob(os.path.join(src, "trace_*.log"))):
        tag = os.path.basename(log)[len("trace_"):-len(".log")]
        form = tag.split("_")[0]
        kernel = KERNEL[form]
        line = [l for l in open(log) if l.startswith("{")]
        bench = json.loads(line[-1]) if line else None
        stats = None
        for path in glob.glob(os.path.join(src, "trace_" + tag, "**", "*kernel_stats.csv"), recursive=True):
            shutil.copy(path, os.path.join(dst, tag + "_kernel_stats.csv"))
            with open(path) as f:
                for row in csv.DictReader(f):
                    if kernel in row["Name"]:
                        stats = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]),
                                 "min_ns": float(row["MinNs"]), "max_ns": float(row["MaxNs"])}
        fetch = counter(os.path.join(src, "fetch_" + tag), "FETCH_SIZE", kernel)
        write = counter(os.path.join(src, "write_" + tag), "WRITE_SIZE", kernel)
        entry = {"kernel": kernel, "kernel_stats": stats}
        if bench:
            rf = bench["roofline"]
            entry.update({"bench_value": bench["value"], "bench_avg_launch_ms": rf["avg_launch_ms"],
                          "algorithmic_bytes_per_launch": rf["algorithmic_bytes_per_launch"],
                          "frac_of_8TBs_hip_events": rf["frac"], "workload": bench["config"]["workload"]})
            if stats:
                entry["frac_of_8TBs_rocprof_avg"] = rf["algorithmic_bytes_per_launch"] / (stats["avg_ns"] * 1e-9) / 8e12
        if fetch and write:
            fb, wb = 2 * 1024 * sum(fetch) / len(fetch), 1024 * sum(write) / len(write)
            entry.update({"launches_counted": len(fetch), "fetch_bytes_corrected": fb, "write_bytes": wb,
                          "bytes_per_launch": fb + wb})
            if bench:
                entry["traffic_over_algorithmic"] = (fb + wb) / bench["roofline"]["algorithmic_bytes_per_launch"]
            parts = tag.split("_")
            if len(parts) == 3:                       # a default launch shape: what bench.py may quote
                traffic[tag] = {"form": form, "n_harmonics": int(parts[1][1:]), "g_grid": int(parts[2][1:]),
                                "kernel": kernel, "bytes_per_launch": fb + wb,
                                "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, " + dst}
        summary[tag] = entry
    with open(os.path.join(dst, "summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    with open(os.path.join(os.path.dirname(dst.rstrip("/")), "hbm_traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
    for tag, e in summary.items():
        print(tag, "avg_ms", None if not e["kernel_stats"] else round(e["kernel_stats"]["avg_ns"] / 1e6, 4),
              "traffic/alg", round(e.get("traffic_over_algorithmic", 0), 4),
              "frac", round(e.get("frac_of_8TBs_rocprof_avg", 0), 4))


if __name__ == "__main__":
    main()
