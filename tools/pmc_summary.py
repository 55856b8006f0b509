"""Summarise the rocprofv3 --pmc passes of tools/pmc_run.sh for one kernel into a small CSV."""
import collections
import csv
import glob
import sys


def main(root, pat, out):
    agg = collections.defaultdict(list)
    dur = []
    for p in sorted(glob.glob(root + "/p*/pmc_counter_collection.csv")):
        for r in csv.DictReader(open(p)):
            if pat in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for p in sorted(glob.glob(root + "/p*/pmc_kernel_trace.csv")):
        for r in csv.DictReader(open(p)):
            if pat in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    lines = ["counter,mean_per_launch,launches"]
    for k in sorted(agg):
        lines.append(f"{k},{sum(agg[k]) / len(agg[k]):.1f},{len(agg[k])}")
    lines.append(f"duration_us(under_pmc),{sum(dur) / max(1, len(dur)):.1f},{len(dur)}")
    txt = "\n".join(lines)
    print(txt)
    open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
