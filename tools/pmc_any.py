"""Average of arbitrary rocprofv3 --pmc counters per (kernel, shape) tag, per launch.

    rocprofv3 --kernel-trace --pmc <counters...> --output-format csv -d DIR -o NAME -- <command>
    python tools/pmc_any.py DIR/**/NAME_counter_collection.csv [--mode=b8] [substring filter of the kernel tag ...]

The tags are those of tools/pmc_traffic.py (kernel name as rocprofv3 prints it + the shape, split by launch order inside one DiT
block); kernels without an entry there keep their bare name."""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_traffic  # noqa: E402
from pmc_traffic import short, tag_of  # noqa: E402


def main():
    cfile, filt = sys.argv[1], sys.argv[2:]
    MODE = "ddim"
    if filt and filt[0].startswith("--mode="):
        MODE, filt = filt[0][7:], filt[1:]
    rows_all = list(csv.DictReader(open(cfile)))
    pmc_traffic.FOLDED = any(short(r["Kernel_Name"]).startswith("gemm144l_dma_kernel<1, 6>") for r in rows_all)
    rows, meta = collections.defaultdict(dict), {}
    for r in rows_all:
        d = int(r["Dispatch_Id"])
        rows[d][r["Counter_Name"]] = float(r["Counter_Value"])
        meta[d] = (r["Kernel_Name"], int(r.get("Grid_Size", 0) or 0))
        if r.get("End_Timestamp"):
            rows[d]["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    seen = collections.Counter()
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(rows):
        tag = tag_of(meta[d][0], meta[d][1], seen, MODE)
        for c, v in rows[d].items():
            agg[tag][c].append(v)
    names = sorted({c for t in agg.values() for c in t if c != "us"})
    print("%-64s %6s %9s " % ("kernel (shape)", "n", "us") + " ".join("%22s" % n[:22] for n in names))
    for tag in sorted(agg):
        if filt and not any(f in tag for f in filt):
            continue
        a = agg[tag]
        n = max(len(v) for v in a.values())
        us = sum(a["us"]) / len(a["us"]) if a.get("us") else float("nan")
        print("%-64s %6d %9.1f " % (tag[:64], n, us) + " ".join("%22.1f" % (sum(a[c]) / len(a[c])) if a.get(c) else "%22s" % "-" for c in names))


if __name__ == "__main__":
    main()
