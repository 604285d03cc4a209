"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output) of `bench.py` into per-kernel HBM traffic per
launch.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes of wide coalesced reads ->
doubled; both counters are in KiB.   usage: python tools/pmc_traffic.py <fetch_csv> <write_csv> <out.json>"""
import collections, csv, json, re, sys

def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]

def load(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc

f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(f) | set(w)):
    if not ("gemm" in k or "attn" in k or "ln_modulate" in k):
        continue
    fk = sum(f.get(k, [0])) / max(len(f.get(k, [])), 1)
    wk = sum(w.get(k, [0])) / max(len(w.get(k, [])), 1)
    out[k] = {"launches_sampled": len(f.get(k, [])), "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk,
              "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    print(f"{v['hbm_bytes_per_launch']/1e6:10.1f} MB/launch  fetch {v['FETCH_SIZE_KiB']:12.0f} KiB  write {v['WRITE_SIZE_KiB']:12.0f} KiB  n={v['launches_sampled']:5d}  {k}")
