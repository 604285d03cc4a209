"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel totals + per-(kernel, grid) breakdown.
usage: python tools/rocprof_summary.py <results.db> [out.txt]"""
import re
import sqlite3
import sys


def short(name: str, n: int = 110) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= n else name[:n - 3] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                      "group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace summary of {sys.argv[1]}  (durations in us; total GPU kernel time {tot/1e3:.1f} us)", file=out)
    print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}  kernel", file=out)
    for name, n, s, a, mn, mx in rows[:40]:
        print(f"{n:7d} {s/1e3:12.1f} {a/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.2f}  {short(name)}", file=out)
    print("\n# per (kernel, grid) breakdown of the primx kernels", file=out)
    rows = db.execute("select name, grid_x, grid_y, count(*), avg(duration), vgpr_count, accum_vgpr_count, lds_size from kernels "
                      "where name like '%gemm%' or name like '%attn_kernel%' or name like '%conv%' or name like '%groupnorm%' group by name, grid_x, grid_y "
                      "order by sum(duration) desc").fetchall()
    print(f"{'calls':>7} {'avg_us':>10} {'grid':>14} {'vgpr':>5} {'agpr':>5} {'lds':>7}  kernel", file=out)
    for name, gx, gy, n, a, vg, ag, lds in rows:
        print(f"{n:7d} {a/1e3:10.2f} {str(gx//256)+'x'+str(gy):>14} {vg:5d} {ag:5d} {lds:7d}  {short(name, 80)}", file=out)

    # ---- idle time between consecutive kernels (same queue, dependent launches): where the step time that is in no kernel goes
    try:
        ks = db.execute('select name, start, "end" from kernels order by start').fetchall()
    except sqlite3.Error as e:  # older rocpd schemas
        print(f"\n# (no gap analysis: {e})", file=out)
        return
    if len(ks) < 2:
        return
    gaps = [(ks[i + 1][1] - ks[i][2], ks[i][0], ks[i + 1][0]) for i in range(len(ks) - 1)]
    small = [g for g in gaps if 0 <= g[0] < 50_000]           # < 50 us: back-to-back launches (larger = host-side pauses)
    span = ks[-1][2] - ks[0][1]
    print(f"\n# gaps between consecutive kernels: {len(gaps)} boundaries over {span/1e3:.0f} us of trace; "
          f"{len(small)} back-to-back (< 50 us) with {sum(g[0] for g in small)/1e3:.1f} us idle in total "
          f"(mean {sum(g[0] for g in small)/max(1, len(small))/1e3:.2f} us), {sum(1 for g in gaps if g[0] < 0)} overlapping", file=out)
    edges = [0, 1000, 2000, 3000, 4000, 6000, 10000, 50000]
    hist = [sum(1 for g in small if lo <= g[0] < hi) for lo, hi in zip(edges[:-1], edges[1:])]
    print("# histogram (us): " + "  ".join(f"[{lo/1e3:g},{hi/1e3:g}) {n}" for lo, hi, n in zip(edges[:-1], edges[1:], hist)), file=out)
    by_pred = {}
    for g, pred, _ in small:
        k = short(pred, 60)
        t = by_pred.setdefault(k, [0, 0])
        t[0] += 1
        t[1] += g
    print("# idle after each kernel (back-to-back boundaries): count, mean us", file=out)
    for k, (n, t) in sorted(by_pred.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"{n:7d} {t/n/1e3:8.2f}  {k}", file=out)


if __name__ == "__main__":
    main()
