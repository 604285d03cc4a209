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
                      "where name like '%gemm_kernel%' or name like '%attn_kernel%' group by name, grid_x, grid_y "
                      "order by sum(duration) desc").fetchall()
    print(f"{'calls':>7} {'avg_us':>10} {'grid':>14} {'vgpr':>5} {'agpr':>5} {'lds':>7}  kernel", file=out)
    for name, gx, gy, n, a, vg, ag, lds in rows:
        print(f"{n:7d} {a/1e3:10.2f} {str(gx//256)+'x'+str(gy):>14} {vg:5d} {ag:5d} {lds:7d}  {short(name, 80)}", file=out)


if __name__ == "__main__":
    main()
