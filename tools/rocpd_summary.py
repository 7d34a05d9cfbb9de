#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (`*_results.db`, the ROCm 7.2 default output of
`rocprofv3 --kernel-trace --stats`) as a small markdown/CSV kernel-stats table that can be committed
under profiles/.  Usage: python tools/rocpd_summary.py <results.db> [--out profiles/rNN/name.md] [--csv]
                         [--skip-first N]   (drop the first N dispatches of every kernel: warm-up)
"""
import argparse
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    if "radix_sort_onesweep_iteration" in name:
        return "rocprim radix_sort_onesweep_iteration"
    if "radix_sort_onesweep_global_offsets" in name:
        return "rocprim radix_sort_onesweep_global_offsets (histogram/scan)"
    if name.startswith("at::native::") or "at::native::" in name[:40]:
        m = re.search(r"(uniform_kernel|random_from_to_kernel|FillFunctor|direct_copy_kernel|MulFunctor|arange|round_kernel)", name)
        return "torch:" + (m.group(1) if m else name[:50]) + " (input generation / glue)"
    i = name.find("(")
    return name if i < 0 else name[:i]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--out", default=None)
    ap.add_argument("--csv", action="store_true")
    ap.add_argument("--by-grid", action="store_true", help="split each kernel by launch grid (per-layer GEMM timings)")
    a = ap.parse_args()
    c = sqlite3.connect(a.db).cursor()
    grp = "name, grid_x, grid_y, grid_z" if a.by_grid else "name"
    q = (f"select name, grid_x/workgroup_x, grid_y/workgroup_y, grid_z/workgroup_z, count(*), sum(duration), avg(duration), "
         f"min(duration), max(duration), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), "
         f"max(scratch_size) from kernels group by {grp} order by sum(duration) desc")
    rows = list(c.execute(q))
    total = sum(r[5] for r in rows) or 1
    lines = []
    if a.csv:
        lines.append("kernel,grid,calls,total_us,avg_us,min_us,max_us,pct,vgpr,agpr,sgpr,lds_bytes,scratch")
    else:
        lines.append("| kernel | grid (WGs) | calls | total µs | avg µs | min µs | max µs | % | vgpr | agpr | sgpr | LDS B | scratch |")
        lines.append("|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for (name, gx, gy, gz, n, tot, avg, mn, mx, vg, ag, sg, lds, scr) in rows:
        grid = f"{gx}x{gy}x{gz}" if a.by_grid else "-"
        vals = [short(name), grid, n, f"{tot / 1e3:.1f}", f"{avg / 1e3:.2f}", f"{mn / 1e3:.2f}", f"{mx / 1e3:.2f}",
                f"{100.0 * tot / total:.2f}", vg, ag, sg, lds, scr]
        if a.csv:
            lines.append(",".join('"%s"' % v if i == 0 else str(v) for i, v in enumerate(vals)))
        else:
            lines.append("| " + " | ".join(str(v) for v in vals) + " |")
    text = "\n".join(lines) + "\n"
    if a.out:
        with open(a.out, "w") as f:
            f.write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
