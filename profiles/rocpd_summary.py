"""Turns a rocprofv3 rocpd database (…_results.db) into the per-kernel stats table committed under
profiles/ (name, calls, total/avg/min/max duration in us, % of kernel time, VGPR/SGPR/LDS)."""
import sqlite3
import sys


def main(db, out=None):
    con = sqlite3.connect(db)
    rows = con.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(sgpr_count), "
        "max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["%-72s %7s %12s %10s %10s %10s %6s %5s %5s %6s %9s %5s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds", "grid_x", "wg_x")]
    for r in rows:
        name = r[0] if len(r[0]) <= 72 else r[0][:69] + "..."
        lines.append("%-72s %7d %12.1f %10.2f %10.2f %10.2f %6.2f %5d %5d %6d %9d %5d" % (
            name, r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot, r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0, r[10] or 0))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
