"""Per-kernel launch counts and durations from a rocprofv3 result database (the .db file rocprofv3 --kernel-trace writes),
one line per kernel and grid:  python tools/rocpd_kernels.py results.db [substring ...]"""
import sqlite3
import sys


def main(path, *subs):
    db = sqlite3.connect(path)
    q = ("select name, grid_x, grid_y, workgroup_x, count(*), avg(end-start)/1000.0, min(end-start)/1000.0, max(end-start)/1000.0 "
         "from kernels group by name, grid_x, grid_y order by 6 desc")
    print("%-72s %14s %6s %9s %9s %9s" % ("kernel", "grid(wg)", "calls", "avg_us", "min_us", "max_us"))
    for name, gx, gy, wx, n, avg, lo, hi in db.execute(q):
        if subs and not any(s in name for s in subs):
            continue
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")
        print("%-72s %14s %6d %9.1f %9.1f %9.1f" % (short[:72], "%dx%d" % (gx // max(wx, 1), gy), n, avg, lo, hi))


if __name__ == "__main__":
    main(*sys.argv[1:])
