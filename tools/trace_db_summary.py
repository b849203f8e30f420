"""Per-kernel summary (count, average / min / max duration in us) of a rocprofv3 --kernel-trace results database (.db, sqlite)."""
import sqlite3
import sys


def main(path, only=None):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    sym = [t for t in tabs if "kernel_symbol" in t][0]
    q = ("select s.kernel_name, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3 from %s d join %s s "
         "on d.kernel_id=s.id group by s.kernel_name order by 3*count(*) desc" % (kd, sym))
    for r in c.execute(q):
        if only and only not in r[0]:
            continue
        print("%-70s %6d %9.1f %9.1f %9.1f" % (r[0][:70], r[1], r[2], r[3], r[4]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
