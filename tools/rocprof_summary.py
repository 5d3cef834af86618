"""Turns a rocprofv3 results .db (rocpd SQLite, ROCm 7.2 default output) into the per-kernel stats
table `rocprofv3 --kernel-trace --stats` prints as CSV: name, calls, total us, average us (the rocpd top_kernels view is in microseconds), %."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    lines = ["Name,Calls,TotalDurationUs,AverageUs,Percentage"]
    for r in rows:
        lines.append('"%s",%d,%.0f,%.0f,%.4f' % (r[0], r[1], r[2], r[3], r[4]))
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    else:
        sys.stdout.write(txt)


if __name__ == "__main__":
    main(*sys.argv[1:])
