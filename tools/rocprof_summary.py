#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite db or *_kernel_stats.csv) as text."""
import glob
import sqlite3
import sys


def main(path, out=None):
    lines = []
    dbs = glob.glob(path + "/**/*.db", recursive=True) if not path.endswith(".db") else [path]
    for db in dbs:
        c = sqlite3.connect(db)
        cols = [r[1] for r in c.execute("pragma table_info(top_kernels)")]
        lines.append("# rocprofv3 --kernel-trace --stats summary (%s)" % db.split("/")[-1])
        lines.append(",".join(cols))
        for row in c.execute("select * from top_kernels"):
            lines.append(",".join(str(x) for x in row))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
