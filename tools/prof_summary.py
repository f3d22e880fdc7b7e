"""Summarise a rocprofv3 rocpd database (or directory of them) into a per-kernel stats CSV
(name, calls, total_us, avg_us, min_us, max_us, pct, vgpr, agpr, lds) — the `--stats` view, committed under profiles/."""
import glob
import os
import sqlite3
import re
import sys


def summarise(db_path):
    db = sqlite3.connect(db_path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name order by sum(duration) desc"
    ).fetchall()
    tot = sum(r[2] for r in rows) or 1
    out = ["name,calls,total_us,avg_us,min_us,max_us,pct,vgpr,agpr,lds_bytes"]
    for r in rows:
        r = (re.sub(r"\s+", " ", r[0])[:150],) + tuple(r[1:])
        out.append('"%s",%d,%.2f,%.3f,%.3f,%.3f,%.2f,%s,%s,%s' % (
            r[0], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot, r[6], r[7], r[8]))
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    src, dst = sys.argv[1], sys.argv[2]
    dbs = [src] if src.endswith(".db") else sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))
    with open(dst, "w") as f:
        for d in dbs:
            f.write(summarise(d))
    print(open(dst).read())
