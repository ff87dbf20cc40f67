"""Dumps a window of a rocprofv3 rocpd kernel trace as text: start (us, relative), duration, queue, kernel name.
Usage: rocpd_timeline.py <db> [skip_fraction=0.6] [count=120]"""
import sqlite3, sys
db = sys.argv[1]; frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6; cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 120
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = con.execute("select start, end, %s, name from kernels order by start" % q).fetchall()
i0 = int(len(rows) * frac)
t0 = rows[i0][0]
for s, e, qq, n in rows[i0:i0 + cnt]:
    print("%10.1f %8.1f  q%-4s %s" % ((s - t0) / 1e3, (e - s) / 1e3, qq, n[:60]))
