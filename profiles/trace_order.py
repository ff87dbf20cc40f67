"""rocprofv3 kernel trace (rocpd sqlite db) -> the launches of the LAST lm_pipeline_run (from its k_coarse on) in order."""
import glob, sqlite3, sys
f = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
db = sqlite3.connect(f)
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = [dict(zip(cols, r)) for r in db.execute("select * from kernels order by start")]
names = [r["name"].split("(")[0].replace("lm::", "").replace("void ", "") for r in rows]
first = sys.argv[2] if len(sys.argv) > 2 else "k_coarse"
last_c = max(i for i, n in enumerate(names) if n.startswith(first))
t0 = rows[last_c]["start"]; prev = t0
for r, n in zip(rows[last_c:], names[last_c:]):
    print("%-22s start %8.1f us  dur %7.1f us  gap %6.1f us  grid %s wg %s lds %s" % (n[:22], (r["start"] - t0) / 1e3, (r["end"] - r["start"]) / 1e3,
          (r["start"] - prev) / 1e3, r.get("grid_x"), r.get("workgroup_x"), r.get("lds_size")))
    prev = r["end"]
