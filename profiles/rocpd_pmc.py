"""Per-kernel mean of every PMC counter found in a set of rocprofv3 rocpd databases (one per --pmc pass)."""
import glob
import sqlite3
import sys
from collections import defaultdict


def main(pattern, out=None, kernels=("k_local_bits", "k_local", "k_coarse")):
    acc = defaultdict(lambda: defaultdict(list))
    for db in sorted(glob.glob(pattern)):
        con = sqlite3.connect(db)
        cols = [r[1] for r in con.execute("pragma table_info('counters_collection')")]
        name_col = "kernel_name" if "kernel_name" in cols else "name"
        rows = con.execute("select %s, counter_name, dispatch_id, sum(value) from counters_collection group by %s, counter_name, dispatch_id"
                           % (name_col, name_col)).fetchall()
        for kname, cname, _disp, val in rows:
            for k in kernels:
                # exactly this kernel, demangled or mangled name, plain or a template instantiation ("k_local(" is not "k_local_bits<5, 4>(")
                if any(t in kname for t in (k + "(", k + "<", "%d%sE" % (len(k), k), "%d%sI" % (len(k), k))):
                    acc[k][cname].append(val)
    lines = []
    for k in kernels:
        lines.append("== %s (mean per dispatch)" % k)
        for cname in sorted(acc[k]):
            v = acc[k][cname]
            lines.append("  %-40s %16.1f   (n=%d)" % (cname, sum(v) / len(v), len(v)))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
