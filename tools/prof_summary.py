"""Summarise a rocprofv3 --kernel-trace database (results.db) per kernel: tools/prof_summary.py DB STEPS [csv_out]."""
import sqlite3
import sys

db, steps = sys.argv[1], int(sys.argv[2])
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                 "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"kernel time total {tot / 1e3:.2f} ms over {steps} steps = {tot / steps / 1e3:.3f} ms/step")
lines = ["name,calls,total_us,avg_us,min_us,max_us,percent"]
for r in rows:
    lines.append(f"\"{r[0]}\",{r[1]},{r[2]:.1f},{r[3]:.2f},{r[4]:.2f},{r[5]:.2f},{100 * r[2] / tot:.2f}")
for r in rows[:26]:
    print(f"{r[0][:100]:100s} n={r[1]:5d} {r[2] / steps:9.1f} us/step {100 * r[2] / tot:5.1f}% avg={r[3]:8.1f}")
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write("\n".join(lines) + "\n")
