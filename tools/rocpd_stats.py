"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) as a per-kernel table (stdout / CSV)."""
import sqlite3
import sys


def main(path, out=None, top=60):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    tot = sum(r[2] for r in rows)
    lines = ["pct,calls,avg_us,total_ms,kernel"]
    for name, calls, dur, avg, pct in rows[:top]:
        lines.append(f"{dur / tot * 100:.2f},{calls},{avg:.1f},{dur / 1e3:.2f},\"{name[:160]}\"")  # view is in microseconds
    lines.append(f"# total kernel time {tot / 1e3:.1f} ms over {sum(r[1] for r in rows)} dispatches, {len(rows)} distinct kernels")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
