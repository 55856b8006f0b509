"""Per-network-evaluation kernel breakdown of the TIMED sample in a rocprofv3 rocpd DB of bench.py
(window = from the 31st to the 60th level-0 FPS launch, i.e. the second sample's 30 evaluations)."""
import sqlite3
import sys


def main(path, out=None, top=45):
    cur = sqlite3.connect(path).cursor()
    fps = [r for r in cur.execute("select start,end from kernels where name like '%fps_kernel<512, 16%' order by start")]
    n = len(fps) // 2
    t0 = fps[n][0]
    t1 = fps[-1][0] + (fps[-1][0] - fps[-2][0])
    rows = list(cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels "
                            "where start>=? and start<? group by name order by 3 desc", (t0, t1)))
    tot = sum(r[2] for r in rows)
    ev = len(fps) - n
    lines = [f"# window {(t1 - t0) / 1e6:.1f} ms = {ev} network evaluations, {(t1 - t0) / 1e6 / ev:.2f} ms/eval wall, "
             f"{tot / ev:.2f} ms/eval kernel time", "pct,calls_per_eval,avg_us,ms_per_eval,kernel"]
    for r in rows[:top]:
        lines.append(f"{r[2] / tot * 100:.2f},{r[1] / ev:.1f},{r[3]:.1f},{r[2] / ev:.3f},\"{r[0][:150]}\"")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
