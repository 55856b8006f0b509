"""Per-network-evaluation kernel breakdown of the TIMED sample in a rocprofv3 rocpd DB of `bench.py --steps 1
--warmup 1`: the level-0 FPS launches mark the evaluations -- 33 for the warm-up sample (3 eager steps around the graph
capture + 30 replays), 30 for the timed sample, then `tail` more from bench.py's roofline section (conv_roofline runs one
eager evaluation to capture the launch it times). Window = the 30 launches before the tail."""
import sqlite3
import sys


def main(path, out=None, top=45, tail=1, T=30, marker="fps_kernel<512, 16"):
    """marker: a kernel launched exactly once per network evaluation (PVDS at 8192 points: the level-0 FPS instance;
    PVDL at 50000 points: `fps_grid_kernel`)"""
    cur = sqlite3.connect(path).cursor()
    fps = [r for r in cur.execute("select start,end from kernels where name like ? order by start", (f"%{marker}%",))]
    tail, T, top = int(tail), int(T), int(top)
    n = len(fps) - tail - T
    t0 = fps[n][0]
    t1 = fps[n + T - 1][0] + (fps[n + T - 1][0] - fps[n + T - 2][0])  # (the roofline launches follow the last evaluation)
    rows = list(cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels "
                            "where start>=? and start<? group by name order by 3 desc", (t0, t1)))
    tot = sum(r[2] for r in rows)
    ev = T
    lines = [f"# window {(t1 - t0) / 1e6:.1f} ms = {ev} network evaluations, {(t1 - t0) / 1e6 / ev:.2f} ms/eval wall, "
             f"{tot / ev:.2f} ms/eval kernel time", "pct,calls_per_eval,avg_us,ms_per_eval,kernel"]
    for r in rows[:top]:
        lines.append(f"{r[2] / tot * 100:.2f},{r[1] / ev:.1f},{r[3]:.1f},{r[2] / ev:.3f},\"{r[0][:150]}\"")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, *sys.argv[3:])
