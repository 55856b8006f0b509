"""Per-launch-shape breakdown of one kernel family in the timed window of a bench.py rocpd DB:
python tools/rocpd_shapes.py <db> <name substring>  (grid is in threads, as rocprofv3 records it)"""
import sqlite3
import sys


def main(path, pat):
    cur = sqlite3.connect(path).cursor()
    fps = [r for r in cur.execute("select start,end from kernels where name like '%fps_kernel<512, 16%' order by start")]
    n = len(fps) // 2
    t0, t1 = fps[n][0], fps[-1][0] + (fps[-1][0] - fps[-2][0])
    ev = len(fps) - n
    rows = cur.execute("select substr(name,1,60), grid_x, grid_y, grid_z, stream_id, count(*), avg(end-start)/1e3, sum(end-start)/1e6 "
                       "from kernels where start>=? and start<? and name like ? group by 1,2,3,4,5 order by 8 desc",
                       (t0, t1, f"%{pat}%"))
    for r in rows:
        print(f"{r[0]:60s} grid {r[1]}x{r[2]}x{r[3]} stream {r[4]} calls/eval {r[5] / ev:.1f} avg {r[6]:.1f} us  {r[7] / ev:.3f} ms/eval")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
