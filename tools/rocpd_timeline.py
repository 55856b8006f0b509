"""Timeline of ONE network evaluation out of a rocprofv3 rocpd DB of `bench.py --steps 1 --warmup 1`: every kernel of the
evaluation in start order with its queue, start offset, duration and the idle time before it on its own queue; then the
totals: wall, busy time per queue, time with NO kernel running on any queue, and the main queue's idle time split into
(a) gaps between back-to-back dependent launches and (b) longer waits (on the other queue's events).
usage: rocpd_timeline.py <db> [evaluation index from the end, default 10] [tail launches, default 1]"""
import sqlite3
import sys


def main(path, back=10, tail=1):
    cur = sqlite3.connect(path).cursor()
    fps = [r[0] for r in cur.execute("select start from kernels where name like '%fps_kernel<512, 16%' order by start")]
    i = len(fps) - int(tail) - int(back)
    t0, t1 = fps[i], fps[i + 1]
    rows = list(cur.execute("select name, queue_id, start, end, grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z) "
                            "from kernels where start>=? and start<? order by start", (t0, t1)))
    last_end = {}
    busy = {}
    gaps = {}
    print(f"# evaluation window {(t1 - t0) / 1e3:.1f} us, {len(rows)} kernels")
    print("queue,start_us,dur_us,idle_before_us,workgroups,kernel")
    for name, q, s, e, wgs in rows:
        idle = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        gaps.setdefault(q, []).append(idle)
        busy[q] = busy.get(q, 0) + (e - s)
        last_end[q] = e
        print(f"{q},{(s - t0) / 1e3:.1f},{(e - s) / 1e3:.1f},{idle:.1f},{wgs},\"{name[:70]}\"")
    # union coverage
    ev = sorted((s, e) for _, _, s, e, _ in rows)
    covered, cur_s, cur_e = 0, ev[0][0], ev[0][1]
    for s, e in ev[1:]:
        if s > cur_e:
            covered += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    covered += cur_e - cur_s
    print(f"# wall {(t1 - t0) / 1e3:.1f} us, some kernel running {covered / 1e3:.1f} us, nothing running {(t1 - t0 - covered) / 1e3:.1f} us")
    for q in busy:
        g = gaps[q]
        small = [x for x in g if x <= 8.0]
        big = [x for x in g if x > 8.0]
        print(f"# queue {q}: {len(g)} kernels, busy {busy[q] / 1e3:.1f} us, gaps <= 8 us: {len(small)} totalling {sum(small):.1f} us, "
              f"longer waits: {len(big)} totalling {sum(big):.1f} us")


if __name__ == "__main__":
    main(*sys.argv[1:])
