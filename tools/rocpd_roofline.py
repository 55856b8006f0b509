"""The bench's roofline launches in a rocprofv3 rocpd DB of `bench.py`: conv_roofline() runs the dominant kernel
3 (warm-up) + 10 (timed) times AFTER the sampler, on dense random input; these are the last 10 launches of that
kernel name. Prints/writes their durations next to the in-sampler launches of the same kernel (whose operands are
sparse, so whole stages are skipped and they run faster)."""
import sqlite3
import sys


def main(path, out=None, pat="conv3d_k3_split_kernel<16, true, 2, true, true>"):
    cur = sqlite3.connect(path).cursor()
    rows = [r[0] / 1e3 for r in cur.execute("select end-start from kernels where name like ? order by start", (f"%{pat}%",))]
    timed, rest = rows[-10:], rows[:-13]
    lines = ["set,launches,avg_us,min_us,max_us",
             f"bench.py conv_roofline timed launches (dense random operand),{len(timed)},{sum(timed) / len(timed):.1f},"
             f"{min(timed):.1f},{max(timed):.1f}",
             f"same kernel inside the sampler (sparse operand: zero stages skipped),{len(rest)},{sum(rest) / max(1, len(rest)):.1f},"
             f"{min(rest):.1f},{max(rest):.1f}"]
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(f"# kernel: {pat}\n" + txt + "\n")


if __name__ == "__main__":
    main(*sys.argv[1:])
