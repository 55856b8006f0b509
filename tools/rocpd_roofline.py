"""The bench's roofline launches in a rocprofv3 rocpd DB of `bench.py`: gemm_roofline() / conv_roofline() run their
kernel 3 (warm-up) + 10 (timed) times AFTER the sampler, on dense random input; these are the last 10 launches of that
kernel name. Prints/writes their durations next to the in-sampler launches of the same kernel name (for the GEMM the
name covers a second, small layer too; for the convolution the in-sampler operands are sparse, so stages are skipped)."""
import sqlite3
import sys


def main(path, out=None, pat="conv3d_k3_split_kernel<16, true, 2, true, true>"):
    cur = sqlite3.connect(path).cursor()
    rows = [r[0] / 1e3 for r in cur.execute("select end-start from kernels where name like ? order by start", (f"%{pat}%",))]
    timed, rest = rows[-10:], rows[:-13]
    lines = ["set,launches,avg_us,min_us,max_us",
             f"bench.py roofline timed launches (dense random operand),{len(timed)},{sum(timed) / len(timed):.1f},"
             f"{min(timed):.1f},{max(timed):.1f}",
             f"same kernel name inside the sampler,{len(rest)},{sum(rest) / max(1, len(rest)):.1f},"
             f"{min(rest or [0]):.1f},{max(rest or [0]):.1f}"]
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(f"# kernel: {pat}\n" + txt + "\n")


if __name__ == "__main__":
    main(*sys.argv[1:])
