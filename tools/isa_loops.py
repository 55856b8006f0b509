"""loop structure of one kernel in /tmp/<obj>.s: backward branches (loops) with body length, MFMA count and VALU count inside;
usage: python tools/isa_loops.py <obj> '<demangled-name substring>'"""
import re, subprocess, sys
s = open(f'/tmp/{sys.argv[1]}.s').read().split('\n')
starts = [(i, l) for i, l in enumerate(s) if re.match(r'^[0-9a-f]+ <.*>:', l)]
names = [l.split('<', 1)[1].rsplit('>', 1)[0] for _, l in starts]
dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
for k, (i, l) in enumerate(starts):
    if sys.argv[2] not in dem[k]:
        continue
    end = starts[k + 1][0] if k + 1 < len(starts) else len(s)
    ins = []  # (addr, text)
    for x in s[i + 1:end]:
        m = re.match(r'^\s+(\S.*?)\s+//\s*([0-9A-Fa-f]+):', x)
        if m:
            ins.append((int(m.group(2), 16), m.group(1)))
    addr2idx = {a: j for j, (a, _) in enumerate(ins)}
    print(dem[k][:120], len(ins), "instructions")
    for j, (a, t) in enumerate(ins):
        m = re.match(r'(s_cbranch_\w+|s_branch)\s+(\d+)', t)
        if m:
            off = int(m.group(2))
            if off >= 32768:
                off -= 65536
            tgt = a + 4 + off * 4
            if tgt <= a and tgt in addr2idx:
                body = ins[addr2idx[tgt]:j + 1]
                mf = sum('v_mfma' in b for _, b in body)
                va = sum(b.startswith('v_') and 'v_mfma' not in b for _, b in body)
                ds = sum(b.startswith('ds_') for _, b in body)
                vm = sum(b.startswith(('buffer_', 'global_', 'flat_')) for _, b in body)
                print(f"  loop [{addr2idx[tgt]}..{j}] {len(body)} instr: {mf} mfma, {va} valu, {ds} ds, {vm} vmem")
    break
