"""instruction / MFMA / scratch counts per kernel of a disassembled object: tools/disasm.sh <obj> ; python tools/isa_count.py <obj> [filter]"""
import re, subprocess, sys
s = open(f'/tmp/{sys.argv[1]}.s').read().split('\n')
flt = sys.argv[2] if len(sys.argv) > 2 else ''
starts = [(i, l) for i, l in enumerate(s) if re.match(r'^[0-9a-f]+ <.*>:', l)]
names = [l.split('<', 1)[1].rsplit('>', 1)[0] for _, l in starts]
dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
for k, (i, l) in enumerate(starts):
    if flt not in dem[k]:
        continue
    end = starts[k + 1][0] if k + 1 < len(starts) else len(s)
    body = s[i:end]
    n = sum(1 for x in body if re.match(r'^\s+[a-z_0-9]+ ', x))
    mf = sum('v_mfma' in x for x in body)
    sc = sum('scratch_' in x for x in body)
    print(f"{n:6d} instr {mf:4d} mfma {sc:4d} scratch  {dem[k][:110]}")
