import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(d['value'], 'gemm', r['ms_per_launch'], 'conv', r['second_kernel']['ms_per_launch'])
