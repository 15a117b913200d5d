import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{d['value']:.1f} {d['unit']}  {d['ms_per_step']:.1f} ms/step")
tot = 0
ns = d.get('kernel_event_steps') or d['steps']
for k in d['kernels']:
    per = k['total_ms'] / ns; tot += per
    tf = k.get('alt_tflops')
    print(f"{k['name']:44s} n/step {k['launches']/ns:6.1f} avg_us {k['avg_us']:9.1f} ms/step {per:8.2f}  {k['achieved']:9.1f} {k['unit']:8s} frac {k['frac']:.3f}" + (f"  {tf:8.1f} TF" if tf else ""))
print("sum of regions", round(tot, 1))
