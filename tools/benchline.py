import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print("value %.3e ms/step %.4f kernel_ms %.4f frac %.3f rounds %d"%(d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"], d["config"]["rounds_per_step"]))
