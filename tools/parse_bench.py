import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("ms/step %.3f  value %.0f Mobs/s  sweep %.0f GB/s frac %.3f" % (d["ms_per_step"], d["value"], d["roofline"]["achieved"], d["roofline"]["frac"]))
print({k: round(v,3) for k,v in d["stage_ms_per_step_rank0"].items()})
