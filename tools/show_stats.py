import json,sys
for f in sys.argv[1:]:
    d=json.load(open(f)); print(f, {k:v for k,v in d["args"].items()})
    for k,v in d["summary"].items():
        if "mean_A" in v: print(f"  {k:18s} A {v['mean_A']:12.5g} ±{v['sd_A']:9.3g}  B {v['mean_B']:12.5g} ±{v['sd_B']:9.3g}  rel {v['rel_diff']:+.4f}")
