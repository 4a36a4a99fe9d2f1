import json,sys
p=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("ms/step", p["ms_per_step"], "resident", p["resident_batch"]["ms_per_step"], "conv_path", round(p["conv_path"]["frac"],3), "roofline", p["roofline"]["kernel"], round(p["roofline"]["frac"],3))
for k,v in p["roofline_kernels"].items():
    print("==",k,"frac %.3f"%v["frac"],"n",v["launches_per_step"],"us/step %.1f"%v["us_per_step"])
    for s in v["shapes"]:
        if "s2" in s["shape"]: print("    ", s["shape"], round(s["us"],1), round(s["TFLOPs"],1))
