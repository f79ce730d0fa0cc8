"""Print the per-kernel breakdown of bench.py JSON lines (files given on the command line)."""
import json
import sys

for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "value", round(d["value"]), d["unit"], "ms/step", round(d["ms_per_step"], 3))
    for k, v in d.get("extra", {}).get("gemm_kernels", {}).items():
        print("   %-52s n=%4d %8.1f us %6.0f TF" % (k, v["launches"], v["avg_us"], v["tflops"]))
    ex = d.get("extra", {})
    if "sampler" in ex:
        print("   sampler", ex["sampler"]["seconds"], "s", ex["sampler"]["dominant_kernel"])
    if "fk_joints" in ex:
        print("   fk", ex["fk_joints"])
    for k in ("lbs_full_fwd", "lbs_full_fwd_bwd"):
        if k in ex:
            print("   " + k, ex[k])
