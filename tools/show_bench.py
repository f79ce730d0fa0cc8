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
        r = ex["sampler"].get("roofline", {})
        print("   sampler %.4f s  %s %.1f us  frac %.3f" % (ex["sampler"]["seconds"], r.get("kernel", "?"), r.get("avg_launch_us", 0.0), r.get("frac", 0.0)))
    for k in ("train_step_bf16x3_mode", "train_step_fp32_mode"):
        if k in ex:
            print("   %-28s %.3f ms / step" % (k, ex[k]["ms_per_step"]))
    if "sampler_bf16x3_mode" in ex:
        print("   sampler_bf16x3_mode          %.3f s" % ex["sampler_bf16x3_mode"]["seconds_scaled_to_n_steps"])
    if "fk_joints" in ex:
        print("   fk_joints %.3f G poses/s  frac %.3f" % (ex["fk_joints"]["poses_per_s_per_gpu"] / 1e9, ex["fk_joints"]["roofline"]["frac"]))
    for k in ("lbs_full_fwd", "lbs_full_fwd_bwd"):
        if k in ex:
            print("   %-18s %.3f ms  runs %s" % (k, ex[k]["ms"], ex[k].get("runs_ms")))
    sb = ex.get("small_batches", {})
    for k, v in sb.items():
        if "ms_per_step" in v:
            print("   %-18s %.4f ms / step (%.3f of the bf16 peak)" % (k, v["ms_per_step"], v["frac_of_mfma_peak"]))
        else:
            print("   %-18s %.4f s = %.0f samples/s" % (k, v["seconds"], v["samples_per_s"]))
