import json, os, subprocess, sys
pols = sys.argv[1:]
for pol in pols:
    env = dict(os.environ)
    if pol != "default":
        env["HSIDM_WIDE_POLICY"] = pol
    r = subprocess.run([sys.executable, "bench.py", "--steps", "30", "--warmup", "4", "--no-cpu-baseline", "--no-modes", "--no-gae", "--no-train", "--no-small", "--no-roofline"],
                       capture_output=True, text=True, env=env)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        p = d["parity"]["fp16"]
        print("%-60s %7.0f steps*batch/s %6.2f ms  latents %.2e cube %.2e dPSNR %.1e dSAM %.2e meets %s" % (pol, d["value"], d["ms_per_step"], p["latents_rel_err"], p["cube_rel_err"], p["dPSNR_dB"], p["dSAM_deg"], p["meets_north_star"]), flush=True)
    except Exception as e:
        print(pol, "FAILED", r.stderr[-500:])
