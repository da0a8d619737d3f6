#!/usr/bin/env python3
"""Board power and clock telemetry while the residual kernel runs back to back for several seconds (runs on the GPU
box through gpurun): starts `bench.py --steps S` as a child and samples rocm-smi (power, shader / memory clocks,
temperature, power cap) a few times per second until it exits.  Writes gpurun_out/power_probe/{samples.jsonl,
summary.json}; the summary is what profiles/ keeps to support the clock-throttling reading of HISTORY.md section 7."""
import json, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "power_probe")
os.makedirs(OUT, exist_ok=True)
steps = os.environ.get("STEPS", "400")


def sample():
    for cmd in (["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--showmaxpower", "--json"],
                ["amd-smi", "metric", "--power", "--clock", "--temperature", "--json"]):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=10)
            if r.returncode == 0 and r.stdout.strip().startswith(("{", "[")):
                return {"tool": cmd[0], "data": json.loads(r.stdout)}
        except Exception as ex:                      # tool missing or not permitted: recorded, not fatal
            last = repr(ex)
    return {"tool": None, "error": locals().get("last", "no smi tool answered")}


idle = sample()
child = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "5", "--no-cpu-baseline"],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
samples = []
t0 = time.time()
while child.poll() is None:
    s = sample()
    s["t"] = time.time() - t0
    samples.append(s)
    time.sleep(0.2)
out = child.stdout.read()
line = [l for l in out.splitlines() if l.startswith("{")]
bench = json.loads(line[-1]) if line else None
with open(os.path.join(OUT, "samples.jsonl"), "w") as f:
    f.write(json.dumps({"idle": idle}) + "\n")
    for s in samples:
        f.write(json.dumps(s) + "\n")


def numbers(d, keys):
    """all numeric leaves of a nested smi answer whose key mentions one of `keys`"""
    found = []
    def walk(x, path=""):
        if isinstance(x, dict):
            for k, v in x.items():
                walk(v, path + "/" + str(k))
        elif isinstance(x, list):
            for v in x:
                walk(v, path)
        else:
            if any(k.lower() in path.lower() for k in keys):
                m = re.search(r"-?\d+(?:\.\d+)?", str(x))
                if m:
                    found.append((path, float(m.group(0))))
    walk(d)
    return found


def series(keys):
    per = {}
    for s in samples:
        if s.get("data") is None:
            continue
        for path, v in numbers(s["data"], keys):
            per.setdefault(path, []).append(v)
    return {p: {"min": min(v), "mean": sum(v) / len(v), "max": max(v), "n": len(v)} for p, v in per.items()}


summary = {"what": f"rocm-smi samples every ~0.2 s while bench.py ran {steps} back-to-back steps (k_dlt4 + k_residual, 50k x 100k)",
           "seconds": time.time() - t0, "samples": len(samples), "tool": samples[0].get("tool") if samples else None,
           "idle_before": idle.get("data"), "power": series(["power"]), "sclk": series(["sclk"]), "mclk": series(["mclk"]),
           "temperature": series(["temp"]),
           "bench": None if bench is None else {"ms_per_step": bench["ms_per_step"], "k_residual_ms": bench["kernel_ms"]["k_residual"],
                                                "residual_kernel_GBps": bench["residual_kernel_GBps"], "frac": bench["roofline"]["frac"]}}
with open(os.path.join(OUT, "summary.json"), "w") as f:
    json.dump(summary, f, indent=1)
print(json.dumps(summary)[:3000])
