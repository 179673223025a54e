"""A/B of whole training steps (bench workload, --no-extras) between library variants, one library per process, round-robin:
    python tools/ab_step.py lib_a.so lib_b.so [--rounds=N]"""
import json, os, subprocess, sys
libs = [a for a in sys.argv[1:] if not a.startswith("--")]
rounds = int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--rounds=")), 2))
for rd in range(rounds):
    for lib in libs:
        env = dict(os.environ, OSU_DREAMER_HIP_LIB=os.path.abspath(lib))
        o = subprocess.run([sys.executable, "bench.py", "--steps", "5", "--warmup", "2", "--no-extras"], env=env, capture_output=True, text=True, timeout=900)
        line = [l for l in o.stdout.splitlines() if l.startswith("{")]
        print(f"[round {rd}] {os.path.basename(lib):28s} ms_per_step", json.loads(line[-1])["ms_per_step"] if line else "FAILED " + o.stderr[-300:], flush=True)
