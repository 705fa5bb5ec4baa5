"""A/B of two BUILDS of the library on one box: alternating subprocesses (each loads one build through ODAM_AMD_LIB), forward of 32 frames.
   python ab_libs.py <libA.so> <libB.so> [more .so ...] [backbone dtype B rounds]   -> per-round medians, and whether the outputs are bit-identical"""
import os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
args = [a for a in sys.argv[1:] if not a.endswith(".so")]
rest = args[:3]
rounds = int(args[3]) if len(args) > 3 else 3
res = {l: [] for l in libs}; dig = {}
for r in range(rounds):
    for lib in libs:
        env = dict(os.environ, ODAM_AMD_LIB=os.path.join(ROOT, lib))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tests/native/fwd_time.py")] + rest, env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.strip().splitlines() if l.strip()][-1] if out.stdout.strip() else "nan nan -"
        if out.returncode != 0: print(out.stderr[-2000:])
        med, mn, d = line.split()
        res[lib].append(float(med)); dig[lib] = d
for lib in libs:
    print(f"{lib}: " + " ".join(f"{t:.2f}" for t in res[lib]) + f"  ms per forward ({' '.join(rest) or 'resnet50 fp32 32'}); outputs {dig[lib]}")
print("outputs bit-identical" if len(set(dig.values())) == 1 else "OUTPUTS DIFFER")
