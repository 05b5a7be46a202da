"""A/B kernel timing of several builds of the library on the same batch: python tools/ab_time.py N lib1 lib2 ..."""
import os, subprocess, sys
n = sys.argv[1]
for lib in sys.argv[2:]:
    env = dict(os.environ, C3POA_LIB=lib)
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "phase_prof.py"), n], env=env, capture_output=True, text=True).stdout
    line = [l for l in out.splitlines() if "ms_total" in l]
    print(os.path.basename(lib), line[0] if line else out[-300:])
