"""Run the unet256 step parity several times and print every quantity that misses its tolerance (atomic summation order
makes the build's result vary in the last bits between runs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import step_parity
name = sys.argv[1] if len(sys.argv) > 1 else 'unet256'
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5):
    rows = step_parity.run(name, check=False)
    bad = [r for r in rows if not r[3]]
    print("run %d: %d rows, %d bad" % (it, len(rows), len(bad)))
    for r in bad:
        print("   ", r)
