"""Puts the secondary-legs table of a bench.py JSON line into DESIGN.md section 5 (in place of the one there):
python tools/design_refresh.py gpurun_out/<run>/bench.json"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tab = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_tables.py"), sys.argv[1]], capture_output=True, text=True, check=True).stdout.split("\n")
i0 = next(i for i, l in enumerate(tab) if l.startswith("| leg |"))
i1 = next(i for i in range(i0, len(tab)) if tab[i].strip() == "")
path = os.path.join(ROOT, "DESIGN.md")
doc = open(path).read().split("\n")
j0 = next(i for i, l in enumerate(doc) if l.startswith("| leg |"))
j1 = next(i for i in range(j0, len(doc)) if doc[i].strip() == "")
doc[j0:j1] = tab[i0:i1]
open(path, "w").write("\n".join(doc))
print("\n".join(tab[:i0]))
