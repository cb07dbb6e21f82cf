import ctypes, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
code = r'''
import ctypes, sys, os
mode, lib = sys.argv[1], sys.argv[2]
if mode == "torch_first":
    import torch
    print("torch avail", torch.cuda.is_available(), torch.version.hip)
    x = torch.ones(4, device="cuda"); print("torch sum", float(x.sum()))
L = ctypes.CDLL(lib)
print("probe rc", L.probe())
print(sorted(set(l.split()[-1] for l in open('/proc/self/maps') if ('amdhip' in l or 'hsa-runtime' in l))))
if mode == "lib_first":
    import torch
    try:
        x = torch.ones(4, device="cuda"); print("torch sum", float(x.sum()))
    except Exception as e: print("torch failed:", repr(e)[:200])
'''
for mode in ("torch_first", "lib_first", "alone"):
    for lib in ("t_default.so", "t_nocomp.so", "t_cov5.so"):
        print("=====", mode, lib, flush=True)
        r = subprocess.run([sys.executable, "-c", code, mode, os.path.join(here, lib)], capture_output=True, text=True, timeout=300)
        print(r.stdout[-1500:], r.stderr[-800:], flush=True)
