"""Timing experiment helper (not a test): frame-kernel time under ADDER_HIP_ABLATE variants."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, json
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "adder-codec-rs_amd"))
import torch, adder_amd as A
W,H,T=1920,1080,100
content = int(os.environ.get("CONTENT","2"))
d_frames = torch.empty((T, W*H), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
A.synth_clip_device(d_frames, content, W, H, 1, num_frames=T, stream=st)
d_ev = torch.empty((int(W*H*T*1.3), 3), dtype=torch.int32, device="cuda")
d_off = torch.zeros(T+1, dtype=torch.int64, device="cuda")
hv = A.HipVideo(W,H,1,time_mode=0,multi_mode=int(os.environ.get("MULTI","1")),delta_t_max=int(os.environ.get("DTM","255")),c_thresh_start=0,c_counter_start=0)
hv.set_crf_parameters(0,10)
best=1e9
for it in range(4):
    hv.reset(); hv.integrate_device(d_frames, d_ev, d_off, stream=st)
    try: n = hv.finish()
    except Exception as e: n = -1
    best=min(best, hv.last_batch_ms()/T*1000)
print(json.dumps({"ablate": os.environ.get("ADDER_HIP_ABLATE","0"), "us_per_frame": round(best,2), "events": n}))
''' % (ROOT, ROOT)
for ab in sys.argv[1:] or ["0", "1", "2", "3", "4", "7"]:
    env = dict(os.environ, ADDER_HIP_ABLATE=ab)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:], flush=True)
