#!/bin/bash
# the bounded Collapse kernel: its GPU tests + microseconds per frame at crf 0 / crf 3 / DeltaT (1080p scene, 300 frames)
cd /root/repo
python -m pytest tests/test_gpu_parity.py -x -q -k "cb_ or crf0 or lossy or config_5 or model_fixtures" > gpurun_out/t.txt 2>&1; grep -v "RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl" gpurun_out/t.txt | tail -4
python - <<'PY'
import os,sys,json
sys.path.insert(0,'.'); sys.path.insert(0,'adder-codec-rs_amd')
import torch, adder_amd as A
W,H,T=1920,1080,300
st=torch.cuda.current_stream().cuda_stream
f=torch.empty((T,W*H),dtype=torch.uint8,device='cuda'); A.synth_clip_device(f,A.CONTENT_SCENE,W,H,1,num_frames=T,stream=st)
e=torch.empty((int(W*H*T*0.6),3),dtype=torch.int32,device='cuda'); o=torch.zeros(T+1,dtype=torch.int64,device='cuda')
for name,crf,tm in (("crf0 abs",(0,0,10),1),("crf3 abs",(2,7,7),1),("crf0 dt",(0,0,10),0)):
    hv=A.HipVideo(W,H,1,time_mode=tm,multi_mode=1,delta_t_max=7650,c_thresh_start=crf[0],c_counter_start=0)
    hv.set_crf_parameters(crf[1],crf[2])
    best=1e9
    for it in range(5):
        hv.reset(); hv.integrate_device(f,e,o,stream=st); n=hv.finish(); best=min(best,hv.last_batch_ms()/T*1000)
    print(name, round(best,2),'us/frame', round(n/(W*H*T),4)); hv.close()
PY
