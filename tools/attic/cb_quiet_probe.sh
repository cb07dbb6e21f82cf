#!/bin/bash
# The quiet-wave path of the bounded Collapse kernel (cb_quiet / cb_step_quiet) A/B against a build without it:
# 1080p default mode at crf 0 / crf 3 / static content / a 900-frame crf-3 run (steady state), C5's shape (4K RGB).
cd /root/repo
python -m pytest tests/test_gpu_parity.py -x -q -k "cb_ or crf0 or lossy or config_5 or model_fixtures or static" > gpurun_out/t.txt 2>&1; grep -v "RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl" gpurun_out/t.txt | tail -3
for lib in "" /root/repo/build/variants/libadder_hip_prev.so; do
echo "== lib=${lib##*/}"
ADDER_HIP_LIB=$lib python - <<'PY'
import os,sys,json
sys.path.insert(0,'.'); sys.path.insert(0,'adder-codec-rs_amd')
import torch, adder_amd as A
st=torch.cuda.current_stream().cuda_stream
def run(name,W,H,C,T,content,crf,tm,reps=4, cont=False):
    f=torch.empty((T,W*H*C),dtype=torch.uint8,device='cuda'); A.synth_clip_device(f,content,W,H,C,num_frames=T,stream=st)
    e=torch.empty((int(W*H*C*T*0.6)+1024,3),dtype=torch.int32,device='cuda'); o=torch.zeros(T+1,dtype=torch.int64,device='cuda')
    hv=A.HipVideo(W,H,C,time_mode=tm,multi_mode=1,delta_t_max=7650,c_thresh_start=crf[0],c_counter_start=0)
    hv.set_crf_parameters(crf[1],crf[2])
    best=1e9
    for it in range(reps):
        hv.reset(); hv.integrate_device(f,e,o,stream=st); n=hv.finish(); best=min(best,hv.last_batch_ms()/T*1000)
    print(name, round(best,2),'us/frame', 'e', round(n/(W*H*C*T),5), flush=True); hv.close(); del f,e,o
run("1080p crf0 abs 300",1920,1080,1,300,A.CONTENT_SCENE,(0,0,10),1)
run("1080p crf3 abs 300",1920,1080,1,300,A.CONTENT_SCENE,(2,7,7),1)
run("1080p crf3 abs 900",1920,1080,1,900,A.CONTENT_SCENE,(2,7,7),1)
run("1080p static crf0 abs 300",1920,1080,1,300,A.CONTENT_STATIC,(0,0,10),1)
run("4K RGB crf3 abs 64",3840,2160,3,64,A.CONTENT_SCENE,(2,7,7),1,reps=3)
PY
done
