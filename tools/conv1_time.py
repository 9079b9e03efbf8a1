#!/usr/bin/env python3
"""The stem's first convolution at cfg-2 (B=32, 80 x 1000 mel frames -> h1 (32, 499, 39, 512) fp32 = 1.28 GB): time and write rate."""
import torch, sys, os
sys.path.insert(0, os.getcwd())
from conformer_amd import _lib
lib=_lib.load()
dev=torch.device("cuda:0")
B,F,T,C=32,80,1000,512
x=torch.randn(B,F,T,device=dev); w1=torch.randn(C,1,3,3,device=dev); b1=torch.randn(C,device=dev)
F1,T1=(F-1)//2,(T-1)//2
h1=torch.empty(B,T1,F1,C,device=dev)
st=torch.cuda.current_stream().cuda_stream
def run(): _lib.check(lib.cfm_subsample_conv1_relu_f32(x.data_ptr(),w1.data_ptr(),b1.data_ptr(),h1.data_ptr(),B,F,T,C,st),"c1")
for _ in range(3): run()
ts=[]
for r in range(7):
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/10*1e3)
print("conv1 median %.1f us min %.1f us  %.2f TB/s written"%(sorted(ts)[3],min(ts),h1.numel()*4/sorted(ts)[3]/1e6))

h16 = torch.empty(B, T1, F1, C, device=dev, dtype=torch.bfloat16)
def run16(): _lib.check(lib.cfm_subsample_conv1_relu_out16_f32(1, x.data_ptr(), w1.data_ptr(), b1.data_ptr(), h16.data_ptr(), B, F, T, C, st), "c1-16")
for _ in range(3): run16()
ts = []
for r in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run16()
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10 * 1e3)
print("conv1 (bf16 h1) median %.1f us min %.1f us  %.2f TB/s written" % (sorted(ts)[3], min(ts), h16.numel() * 2 / sorted(ts)[3] / 1e6))
