import sys, math, time
sys.path.insert(0, '.')
import torch
import dasr_amd
from dasr_amd import synth
from dasr_amd.depthnet import DepthNet
from oracle import depthnet_oracle as O
dev='cuda'
cfg=O.make_cfg()
net=DepthNet(which_ResBlk_depth=list(range(14)), nb=16, scale=8, depth_latent_ch=256)
synth.closed_form_fill_(net.state_dict().items())
net=net.to(dev)
lq,gt,dm,mk=synth.seeded_batch(0,1,128,160,8)
res={}
for name,dt in (('o32',torch.float32),('o64',torch.float64)):
    t=time.time()
    sd={k:v.detach().cpu().to(dt).clone().requires_grad_(True) for k,v in net.state_dict().items()}
    ref=O.depthnet_forward(sd,cfg,lq.to(dt),dm.to(dt),mk.to(dt))
    wgt=torch.cos(torch.arange(ref.numel(),dtype=dt)*0.013).reshape(ref.shape)
    (ref*wgt).sum().backward()
    res[name]=(ref.detach(),{k:v.grad for k,v in sd.items()})
    print(name,'done',time.time()-t,flush=True)
sr=net(lq.to(dev),dm.to(dev),mk.to(dev))
wgt=torch.cos(torch.arange(sr.numel(),dtype=torch.float32)*0.013).reshape(sr.shape).to(dev)
(sr*wgt).sum().backward()
hip=(sr.detach().cpu(),{k:(p.grad.detach().cpu() if p.grad is not None else None) for k,p in net.named_parameters()})
def fwd(a,b): return (a.double()-b.double()).abs().max().item()
print('fwd hip-o32',fwd(hip[0],res['o32'][0]),'hip-o64',fwd(hip[0],res['o64'][0]),'o32-o64',fwd(res['o32'][0],res['o64'][0]))
def rel(A,B,filt=lambda k:True):
    num=den=0
    for k in A:
        if A[k] is None or B[k] is None or not filt(k): continue
        if 'conv1.0.bias' in k or 'conv2.0.bias' in k: continue
        num+=(A[k].double()-B[k].double()).pow(2).sum().item(); den+=B[k].double().pow(2).sum().item()
    return math.sqrt(num/den)
print('grad rel L2: hip-o32',rel(hip[1],res['o32'][1]),'hip-o64',rel(hip[1],res['o64'][1]),'o32-o64',rel(res['o32'][1],res['o64'][1]))
groups=['encoder','head','depth-residual1.','depth-residual7.','depth-residual13.','upscale1','classic-residual15','upscale2','classic-residual16','upscale3','conv_output']
for gname in groups:
    f=lambda k,g=gname:k.startswith(g)
    print('%-22s hip-o64 %.3e   o32-o64 %.3e'%(gname,rel(hip[1],res['o64'][1],f),rel(res['o32'][1],res['o64'][1],f)))
