import ctypes as C, numpy as np, torch
from opendpd_amd import CoreModel, _lib
from opendpd_amd.train_funcs import FusedAdamW, fused_train_step
lib=_lib.load()
torch.manual_seed(0)
net=CoreModel(2,5,1,"gru").cuda()
B,T=3,9
x=(torch.rand(B,T,2)-0.5).cuda(); t=(torch.randn(B,T,2)*0.3).cuda()
res={}
for gp in (-1,0):
    lib.odpd_set_tuning(b"gp_max_batch", C.c_int64(gp))
    opt=FusedAdamW(net, lr=0.0, weight_decay=0.0)
    loss=fused_train_step(opt,x,t,"l2",0.0)
    res[gp]=(float(loss), opt.grad[:-4].cpu().numpy().copy())
print(res[-1][0],res[0][0])
o=0
for k,p in net.named_parameters():
    n=p.numel(); a=res[-1][1][o:o+n]; b=res[0][1][o:o+n]; o+=n
    print(k, np.abs(a-b).max()/max(np.abs(b).max(),1e-9))
    if "bias_ih" in k or "bias_hh" in k: print(a.reshape(3,5)); print(b.reshape(3,5))
    if "weight_ih" in k: print(a.reshape(3,5,2)[:, :2]); print(b.reshape(3,5,2)[:, :2])
