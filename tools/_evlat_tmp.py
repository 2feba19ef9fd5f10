import torch, numpy as np, itertools
from opendpd_amd import CoreModel
torch.manual_seed(0)
net = CoreModel(2,1,1,"lstm").eval()
sd = {k:v.numpy().astype(np.float64) for k,v in net.state_dict().items()}
for k,v in sd.items(): print(k, v.ravel())
x = torch.randn(1,300,2)*0.3
net = net.cuda()
with torch.no_grad(): y = net(x.cuda())
y2 = net(x.cuda().requires_grad_(True)).detach()
print(y[0,0].cpu().numpy(), y2[0,0].cpu().numpy())
ks = list(sd)
wih = [v for k,v in sd.items() if "weight_ih" in k][0]; bih=[v for k,v in sd.items() if "bias_ih" in k][0]; bhh=[v for k,v in sd.items() if "bias_hh" in k][0]
wout=[v for k,v in sd.items() if "fc_out.weight" in k][0]
a = wih @ x[0,0].numpy().astype(np.float64) + bih + bhh
sg = 1/(1+np.exp(-a)); th = np.tanh(a)
act = np.array([sg[0], sg[1], th[2], sg[3]])
for perm in itertools.product(range(4), repeat=4):
    i,f,g,o = (act[p] for p in perm)
    h = o*np.tanh(i*g)
    yy = wout[:,0]*h
    if abs(yy[0]-y[0,0,0].item()) < 1e-5: print("eval matches perm", perm)
    if abs(yy[0]-y2[0,0,0].item()) < 1e-5: print("ref matches perm", perm)
