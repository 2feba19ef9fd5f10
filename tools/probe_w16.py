import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from tests.golden_util import Fixture
from tests.test_oracle_golden import QAT
from tests.test_quant_gpu import _qmodel
for name, bb, bits in QAT:
    if bits == 8: continue
    fx = Fixture(name)
    x = torch.from_numpy(fx["x"]).cuda()
    for prefix, ytr, yev in (("sd", "y", "y_eval"), ("sd3", "y_p3_train", "y_p3_eval")):
        q = _qmodel(fx, bb, bits, prefix)
        q.train()
        with torch.no_grad(): yt = q(x).cpu().numpy()
        q.eval()
        with torch.no_grad(): ye = q(x).cpu().numpy()
        dt, de = np.abs(yt - fx[ytr]), np.abs(ye - fx[yev])
        print(name, prefix, "train max/LSB %.3f n>0.5LSB %d of %d | eval max/LSB %.3f ndiff %d" % (dt.max() * 2**14, (dt > 2**-15).sum(), dt.size, de.max() * 2**14, (de > 0).sum()), "ymax", np.abs(fx[ytr]).max())
