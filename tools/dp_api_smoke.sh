#!/bin/bash
# two ranks sharing the one GPU over gloo through the API (what tests/test_e2e_gpu.py::test_data_parallel_api_run_equals_single_process does), stderr shown
set -u
WD=$(mktemp -d); cd $WD
python3 - <<PY
import os, numpy as np, pandas as pd
d = dict(np.load(os.path.join("$GRAFT_REPO_ROOT", "tests", "golden", "dpa200_dataset.npz")))
ds = os.path.join("$WD", "datasets", "DPA_200MHz"); os.makedirs(ds)
open(os.path.join(ds, "spec.json"), "w").write(str(d.pop("spec")))
for k, v in d.items(): pd.DataFrame(v, columns=["I", "Q"]).to_csv(os.path.join(ds, f"{k}.csv"), index=False)
PY
cat > run.py <<PY
import os, sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
os.environ["OPENDPD_DATASETS"] = "$WD/datasets"
import opendpd_amd as od
kw = dict(dataset_name="DPA_200MHz", PA_backbone="dgru", PA_hidden_size=9, frame_length=50, batch_size=64, lr=1e-3, seed=0, accelerator="cuda", n_epochs=2)
res = od.train_pa(**kw)
print("DONE", res["model_path"])
PY
OPENDPD_DIST_BACKEND=gloo OPENDPD_DIST_SINGLE_DEVICE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 run.py 2>&1 | grep -v "amdgpu.ids" | grep -B30 "Error\|error" | head -60
