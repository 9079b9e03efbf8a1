import sys, os, collections, traceback
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from model.modules.encoder import Encoder
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
class Sites(TorchDispatchMode):
    def __init__(s): super().__init__(); s.c = collections.Counter()
    def __torch_dispatch__(s, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(("aten.view", "aten._unsafe_view", "aten.detach", "aten.alias", "aten.t.", "aten.transpose", "aten.slice", "aten.select", "aten.as_strided", "aten.expand", "aten.unsqueeze", "aten.squeeze", "aten.permute", "aten.reshape", "aten.empty", "aten.sym_", "aten.is_", "aten.lift_fresh", "aten._local_scalar")):
            site = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "conformer_amd" in fr.filename or "/model/" in fr.filename:
                    site = f"{fr.filename.split('repo/')[-1]}:{fr.lineno}"; break
            s.c[(name, site)] += 1
        return func(*args, **(kwargs or {}))
dev = torch.device("cuda:0")
enc = Encoder(80, 16, 512, 8, 31, 0.0).to(dev).eval()
x = torch.randn(32, 80, 1000, device=dev); L = torch.full((32,), 1000, dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(2): enc(x, L)
    with Sites() as s: enc(x, L)
for (n, site), k in s.c.most_common(20): print(k, n, site)
