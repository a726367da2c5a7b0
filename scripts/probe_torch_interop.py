import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
order = sys.argv[1] if len(sys.argv) > 1 else "torch_first"
if order == "torch_first":
    import torch
    print("torch", torch.__version__, "cuda avail", torch.cuda.is_available(), torch.cuda.get_device_name(0))
    x = torch.ones(4, device="cuda")
from muopdb_amd import lib as L
from muopdb_amd.index import FlatIndex
ctx = L.Context(0)
rng = np.random.default_rng(0)
base = rng.standard_normal((5000, 128)).astype(np.float32)
q = rng.standard_normal((8, 128)).astype(np.float32)
idx = FlatIndex(ctx, base)
ids, dist, _ = idx.search(q, 10)
print("host-mode ids[0]", ids[0][:5])
if order != "torch_first":
    import torch
    print("torch after: cuda avail", torch.cuda.is_available())
tq = torch.from_numpy(q).cuda()
tids = torch.zeros((8, 10), dtype=torch.int32, device="cuda")
tdist = torch.zeros((8, 10), dtype=torch.float32, device="cuda")
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
idx.search_device(tq.data_ptr(), 8, 10, tids.data_ptr(), tdist.data_ptr())
ctx.sync()
print("device-mode equal:", np.array_equal(tids.cpu().numpy().view(np.uint32), ids), np.array_equal(tdist.cpu().numpy(), dist))
os.system("cat /proc/%d/maps | grep -E 'libamdhip64|libhsa-runtime' | awk '{print $6}' | sort -u" % os.getpid())
