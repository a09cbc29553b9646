import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from city2ba_amd import device as D
dev = torch.device("cuda", 0)
f0 = torch.cuda.mem_get_info()[0]
for n, att, thr in ((2412812, 48, 6800.0), (2412812, 8, 6800.0), (19302494, 48, 7000.0)):
    o = D.JacobianOutputs(n, dev, max_attempts=att, fast_store_GBs=thr)
    print("n %d attempts %d: kept %.0f GB/s (#%d of %d tried): %s" % (n, att, o.store_GBs, o.chosen, len(o.log), [int(x / 100) for x in o.log]), flush=True)
    del o
torch.cuda.synchronize()
print("free memory before / after: %.2f / %.2f GB" % (f0 / 1e9, torch.cuda.mem_get_info()[0] / 1e9))
