#!/usr/bin/env python3
"""Does any hipExtMallocWithFlags flavour give reliably fast streaming stores?  For each flag, several allocations of
the Jacobian outputs' sizes; the store pattern's rate in each."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from city2ba_amd import _lib as L  # noqa: E402

torch.cuda.init()
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
lib = L.lib()
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"), mode=C.RTLD_GLOBAL)
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipFree.argtypes = [C.c_void_p]
n = 19302494
sizes = (n * 16, n * 144, n * 48)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def rate(ptrs):
    for _ in range(2):
        L.check(lib.c2b_calib_store_pattern(n, ptrs[0], ptrs[1], ptrs[2], st))
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(4):
        L.check(lib.c2b_calib_store_pattern(n, ptrs[0], ptrs[1], ptrs[2], st))
    e.record()
    torch.cuda.synchronize()
    return n * 208 / (s.elapsed_time(e) / 4 * 1e-3) / 1e9


for name, flag in (("default", 0), ("finegrained", 1), ("uncached", 3), ("contiguous", 4)):
    rates, held = [], []
    for attempt in range(5):
        ptrs = []
        ok = True
        for sz in sizes:
            p = C.c_void_p()
            rc = hip.hipExtMallocWithFlags(C.byref(p), sz, flag)
            if rc != 0:
                ok = False
                break
            ptrs.append(p)
        if not ok:
            rates.append("alloc rc=%d" % rc)
            for p in ptrs:
                hip.hipFree(p)
            break
        rates.append(round(rate(ptrs), 0))
        held.append(ptrs)
    for ptrs in held:
        for p in ptrs:
            hip.hipFree(p)
    print("%-12s store GB/s per allocation: %s" % (name, rates))
# torch's allocator for comparison
r = []
held = []
for attempt in range(5):
    t = [torch.empty(sz // 8, dtype=torch.float64, device=dev) for sz in sizes]
    r.append(round(rate([C.c_void_p(x.data_ptr()) for x in t]), 0))
    held.append(t)
print("%-12s store GB/s per allocation: %s" % ("torch.empty", r))
