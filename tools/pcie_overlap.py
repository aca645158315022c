#!/usr/bin/env python3
"""Do uploads and downloads overlap on this box?  One 640 MB pinned upload and one 160 MB pinned download, alone
and together on two streams (hipMemcpyAsync through torch); prints GB/s of each leg."""
import time

import torch

up_h = torch.empty(160_000_000, dtype=torch.float32).pin_memory()
dn_h = torch.empty(40_000_000, dtype=torch.float32).pin_memory()
up_d = torch.empty_like(up_h, device="cuda")
dn_d = torch.empty_like(dn_h, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def leg(do_up, do_dn, reps=6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if do_up:
            with torch.cuda.stream(s1):
                up_d.copy_(up_h, non_blocking=True)
        if do_dn:
            with torch.cuda.stream(s2):
                dn_h.copy_(dn_d, non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for _ in range(2):
    a, b, c = leg(True, False), leg(False, True), leg(True, True)
    print(f"upload 640 MB alone {a * 1e3:6.2f} ms ({0.64 / a:5.1f} GB/s)   download 160 MB alone {b * 1e3:6.2f} ms ({0.16 / b:5.1f} GB/s)   "
          f"both {c * 1e3:6.2f} ms (sum of the two alone {1e3 * (a + b):6.2f} ms)")
