"""What a plain device-to-device copy reaches when its source streams out of HBM (a 4 GB source walked 39 MB at a time: nothing is found in
the 256 MB Infinity Cache) and when it is cache-resident (the same 39 MB again and again) -- the practical ceiling for the replay gather.
    python tools/copy_ceiling.py [MB=39]"""
import sys
import torch
mb = float(sys.argv[1]) if len(sys.argv) > 1 else 39.0
n = int(mb * 1e6 // 4)
big = torch.empty(int(4e9 // 4), device="cuda:0").normal_()
dst_big = torch.empty(int(2e9 // 4), device="cuda:0")
slots, dslots = big.numel() // n, dst_big.numel() // n
for name, resident in (("HBM-streaming source and destination", False), ("cache-resident (same 39 MB every time)", True)):
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        K = 60
        torch.cuda.synchronize()
        e0.record()
        for i in range(K):
            s = 0 if resident else (i * 7 + rep * 13) % slots
            d = 0 if resident else (i * 5 + rep * 11) % dslots
            dst_big[d * n:(d + 1) * n].copy_(big[s * n:(s + 1) * n])
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / K
        print("%-44s %6.1f MB read + %6.1f MB written: %6.2f us per copy = %5.2f TB/s (read + write)" % (name, mb, mb, us, 2 * mb / us))
