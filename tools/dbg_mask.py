import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from shared_simd_scan_amd import ScanEngine
from oracle import oracle
O = oracle()
eng = ScanEngine(0)
def bad_tiles(bm, n):
    got = np.unpackbits(bm.cpu().numpy(), bitorder="little")[:n]
    bad = np.nonzero(got == 0)[0]
    return sorted(set((bad // 8192).tolist()))
for K in (0, 1, 2, 4):
    eng.set_option("scan_burst", K)
    for c in (1, 2, 3, 9, 17):
        for n in (16461, 8192 * 9 + 5, 8192 * 40 + 3):
            vals = O.gen_values("splitmix", n, c, 99)
            col = eng.compress(torch.from_numpy(vals.astype(np.int32)).cuda(), c)
            nb_prev = nb_and = nb_plain = 0
            for rep in range(30):
                prev, _ = eng.scan_where("<=", (1 << c) - 1, col)
                bp = bad_tiles(prev, n)
                bm, hits = eng.scan_where(">=", 0, col, and_mask=prev)
                ba = bad_tiles(bm, n)
                nb_prev += bool(bp); nb_and += bool(ba)
                if (bp or ba) and nb_prev + nb_and <= 2:
                    print("   rep", rep, "prev bad tiles", bp[:8], "and bad tiles", ba[:8], "hits", int(hits.item()))
            print(f"K={K} c={c} n={n}: prev wrong {nb_prev}/30, and wrong {nb_and}/30", flush=True)
