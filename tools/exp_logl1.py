import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import exp_cols32 as e
for name, B, N, W in (("cfg2 3 x 2^20 x 64", 3, 1 << 20, 64), ("3 x 2^19 x 64", 3, 1 << 19, 64), ("cfg1 3 x 2^18 x 1", 3, 1 << 18, 1)):
    print("==", name, flush=True)
    for vn, o in (("default", {}), ("logl1=10", {"logl1": 10}), ("logl1=8", {"logl1": 8}), ("logl1=7", {"logl1": 7})):
        try:
            ms, fam, out = e.run(B, N, W, o)
        except Exception as ex:
            print("  ", vn, "failed:", ex); continue
        ks = "  ".join(f"{k} {v['ms'] / max(v['launches'], 1) * 1e3:.0f}us" for k, v in fam.items())
        print(f"   {vn:10s} {ms:8.4f} ms | {ks}", flush=True)
