"""GPU box: four-step column kernels with 8-column tiles (64-byte row segments, option col_logt = 3: twice the workgroups per CU)
against the 16-column default, with the default and a shorter / longer column length, on cfg2, 3 x 2^19 x 64, 8 x 2^18 x 8 and cfg1."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import exp_cols32 as e
for name, B, N, W in (("cfg2 3 x 2^20 x 64", 3, 1 << 20, 64), ("3 x 2^19 x 64", 3, 1 << 19, 64), ("8 x 2^18 x 8", 8, 1 << 18, 8), ("cfg1 3 x 2^18 x 1", 3, 1 << 18, 1)):
    print("==", name, flush=True)
    ref = None
    for vn, o in (("default", {}), ("col_logt=3", {"col_logt": 3}), ("col_logt=3 logl1=10", {"col_logt": 3, "logl1": 10}),
                  ("col_logt=3 logl1=8", {"col_logt": 3, "logl1": 8}), ("logl1=8", {"logl1": 8})):
        try:
            ms, fam, out = e.run(B, N, W, o)
        except Exception as ex:
            print("  ", vn, "failed:", ex); continue
        if ref is None:
            ref = out
        same = all((a == b).all() for a, b in zip(out[:1], ref[:1]))
        ks = "  ".join(f"{k} {v['ms'] / max(v['launches'], 1) * 1e3:.0f}us" for k, v in fam.items())
        print(f"   {vn:22s} {ms:8.4f} ms | lag_int equal {same} | {ks}", flush=True)
