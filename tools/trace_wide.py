"""Per-workgroup timeline of the two wide kernels (pvk_dec, pvk_enc_bwd) of the last train step: build the engine with
-DPV_WTRACE into another file and point PVAE_LIB at it:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DPV_WTRACE -o tools/_build/libpvae_wtrace.so popvae_amd/csrc/pvae_engine.hip
  PVAE_LIB=tools/_build/libpvae_wtrace.so python tools/trace_wide.py [n_snps]
Points per workgroup (100 MHz wall clock): start, staged (LDS tile + barrier), sweep done, end; plus where it ran."""
import ctypes as C, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popvae_amd import engine as E
L = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
N, W, D = 200, 128, 6
g = np.random.default_rng(0)
packed = (g.integers(0, 256, size=(N, (L + 31) // 32 * 8), dtype=np.uint8) & 0x55)
e = E.Engine(n_snps=L, width=W, depth=D, latent_dim=2, batch_size=32, seed=1)
e.load_genotypes(packed); e.init_weights(1); e.lr = 1e-6
rows = np.tile(np.arange(32, dtype=np.int32), 12)
e.train_epoch(rows, np.zeros(0, np.int32))
e.synchronize()
n = 2 * 1024 * 6
buf = (C.c_ulonglong * n)()
E.lib().pvae_debug_wtrace(buf, n)
a = np.array(buf[:], dtype=np.int64).reshape(2, 1024, 6)
for k, name in enumerate(("dec", "enc_bwd")):
    t = a[k]
    used = t[:, 0] > 0
    ids = np.flatnonzero(used)
    hw, xcc = t[used, 4], t[used, 5] & 15
    cu = ((hw >> 8) & 15) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)  # (xcc, se, sh, cu)
    t = t[used, :4].astype(np.float64) * 0.01  # us
    t -= t[:, 0].min()
    print(f"{name}: {len(t)} workgroups on {len(set(cu.tolist()))} distinct CUs, kernel span {t[:, 3].max():.1f} us")
    for j, pt in enumerate(("start", "staged", "sweep_done", "end")):
        c = t[:, j]
        print(f"  {pt:10s} min {c.min():6.1f}  p10 {np.percentile(c, 10):6.1f}  median {np.median(c):6.1f}  p90 {np.percentile(c, 90):6.1f}  max {c.max():6.1f}")
    per_cu = {}
    for i in range(len(t)):
        per_cu.setdefault(int(cu[i]), []).append(i)
    for cnt in sorted(set(len(v) for v in per_cu.values())):
        ends = [t[i, 3] for v in per_cu.values() if len(v) == cnt for i in v]
        print(f"  CUs holding {cnt} workgroup(s): {sum(1 for v in per_cu.values() if len(v) == cnt)}; their workgroups end at median {np.median(ends):.1f}, max {np.max(ends):.1f} us")
    xs = sorted(set((cu >> 8).tolist()))
    print("  median / max end per XCD:", [(round(float(np.median(t[(cu >> 8) == x, 3])), 1), round(float(t[(cu >> 8) == x, 3].max()), 1)) for x in xs])
    nq = 8
    edges = np.linspace(0, len(t), nq + 1).astype(int)
    print("  median end per eighth of the workgroup indices:", [round(float(np.median(t[edges[i]:edges[i + 1], 3])), 1) for i in range(nq)])
    se = (cu >> 5) & 7
    print("  median end per shader engine id:", {int(x): round(float(np.median(t[se == x, 3])), 1) for x in sorted(set(se.tolist()))})
    print("  workgroups per XCD:", [int(np.sum((cu >> 8) == x)) for x in xs], " first 12 workgroups' (xcc, cu):", [(int(c_ >> 8), int(c_ & 255)) for c_ in cu[:12]])
