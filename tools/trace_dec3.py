"""Cycles per phase of a pvk_dec3 visit (wave 0 of the first 256 workgroups), shader clock:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DPV_WTRACE -DPV_DEC3_PROF -o tools/_build/libpvae_d3prof.so popvae_amd/csrc/pvae_engine.hip
  PVAE_LIB=tools/_build/libpvae_d3prof.so python tools/trace_dec3.py [n_snps] [batch]"""
import ctypes as C, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popvae_amd import engine as E
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
N, W, D = 256, 128, 6
g = np.random.default_rng(0)
packed = (g.integers(0, 256, size=(N, (L + 31) // 32 * 8), dtype=np.uint8) & 0x55)
e = E.Engine(n_snps=L, width=W, depth=D, latent_dim=2, batch_size=B, seed=1)
e.load_genotypes(packed); e.init_weights(1); e.lr = 1e-6
rows = np.arange(B, dtype=np.int32)
for _ in range(int(os.environ.get("STEPS", "200"))):  # long enough for the clock to settle under load
    e.train_step(rows)
e.synchronize()
n = 2 * 1024 * 6
buf = (C.c_ulonglong * n)()
E.lib().pvae_debug_wtrace(buf, n)
a = np.array(buf[:256 * 12], dtype=np.int64).reshape(256, 12)
a = a[a[:, 10] > 0]
per = a[:, :10] / a[:, 10:11]
names = ["requests", "logits", "sigmoid", "waitB+stage", "sync C", "dg", "dV+post", "sync A", "update", "rotate"]
if B > 64 and not os.environ.get("PVAE_NO_DEC3B"):  # pvk_dec3b's stamps
    names = ["requests", "logits", "sigmoid", "waitB+posts", "sync C", "dV", "dg", "-", "update", "rotate"]
print(f"L={L} B={B}: {len(a)} workgroups, visits per wave {a[:, 10].mean():.1f}; cycles per visit (mean / p10 / p90 over workgroups)")
for i, nm in enumerate(names):
    print(f"  {nm:12s} {per[:, i].mean():8.0f} {np.percentile(per[:, i], 10):8.0f} {np.percentile(per[:, i], 90):8.0f}")
print(f"  {'total':12s} {per.sum(1).mean():8.0f}")
clk = a[:, :10].sum(1) / a[:, 11] * 100.0  # MHz
print(f"  in-kernel clock (s_memtime / s_memrealtime x 100 MHz): median {np.median(clk):.0f} MHz, p10 {np.percentile(clk, 10):.0f}, p90 {np.percentile(clk, 90):.0f}")
ef = np.array(buf[4096:4096 + 2048], dtype=np.int64).reshape(1024, 2)
ef = ef[ef[:, 1] > 0]
if len(ef):
    clk = ef[:, 0] / ef[:, 1] * 100.0
    nblk = (L + 31) // 32
    T = 2 if B <= 64 else 4
    per_wave = nblk * 4 * 16 * T / (len(ef) * 4)  # matrix instructions per wave
    print(f"pvk_enc_fwd<4,{T}>: {len(ef)} workgroups, sweep {np.median(ef[:, 1]) / 100:.1f} us (median), in-kernel clock median {np.median(clk):.0f} MHz "
          f"(p10 {np.percentile(clk, 10):.0f}, p90 {np.percentile(clk, 90):.0f}); {np.median(ef[:, 0]) / per_wave:.1f} cycles per matrix instruction per wave")
