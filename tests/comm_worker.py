"""Child process of tests/test_gpu_comm.py: the engine's own RCCL communicator with ONE rank on the test box's GPU, in the
library order of real use (torch first: popvae_amd.dist / bench.py import it before the engine, so PyTorch's librccl and HIP
runtime are the ones in the process).  Prints OK and exits 0 when the communicator path equals the plain engine bit for bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

torch.cuda.init()
import numpy as np  # noqa: E402

from popvae_amd import engine, prep  # noqa: E402
from tests.test_gpu_parity import _problem  # noqa: E402

N, L, W, D, K = 120, 3000, 128, 3, 2
dc, w = _problem(N, L, W, D, K, seed=31)
g = np.random.default_rng(3)
perm = g.permutation(70).astype(np.int32)
val = np.arange(70, 120, dtype=np.int32)  # two validation batches: they share one pass of the read-only multi-tile kernels
outs = []
for with_comm in (False, True):
    e = engine.Engine(n_snps=L, width=W, depth=D, latent_dim=K, batch_size=32, seed=5)
    if with_comm:
        e.comm_init(engine.Engine.comm_unique_id())
        n = e.exchange_buffer(0)[1]
        e.exchange_write(0, np.arange(n, dtype=np.float32))
        e.comm_allreduce(0)  # a one-rank SUM is the identity
        assert np.array_equal(e.exchange_read(0), np.arange(n, dtype=np.float32))
    e.load_genotypes(prep.pack_dc(dc))
    e.set_weights([x.astype(np.float32) for x in w])
    res = [e.train_epoch(perm, val) for _ in range(2)]
    res.append(e.train_step(perm[:32]))
    res.append(e.eval_step(val[:32]))
    mu, _, _ = e.encode(np.arange(N, dtype=np.int32))
    outs.append((res, e.get_weights(), mu))
    e.close()
# batches above 32 rows: the two exchanges of a pass carry all its row tiles (48 rows: one pass of two tiles; 60 rows at
# batch_size 160 would be one pass of four; validation batches: 32-row tiles), one Adam pass per batch
# ("p2p": the one-shot exchange with this rank as its only peer -- its own inbox, both parities, exchanges of 32 T rows)
for B in (48, 160):
    big = []
    for with_comm in (False, True, "p2p"):
        e = engine.Engine(n_snps=L, width=W, depth=D, latent_dim=K, batch_size=B, seed=5)
        if with_comm == "p2p":
            e.p2p_connect([e.p2p_prepare()])
        elif with_comm:
            e.comm_init(engine.Engine.comm_unique_id())
        e.load_genotypes(prep.pack_dc(dc))
        e.set_weights([x.astype(np.float32) for x in w])
        res = [e.train_epoch(perm, val), e.train_step(perm[:min(B, 70)]), e.eval_step(val[:min(B, 50)])]
        if with_comm == "p2p":
            assert not e.p2p_gave_up()
        big.append((res, e.get_weights()))
        e.close()
    for other in big[1:]:
        assert big[0][0] == other[0], (B, big[0][0], other[0])
        for a, b in zip(big[0][1], other[1]):
            assert np.array_equal(a, b)
# widths above 128 (BASELINE config 3 is width 512, SNP-sharded): the grouped decoder pass and the per-layer narrow kernels
# behind the in-stream all-reduces -- exchange buffers of 32 x 256 / 32 x 512 (+ 32) floats -- with a one-rank RCCL communicator
# and with the one-shot exchange (this rank as its only peer): epochs with two validation batches, single steps, encode
for W2, D2 in ((256, 3), (512, 2)):
    dc2, w2 = _problem(N, 2500, W2, D2, K, seed=37)
    wide = []
    for with_comm in (False, True, "p2p"):
        e = engine.Engine(n_snps=2500, width=W2, depth=D2, latent_dim=K, batch_size=32, seed=5)
        if with_comm == "p2p":
            e.p2p_connect([e.p2p_prepare()])
        elif with_comm:
            e.comm_init(engine.Engine.comm_unique_id())
        e.load_genotypes(prep.pack_dc(dc2))
        e.set_weights([x.astype(np.float32) for x in w2])
        res = [e.train_epoch(perm, val) for _ in range(2)]
        res.append(e.train_step(perm[:32]))
        res.append(e.train_step(perm[:11]))
        res.append(e.eval_step(val[:32]))
        mu, _, _ = e.encode(np.arange(N, dtype=np.int32))
        if with_comm == "p2p":
            assert not e.p2p_gave_up()
        wide.append((res, e.get_weights(), mu))
        e.close()
    for other in wide[1:]:
        assert wide[0][0] == other[0], (W2, wide[0][0], other[0])
        assert all(np.isfinite(x) for r_ in other[0] for x in r_), other[0]
        for a, b in zip(wide[0][1], other[1]):
            assert np.array_equal(a, b)
        assert np.array_equal(wide[0][2], other[2])
(ra, wa, ma), (rb, wb, mb) = outs
assert ra == rb, (ra, rb)
for a, b in zip(wa, wb):
    assert np.array_equal(a, b)
assert np.array_equal(ma, mb)
print("OK", flush=True)
