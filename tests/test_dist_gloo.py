"""world_size 2 and 3 over gloo on the CPU: the product's sharding host protocol (popvae_amd.dist.ShardedEngine:
shard bounds, genotype / weight slicing, stage A -> all-reduce -> stage B -> all-reduce -> stage C) driven with
a NumPy stand-in engine per rank, against the unsharded oracle."""
import numpy as np
import pytest

from popvae_amd import dist as D
from tests import dist_helpers as H


def test_shard_bounds_and_packed_slices():
    for L, world in [(3887, 2), (3887, 8), (1_000_000, 8), (100, 3), (64, 2)]:
        b = D.shard_bounds(L, world)
        assert b[0][0] == 0 and b[-1][1] == L
        for (lo, hi), (lo2, _) in zip(b, b[1:]):
            assert hi == lo2 and lo % 32 == 0
    dc, _ = H.make_problem(N=9, L=333, W=32, D=1, K=1)
    from oracle import popvae_oracle as O
    packed = O.pack2bit(dc)
    for lo, hi in D.shard_bounds(333, 3):
        sh = D.shard_packed(packed, 333, lo, hi)
        assert sh.shape[1] % 8 == 0
        assert np.array_equal(O.unpack2bit(sh, hi - lo) * 0.5, dc[:, lo:hi])
        assert not O.unpack2bit(sh, sh.shape[1] * 4)[:, hi - lo:].any()   # padding codes are zero


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_protocol_matches_unsharded_oracle(tmp_path, world):
    import torch.multiprocessing as mp
    dims = (40, 333, 32, 3, 2)
    out = str(tmp_path / "res")
    port = 29500 + int(np.random.default_rng().integers(0, 2000))
    mp.spawn(H.sharded_worker, args=(world, port, out, False, dims), nprocs=world, join=True)
    res = [np.load(out + f".{r}.npy", allow_pickle=True)[0] for r in range(world)]
    ref = H.reference_run(dims)
    H.compare(res, ref, dims, rtol=1e-5, wtol=1e-6)   # float32 exchange buffers between float64 stages
