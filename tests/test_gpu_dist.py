"""Two ranks sharing the one GPU of the test box (gloo moves the exchange buffers): the real engine's staged
C-ABI entry points (pvae_stage_a/b/c, pvae_exchange_buffer, pvae_stage_encode) under SNP sharding, against the
unsharded oracle.  RCCL itself needs one GPU per rank and is exercised by bench.py --gpus N."""
import numpy as np
import pytest

from tests import dist_helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dims,comm", [((40, 1000, 128, 3, 2), "rccl"), ((40, 700, 256, 2, 2), "torch"),
                                       ((40, 900, 128, 2, 2), "rccl-inject"),
                                       # width 512 (BASELINE config 3 is SNP-sharded at that width): grouped decoder pass,
                                       # per-layer narrow kernels, exchange buffers of 32 x 512 (+ 32) floats
                                       ((40, 900, 512, 2, 2), "rccl-inject")])
def test_two_rank_snp_sharding_on_one_gpu(tmp_path, monkeypatch, dims, comm):
    """comm = "torch": host-driven exchanges from the start.  comm = "rccl": the ranks first consider the engine-owned RCCL
    communicator, find that they share a device (RCCL wants one per rank; a doomed ncclCommInitRank would leave half-built
    state behind) and must AGREE on the host-driven path -- the agreement is what is tested here.  comm = "rccl-inject":
    the branch taken on DISTINCT devices when the communicator cannot be set up on one rank -- the device comparison is
    skipped, rank 1 reports librccl as unusable, and both ranks must fall back together before anyone enters the collective
    ncclCommInitRank."""
    import torch.multiprocessing as mp
    monkeypatch.setenv("PVAE_COMM", "rccl" if comm.startswith("rccl") else comm)
    if comm == "rccl-inject":
        monkeypatch.setenv("PVAE_COMM_ASSUME_DISTINCT", "1")
        monkeypatch.setenv("PVAE_COMM_INJECT_FAIL", "1")
    out = str(tmp_path / "res")
    port = 29500 + int(np.random.default_rng().integers(0, 2000))
    mp.spawn(H.sharded_worker, args=(2, port, out, True, dims), nprocs=2, join=True)
    res = [np.load(out + f".{r}.npy", allow_pickle=True)[0] for r in range(2)]
    for r in res:
        assert r["native"] is False and r["comm"]["path"] == "host-staged", r["comm"]
        if comm == "rccl-inject":
            assert "librccl not usable" in r["comm"]["note"], r["comm"]
        if comm == "rccl":
            assert "share a device" in r["comm"]["note"], r["comm"]
    ref = H.reference_run(dims)
    H.compare(res, ref, dims, rtol=1e-4, wtol=2e-2 * 4e-3)


@pytest.mark.parametrize("dims", [(40, 1000, 128, 3, 2), (40, 700, 256, 2, 2), (40, 900, 512, 2, 2)])
def test_two_rank_one_shot_exchange_on_one_gpu(tmp_path, monkeypatch, dims):
    """PVAE_COMM=p2p: the two exchanges of a step go through pvk_p2p_allreduce -- every rank writes its partial into every rank's
    hipIpc-mapped inbox and sums the senders in rank order -- instead of RCCL.  Two PROCESSES on the one GPU of the test box (on a
    node the same mappings cross xGMI): handles carried by torch.distributed, the agreed self-test, then the whole sharded protocol
    with the engine's own calls (train_step, encode, eval_step ... exchange inside the engine) against the unsharded oracle."""
    import torch.multiprocessing as mp
    monkeypatch.setenv("PVAE_COMM", "p2p")
    out = str(tmp_path / "res")
    port = 29500 + int(np.random.default_rng().integers(0, 2000))
    mp.spawn(H.sharded_worker, args=(2, port, out, True, dims), nprocs=2, join=True)
    res = [np.load(out + f".{r}.npy", allow_pickle=True)[0] for r in range(2)]
    for r in res:
        assert r["native"] is True and r["comm"]["path"] == "p2p" and r["comm"]["comm_nranks"] == 2, r["comm"]
    # both ranks hold bit-identical replicated tensors (the senders are summed in rank order on every rank)
    for a, b in zip(res[0]["weights"][1:-2], res[1]["weights"][1:-2]):
        assert np.array_equal(a, b)
    ref = H.reference_run(dims)
    H.compare(res, ref, dims, rtol=1e-4, wtol=2e-2 * 4e-3)


def test_one_shot_exchange_that_gives_up_fails_loudly(tmp_path, monkeypatch):
    """A rank that stalls past the exchange's poll bound must end the run, not poison it: the waiting rank's buffer is NaN-filled
    and its status word raised (pvk_p2p_allreduce), and ShardedEngine.check_exchange -- called once per epoch and per encode --
    raises on every rank (the status used to be read in the set-up self-test only)."""
    import torch.multiprocessing as mp
    monkeypatch.setenv("PVAE_COMM", "p2p")
    out = str(tmp_path / "res")
    port = 29500 + int(np.random.default_rng().integers(0, 2000))
    mp.spawn(H.p2p_giveup_worker, args=(2, port, out), nprocs=2, join=True)
    res = [np.load(out + f".{r}.npy", allow_pickle=True)[0] for r in range(2)]
    for r in res:
        assert r["native"] is True and r["path"] == "p2p" and r["clean_before"] is True, r
        assert r["raised"] is True, r
    assert res[0]["nan"] is True
