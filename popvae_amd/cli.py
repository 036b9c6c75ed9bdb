"""Command line of the reference (/root/reference/popvae.py:10-173) in front of the HIP engine.

Same flags, same defaults, same output files; what changes is what runs underneath:

* genotype front end      popvae.py:195-269 -> ``popvae_amd.prep`` (native VCF reader, bit-exact legacy RNG)
* model + fit + callbacks popvae.py:383-526 -> ``popvae_amd.engine`` / ``popvae_amd.fit`` (libpvae.so)
* --search_network_sizes  popvae.py:275-380 -> one network per GPU when launched under torch.distributed.run

Launched as ``python -m torch.distributed.run --nproc-per-node N popvae.py ...`` the model's SNP axis is
sharded over the N GPUs (popvae_amd.dist); rank 0 writes the files.
"""
from __future__ import annotations

import argparse
import os
import re
import subprocess
import sys
import time

import numpy as np


GRID_TIMINGS = []  # (width, depth, seconds, epochs) per --search_network_sizes cell of the last run in this process (measurement)


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--infile", help="path to input genotypes in vcf (.vcf | .vcf.gz), zarr, or .popvae.hdf5 format.")
    p.add_argument("--out", default="vae", help="path for saving output")
    p.add_argument("--patience", default=50, type=int, help="training patience. default=50")
    p.add_argument("--max_epochs", default=500, type=int, help="max training epochs. default=500")
    p.add_argument("--batch_size", default=32, type=int, help="batch size. default=32")
    p.add_argument("--save_allele_counts", default=False, action="store_true",
                   help="save allele counts and and sample IDs to out+'.popvae.hdf5'.")
    p.add_argument("--save_weights", default=False, action="store_true", help="save model weights to out+weights.hdf5.")
    p.add_argument("--seed", default=None, type=int, help="random seed. default: None")
    p.add_argument("--train_prop", default=0.9, type=float,
                   help="proportion of samples to use for training (vs validation). default: 0.9")
    p.add_argument("--search_network_sizes", default=False, action="store_true",
                   help="run grid search over network sizes and use the network with minimum validation loss.")
    p.add_argument("--width_range", default="32,64,128,256,512", type=str)
    p.add_argument("--depth_range", default="3,6,10,20", type=str)
    p.add_argument("--depth", default=6, type=int, help="number of hidden layers. default=6.")
    p.add_argument("--width", default=128, type=int, help="nodes per hidden layer. default=128")
    p.add_argument("--gpu_number", default="0", type=str, help="gpu number to use for training. default: 0")
    p.add_argument("--prediction_freq", default=5, type=int,
                   help="print predictions during training every --prediction_freq epochs.")
    p.add_argument("--max_SNPs", default=None, type=int, help="If not None, randomly select --max_SNPs variants to run.")
    p.add_argument("--latent_dim", default=2, type=int, help="N latent dimensions to fit. default: 2")
    p.add_argument("--PCA", default=False, action="store_true")
    p.add_argument("--n_pc_axes", default=20, type=int)
    p.add_argument("--PCA_scaler", default="Patterson", type=str)
    p.add_argument("--test", default=False, action="store_true", help="fit with the test data")
    p.add_argument("--plot", default=False, action="store_true")
    p.add_argument("--metadata", default=None, help="path to tab-delimited metadata file with column 'sampleID'.")
    # not a flag of the reference: start from exported weights (a Keras <out>_weights.hdf5 / .npz), for per-step parity runs
    p.add_argument("--bce_mode", default="logits", choices=("logits", "probs"),
                   help="(popvae_amd) form of Keras' binary cross-entropy: from the logits (Keras in graph mode, default) or "
                        "from probabilities clipped to [1e-7, 1 - 1e-7]; they differ where |logit| > 16")
    p.add_argument("--init_weights", default=None,
                   help="(popvae_amd) initialise the model from a Keras weight file (.hdf5) or .npz instead of glorot_uniform")
    return p


def load_genotypes(infile, rng, save_allele_counts=False, log=print):
    """popvae.py:195-248 -> (packed 2-bit matrix [N][row_bytes], n_snps, samples)."""
    from . import prep
    if infile.endswith(".popvae.hdf5"):
        from .weights_io import read_popvae_cache
        dc, samples = read_popvae_cache(infile)  # popvae.py:205-209 (plain-Python HDF5 reader, no h5py needed)
        return prep.pack_dc(dc), dc.shape[1], samples
    if infile.endswith(".zarr"):
        try:
            import zarr
        except ImportError as ex:
            raise RuntimeError("zarr input needs the zarr package, which is not importable here") from ex
        callset = zarr.open_group(infile, mode="r")
        gen = np.asarray(callset["calldata/GT"][:], dtype=np.int8)
        samples = np.asarray(callset["samples"][:], dtype=object)
    elif infile.endswith(".vcf") or infile.endswith(".vcf.gz"):
        gen, samples = prep.read_vcf(infile)
    else:
        raise RuntimeError(f"unrecognised --infile extension: {infile}")
    log("counting alleles")
    log("dropping non-biallelic sites")
    log("dropping singletons")
    log("filling missing data with rbinom(2,derived_allele_frequency)")
    packed, n_snps, _, _, _ = prep.filter_impute_pack(gen, rng)
    if save_allele_counts:
        log("saving derived counts for reanalysis")
        from .weights_io import write_popvae_cache
        write_popvae_cache(infile + ".popvae.hdf5", prep.unpack_dc(packed, n_snps), samples)
    return packed, n_snps, samples


def test_infile():
    """The genotypes --test runs on (popvae.py:175-180): data/pabu/pabu_test_genotypes.vcf relative to the working directory,
    as in the reference; where that file is absent, the copy of the same fixture this repository keeps (gzip-compressed)
    under tests/golden/."""
    ref = os.path.join("data", "pabu", "pabu_test_genotypes.vcf")
    if os.path.exists(ref):
        return ref
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    alt = os.path.join(here, "tests", "golden", "pabu_test_genotypes.vcf.gz")
    if os.path.exists(alt):
        return alt
    raise SystemExit(f"--test: neither {ref} nor {alt} exists")


def main(argv=None):
    args = build_parser().parse_args(argv)
    depth_range = np.array([int(x) for x in re.split(",", args.depth_range)])
    width_range = np.array([int(x) for x in re.split(",", args.width_range)])

    if args.test:  # popvae.py:175-180
        os.makedirs("out", exist_ok=True)  # (the reference ships an empty out/ directory)
        rc = subprocess.run([sys.executable, os.path.abspath(sys.argv[0]), "--infile", test_infile(),
                             "--out", "out/pabu_test", "--seed", "42"]).returncode
        raise SystemExit(rc)
    if args.plot and args.metadata is None:
        print("ERROR: `--plot` argument requires `--metadata`")
        raise SystemExit(0)
    if args.infile is None:
        raise SystemExit("--infile is required")

    from . import fit as F
    from . import prep
    from .engine import Engine

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    log = print if rank == 0 else (lambda *a, **k: None)
    if world > 1:
        device_id = local_rank
    else:
        # popvae.py:187 selects the device through CUDA_VISIBLE_DEVICES; here it is the HIP ordinal
        device_id = int(args.gpu_number.split(",")[0]) if args.gpu_number.strip() else 0

    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        # PVAE_DIST_BACKEND=gloo: several ranks on the GPUs that are there (tests / rehearsals on a one-GPU box)
        backend = os.environ.get("PVAE_DIST_BACKEND", "nccl")
        if backend != "nccl":
            local_rank %= max(1, torch.cuda.device_count())
            device_id = local_rank
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)

    host_pg = None
    if world > 1:  # Python objects (grid results, weight shards) travel over a CPU group with a day-long timeout
        import datetime
        host_pg = dist.new_group(backend="gloo", timeout=datetime.timedelta(hours=24))

    seed = args.seed
    if seed is not None:
        os.environ["PYTHONHASHSEED"] = str(seed)
        run_seed = seed
    else:
        # popvae.py:189 leaves numpy/TF on OS entropy; every rank must still agree on the split, the
        # shuffle and the noise, so rank 0 draws one seed for the run
        box = [int.from_bytes(os.urandom(4), "little")]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        run_seed = box[0]
    rng = prep.LegacyRNG(run_seed)  # popvae.py:189-193 (np.random.seed)

    log("\nloading genotypes")
    packed, n_snps, samples = load_genotypes(args.infile, rng, args.save_allele_counts and rank == 0, log)
    if args.max_SNPs is not None:  # popvae.py:250-252
        log("subsetting to " + str(args.max_SNPs) + " SNPs")
        cols = rng.choice(range(n_snps), args.max_SNPs, replace=False)
        packed = prep.gather_columns(packed, cols)
        n_snps = args.max_SNPs
    log("running train/test splits")
    ninds = packed.shape[0]
    train, test = prep.train_test_split(ninds, args.train_prop, rng)
    log("validation samples:" + np.asarray(samples, dtype=object)[test])
    log("running on " + str(n_snps) + " SNPs")
    if n_snps * 1e-3 > 300:
        # not a change of behaviour: the reference (Keras Adam, lr 1e-3) does the same
        log(f"WARNING: with {n_snps} SNPs the first Adam steps at Keras' default learning rate 1e-3 move the first layer's "
            "pre-activations by ~n_snps * lr: at 1M SNPs the loss was seen to grow by four orders of magnitude within one step "
            "(float32 oracle and GPU engine alike) and, on some data, to turn inf / nan.  If it does, thin the input with --max_SNPs.")

    eng_seed = run_seed
    shuffle_seed = run_seed

    def make_single(width, depth):
        e = Engine(n_snps=n_snps, width=width, depth=depth, latent_dim=args.latent_dim, batch_size=args.batch_size,
                   seed=eng_seed, device_id=device_id)
        e.load_genotypes(packed)
        e.init_weights(eng_seed)
        e.set_bce_mode(args.bce_mode)
        return e

    width, depth = args.width, args.depth
    if args.search_network_sizes:  # popvae.py:275-380
        log("running grid search on network sizes")
        gather = None
        if world > 1:
            # every rank runs a different set of cells, and early stopping makes their run times data dependent: the
            # results meet over a CPU (gloo) group with a day-long timeout, not in a RCCL collective under the process
            # group watchdog's 10 minutes
            def gather(obj):
                outl = [None] * world
                dist.all_gather_object(outl, obj, group=host_pg)
                return outl
        _, width, depth = F.grid_search(make_single, width_range, depth_range, train, test, args.max_epochs,
                                        args.patience, n_snps, out=args.out,
                                        perm_fn_factory=lambda i: F.default_perm_fn(None if shuffle_seed is None
                                                                                    else shuffle_seed + 1 + i),
                                        rank=rank, world=world, gather=gather, log=log, timings=GRID_TIMINGS)
        log("best parameters:\nwidth = " + str(width) + "\ndepth = " + str(depth))

    if world > 1:
        from .dist import ShardedEngine
        eng = ShardedEngine(n_snps=n_snps, width=width, depth=depth, latent_dim=args.latent_dim,
                            batch_size=args.batch_size, seed=eng_seed, device_id=device_id)
        log(f"SNP axis sharded over {world} ranks; exchanges: {eng.comm_report()['note']}")
        eng.load_genotypes(packed)
        eng.init_weights(eng_seed)
        eng.eng.set_bce_mode(args.bce_mode)
    else:
        eng = make_single(width, depth)
    if args.init_weights:  # vae.load_weights before fit (per-step parity from a Keras export)
        from .weights_io import load_weights_file
        log("initialising from " + args.init_weights)
        eng.set_weights(load_weights_file(args.init_weights))

    t1 = time.time()
    history, cbs = F.fit(eng, train, test, max_epochs=args.max_epochs, patience=args.patience,
                         prediction_freq=args.prediction_freq, n_samples=ninds, samples=samples, out=args.out,
                         perm_fn=F.default_perm_fn(shuffle_seed), verbose=1 if rank == 0 else 0, log=log,
                         write_preds=rank == 0)
    vaetime = time.time() - t1
    log("VAE run time: " + str(vaetime) + " seconds")

    if rank == 0:
        F.write_history(args.out + "_history.txt", history)
    # popvae.py:507-523: best weights -> latent coordinates of all samples
    eng.restore_best()
    pred = eng.encode(np.arange(ninds, dtype=np.int32))
    full_weights = None
    if args.save_weights:  # under sharding the SNP-axis tensors are gathered to rank 0 (collective: every rank calls it)
        full_weights = eng.get_weights_full(host_pg) if world > 1 else eng.get_weights()
    if rank == 0:
        F.write_latent_coords(args.out + "_latent_coords.txt", pred, samples, args.latent_dim)
        if args.save_weights and full_weights is not None:  # popvae.py:429-435, 525-526: the best epoch's weights stay on disk
            from .weights_io import save_weights_hdf5
            log("weights saved to " + save_weights_hdf5(args.out + "_weights.hdf5", full_weights, depth))
        try:
            F.plot_history(args.out + "_history.pdf", history)
        except Exception as ex:  # matplotlib absent or headless failure: the reference's pdf is cosmetic
            log(f"history plot skipped: {ex}")
        if args.plot:
            log("--plot needs bokeh (scripts/plotvae.py in the reference); skipped")
    if args.PCA:  # popvae.py:528-540, 561-563 (every rank contributes the Gram matrix of its own SNP columns)
        from . import pca as P
        t1 = time.time()
        log("running PCA")
        coords, _ = eng.pca(args.PCA_scaler, args.n_pc_axes) if world > 1 else P.pca(eng, args.PCA_scaler, args.n_pc_axes)
        if rank == 0:
            P.write_pca(args.out + "_pca.txt", coords, samples)
        pcatime = time.time() - t1
        log("PCA run time: " + str(pcatime) + " seconds")
        if rank == 0:
            np.savetxt(X=np.array([vaetime, pcatime]), fname=args.out + "_runtimes.txt")
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
