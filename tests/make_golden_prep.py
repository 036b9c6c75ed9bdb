"""Generates tests/golden/pabu_prep_seed42.json from the oracle (numpy legacy RNG called directly =
the reference's own RNG library) on the reference's fixture VCF.  Run in the build container only
(/root/reference is not on the GPU box):  python tests/make_golden_prep.py
The VCF itself travels as tests/golden/pabu_test_genotypes.vcf.gz (gzip -9 of the reference's data file, a fixture the
reference's own --test run holds) together with its sample sheet pabu_test_sample_data.txt."""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import popvae_oracle as O  # noqa: E402

VCF = "/root/reference/data/pabu/pabu_test_genotypes.vcf"
gen, samples = O.read_vcf_gt(VCF)
np.random.seed(42)
dc, keep = O.filter_and_impute(gen)
train, test = O.train_test_split(dc.shape[0], 0.9)
pk = O.pack2bit(dc)
out = {
    "source": "oracle on data/pabu/pabu_test_genotypes.vcf, np.random.seed(42) (popvae.py --test)",
    "n_variants": int(gen.shape[0]), "n_samples": int(gen.shape[1]),
    "n_missing_alleles": int((gen < 0).sum()),
    "gt_sha256": hashlib.sha256(gen.tobytes()).hexdigest(),
    "samples_first3": list(samples[:3]), "samples_last": samples[-1],
    "n_snps": int(dc.shape[1]), "keep_sha256": hashlib.sha256(keep.astype(np.int64).tobytes()).hexdigest(),
    "code_counts": [int((dc == v).sum()) for v in (0.0, 0.5, 1.0)],
    "packed_sha256": hashlib.sha256(pk.tobytes()).hexdigest(),
    "train": [int(x) for x in train], "test": [int(x) for x in test],
    "next_randint_1e7": int(np.random.randint(10e6)),
}
with open(os.path.join(os.path.dirname(__file__), "golden", "pabu_prep_seed42.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(out["n_snps"], out["packed_sha256"])
