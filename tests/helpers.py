"""Shared test helpers (no reference code; reads only tests/golden)."""
import gzip
import os
import random

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Rec:
    """Minimal stand-in for a Bio.SeqRecord: .id, .seq, len()."""

    def __init__(self, seq, id):
        self.seq = seq
        self.id = id

    def __len__(self):
        return len(self.seq)


def read_fasta_gz(name):
    recs, rid, chunks = [], None, []
    with gzip.open(os.path.join(GOLDEN, name), "rt") as fh:
        for line in fh:
            line = line.strip()
            if line.startswith(">"):
                if rid is not None:
                    recs.append(Rec("".join(chunks), rid))
                rid, chunks = line[1:].split()[0], []
            elif line:
                chunks.append(line)
    if rid is not None:
        recs.append(Rec("".join(chunks), rid))
    return recs


def carsonella():
    return read_fasta_gz("Carsonella_ruddii.fasta.gz")


def random_pool(n, L, seed, planted=0.01, indel=False):
    """n random L-mers plus a fraction of planted near-duplicates (1-3 substitutions, optional indel)."""
    rng = random.Random(seed)
    seqs = ["".join(rng.choice("ACGT") for _ in range(L)) for _ in range(n)]
    for _ in range(int(n * planted)):
        src = list(seqs[rng.randrange(n)])
        for p in rng.sample(range(L), rng.randint(1, 3)):
            src[p] = rng.choice("ACGT")
        if indel and rng.random() < 0.5:
            p = rng.randrange(L)
            src = src[:p] + [rng.choice("ACGT")] + src[p:-1]
        seqs[rng.randrange(n)] = "".join(src)
    return seqs


def targets_frame(targets, dtype="hamming", pam_orientation=False):
    """A minimal targets dataframe of the shape PamTarget.find_targets returns."""
    import pandas as pd
    n = len(targets)
    df = pd.DataFrame({
        "target": list(targets), "exact_pam": ["AGG"] * n, "start": np.arange(n, dtype=np.uint32) * 30,
        "stop": np.arange(n, dtype=np.uint32) * 30 + len(targets[0]), "strand": [True] * n,
        "pam_orientation": [pam_orientation] * n, "target_seq30": ["N" * 30] * n, "seqid": ["s"] * n,
        "seedseq": [np.nan] * n, "isseedduplicated": [np.nan] * n, "hasrestrictionsite": [np.nan] * n,
        "dtype": [dtype] * n})
    return df.astype({"target": "str", "exact_pam": "category", "start": "uint32", "stop": "uint32",
                      "strand": "bool", "pam_orientation": "bool", "seqid": "category"})
