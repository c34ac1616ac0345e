"""Synthetic workloads of SURVEY.md section 8(d): i.i.d. genomes with a given GC content, scanned with
the package's own PAM scanner (no real E. coli / B. subtilis / yeast genome exists offline)."""
import numpy as np

from . import core

# name: (record lengths, gc, seed, pam, orientation, guidelen, metric, k, editdist)
WORKLOADS = {
    # BASELINE.json configs[1]: E. coli K-12 MG1655 stand-in
    "s1": dict(records=[4_641_652], gc=0.508, seed=1, pam="NGG", orientation="3prime", guidelen=20,
               metric="hamming", k=10, editdist=2,
               label="S1: synthetic 4.64 Mb gc 0.508 (E. coli K-12 stand-in), NGG 3prime, L=20, hamming, k=10"),
    # configs[2]: B. subtilis 168 stand-in, Cas12a
    "s2": dict(records=[4_215_606], gc=0.435, seed=2, pam="TTTV", orientation="5prime", guidelen=23,
               metric="leven", k=5, editdist=3,
               label="S2: synthetic 4.22 Mb gc 0.435 (B. subtilis 168 stand-in), TTTV 5prime, L=23, leven, k=5"),
    # configs[3]: S. cerevisiae S288C stand-in, 17 records
    "s3": dict(records=[230_218, 813_184, 316_620, 1_531_933, 576_874, 270_161, 1_090_940, 562_643,
                        439_888, 745_751, 666_816, 1_078_177, 924_431, 784_333, 1_091_291, 948_066,
                        85_779],
               gc=0.383, seed=3, pam="NGG", orientation="3prime", guidelen=20, metric="hamming", k=5,
               editdist=2,
               label="S3: synthetic 12.16 Mb in 17 records gc 0.383 (S. cerevisiae stand-in), NGG 3prime, L=20, hamming, k=5"),
    # configs[4]: 100 Mb uniform genome, degenerate PAM
    "s4": dict(records=[100_000_000], gc=0.5, seed=4, pam="NNGRRT", orientation="3prime", guidelen=20,
               metric="hamming", k=20, editdist=2,
               label="S4: synthetic 100 Mb uniform, NNGRRT 3prime, L=20, hamming, k=20"),
    # small variant for tests / smoke
    "tiny": dict(records=[200_000], gc=0.45, seed=9, pam="NGG", orientation="3prime", guidelen=20,
                 metric="hamming", k=10, editdist=2, label="tiny: synthetic 200 kb, NGG 3prime, L=20"),
}


def synth_genome(length, gc, rng):
    """i.i.d. bases, P(G)=P(C)=gc/2, P(A)=P(T)=(1-gc)/2, as base codes A0 C1 G2 T3."""
    p = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])
    return rng.choice(4, size=length, p=p).astype(np.uint8)


def scan_codes(codes, pam, orientation, L):
    """PAM scan on a base-code array (the array-level path of core.scan_targets): returns the (n, L)
    target codes in the reference's row order (forward matches, then reverse matches)."""
    seq = np.frombuffer(b"ACGT", np.uint8)[codes].tobytes().decode("ascii")
    return core.scan_targets(seq, pam, orientation, L)["codes"]


def make_pool(name, seed_offset=0):
    """-> dict(packed_all uint64 [N_q] in frame order, packed_db uint64 [N_db] first-occurrence unique,
    guidelen, metric, k, label).  With no restriction enzymes every row is a query (core.py:495)."""
    w = WORKLOADS[name]
    rng = np.random.Generator(np.random.PCG64(w["seed"] + seed_offset))
    parts = []
    for rec_len in w["records"]:
        codes = synth_genome(rec_len, w["gc"], rng)
        parts.append(core.pack_codes(scan_codes(codes, w["pam"], w["orientation"], w["guidelen"])))
    packed_all = np.concatenate(parts)
    _, first = np.unique(packed_all, return_index=True)
    packed_db = packed_all[np.sort(first)]
    return dict(packed_all=packed_all, packed_db=packed_db, guidelen=w["guidelen"], metric=w["metric"],
                k=w["k"], editdist=w["editdist"], label=w["label"], name=name)
