"""Workloads of SURVEY.md section 8(d): i.i.d. genomes with a given GC content, scanned with the package's own PAM
scanner (no real E. coli / B. subtilis / yeast genome exists offline), and the one real genome of BASELINE size that
the reference ships as data (P. aeruginosa PAO1, as a 2-bit fixture under tests/golden/)."""
import gzip
import os

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
    # SURVEY.md section 8 "offline real data": P. aeruginosa PAO1 (NC_002516, 6 264 404 bp, GC 66.6 %), the genome in the
    # reference's data directory; the sequence travels as tests/golden/pao1_genome.2bit.gz (tests/golden/make_pao1.py)
    "pao1": dict(fixture="pao1_genome.2bit.gz", records=[6_264_404], gc=0.666, seed=0, pam="NGG", orientation="3prime",
                 guidelen=20, metric="hamming", k=10, editdist=2,
                 label="PAO1: REAL genome P. aeruginosa PAO1 NC_002516 (6 264 404 bp, gc 0.666), NGG 3prime, L=20, hamming, k=10"),
    # ... and the same genome under BASELINE configs[2]'s parameters (Cas12a, Levenshtein)
    "pao1_leven": dict(fixture="pao1_genome.2bit.gz", records=[6_264_404], gc=0.666, seed=0, pam="TTTV", orientation="5prime",
                       guidelen=23, metric="leven", k=5, editdist=3,
                       label="PAO1-leven: REAL genome P. aeruginosa PAO1 NC_002516 (6 264 404 bp, gc 0.666), TTTV 5prime, L=23, leven, k=5"),
    # ... and its 1.16 M NGG targets cut at L = 23 under Levenshtein: ten times configs[2]'s pool, real repeats and codon bias
    # (the displaced-block partition K2P with two levels; profiles/r03/y2_real_genome_leven_k2p_two_levels.txt)
    "pao1_l23": dict(fixture="pao1_genome.2bit.gz", records=[6_264_404], gc=0.666, seed=0, pam="NGG", orientation="3prime",
                     guidelen=23, metric="leven", k=5, editdist=3,
                     label="PAO1-L23: REAL genome P. aeruginosa PAO1 NC_002516 (6 264 404 bp, gc 0.666), NGG 3prime, L=23, leven, k=5"),
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


def read_genome(path):
    """Records of a FASTA or GenBank file (plain or gzipped) as a list of (id, upper-case sequence) -- the input
    step of the reference (core.py:1065-1090 `get_fastas`: Bio.SeqIO.parse + upper()), restated without Biopython:
    FASTA = '>' header lines + sequence lines; GenBank = LOCUS / VERSION headers and the ORIGIN ... // blocks."""
    with open(path, "rb") as fh:
        gz = fh.read(2) == b"\x1f\x8b"
    opener = gzip.open if gz else open
    recs, rid, chunks, in_origin, fmt = [], None, [], False, None
    with opener(path, "rt") as fh:
        for line in fh:
            if fmt is None:
                if not line.strip():
                    continue
                fmt = "fasta" if line.startswith(">") else "genbank"
            if fmt == "fasta":
                if line.startswith(">"):
                    if rid is not None:
                        recs.append((rid, "".join(chunks).upper()))
                    rid, chunks = (line[1:].split() or ["record%d" % len(recs)])[0], []
                else:
                    chunks.append(line.strip())
            else:
                if line.startswith("LOCUS"):
                    rid, chunks, in_origin = line.split()[1], [], False
                elif line.startswith("VERSION") and len(line.split()) > 1:
                    rid = line.split()[1]
                elif line.startswith("ORIGIN"):
                    in_origin = True
                elif line.startswith("//"):
                    if rid is not None:
                        recs.append((rid, "".join(chunks).upper()))
                    rid, chunks, in_origin = None, [], False
                elif in_origin:
                    chunks.append("".join(line.split()[1:]))
    if fmt == "fasta" and rid is not None:
        recs.append((rid, "".join(chunks).upper()))
    return recs


def read_2bit(path, length):
    """Sequence of a tests/golden/*.2bit.gz fixture: 2 bits per base (A0 C1 G2 T3), base i in bits [2(i%4), 2(i%4)+1]
    of byte i//4, `length` bases."""
    with gzip.open(path, "rb") as fh:
        b = np.frombuffer(fh.read(), np.uint8)
    if b.shape[0] * 4 < length:
        raise ValueError("%s holds %d bases, %d expected" % (path, b.shape[0] * 4, length))
    codes = np.stack([(b >> s) & 3 for s in (0, 2, 4, 6)], axis=1).reshape(-1)[:length]
    return np.frombuffer(b"ACGT", np.uint8)[codes].tobytes().decode("ascii")


def fixture_path(fname):
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", fname)


def real_genome_path(name):
    """Path of a real genome for workload `name`, if the box has one: environment variable GM_GENOME_<NAME>
    (e.g. GM_GENOME_S1=/data/ecoli_k12_mg1655.gbk.gz).  SURVEY.md 8(d): real E. coli / B. subtilis / yeast files are
    used iff present on the GPU box; otherwise the synthetic stand-ins are, and the bench line says which."""
    p = os.environ.get("GM_GENOME_" + name.upper(), "")
    return p if p and os.path.exists(p) else None


def make_pool(name, seed_offset=0):
    """-> dict(packed_all uint64 [N_q] in frame order, packed_db uint64 [N_db] first-occurrence unique,
    guidelen, metric, k, label).  With no restriction enzymes every row is a query (core.py:495)."""
    w = WORKLOADS[name]
    rng = np.random.Generator(np.random.PCG64(w["seed"] + seed_offset))
    parts = []
    label = w["label"]
    real = real_genome_path(name) if seed_offset == 0 else None
    if "fixture" in w:
        path = fixture_path(w["fixture"])
        if not os.path.exists(path):
            raise FileNotFoundError("workload %s needs the fixture %s" % (name, path))
        seq = read_2bit(path, w["records"][0])
        parts.append(core.pack_codes(core.scan_targets(seq, w["pam"], w["orientation"], w["guidelen"])["codes"]))
        real = path
    elif real:
        total = 0
        for _, seq in read_genome(real):
            total += len(seq)
            parts.append(core.pack_codes(core.scan_targets(seq, w["pam"], w["orientation"], w["guidelen"])["codes"]))
        label = "%s: REAL genome %s (%d bp), %s %s, L=%d, %s, k=%d" % (
            name.upper(), os.path.basename(real), total, w["pam"], w["orientation"], w["guidelen"], w["metric"], w["k"])
    else:
        for rec_len in w["records"]:
            codes = synth_genome(rec_len, w["gc"], rng)
            parts.append(core.pack_codes(scan_codes(codes, w["pam"], w["orientation"], w["guidelen"])))
    packed_all = np.concatenate(parts)
    _, first = np.unique(packed_all, return_index=True)
    packed_db = packed_all[np.sort(first)]
    return dict(packed_all=packed_all, packed_db=packed_db, guidelen=w["guidelen"], metric=w["metric"],
                k=w["k"], editdist=w["editdist"], label=label, name=name, real_genome=real)
