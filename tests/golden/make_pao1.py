#!/usr/bin/env python3
"""Real-genome fixture at the scale of the BASELINE configs: P. aeruginosa PAO1 (NC_002516, 6 264 404 bp, GC 66 %),
the genome the reference ships as data (guidemaker/data/Pseudomonas_aeruginosa.gbk.gz; SURVEY.md section 8's "offline
real data" row).

Runs ONLY in the build container (needs /root/reference), like make_golden.py, whose by-file-path import of the
reference's core.py (third-party modules absent from the image replaced by inert placeholders) it reuses.  Writes
  * pao1_genome.2bit.gz   the record's sequence, 2 bits per base (A0 C1 G2 T3, base i in bits [2(i%4), 2(i%4)+1] of
                          byte i//4); data, not code.  The record has no base outside ACGT (checked here).
  * pao1_summary.json     what the REFERENCE's PamTarget.find_targets / check_restriction_enzymes /
                          find_unique_near_pam make of it (NGG 3prime, L=20, lsr=10, NRAGCA): row counts, SHA-256 of
                          every column, first and last rows.
"""
import gzip
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402

GBK = f"{mg.REF}/guidemaker/data/Pseudomonas_aeruginosa.gbk.gz"


def read_genbank_sequences(path):
    """(id, sequence) of every record: LOCUS name + the ORIGIN block (digits and blanks dropped, upper-cased)."""
    recs, name, chunks, in_origin = [], None, [], False
    with gzip.open(path, "rt") as fh:
        for line in fh:
            if line.startswith("LOCUS"):
                name = line.split()[1]
            elif line.startswith("VERSION") and len(line.split()) > 1:
                name = line.split()[1]          # Biopython's record.id is the VERSION accession
            elif line.startswith("ORIGIN"):
                in_origin, chunks = True, []
            elif line.startswith("//"):
                recs.append((name, "".join(chunks).upper()))
                in_origin = False
            elif in_origin:
                chunks.append("".join(line.split()[1:]))
    return recs


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def sha_strings(values):
    return hashlib.sha256(("\n".join(values) + "\n").encode()).hexdigest()


def main():
    recs = read_genbank_sequences(GBK)
    assert len(recs) == 1, [r[0] for r in recs]
    rid, seq = recs[0]
    assert set(seq) <= set("ACGT"), sorted(set(seq))
    codes = np.frombuffer(seq.encode(), np.uint8)
    lut = np.zeros(256, np.uint8)
    for i, c in enumerate(b"ACGT"):
        lut[c] = i
    c = lut[codes]
    pad = (-len(c)) % 4
    c4 = np.concatenate([c, np.zeros(pad, np.uint8)]).reshape(-1, 4)
    packed = (c4[:, 0] | (c4[:, 1] << 2) | (c4[:, 2] << 4) | (c4[:, 3] << 6)).astype(np.uint8)
    path = os.path.join(mg.OUT, "pao1_genome.2bit.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as fh:
        fh.write(packed.tobytes())
    print("wrote", path, len(packed), "bytes before gzip")

    mg._install_placeholders()
    core = mg._load("guidemaker.core", f"{mg.REF}/guidemaker/core.py")
    pt = core.PamTarget("NGG", "3prime", "hamming")
    df = pt.find_targets(seq_record_iter=[mg._Rec(seq, rid)], target_len=20)
    tp = core.TargetProcessor(targets=df, lsr=10, editdist=2, knum=10)
    tp.check_restriction_enzymes(["NRAGCA"])
    tp.find_unique_near_pam()
    t = tp.targets
    q = t.loc[(t["isseedduplicated"] == False) | (t["hasrestrictionsite"] == False)]   # noqa: E712  (core.py:495)
    rows = lambda fr: [[r.target, str(r.exact_pam), int(r.start), int(r.stop), bool(r.strand), r.target_seq30,   # noqa: E731
                        bool(r.isseedduplicated), bool(r.hasrestrictionsite)] for r in fr.itertuples()]
    out = {
        "provenance": dict(mg.PROVENANCE, generator="tests/golden/make_pao1.py",
                           source="guidemaker/data/Pseudomonas_aeruginosa.gbk.gz (NC_002516, data shipped by the reference)"),
        "record": {"id": rid, "length": len(seq), "sha256_ascii": hashlib.sha256(seq.encode()).hexdigest(),
                   "gc": (seq.count("G") + seq.count("C")) / len(seq)},
        "params": {"pam": "NGG", "orientation": "3prime", "target_len": 20, "lsr": 10, "restriction": ["NRAGCA"]},
        "n_targets": int(len(t)), "n_unique": int(t["target"].nunique()), "n_query": int(len(q)),
        "n_seed_unique": int((t["isseedduplicated"] == False).sum()),      # noqa: E712
        "n_restriction": int((t["hasrestrictionsite"] == True).sum()),     # noqa: E712
        "sha256": {
            "target": sha_strings(t["target"].tolist()),
            "exact_pam": sha_strings([str(x) for x in t["exact_pam"].tolist()]),
            "target_seq30": sha_strings(t["target_seq30"].tolist()),
            "start_u32": sha(t["start"].to_numpy().astype(np.uint32)),
            "stop_u32": sha(t["stop"].to_numpy().astype(np.uint32)),
            "strand_u8": sha(t["strand"].to_numpy().astype(np.uint8)),
            "isseedduplicated_u8": sha(t["isseedduplicated"].to_numpy().astype(np.uint8)),
            "hasrestrictionsite_u8": sha(t["hasrestrictionsite"].to_numpy().astype(np.uint8)),
        },
        "columns": ["target", "exact_pam", "start", "stop", "strand", "target_seq30", "isseedduplicated",
                    "hasrestrictionsite"],
        "first_rows": rows(t.head(5)), "last_rows": rows(t.tail(5)),
    }
    mg.dump("pao1_summary.json", out)
    print({k: out[k] for k in ("n_targets", "n_unique", "n_query", "n_seed_unique", "n_restriction")})


if __name__ == "__main__":
    main()
