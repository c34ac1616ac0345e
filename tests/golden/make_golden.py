#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference itself.

Runs ONLY in the build container (needs /root/reference); nothing here is imported
by the product, the tests or the bench.  The reference travels nowhere: this script
imports two of its modules *by file path*, feeds them inputs, and writes
inputs + outputs as data files.

What is imported
  * /root/reference/guidemaker/cfd_score_calculator.py  (stdlib only)  -> calc_cfd
  * /root/reference/guidemaker/core.py with the third-party modules that are absent
    from this image (nmslib, Bio, pybedtools, altair, onnxruntime, importlib_resources)
    replaced by inert placeholders in sys.modules.  Only pure-Python code paths are
    exercised: PamTarget.find_targets, TargetProcessor.check_restriction_enzymes,
    find_unique_near_pam, _one_hot_encode, extend_ambiguous_dna.  The k-NN itself
    (nmslib) can NOT be run here; k-NN expectations in the fixtures are limited to the
    known answers asserted by the reference's own tests (tests/test_core.py:126,346-347).
  Caveats recorded in every fixture's "provenance": Bio.Seq.reverse_complement is the
  placeholder's (str.translate + reverse); `regex` is 2026.7.19, not the pinned 2020.11.13.
"""
import gzip
import importlib.util
import json
import os
import random
import sys
import types

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

PROVENANCE = {
    "generator": "tests/golden/make_golden.py",
    "reference": "USDA-ARS-GBRU/GuideMaker v0.4.2 at /root/reference (imported by file path)",
    "caveats": [
        "Bio.Seq replaced by a placeholder whose reverse_complement is translate+reverse",
        "regex module 2026.7.19 instead of pinned 2020.11.13",
        "nmslib absent: no k-NN output of the reference is contained in these fixtures",
    ],
}


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


# ---------------------------------------------------------------- placeholders
class _Seq(str):
    _T = str.maketrans("ACGTMRWSYKVHDBXNacgtmrwsykvhdbxn", "TGCAKYWSRMBDHVXNtgcakywsrmbdhvxn")

    def reverse_complement(self):
        return _Seq(str(self).translate(self._T)[::-1])


class _Rec:
    def __init__(self, seq, id):
        self.seq = _Seq(seq)
        self.id = id

    def __len__(self):
        return len(self.seq)


def _install_placeholders():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("nmslib")
    bio = mod("Bio")
    seqmod = mod("Bio.Seq", Seq=_Seq)
    bio.Seq = seqmod
    bio.SeqIO = mod("Bio.SeqIO")
    mod("Bio.SeqUtils", gc_fraction=lambda s: (s.count("G") + s.count("C")) / max(1, len(s)))
    mod("pybedtools", BedTool=object)
    mod("altair")
    mod("onnxruntime")
    gm = mod("guidemaker")
    gm.__path__ = []  # mark as package
    gm.doench_predict = mod("guidemaker.doench_predict")
    gm.cfd_score_calculator = _load(
        "guidemaker.cfd_score_calculator", f"{REF}/guidemaker/cfd_score_calculator.py")


def read_fasta(path):
    recs, name, chunks = [], None, []
    with open(path) as fh:
        for line in fh:
            line = line.strip()
            if line.startswith(">"):
                if name is not None:
                    recs.append(_Rec("".join(chunks), name))
                name, chunks = line[1:].split()[0], []
            elif line:
                chunks.append(line)
    if name is not None:
        recs.append(_Rec("".join(chunks), name))
    return recs


def dump(name, obj):
    path = os.path.join(OUT, name)
    if name.endswith(".gz"):
        with gzip.GzipFile(path, "wb", mtime=0) as fh:
            fh.write(obj.encode())
    else:
        with open(path, "w") as fh:
            json.dump(obj, fh, indent=1, sort_keys=True)
            fh.write("\n")
    print("wrote", path)


def main():
    _install_placeholders()
    cfd = sys.modules["guidemaker.cfd_score_calculator"]
    core = _load("guidemaker.core", f"{REF}/guidemaker/core.py")

    # ---- 1. CFD weight table (data: guidemaker/data/cfd_data.json, CC0) -----------------
    mm, pam = cfd.get_mm_pam_scores()
    # table[pos-1][r][d] with r = guide RNA base index in ACGU, d = DNA base index in ACGT
    # (d is the base that appears in the key, i.e. the COMPLEMENT of the off-target base);
    # absent keys are the Watson-Crick pairings, never looked up by calc_cfd -> stored as 1.0
    table = [[[1.0] * 4 for _ in range(4)] for _ in range(20)]
    for key, val in mm.items():
        pair, pos = key.split(",")
        r = "ACGU".index(pair[1])
        d = "ACGT".index(pair[4])
        table[int(pos) - 1][r][d] = val
    dump("cfd_table.json", {"provenance": PROVENANCE, "layout": "mm[pos-1][rRNA ACGU][dDNA ACGT]",
                            "mm": table, "n_keys": len(mm), "pam_unused": pam})

    # ---- 2. CFD known answers -----------------------------------------------------------
    rng = random.Random(20240607)
    cases = [
        ("GCATGCACAGCTAGCATGCATGCAGCT", "GCATGCACAGCTAGCATGCATGCAGCG"),   # tests/test_core.py:266
        ("AAATGGTACGTTATGTGTTA", "AACAGTAAAATGGTTTAATG"),
        ("ACGTACGTAC", "ACGTACGTAA"),
        ("ACGTACGTACGTACGTACGT", "ACGTACGTACGTACGTACGT"),
    ]
    for L in range(10, 28):
        for _ in range(70):
            g = "".join(rng.choice("ACGT") for _ in range(L))
            o = list(g)
            nmut = rng.choice([0, 1, 1, 2, 2, 3, 4, 6, L])
            for p in rng.sample(range(L), min(nmut, L)):
                o[p] = rng.choice("ACGT")
            cases.append((g, "".join(o)))
    recs = []
    for g, o in cases:
        v = cfd.calc_cfd(g, o, mm_scores=mm)
        recs.append({"guide": g, "off": o, "cfd": float(v), "cfd_hex": float(v).hex()})
    dump("cfd_kat.json", {"provenance": PROVENANCE, "cases": recs})

    # ---- 3. PamTarget.find_targets known answers on small contigs ------------------------
    small = {}
    seq1 = ("AATGATCTGGATGCACATGCACTGCTCCAAGCTGCATGAAAAGTACAAAGCACGTTATTAGATGGTAACAATGATCTGGATGCACATG"
            "CACTGCTCCAAGCTGCATGAAAAGTACAAAGCACGTTATTAGATGGTGGGAAC")
    distseq = ("CGTAGCTAGTCACTAGCTGACAGCAAGGTTTTTCGTAGCTAGACACTAGCTGACAGCAAGGTTTTTTCGTAGCTAGTCACTAGCTG"
               "ACTAGCAAGG")
    rng2 = random.Random(7)
    rnd = "".join(rng2.choice("ACGT") for _ in range(600)) + "NNNN" + \
          "".join(rng2.choice("ACGT") for _ in range(300))
    for cname, seq, pam_, orient, L in [
        ("testseq1_NGG_5p_L6", seq1, "NGG", "5prime", 6),
        ("testseq1_NGG_3p_L6", seq1, "NGG", "3prime", 6),
        ("distseq_NGG_3p_L20", distseq, "NGG", "3prime", 20),
        ("rnd_TTTV_5p_L23", rnd, "TTTV", "5prime", 23),
        ("rnd_NNGRRT_3p_L20", rnd, "NNGRRT", "3prime", 20),
        ("rnd_NGG_5p_L10", rnd, "NGG", "5prime", 10),
        ("rnd_NGG_3p_L27", rnd, "NGG", "3prime", 27),
    ]:
        pt = core.PamTarget(pam_, orient, "hamming")
        df = pt.find_targets(seq_record_iter=[_Rec(seq, cname)], target_len=L)
        small[cname] = {
            "seq": seq, "pam": pam_, "orientation": orient, "target_len": L,
            "rows": [[r.target, str(r.exact_pam), int(r.start), int(r.stop), bool(r.strand),
                      bool(r.pam_orientation), r.target_seq30] for r in df.itertuples()],
        }
    dump("find_targets_kat.json", {"provenance": PROVENANCE,
                                   "columns": ["target", "exact_pam", "start", "stop", "strand",
                                               "pam_orientation", "target_seq30"],
                                   "cases": small})

    # ---- 4. Carsonella: find_targets + filters (the hot path's real-data input) ----------
    carso = read_fasta(f"{REF}/tests/test_data/Carsonella_ruddii.fasta")
    summary = {}
    for tag, orient in [("NGG5p", "5prime"), ("NGG3p", "3prime")]:
        pt = core.PamTarget("NGG", orient, "hamming")
        df = pt.find_targets(seq_record_iter=carso, target_len=20)
        tp = core.TargetProcessor(targets=df, lsr=10, editdist=2, knum=10)
        tp.check_restriction_enzymes(["NRAGCA"])
        tp.find_unique_near_pam()
        t = tp.targets
        lines = ["target\tstart\tstop\tstrand\tisseedduplicated\thasrestrictionsite"]
        for r in t.itertuples():
            lines.append("%s\t%d\t%d\t%d\t%d\t%d" % (r.target, r.start, r.stop, int(r.strand),
                                                   int(r.isseedduplicated), int(r.hasrestrictionsite)))
        dump(f"carsonella_targets_{tag}.tsv.gz", "\n".join(lines) + "\n")
        q = t.loc[(t["isseedduplicated"] == False) | (t["hasrestrictionsite"] == False)]
        summary[tag] = {"n_targets": int(len(t)), "n_unique": int(t["target"].nunique()),
                        "n_query": int(len(q)), "first_target": t["target"].iat[0],
                        "n_bed": int((t["isseedduplicated"] == False).sum())}
    dump("carsonella_summary.json", {"provenance": PROVENANCE, "summary": summary,
                                     "filters": {"lsr": 10, "restriction": ["NRAGCA"]}})

    # ---- 5. the reference tests' own k-NN known answers (tests/test_core.py) -------------
    pt = core.PamTarget("NGG", "3prime", "hamming")
    dfh = pt.find_targets(seq_record_iter=[_Rec(distseq, "distseq")], target_len=20)
    tp = core.TargetProcessor(targets=dfh, lsr=0, editdist=1, knum=3)
    dump("knn_kat.json", {
        "provenance": PROVENANCE,
        "three_row": {   # tests/test_core.py:67-82,116-126
            "targets": ["AAATGGTACGTTATGTGTTA", "AAATGGTACGTTATGTGTTA", "AACAGTAAAATGGTTTAATG"],
            "lsr": 10, "editdist": 2, "knum": 2, "restriction": ["NRAGCA"], "dtype": "hamming",
            "pam_orientation": False,
            "expect": {"key": "AAATGGTACGTTATGTGTTA", "dist_index": 1, "dist_value": 12},
        },
        "levin": {       # tests/test_core.py:319-347
            "seq": distseq, "pam": "NGG", "orientation": "3prime", "target_len": 20,
            "targets": dfh["target"].tolist(),
            "lsr": 0, "editdist": 1, "knum": 3,
            "expect_key": "CTAGTCACTAGCTGACAGCA",
            "expect_leven_dist": [0, 1, 2], "expect_hamming_dist": [0, 1, 16],
        },
        "one_hot": {     # core.py:379-386
            "in": ["ACGT", "TTGA"], "out": tp._one_hot_encode(["ACGT", "TTGA"]),
        },
        "extend_ambiguous_dna": {"NGG": core.extend_ambiguous_dna("NGG"),
                                 "NRAGCA": core.extend_ambiguous_dna("NRAGCA")},
    })


if __name__ == "__main__":
    main()
