"""Host-side mirror of GuideMaker's operator interface for the neighbour-search path.

Same class / method names, arguments and error behaviour as the reference's
``guidemaker/core.py`` (v0.4.2), so code and tests written against ``guidemaker.core`` run
against this module; the work happens in libgmhip.so on an MI355X.  Mirrored symbols:

  PamTarget(pam, pam_orientation, dtype).find_targets   reference core.py:39-292  (host, numpy)
  TargetProcessor.__init__/__len__/__str__               reference core.py:304-352
  TargetProcessor.check_restriction_enzymes              reference core.py:354-377 (host, pandas)
  TargetProcessor.find_unique_near_pam                   reference core.py:388-416 (host, pandas)
  TargetProcessor.create_index                           reference core.py:418-467 -> gm_index_create
  TargetProcessor.get_neighbors                          reference core.py:471-523 -> gm_knn_query
  TargetProcessor.export_bed                             reference core.py:525-543
  TargetProcessor.get_control_seqs                       reference core.py:545-633 -> gm_nearest_dist
  cfd_score(df)                                          reference core.py:1129-1148 -> gm_cfd_batch
  format_neighbor_columns(tl, guides)                    reference core.py:913-931 (the two apply() closures of
                                                         Annotation._format_guide_table), vectorised

Deliberate, documented differences (DESIGN.md "Parity contract"):
  * the search is exact, not HNSW: the NMSLIB block of the config is read and ignored;
  * the index order is first occurrence in the targets frame, not ``list(set(...))`` hash order,
    and ties are ordered (distance, index);
  * ``neighbors[...]["seqs"]`` holds the TRUE neighbour sequences (the reference looks nmslib ids
    up in the wrong table, core.py:513);
  * there is no CPU fallback: without the HIP library / a GPU every search call raises.
"""
import hashlib
import itertools
import json
import logging
import os
import statistics
import time
from collections.abc import Mapping
from copy import deepcopy
from itertools import product
from typing import List

import numpy as np
import pandas as pd
import yaml

from . import _lib

logger = logging.getLogger("guidemaker")      # the reference's logger name (cli.py:20-22)

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
CONFIG_PATH = os.path.join(_DATA, "config_default.yaml")

_IUPAC = {
    "A": "A", "C": "C", "G": "G", "T": "T", "M": "AC", "R": "AG", "W": "AT", "S": "CG", "Y": "CT",
    "K": "GT", "V": "ACG", "H": "ACT", "D": "AGT", "B": "CGT", "X": "GATC", "N": "GATC",
}
_COMP = str.maketrans("ACGTMRWSYKVHDBXN", "TGCAKYWSRMBDHVXN")
_CODE = np.full(256, 4, np.uint8)
for _i, _c in enumerate("ACGT"):
    _CODE[ord(_c)] = _i


def reverse_complement(seq: str) -> str:
    return seq.translate(_COMP)[::-1]


def extend_ambiguous_dna(seq: str) -> List[str]:
    """All concrete sequences of an IUPAC string (reference core.py:1093-1124, same order)."""
    return ["".join(p) for p in product(*[_IUPAC[b] for b in seq])]


# ------------------------------------------------------------------------------------------------
# PAM scan (producer of the hot path's input; host side, vectorised numpy)
# ------------------------------------------------------------------------------------------------
def _match_positions(codes: np.ndarray, pam: str) -> np.ndarray:
    """Start offsets of all (overlapping) occurrences of the IUPAC pattern; non-ACGT never matches."""
    n, plen = codes.shape[0], len(pam)
    if n < plen:
        return np.empty((0,), np.int64)
    ok = np.ones(n - plen + 1, dtype=bool)
    for p, sym in enumerate(pam):
        allowed = np.zeros(5, dtype=bool)
        for b in _IUPAC[sym]:
            allowed["ACGT".index(b)] = True
        ok &= allowed[codes[p:n - plen + 1 + p]]
    return np.flatnonzero(ok).astype(np.int64)


def _windows(codes: np.ndarray, starts: np.ndarray, width: int) -> np.ndarray:
    return codes[starts[:, None] + np.arange(width)[None, :]]


def scan_targets(seq: str, pam: str, pam_orientation: str, target_len: int):
    """Array-level PAM scan of one record.

    Returns dict with 'codes' (n, L) uint8 base codes of the targets as reported (i.e. reverse
    complemented for the minus strand), 'start', 'stop' (int64), 'strand' (bool), 'mstart', 'mend'
    (match coordinates on the forward sequence).  Row order = forward matches ascending, then reverse
    matches ascending, as the reference emits them (core.py:252-284).
    """
    codes = _CODE[np.frombuffer(seq.encode("ascii", "replace"), dtype=np.uint8)]
    n, plen, L = codes.shape[0], len(pam), target_len
    out = {k: [] for k in ("codes", "start", "stop", "strand", "mstart", "mend")}
    for forward in (True, False):
        pat = pam if forward else reverse_complement(pam)
        ms = _match_positions(codes, pat)
        me = ms + plen
        # target lies after the match (5prime forward / 3prime reverse) or before it
        after = (pam_orientation == "5prime") == forward
        tstart = me if after else ms - L
        keep = (tstart >= 0) & (tstart + L <= n)
        ms, me, tstart = ms[keep], me[keep], tstart[keep]
        win = _windows(codes, tstart, L) if tstart.size else np.empty((0, L), np.uint8)
        good = (win < 4).all(axis=1)
        ms, me, tstart, win = ms[good], me[good], tstart[good], win[good]
        if not forward:
            win = (3 - win)[:, ::-1]
        out["codes"].append(win)
        out["start"].append(tstart)
        out["stop"].append(tstart + L)
        out["strand"].append(np.full(tstart.shape, forward, dtype=bool))
        out["mstart"].append(ms)
        out["mend"].append(me)
    return {k: np.concatenate(v) for k, v in out.items()}


def scan_targets_gpu(seq: str, pam: str, pam_orientation: str, target_len: int, device: int = 0):
    """The same scan on the GPU (libgmhip gm_pam_count / gm_pam_emit).  Returns the dict of scan_targets
    with 'packed' (uint64, the boundary format of include/gmhip.h) in place of 'codes'."""
    five = pam_orientation == "5prime"
    packed, tstart, mstart, n_fwd = _lib.pam_scan(seq, pam, five, target_len, device)
    strand = np.zeros(packed.shape, dtype=bool)
    strand[:n_fwd] = True
    return {"packed": packed, "start": tstart, "stop": tstart + target_len, "strand": strand,
            "mstart": mstart, "mend": mstart + len(pam)}


_RC_BYTES = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTMRWSYKVHDBXNacgtmrwsykvhdbxn", b"TGCAKYWSRMBDHVXNtgcakywsrmbdhvxn"):
    _RC_BYTES[_a] = _b


def _cut_windows(seq: str, rc: str, lo: np.ndarray, width: int, forward: np.ndarray) -> List[str]:
    """seq[lo:lo+width] per row (reverse-complemented where forward is False) as a list of str.  rc = the reverse
    complement of the whole record: a reverse row's window is a plain slice of it.  Rows are slices of two Python
    strings (no per-row array work); a forward window that leaves the record keeps python slice semantics (negative
    start wraps, end clips), a reverse one is cut from the forward strand and complemented, as the reference does."""
    n = len(seq)
    fw = np.flatnonzero(forward)
    rv = np.flatnonzero(~forward)
    out = [None] * lo.shape[0]
    a_fw = lo[fw].tolist()
    for i, a in zip(fw.tolist(), a_fw):
        out[i] = seq[a:a + width]
    lo_rv = lo[rv]
    inside = (lo_rv >= 0) & (lo_rv + width <= n)
    if inside.all():
        for i, a in zip(rv.tolist(), (n - lo_rv - width).tolist()):
            out[i] = rc[a:a + width]
    else:
        for i, a, ok in zip(rv.tolist(), lo_rv.tolist(), inside.tolist()):
            out[i] = rc[n - a - width:n - a] if ok else reverse_complement(seq[a:a + width])
    return out


def codes_to_strings(codes: np.ndarray) -> List[str]:
    if codes.shape[0] == 0:
        return []
    asc = np.frombuffer(b"ACGT", np.uint8)[codes]
    L = codes.shape[1]
    return np.ascontiguousarray(asc).view("S%d" % L).ravel().astype(str).tolist()


def pack_codes(codes: np.ndarray) -> np.ndarray:
    """(n, L) base codes -> uint64, base p in bits [2p, 2p+1] (the layout of include/gmhip.h)."""
    L = codes.shape[1]
    shifts = (2 * np.arange(L, dtype=np.uint64))[None, :]
    return (codes.astype(np.uint64) << shifts).sum(axis=1, dtype=np.uint64)


class PamTarget:
    """A PAM plus the targets it defines (mirror of reference core.py:39-292)."""

    def __init__(self, pam: str, pam_orientation: str, dtype: str) -> None:
        for letter in pam.upper():
            assert letter in _IUPAC
        assert pam_orientation in ["3prime", "5prime"]
        self.pam: str = pam.upper()
        self.pam_orientation: str = pam_orientation
        self.dtype: str = dtype

    def __str__(self) -> str:
        return "A PAM object: {self.pam}".format(self=self)

    def find_targets(self, seq_record_iter: object, target_len: int) -> pd.DataFrame:
        """All targets next to the PAM on both strands, as the reference's dataframe (core.py:83-292): same rows,
        row order, columns and dtypes.  The strings of the frame (target, exact PAM, 30-mer context) are cut as slices
        of the record and of its reverse complement; the columns are assembled once, in their final dtypes."""
        five = self.pam_orientation == "5prime"
        plen = len(self.pam)
        cols = {k: [] for k in ("target", "exact_pam", "start", "stop", "strand", "target_seq30", "seqcode")}
        packed_parts, seqids = [], []
        # the PAM scan runs on the GPU when one is visible (and the guide fits the packed format);
        # the numpy scanner is the host implementation of the same step
        on_gpu = target_len <= _lib.MAX_GUIDELEN and plen <= 16 and _lib.device_count() > 0
        for record in seq_record_iter:
            seq = str(record.seq)
            if on_gpu:
                res = scan_targets_gpu(seq, self.pam, self.pam_orientation, target_len)
            else:
                res = scan_targets(seq, self.pam, self.pam_orientation, target_len)
            if res["start"].size == 0:
                continue
            strand = res["strand"]
            ms, me, ts = res["mstart"], res["mend"], res["start"]
            # 30-mer context: rows whose window leaves the record keep python slice semantics (a negative start
            # wraps, the end clips: core.py:156,184,211,237)
            lo = np.where(strand == five, ms - 3, me - 27)
            rc = reverse_complement(seq)
            cols["target_seq30"] += _cut_windows(seq, rc, lo, 30, strand)
            cols["exact_pam"] += _cut_windows(seq, rc, ms, plen, strand)
            cols["target"] += _cut_windows(seq, rc, ts, target_len, strand)
            cols["start"].append(ts)
            cols["stop"].append(res["stop"])
            cols["strand"].append(strand)
            cols["seqcode"].append(np.full(strand.shape, len(seqids), dtype=np.int32))
            seqids.append(record.id)
            if target_len <= _lib.MAX_GUIDELEN:
                packed_parts.append(res["packed"] if on_gpu else pack_codes(res["codes"]))
        if not seqids:
            raise ValueError("No objects to concatenate")   # what pd.concat([]) raises in the reference
        n = len(cols["target"])
        strand = np.concatenate(cols["strand"])
        # 'seqid' as a category: categories sorted and unique, as astype("category") makes them
        cats = sorted(set(seqids))
        code_of = np.array([cats.index(x) for x in seqids], dtype=np.int32)
        df_targets = pd.DataFrame({
            "target": pd.Series(cols["target"], dtype=object),
            "exact_pam": pd.Categorical(cols["exact_pam"]),
            "start": np.concatenate(cols["start"]).astype(np.uint32),
            "stop": np.concatenate(cols["stop"]).astype(np.uint32),
            "strand": strand.astype(bool),
            "pam_orientation": np.full(n, five, dtype=bool),
            "target_seq30": pd.Series(cols["target_seq30"], dtype=object),
            "seqid": pd.Categorical.from_codes(code_of[np.concatenate(cols["seqcode"])], categories=cats),
            # the reference assigns NaN and casts: seedseq -> the string 'nan', isseedduplicated -> True (core.py:287-289)
            "seedseq": pd.Series(np.full(n, "nan", dtype=object), dtype=object),
            "hasrestrictionsite": np.full(n, np.nan),
            "isseedduplicated": np.ones(n, dtype=bool),
            "dtype": pd.Categorical.from_codes(np.zeros(n, dtype=np.int8), categories=[self.dtype]),
        })
        if packed_parts:
            # side-car for TargetProcessor: the 2-bit packed form of the 'target' column (the boundary format of
            # include/gmhip.h), so that the filters, create_index and get_neighbors do not re-encode the strings
            # (the array itself lives in this module's side-car cache; the frame carries a string token -- arrays in
            # `attrs` break pd.concat / pd.merge, which compare attrs dicts with ==, and are deep-copied per operation)
            df_targets.attrs["gm_packed_token"] = _packed_side_car_put(np.concatenate(packed_parts))
            df_targets.attrs["gm_guidelen"] = int(target_len)
        return df_targets


# ------------------------------------------------------------------------------------------------
# TargetProcessor
# ------------------------------------------------------------------------------------------------
def _load_cfd_table():
    with open(os.path.join(_DATA, "cfd_mm_table.json")) as fh:
        return np.asarray(json.load(fh)["mm"], dtype=np.float64)


class NeighborMap(Mapping):
    """Read-only dict view of the k-NN result with the reference's schema

        neighbors[seq] == {"target": seq, "neighbors": {"seqs": [...], "dist": [...]}}

    (reference core.py:513-523).  The reference materialises one dict + two lists per retained guide in
    a Python loop; at 10^6 guides that loop costs seconds while the GPU scan takes milliseconds, so the
    entries are built on access from the result arrays instead.  Duplicate candidate strings collapse
    onto one key, the last occurrence winning, as with the reference's dict assignment.  Vectorised
    consumers use .keys_array / .row_of / .ids / .dists / .counts directly.
    """

    def __init__(self, query_targets, keep, ids, dists, counts, db_targets, q_packed=None, db_packed=None,
                 guidelen=None):
        self._q = query_targets                  # list[str], candidate order
        self.keep = np.asarray(keep, dtype=np.int64)
        self.ids, self.dists, self.counts = ids, dists, counts
        self._db = db_targets                    # object array: index id -> sequence
        self._row = None                         # key -> row in the result arrays (lazy)
        self._qp, self._dbp, self._L = q_packed, db_packed, guidelen      # packed forms (optional)

    def packed_pairs(self, keys):
        """(guides uint64 [n], similar guides uint64 [n, k], guide length) of the given keys when the packed forms
        are known and every row has k hits; None otherwise."""
        if self._qp is None or self._dbp is None or not len(keys):
            return None
        rowmap = self._rows()
        rows = np.fromiter((rowmap[g] for g in keys), dtype=np.int64, count=len(keys))
        if not (self.counts[rows] == self.ids.shape[1]).all():
            return None
        return self._qp[rows], self._dbp[self.ids[rows]], self._L

    def _rows(self):
        if self._row is None:
            q = self._q
            self._row = {q[i]: i for i in self.keep.tolist()}      # later rows overwrite earlier ones
        return self._row

    def row_of(self, key):
        return self._rows()[key]

    @property
    def keys_array(self):
        """Retained candidate sequences (object array, candidate order, duplicates kept)."""
        return np.asarray(self._q, dtype=object)[self.keep]

    def __getitem__(self, key):
        i = self._rows()[key]
        c = int(self.counts[i])
        return {"target": key, "neighbors": {"seqs": self._db[self.ids[i, :c]].tolist(),
                                             "dist": self.dists[i, :c].tolist()}}

    def __iter__(self):
        return iter(self._rows())

    def __len__(self):
        return len(self._rows())

    def __contains__(self, key):
        return key in self._rows()

    def __repr__(self):
        return "NeighborMap(%d guides)" % len(self)


class TargetProcessor:
    """A set of guide RNA targets and its MI355X-resident neighbour index."""

    def __init__(self, targets: pd.DataFrame, lsr: int, editdist: int = 2, knum: int = 2,
                 device: int = None, shard_candidates: bool = None) -> None:
        self.targets = targets
        self.lsr: int = lsr
        self.editdist: int = editdist
        self.knum: int = knum
        self.nmslib_index: object = None      # name kept; holds a guidemaker_amd._lib.Index
        self.neighbors: dict = {}
        self.closest_neighbor_df: pd.DataFrame = None
        self.ncontrolsearched: int = None
        self.gc_percent: float = None
        self.genomesize: float = None
        self.pam_orientation: bool = targets['pam_orientation'].iat[0]
        self._device_arg, self._shard_arg = device, shard_candidates
        self.device = 0 if device is None else device
        # multi-GPU: one process per GPU, every process holds the SAME targets frame; opt in with
        # shard_candidates=True (or GM_SHARD_CANDIDATES=1) -- an initialised process group alone does not switch it
        # on, because ranks that work on different genomes must not enter a collective together
        if shard_candidates is None:
            shard_candidates = os.environ.get("GM_SHARD_CANDIDATES", "0") not in ("", "0")
        self.shard_candidates = bool(shard_candidates)
        self._db_targets = None               # index id -> sequence
        self._packed_cache = None             # ((id(frame), rows), (packed, L) or None): see _packed_targets
        self._db_packed = None                # packed form of _db_targets
        self._guidelen = None
        self.last_knn = None                  # (ids, dists, counts) of the last get_neighbors

    def __str__(self) -> str:
        return "TargetList: contains a set of {} potential PAM targets".format(len(self.targets))

    def __len__(self) -> int:
        return len(self.targets)

    def _is_hamming(self) -> bool:
        # anything but "hamming" selects Levenshtein (reference core.py:448,496,509)
        return self.targets['dtype'].iat[0] == "hamming"

    def _packed_targets(self):
        """(packed uint64 [len(targets)], guide length) of the frame's 'target' column, encoded ONCE per frame: taken
        from the side-car find_targets keeps for the frame (token in frame.attrs, packed_side_car(); checked against the
        strings on up to 256 probe rows), else packed here; cached until the column object or its length changes.  None when the targets have no
        common length <= 27 or contain non-ACGT letters."""
        t = self.targets['target']
        n = len(t)
        key = (id(self.targets), n)
        if self._packed_cache is not None and self._packed_cache[0] == key:
            return self._packed_cache[1]
        out = None
        if n:
            L = len(t.iat[0])
            side = packed_side_car(self.targets)
            if (side is not None and self.targets.attrs.get("gm_guidelen") == L and len(side) == n
                    and 1 <= L <= _lib.MAX_GUIDELEN):
                # probe rows: the ends and up to 256 evenly spaced rows (a full check would cost as much as packing)
                probe = sorted(set(np.linspace(0, n - 1, num=min(n, 256), dtype=np.int64).tolist()) | {0, n // 3, n // 2, n - 1})
                try:
                    if (_lib.pack([t.iat[i] for i in probe], L) == side[probe]).all():
                        out = (np.ascontiguousarray(side, dtype=np.uint64), L)
                except ValueError:
                    out = None
            if out is None and 1 <= L <= _lib.MAX_GUIDELEN:
                try:
                    out = (_lib.pack(t.tolist(), L), L)
                except ValueError:
                    out = None
        self._packed_cache = (key, out)
        return out

    def _packed_for_filters(self):
        """The packed targets when the GPU filters apply (a GPU is visible, one common length <= 27, ACGT only), else
        None -> host (pandas) form of the same filter."""
        if len(self.targets) == 0 or _lib.device_count() <= 0:
            return None
        return self._packed_targets()

    def check_restriction_enzymes(self, restriction_enzyme_list: list = []) -> None:
        """Flag targets containing a restriction site or its reverse complement (reference
        core.py:354-377).  On the GPU the IUPAC sites are matched on the packed words (gm_site_flags)."""
        records = []
        for record in set(restriction_enzyme_list):
            for letter in record.upper():
                assert letter in _IUPAC
            records += [record.upper(), reverse_complement(record.upper())]
        if len(records) == 0:
            self.targets['hasrestrictionsite'] = False
            return
        pk = self._packed_for_filters() if len(records) <= 64 else None
        if pk is not None:
            self.targets['hasrestrictionsite'] = _lib.site_flags(pk[0], pk[1], records, self.device)
            return
        sites = []
        for record in records:
            sites += extend_ambiguous_dna(record)
        self.targets['hasrestrictionsite'] = self.targets['target'].str.contains('|'.join(sites))

    def _one_hot_encode(self, seq_list: List[object]) -> np.ndarray:
        """Encoding step of the reference (core.py:379-386); here: 2-bit packing for the GPU."""
        return _lib.pack(seq_list, self._target_len(seq_list))

    def _target_len(self, seq_list=None) -> int:
        if self._guidelen is not None:
            return self._guidelen
        if seq_list is not None and len(seq_list):
            return len(seq_list[0])
        return len(self.targets['target'].iat[0])

    def find_unique_near_pam(self) -> None:
        """Flag targets whose PAM-proximal seed (length lsr) already occurred earlier in the frame
        (reference core.py:388-416).  On the GPU: gm_seed_duplicated on the packed words."""
        cached = self._packed_targets()       # (the copy below holds the same rows: carry the encoded column over)
        self.targets = deepcopy(self.targets)
        self._packed_cache = ((id(self.targets), len(self.targets)), cached)
        t = self.targets['target']
        if self.lsr == 0:
            seed = t
        elif self.pam_orientation == True:   # 5prime: PAM sits in front of the target
            seed = t.str[0:self.lsr]
        else:
            seed = t.apply(lambda s: s[(len(s) - self.lsr):])
        self.targets['seedseq'] = seed
        pk = self._packed_for_filters()
        if pk is not None:
            L = pk[1]
            if self.lsr == 0:
                start, count = 0, L
            elif self.pam_orientation == True:
                start, count = 0, min(self.lsr, L)
            else:      # s[(len - lsr):] -- a negative start counts from the end, as Python slices do
                start = L - self.lsr
                if start < 0:
                    start = max(L + start, 0)
                count = L - start
            self.targets['isseedduplicated'] = _lib.seed_duplicated(pk[0], L, start, count, self.device)
            return
        self.targets['isseedduplicated'] = self.targets['seedseq'].duplicated()

    def create_index(self, configpath: str, num_threads=2):
        """Upload the de-duplicated targets to HBM (replaces the nmslib HNSW build)."""
        with open(configpath) as cf:
            config = yaml.safe_load(cf)
        # read for interface compatibility; an exact scan has no M / efC / post
        _ = config['NMSLIB']['M'], config['NMSLIB']['efc'], config['NMSLIB']['post']
        # optional GPU block (absent from the reference's config files): device ordinal / sharding, unless the
        # constructor already said
        gpu_cfg = config.get('GPU') or {}
        if self._device_arg is None and 'DEVICE' in gpu_cfg:
            self.device = int(gpu_cfg['DEVICE'])
        if self._shard_arg is None and 'SHARD_CANDIDATES' in gpu_cfg and "GM_SHARD_CANDIDATES" not in os.environ:
            self.shard_candidates = bool(gpu_cfg['SHARD_CANDIDATES'])
        pk = self._packed_targets()
        if pk is not None:
            # de-duplicate on the packed words: first-occurrence order (the reference's list(set(...)) order is
            # arbitrary, core.py:446)
            _, first = np.unique(pk[0], return_index=True)
            first.sort()
            packed = pk[0][first]
            self._db_targets = self.targets['target'].to_numpy(dtype=object)[first]
        else:
            self._db_targets = np.asarray(pd.unique(self.targets['target']), dtype=object)
            packed = None
        self._guidelen = len(self._db_targets[0])
        if self._guidelen > _lib.MAX_GUIDELEN:
            raise ValueError("guide length %d exceeds %d" % (self._guidelen, _lib.MAX_GUIDELEN))
        metric = _lib.METRIC_HAMMING if self._is_hamming() else _lib.METRIC_LEVEN
        if packed is None:
            packed = _lib.pack(self._db_targets.tolist(), self._guidelen)
        self._db_packed = packed
        if self.nmslib_index is not None:
            self.nmslib_index.close()
        t0 = time.perf_counter()
        self.nmslib_index = _lib.Index(packed, self._guidelen, metric, self.device)
        logger.info("create_index: N_db=%d unique targets of %d (guide length %d, %s) resident on GPU %d in %.3f s",
                    len(self._db_targets), len(self.targets), self._guidelen,
                    "hamming" if metric == _lib.METRIC_HAMMING else "leven", self.device, time.perf_counter() - t0)

    def _query_targets(self) -> list:
        # the OR is the reference's (core.py:495)
        t = self.targets
        return t.loc[(t['isseedduplicated'] == False) | (t['hasrestrictionsite'] == False)]['target'].tolist()

    def get_neighbors(self, configpath, num_threads=2) -> None:
        """k nearest targets of every candidate guide; keep guides whose 2nd hit is >= editdist away.

        Writes self.neighbors[seq] = {"target": seq, "neighbors": {"seqs": [...], "dist": [...]}}.
        """
        with open(configpath) as cf:
            config = yaml.safe_load(cf)
        _ = config['NMSLIB']['ef']
        t = self.targets
        qmask = ((t['isseedduplicated'] == False) | (t['hasrestrictionsite'] == False)).to_numpy()      # core.py:495
        unique_targets = t['target'].to_numpy(dtype=object)[qmask].tolist()
        pk = self._packed_targets()
        qpacked = pk[0][qmask] if pk is not None else _lib.pack(unique_targets, self._guidelen)
        t0 = time.perf_counter()
        sharded = False
        if self.shard_candidates:
            from . import distributed
            sharded = distributed.is_distributed()
        if sharded:
            # one process per GPU: scan this rank's shard of the candidates on the device, one all-gather of the lists
            ids, dists, counts = distributed.sharded_knn_index(self.nmslib_index, qpacked, self.knum, self.device)
        else:
            ids, dists, counts = self.nmslib_index.knn(qpacked, self.knum)
        dt = time.perf_counter() - t0
        n_q, n_db = len(unique_targets), len(self._db_targets)
        logger.info("get_neighbors: N_q=%d candidates x N_db=%d targets, k=%d, %s: %.3f s (%.3e guide pairs/s)%s",
                    n_q, n_db, self.knum, "hamming" if self._is_hamming() else "leven", dt,
                    n_q * n_db / max(dt, 1e-9), " [sharded]" if sharded else "")
        self.last_knn = (ids, dists, counts)
        if len(unique_targets) and (self.knum < 2 or int(counts.min()) < 2):
            raise IndexError("list index out of range")      # editdist[1] in the reference (core.py:512)
        # reference: hamming compares the one-hot distance (2 x mismatches) with 2 * editdist and
        # reports int(x / 2); in true units both reduce to the same test / value
        keep = np.flatnonzero(dists[:, 1] >= self.editdist) if len(unique_targets) else np.empty(0, np.int64)
        self.neighbors = NeighborMap(unique_targets, keep, ids, dists, counts, self._db_targets, q_packed=qpacked,
                                     db_packed=self._db_packed, guidelen=self._guidelen)

    def export_bed(self) -> object:
        """BED-like frame (chrom, chromstart, chromend, name, strand) of the targets whose seed is unique, ordered by
        record and start -- what the reference's export_bed returns (core.py:525-543)."""
        t = self.targets
        rows = ~t["isseedduplicated"].to_numpy(dtype=bool)
        bed = pd.DataFrame({
            "chrom": t["seqid"].array[rows],      # (stays categorical: the sort below follows the category order)
            "chromstart": t["start"].to_numpy()[rows],
            "chromend": t["stop"].to_numpy()[rows],
            "name": t["target"].to_numpy()[rows],
            "strand": pd.Series(np.where(t["strand"].to_numpy(dtype=bool)[rows], "+", "-"), dtype=object).to_numpy(),
        }, index=t.index[rows])
        return bed.sort_values(by=["chrom", "chromstart"])

    @staticmethod
    def _draw_control_codes(count: int, length: int, gc: float) -> np.ndarray:
        """`count` random sequences as (count, length) base codes (A0 C1 G2 T3), consuming numpy's global
        RNG exactly like the reference's per-sequence np.random.choice(["G","C","A","T"], size=length,
        p=...) calls (core.py:590-592): choice() draws one uniform per base and maps it through the
        cumulative distribution."""
        p = np.array([gc / 2, gc / 2, (1 - gc) / 2, (1 - gc) / 2], dtype=np.float64)
        cdf = p.cumsum()
        cdf /= cdf[-1]
        u = np.random.random_sample((count, length))
        idx = cdf.searchsorted(u, side='right')               # index into ["G", "C", "A", "T"]
        return np.array([2, 1, 0, 3], dtype=np.uint8)[idx]

    def _control_match_probability(self, gc: float) -> float:
        """P(a random control carries the same base as a pool target at a position) = sum_b f_pool(b) f_control(b), the
        pool's composition measured on (a sample of) its packed targets, the controls' from the GC fraction they are
        drawn with.  0.25 for uniform bases; larger on a GC-skewed genome, where mismatches are rarer."""
        packed = getattr(self, "_db_packed", None)
        if packed is None or len(packed) == 0:
            return 0.25
        sample = np.asarray(packed[:: max(1, len(packed) // 200_000)], dtype=np.uint64)
        shifts = (2 * np.arange(self._guidelen, dtype=np.uint64))[None, :]
        codes = ((sample[:, None] >> shifts) & np.uint64(3)).astype(np.uint8)      # A0 C1 G2 T3
        f_pool = np.bincount(codes.reshape(-1), minlength=4).astype(np.float64)
        f_pool /= f_pool.sum()
        f_ctrl = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])
        return float((f_pool * f_ctrl).sum())

    @staticmethod
    def _draw_control_seqs(count: int, length: int, gc: float) -> List[str]:
        """The same draw as strings."""
        return codes_to_strings(TargetProcessor._draw_control_codes(count, length, gc))

    def get_control_seqs(self, seq_record_iter: object, configpath, length: int = 20, n: int = 10,
                         num_threads: int = 2) -> pd.DataFrame:
        """Random sequences with the genome's GC content that are farthest from every target."""
        with open(configpath) as fh:
            control_cfg = yaml.safe_load(fh)["CONTROL"]
        wanted_dist = control_cfg["MINIMUM_HMDIST"]
        multiples = list(control_cfg["CONTROL_SEARCH_MULTIPLE"])
        largest_multiple = max(multiples)
        # Opt-in (GPU: CONTROL_FAIL_FAST: true in the config; the reference has no such key): give up with the reference's
        # IndexError as soon as TWO consecutive draws' exact nearest-distance histograms rule the largest draw out, instead
        # of drawing every multiple first.  Off by default: the mirror then walks CONTROL_SEARCH_MULTIPLE exactly as the
        # reference does (core.py:586-620) and leaves the global RNG in the state the reference would.
        with open(configpath) as fh:
            fail_fast = bool((yaml.safe_load(fh).get("GPU") or {}).get("CONTROL_FAIL_FAST", False))
        ruled_out = 0

        # genome-wide GC fraction, length-weighted over the records; per record Bio.SeqUtils.gc_fraction's default:
        # (G+C+S) / (G+C+S+A+T+W+U), other letters ignored (reference core.py:570-578)
        genome_len, gc_weighted = 0, 0.0
        for record in seq_record_iter:
            s = str(record.seq)
            strong = sum(map(s.count, "CGScgs"))
            weak = sum(map(s.count, "ATWUatwu"))
            gc_weighted += (strong / (strong + weak) if strong + weak else 0) * len(s)
            genome_len += len(s)
        gc = gc_weighted / genome_len
        self.gc_percent = gc * 100
        self.genomesize = genome_len / (1024 * 1024)

        if self.nmslib_index is None:
            raise ValueError("get_control_seqs: no index -- call create_index() first")
        if length != self._guidelen:
            # (the kernels compare equal-length sequences; nmslib's leven would accept other lengths)
            raise ValueError("get_control_seqs: length=%d differs from the indexed guide length %d"
                             % (length, self._guidelen))
        minimum_hmdist = 0
        sm_count = 0
        search_mult = 0
        n_searched = 0
        t0 = time.perf_counter()
        hamming = self._is_hamming()
        # loop condition (and its walk off the end of the list) as in the reference, core.py:586
        while minimum_hmdist < wanted_dist or search_mult == largest_multiple:
            if sm_count >= len(multiples):
                # the reference's CONTROL_SEARCH_MULTIPLE[sm_count] raises a bare "list index out of range" here
                # (core.py:587, re-raised at :619-620); same exception type, with the cause spelled out
                raise IndexError(
                    "list index out of range: no %d control sequences at distance >= CONTROL.MINIMUM_HMDIST=%s "
                    "from every target after CONTROL.CONTROL_SEARCH_MULTIPLE=%s x n draws (best minimum %d over "
                    "%d targets; the search is exact, so on large genomes lower MINIMUM_HMDIST in the config)"
                    % (n, wanted_dist, multiples, minimum_hmdist, len(self._db_targets)))
            search_mult = multiples[sm_count]
            # candidates stay numeric (codes -> packed words) until the n winners are known: at the
            # largest multiple this is 10^4 * n sequences
            codes = self._draw_control_codes(n * search_mult, length, gc)
            cpacked = pack_codes(codes)
            sharded = False
            if self.shard_candidates:
                from . import distributed
                sharded = distributed.is_distributed()
            if sharded:
                # one process per GPU: every rank has drawn the same sequences (same seed, same stream of the
                # global RNGs -- checked by fingerprint), scans its shard of them, one all-gather of int32 distances
                d1 = distributed.sharded_nearest_index(self.nmslib_index, cpacked, self.device)
            else:
                d1 = self.nmslib_index.nearest(cpacked)
            # sorted(..., reverse=True, key=dist) is stable: ties stay in draw order
            order = np.argsort(-d1.astype(np.int64), kind="stable")[0:n]
            sort_seq = codes_to_strings(codes[order])
            if hamming:
                sort_dist = [float(d1[i]) for i in order.tolist()]   # reference: raw / 2 -> float
            else:
                sort_dist = [int(d1[i]) for i in order.tolist()]
            minimum_hmdist = int(min(sort_dist))
            sm_count += 1
            n_searched += n * search_mult
            if fail_fast and minimum_hmdist < wanted_dist and sm_count < len(multiples):
                # The search is exact, so the nearest-distance histogram of a draw says whether the LARGEST multiple could
                # yield n sequences at >= MINIMUM_HMDIST (the reference's HNSW misses neighbours and reports too-large
                # distances, which is how its loop ends; an exact search on a large genome draws 1.1 10^4 n more
                # sequences and then walks off the list all the same).  The extrapolation is a model (independent
                # positions with the MEASURED match probability of controls against the pool): two draws in a row must
                # agree, with a factor of a hundred to spare, before the search is abandoned.
                p_match = self._control_match_probability(gc)
                expect = _controls_expected_at(d1, wanted_dist, length, n * largest_multiple, len(self._db_targets), p_match=p_match)
                ruled_out = ruled_out + 1 if expect < 0.01 * n else 0
                if ruled_out >= 2:
                    raise IndexError(
                        "list index out of range: no %d control sequences at distance >= CONTROL.MINIMUM_HMDIST=%s "
                        "from every target can be expected within CONTROL.CONTROL_SEARCH_MULTIPLE=%s x n draws: the "
                        "exact nearest-distance histograms of the first %d draws (best minimum %d over %d targets) put "
                        "%.3g such sequences into the largest draw (the search is exact, so on large genomes lower "
                        "MINIMUM_HMDIST in the config; GPU.CONTROL_FAIL_FAST asked for this early answer)"
                        % (n, wanted_dist, multiples, n_searched, minimum_hmdist, len(self._db_targets), expect))
        dt = time.perf_counter() - t0
        logger.info("get_control_seqs: %d random %d-mers x N_db=%d targets in %.3f s (%.3e guide pairs/s), minimum "
                    "distance %s", n_searched, length, len(self._db_targets), dt,
                    n_searched * len(self._db_targets) / max(dt, 1e-9), min(sort_dist))

        self.ncontrolsearched = search_mult * n      # (the LAST multiple's draw, as the reference reports it)
        table = pd.DataFrame({
            "name": ["Cont-" + hashlib.md5(seq.encode()).hexdigest() for seq in sort_seq],
            "Sequences": sort_seq,
            "Hamming distance": sort_dist,
        })
        return (min(sort_dist), statistics.median(sort_dist), table)


def _controls_expected_at(d1, wanted, L, draws, n_db, min_count=30, p_match=0.25) -> float:
    """Expected number of random controls at nearest distance >= `wanted` among `draws` draws, from the exact nearest
    distances d1 of one draw.  P(nearest >= s) = exp(-lambda(s - 1)), lambda(r) = expected targets within r of a random
    sequence; it is read off the draw at the largest s <= wanted that still has `min_count` sequences, and carried to
    `wanted` with the growth of lambda under independent positions that mismatch with probability q = 1 - p_match,
    lambda(r) / lambda(r - 1) = P(Bin(L, q) <= r) / P(Bin(L, q) <= r - 1); p_match is the MEASURED match probability of
    controls against the pool (TargetProcessor._control_match_probability; 0.25 = uniform bases).  With uniform bases on a
    GC-rich pool the growth -- it scales with q / (1 - q): 3 at q = 0.75, 2.45 at 70 % GC -- would be overestimated and
    the expected controls come out more than ten times too low; clustered pools (repeats) grow slower still
    than independent positions, so this remains a MODEL, not a bound: the caller asks two draws to agree with a factor
    of a hundred to spare, and only when the config opts in."""
    d1 = np.asarray(d1, dtype=np.int64)
    m = int(d1.shape[0])
    if m == 0:
        return float(draws)
    wanted = int(wanted)
    tail = np.array([(d1 >= s).sum() for s in range(wanted + 1)], dtype=np.int64)
    if tail[wanted] >= min_count:
        return float(draws) * float(tail[wanted]) / m
    s_star = max((s for s in range(1, wanted + 1) if tail[s] >= min_count), default=0)
    if s_star == 0:                       # (every control hits a target exactly: nothing to extrapolate from)
        return 0.0 if n_db > 0 else float(draws)
    lam = -np.log(min(float(tail[s_star]) / m, 1.0 - 0.5 / m))
    q = min(max(1.0 - float(p_match), 1e-6), 1.0 - 1e-6)

    def within(r):                        # P(Bin(L, q) <= r)
        from math import comb
        return sum(comb(L, i) * q ** i * (1.0 - q) ** (L - i) for i in range(0, min(r, L) + 1))

    for s in range(s_star + 1, wanted + 1):
        lam *= within(s - 1) / max(within(s - 2), 1e-300)
    return float(draws) * float(np.exp(-lam))


# ------------------------------------------------------------------------------------------------
# neighbors -> output-table columns
# ------------------------------------------------------------------------------------------------
def _join_rows_fixed(mat: np.ndarray) -> List[str]:
    """(n, w) uint8 -> list of n str: one decode of the whole buffer, then plain slices (a fixed-width unicode array
    would take 4 bytes per character)."""
    n, w = mat.shape
    if n == 0:
        return []
    big = np.ascontiguousarray(mat).tobytes().decode("ascii")
    return [big[a:a + w] for a in range(0, n * w, w)]


def format_neighbor_columns(targetprocessor_object, guide_sequences):
    """The two neighbour columns of the reference's output table, without its per-row Python closures.

    Reference: Annotation._format_guide_table (core.py:913-931) builds, for every row of the guide table,
        'Similar guides'          = ";".join(tl.neighbors[seq]["neighbors"]["seqs"])
        'Similar guide distances' = ";".join(str(i) for i in tl.neighbors[seq]["neighbors"]["dist"])
    with two DataFrame.apply calls that go through the neighbors dict row by row.  Here both columns come straight
    from the k-NN result arrays: one id -> sequence take, one fixed-width byte matrix for the sequences (every row
    with k hits has the same width) and one ragged byte buffer for the distances (one or two digits each).
    Returns (similar_guides, similar_guide_distances), lists aligned with guide_sequences, byte-identical to the
    reference's strings.  A sequence that is not a key of .neighbors raises KeyError, as the dict lookup would.
    """
    nb = targetprocessor_object.neighbors
    guides = list(guide_sequences)
    if not isinstance(nb, NeighborMap):      # a plain dict (e.g. filled by hand): the reference's own loops
        return ([";".join(nb[g]["neighbors"]["seqs"]) for g in guides],
                [";".join(str(i) for i in nb[g]["neighbors"]["dist"]) for g in guides])
    n = len(guides)
    if n == 0:
        return [], []
    rowmap = nb._rows()
    rows = np.fromiter((rowmap[g] for g in guides), dtype=np.int64, count=n)
    ids, dists, counts = nb.ids[rows], nb.dists[rows], nb.counts[rows]
    k = ids.shape[1]
    if nb._dbp is not None and nb._L is not None and 1 <= nb._L <= _lib.MAX_GUIDELEN:
        # the index's targets are at hand in packed form: both columns natively (libgmhip, host threads)
        return _lib.format_neighbor_rows(nb._dbp, nb._L, ids, dists, counts)
    db = nb._db
    L = len(db[0]) if len(db) else 0
    full = counts == k
    sim = np.empty(n, dtype=object)
    dst = np.empty(n, dtype=object)
    if full.any():
        fi = np.flatnonzero(full)
        # sequences: k fields of L bytes with ';' between them -- one fixed-width byte matrix
        db_bytes = np.frombuffer("".join(db.tolist()).encode("ascii"), dtype=np.uint8).reshape(-1, L)
        mat = np.full((fi.size, k * (L + 1)), ord(";"), dtype=np.uint8)
        mat.reshape(fi.size, k, L + 1)[:, :, :L] = db_bytes[ids[fi]]
        sim[fi] = _join_rows_fixed(mat[:, :k * (L + 1) - 1])
        # distances: [tens digit][ones digit][';'] per hit, tens digit only for d >= 10, no ';' after the last hit
        d = dists[fi].astype(np.int64)
        parts = np.empty((fi.size, k, 3), dtype=np.uint8)
        parts[:, :, 0] = ord("0") + d // 10
        parts[:, :, 1] = ord("0") + d % 10
        parts[:, :, 2] = ord(";")
        use = np.ones((fi.size, k, 3), dtype=bool)
        use[:, :, 0] = d >= 10
        use[:, k - 1, 2] = False
        if (d >= 100).any():                 # (cannot happen for guides of <= 27 bases; keep the general path honest)
            full_rows = fi.tolist()
            dst[fi] = [";".join(str(int(x)) for x in dists[r]) for r in full_rows]
        else:
            flat = parts[use].tobytes().decode("ascii")
            ends = np.cumsum(use.sum(axis=(1, 2)))
            starts = ends - use.sum(axis=(1, 2))
            dst[fi] = [flat[a:b] for a, b in zip(starts.tolist(), ends.tolist())]
    for r in np.flatnonzero(~full).tolist():          # rows with fewer than k hits (index smaller than k)
        c = int(counts[r])
        sim[r] = ";".join(db[ids[r, :c]].tolist())
        dst[r] = ";".join(str(int(x)) for x in dists[r, :c])
    return sim.tolist(), dst.tolist()


def neighbor_table(targetprocessor_object) -> pd.DataFrame:
    """Frame with one row per retained guide (the keys of .neighbors, first-seen order) and the columns
    'Guide sequence', 'Similar guides', 'Similar guide distances' -- what _format_guide_table (core.py:925-931) adds to
    the guide table, ready for cfd_score()."""
    nb = targetprocessor_object.neighbors
    keys = list(nb.keys())
    sim, dst = format_neighbor_columns(targetprocessor_object, keys)
    frame = pd.DataFrame({"Guide sequence": keys, "Similar guides": sim, "Similar guide distances": dst})
    pk = getattr(nb, "packed_pairs", None)
    if pk is not None:
        # side-car for cfd_score: the packed guides and their packed similar guides (n x k), so that the strings just
        # built are not parsed and re-encoded.  The arrays stay in a private cache of this module; the frame carries a
        # small string token only (pandas compares and deep-copies `attrs` on concat / merge / every frame operation:
        # arrays in there break `pd.concat` and cost a copy per operation), and cfd_score uses the side-car only when a
        # hash over EVERY row of both columns still matches.
        side = pk(keys)
        if side is not None:
            frame.attrs["gm_cfd_token"] = _side_car_put(side, keys, sim)
    return frame


def _packed_side_car_put(packed) -> str:
    token = "gm-packed-%d" % next(_SIDE_CAR_SERIAL)      # (a monotonic serial: an evicted token can never name a later entry)
    while len(_PACKED_SIDE_CARS) >= _SIDE_CAR_LIMIT:
        _PACKED_SIDE_CARS.pop(next(iter(_PACKED_SIDE_CARS)))
    _PACKED_SIDE_CARS[token] = packed
    return token


def packed_side_car(frame):
    """The 2-bit packed 'target' column find_targets kept for this frame (uint64 [len(frame)]), or None: the frame was
    not made by find_targets, or its side-car has been dropped from the cache (the last few frames are kept)."""
    token = frame.attrs.get("gm_packed_token") if hasattr(frame, "attrs") else None
    return _PACKED_SIDE_CARS.get(token) if isinstance(token, str) else None


_PACKED_SIDE_CARS = {}   # token -> packed 'target' column of a find_targets frame
_SIDE_CARS = {}          # token -> (packed guides, packed similar guides, L, hash of the guide column, of the similar-guides column)
_SIDE_CAR_LIMIT = 4      # frames whose side-car is kept (oldest dropped first)
_SIDE_CAR_SERIAL = itertools.count(1)


def _column_hash(col) -> int:
    """Hash over every row of a string column (row boundaries included): one C-level join + one hash."""
    return hash("\n".join(col))


def _side_car_put(side, guides, sim_col) -> str:
    gpk, spk, L = side
    token = "gm-cfd-%d" % next(_SIDE_CAR_SERIAL)
    while len(_SIDE_CARS) >= _SIDE_CAR_LIMIT:
        _SIDE_CARS.pop(next(iter(_SIDE_CARS)))
    _SIDE_CARS[token] = (gpk, spk, L, _column_hash(guides), _column_hash(sim_col))
    return token


# ------------------------------------------------------------------------------------------------
# CFD
# ------------------------------------------------------------------------------------------------
def _pack_rna(seqs: List[str], L: int) -> np.ndarray:
    """Pack equal-length sequences; case and U/T are normalised on the joined string (one C-level pass)."""
    joined = "".join(seqs).upper().replace('U', 'T')
    if len(joined) != len(seqs) * L:
        raise ValueError("all sequences must have length %d" % L)
    return _lib.pack_bytes(joined.encode("ascii"), len(seqs), L)


def _pack_by_length(seqs: List[str], lens: np.ndarray) -> np.ndarray:
    """Pack sequences of mixed (valid) lengths; returns uint64 array aligned with seqs."""
    uniq = np.unique(lens).tolist()
    if len(uniq) == 1:
        return _pack_rna(seqs, uniq[0])
    out = np.empty((len(seqs),), np.uint64)
    for L in uniq:
        sel = np.flatnonzero(lens == L)
        out[sel] = _pack_rna([seqs[i] for i in sel.tolist()], L)
    return out


def calc_cfd_pairs(guides: List[str], offs: List[str], device: int = 0) -> np.ndarray:
    """CFD score of equal-length (guide, off-target) pairs on the GPU (calc_cfd, batched)."""
    n = len(guides)
    out = np.empty((n,), np.float64)
    if not n:
        return out
    glen = np.fromiter(map(len, guides), dtype=np.int64, count=n)
    olen = np.fromiter(map(len, offs), dtype=np.int64, count=n)
    bad = np.flatnonzero(glen != olen)
    if bad.size:
        i = int(bad[0])
        raise AssertionError("The lengths wt and off differ: wt = {}, off = {}".format(str(glen[i]), str(olen[i])))
    mm = _load_cfd_table()
    gp = _pack_by_length(guides, glen)
    op = _pack_by_length(offs, olen)
    for L in np.unique(glen).tolist():
        sel = np.flatnonzero(glen == L)
        out[sel] = _lib.cfd_batch(gp[sel], op[sel], L, mm, device)
    return out


def calc_cfd(wt: str, off: str, mm_scores=None) -> float:
    """Single-pair form of the reference's cfd_score_calculator.calc_cfd (runs on the GPU)."""
    return float(calc_cfd_pairs([wt], [off])[0])


def _split_similar(sim_col: List[str], L: int):
    """'Similar guides' strings -> (per_row counts, packed off-targets) without splitting row by row, when every
    listed sequence has the guide length L: the ';' are dropped from ONE joined string and the bytes packed in one
    call.  Returns None when a row does not fit that shape (the general path then reports the mismatch)."""
    n = len(sim_col)
    lens = np.fromiter(map(len, sim_col), dtype=np.int64, count=n)
    per_row = np.fromiter((s.count(';') for s in sim_col), dtype=np.int64, count=n) + 1
    if not (lens == per_row * (L + 1) - 1).all():
        return None
    flat = "".join(sim_col).replace(";", "").upper().replace("U", "T")
    total = int(per_row.sum())
    if len(flat) != total * L:
        return None
    try:
        return per_row, _lib.pack_bytes(flat.encode("ascii"), total, L)
    except ValueError:
        return None


def _cfd_side_car(df, guides, sim_col):
    """The packed (guides, similar guides, L) neighbor_table kept for this frame (token in df.attrs) -- used only if it
    still describes the frame: same number of rows and the same hash over EVERY row of 'Guide sequence' and of
    'Similar guides' as when it was made (a frame that was filtered, sorted, edited or concatenated since falls back to
    the strings)."""
    token = df.attrs.get("gm_cfd_token") if hasattr(df, "attrs") else None
    side = _SIDE_CARS.get(token) if isinstance(token, str) else None
    if side is None:
        return None
    gpk, spk, L, h_guides, h_sim = side
    n = len(guides)
    if n == 0 or gpk.shape[0] != n or spk.shape[0] != n:
        return None
    try:
        if _column_hash(guides) != h_guides or _column_hash(sim_col) != h_sim:
            return None
    except TypeError:      # (a non-string cell)
        return None
    return gpk, spk, L


def cfd_score(df, device: int = 0):
    """Adds 'CFD Similar Guides' (list of str(float)) and 'Max CFD' (reference core.py:1129-1148).

    All (guide, similar guide) pairs of the frame go to the GPU in one batch; each distinct guide is packed once;
    the str(float) formatting runs once per DISTINCT score."""
    guides = df['Guide sequence'].tolist()
    sim_col = df['Similar guides'].tolist()
    glen_row = np.fromiter(map(len, guides), dtype=np.int64, count=len(guides))
    fast = None
    side = _cfd_side_car(df, guides, sim_col)
    if side is None and len(guides) and (glen_row == glen_row[0]).all() and 1 <= int(glen_row[0]) <= _lib.MAX_GUIDELEN:
        fast = _split_similar(sim_col, int(glen_row[0]))
    if side is not None:
        gpk, spk, L = side
        per_row = np.full(len(guides), spk.shape[1], dtype=np.int64)
        n = int(spk.size)
        scores = _lib.cfd_batch(np.repeat(gpk, spk.shape[1]), spk.reshape(-1), L, _load_cfd_table(), device)
    elif fast is not None:
        per_row, op = fast
        L = int(glen_row[0])
        n = int(per_row.sum())
        scores = np.empty((n,), np.float64)
        if n:
            gp = np.repeat(_pack_rna(guides, L), per_row)
            scores[:] = _lib.cfd_batch(gp, op, L, _load_cfd_table(), device)
    else:
        sims = [s.split(';') for s in sim_col]
        per_row = np.fromiter(map(len, sims), dtype=np.int64, count=len(sims))
        offs = list(itertools.chain.from_iterable(sims))
        n = len(offs)
        olen = np.fromiter(map(len, offs), dtype=np.int64, count=n)
        glen = np.repeat(glen_row, per_row)
        bad = np.flatnonzero(glen != olen)
        if bad.size:
            i = int(bad[0])
            raise AssertionError("The lengths wt and off differ: wt = {}, off = {}".format(str(glen[i]), str(olen[i])))
        scores = np.empty((n,), np.float64)
        if n:
            mm = _load_cfd_table()
            gp = np.repeat(_pack_by_length(guides, glen_row), per_row)
            op = _pack_by_length(offs, olen)
            for L in np.unique(olen).tolist():
                sel = np.flatnonzero(olen == L)
                scores[sel] = _lib.cfd_batch(gp[sel], op[sel], L, mm, device)
    ends = np.cumsum(per_row)
    starts = ends - per_row
    # str(float), as the reference formats them (CFD products are almost all distinct: nothing to share)
    flat_l = _lib.format_doubles(scores)      # Python's float repr, formatted natively on the host's threads
    if len(per_row) and (per_row == per_row[0]).all():
        w = int(per_row[0])
        lists = [flat_l[a:a + w] for a in range(0, n, w)]
    else:
        lists = [flat_l[a:b] for a, b in zip(starts.tolist(), ends.tolist())]
    df['CFD Similar Guides'] = pd.Series(lists, index=df.index, dtype=object)
    # ''.split(';') yields [''] so every row has at least one entry, as in the reference
    df['Max CFD'] = np.maximum.reduceat(scores, starts) if n else np.empty((0,), np.float64)
    return df
