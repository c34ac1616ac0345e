import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def load_carsonella(tag):
    """-> dict of column lists from tests/golden/carsonella_targets_<tag>.tsv.gz"""
    with gzip.open(os.path.join(GOLDEN, "carsonella_targets_%s.tsv.gz" % tag), "rt") as fh:
        header = fh.readline().rstrip("\n").split("\t")
        cols = {h: [] for h in header}
        for line in fh:
            f = line.rstrip("\n").split("\t")
            cols["target"].append(f[0])
            cols["start"].append(int(f[1]))
            cols["stop"].append(int(f[2]))
            cols["strand"].append(bool(int(f[3])))
            cols["isseedduplicated"].append(bool(int(f[4])))
            cols["hasrestrictionsite"].append(bool(int(f[5])))
    return cols


@pytest.fixture(scope="session")
def cfd_table():
    return load_json("cfd_table.json")["mm"]


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc
