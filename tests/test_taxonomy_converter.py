"""taxonomy list -> <name>_data.txt / <name>_tree.txt / refkey (tools/taxonomy_from_list.py).  The reference
ships no code for this step, only its input (a list) and its outputs for the mitochondria DB; the fixture is the first
400 strains of that list with what the shipped files say about them (ids are given in first-appearance order, so a
prefix of the list yields a prefix of the ids).  Where /root/reference exists the whole list is compared too."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import taxonomy_from_list as tfl  # noqa: E402

FIX = os.path.join(ROOT, "tests", "golden", "taxonomy_list")


def _edges(path):
    return sorted(tuple(int(x) for x in l.split()) for l in open(path) if l.strip())


def test_excerpt_matches_the_reference_files(tmp_path):
    out = os.path.join(str(tmp_path), "mito")
    par, strains, names, under = tfl.write_db_files(os.path.join(FIX, "list_excerpt.txt"), 6, 6, out)
    assert open(out + "_data.txt").read() == open(os.path.join(FIX, "expected_data.txt")).read()
    assert _edges(out + "_tree.txt") == _edges(os.path.join(FIX, "expected_tree.txt"))
    got = [l.rstrip("\r\n").split("\t") for l in open(out + "_refkey.txt", newline="")]
    exp = [l.rstrip("\n").split("\t") for l in open(os.path.join(FIX, "expected_refkey_names.txt"))]
    assert [g[:2] for g in got] == exp
    assert got[0][6] == "strains" and all(len(g) == 7 for g in got)
    assert sum(int(g[6]) for g in got[1:] if par[int(g[0])] == 1 and int(g[0]) > 1) == len(strains)  # top ranks partition the strains
    # the tree file round-trips through the loader's rule (parent[child] = parent, default root)
    par2 = np.ones(par.size, np.int32)
    for p_, c in _edges(out + "_tree.txt"):
        par2[c] = p_
    assert np.array_equal(par, par2)


def test_whole_mitochondria_list_when_reference_is_present(tmp_path):
    ref = "/root/reference"
    if not os.path.exists(os.path.join(ref, "mitochondria_list.txt")):
        pytest.skip("reference data not present (GPU box)")
    out = os.path.join(str(tmp_path), "mito")
    par, strains, names, under = tfl.write_db_files(os.path.join(ref, "mitochondria_list.txt"), 6, 6, out)
    assert open(out + "_data.txt").read() == open(os.path.join(ref, "mitochondria_data.txt")).read()
    assert _edges(out + "_tree.txt") == _edges(os.path.join(ref, "mitochondria_tree.txt"))
    exp = [l.rstrip("\r\n").split("\t") for l in open(os.path.join(ref, "mitochondria_refkey.txt"), newline="") if l.strip()]
    got = [l.rstrip("\r\n").split("\t") for l in open(out + "_refkey.txt", newline="")]
    assert [g[:2] for g in got] == [e[:2] for e in exp]          # ids and names
    assert [g[6] for g in got[3:]] == [e[6] for e in exp[3:]]    # strains under every node
    z = np.load(os.path.join(ROOT, "kmer_id_amd", "data", "taxonomy_mito.npz"))
    assert np.array_equal(par, z["parent"])
