#!/usr/bin/env python3
"""Generates tests/golden/count_cases.json.

Provenance of the expected outputs
  * case `ref_doc_GGAGCT_k3` is the reference's own known-answer table
    (documentation/source/reference.rst:545-568): 3-mers of GGAGCT, canonical
    under A<C<T<G, stored order AGC(2) CTC(1) TCC(1).  Its `expected` list was
    typed from that table by hand and is NOT produced by our oracle; the
    oracle must reproduce it (tests/test_oracle.py).
  * every other case is produced by oracle/ (orc_count_brute), i.e. by our
    restatement -- the reference cannot be built here (its meryl-utility
    submodule is absent), so these pin the HIP path to the oracle, and the
    oracle to the reference only through the case above plus the ordering rule
    of src/tests/test-operations.pl:114-118.  Inputs follow the reference's
    own test generators where it has them (src/tests/test-build.pl:9-62: A, AC,
    ACG, ACGT repeats at k=22) and SURVEY.md 8(c)'s edge-case list.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

MODES = {"canonical": 0, "forward": 1, "reverse": 2}


def run(name, bases, k, mode="canonical", note=""):
    hi, lo, cn, ni = oracle.count_brute(bases, k, MODES[mode])
    exp = [[oracle.kmer_to_string(h, l, k), int(c)] for h, l, c in zip(hi, lo, cn)]
    return {"name": name, "k": k, "mode": mode, "bases": bases, "n_instances": int(ni), "expected": exp,
            "source": "oracle", "note": note}


def main():
    cases = []
    cases.append({"name": "ref_doc_GGAGCT_k3", "k": 3, "mode": "canonical", "bases": "GGAGCT", "n_instances": 4,
                  "expected": [["AGC", 2], ["CTC", 1], ["TCC", 1]], "source": "reference.rst:545-568",
                  "note": "the reference's only in-tree known-answer vector"})
    # the reference's low-complexity generators, src/tests/test-build.pl:9-62 (k=22)
    for unit in ("A", "AC", "ACG", "ACGT"):
        cases.append(run("testbuild_%s_repeat_k22" % unit, (unit * 60)[:60] + ".", 22,
                         note="test-build.pl generator"))
    seq = "ACGTTGCATGTCGCATGATGCATGAGAGCTACGTTGCATGNACGTAGCTAGCTAGTCGATCGATCGTAGCTAGCTAGCTGATCG"
    cases.append(run("n_breaks_kmer_k6", seq + ".", 6, note="N resets the rolling k-mer"))
    cases.append(run("lower_case_k6", seq.lower() + ".", 6, note="lower case counts like upper case"))
    cases.append(run("mixed_case_k16", "".join(c.lower() if i % 3 else c for i, c in enumerate(seq)) + ".", 16))
    cases.append(run("shorter_than_k", "ACGTACGTAC.", 21, note="no k-mer at all"))
    cases.append(run("empty", "", 21))
    cases.append(run("only_breakers", "....", 5))
    cases.append(run("two_reads_no_span_k5", "AAAAC.GTTTT.", 5, note="AAAACGTTTT would add spurious k-mers across the breaker"))
    cases.append(run("palindromes_k4", "ACGTACGTTTAAATGCATCGCGAT.", 4, note="even k: f == r happens (ACGT, TTAA, GCGC...)"))
    cases.append(run("forward_k5", seq + ".", 5, "forward"))
    cases.append(run("reverse_k5", seq + ".", 5, "reverse"))
    long_seq = (seq.replace("N", "T") * 3)
    for k in (1, 2, 16, 17, 21, 31, 32):
        cases.append(run("k%d" % k, long_seq + "." + long_seq[::-1] + ".", k))
    for k in (33, 51, 64):
        cases.append(run("k%d_wide" % k, long_seq + "." + long_seq[::-1] + ".", k, note="128-bit keys"))
    cases.append(run("iupac_and_gaps_k4", "ACGTRYKMACGT-ACGTNNACGU.acgt*ACGT", 4, note="anything not ACGTacgt breaks"))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "count_cases.json")
    with open(out, "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py", "cases": cases}, f, indent=0)
    print("wrote %d cases to %s" % (len(cases), out))


if __name__ == "__main__":
    main()
