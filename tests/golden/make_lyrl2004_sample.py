#!/usr/bin/env python3
"""Writes tests/golden/lyrl2004_sample/: 24 documents in the FORMATTING of the LYRL2004 distribution of RCV1-v2
(utils/Dataset.scala:47-50 names the files) -- not the Reuters data, which is neither in the reference repository
nor reachable from here:

  lyrl2004_vectors_train.dat / _test_pt0..3.dat   `<did>  <fid>:<w> <fid>:<w> ...`  two spaces after the document
      id, feature ids ascending, cosine-normalised log-TF-IDF weights printed with up to 16 significant digits
  rcv1-v2.topics.qrels                             `<code> <did> 1`, grouped by document, a document's codes in
      code order (so `CCAT` comes before `ECAT`/`GCAT`/`MCAT` and after `C15`...)

Deterministic (seed 2004).  The expected CSR of the reference's loader (Dataset.rcv1, last qrels line per document
wins, tokens 2.. of a line are the features) is produced independently by oracle/ref_loader.py.
"""
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "lyrl2004_sample")
FILES = ["lyrl2004_vectors_train.dat"] + ["lyrl2004_vectors_test_pt%d.dat" % d for d in range(4)]
CODES = ["C15", "C151", "C152", "C31", "CCAT", "E12", "E21", "ECAT", "G15", "GCAT", "GPOL", "M11", "M14", "MCAT"]


def main():
    rnd = random.Random(2004)
    os.makedirs(OUT, exist_ok=True)
    did = 2286
    docs = []
    for f in range(5):
        for _ in range(8 if f == 0 else 4):
            n = rnd.choice([1, 3, 7, 19, 42, 77, 130])
            fids = sorted(rnd.sample(range(1, 47237), n))
            raw = [rnd.lognormvariate(0.0, 0.6) for _ in fids]
            norm = sum(x * x for x in raw) ** 0.5
            docs.append((f, did, [(i, x / norm) for i, x in zip(fids, raw)]))
            did += rnd.choice([1, 1, 2, 5])
    for f, name in enumerate(FILES):
        with open(os.path.join(OUT, name), "w") as fh:
            for ff, d, feats in docs:
                if ff == f:
                    fh.write("%d  %s\n" % (d, " ".join("%d:%s" % (i, repr(x)) for i, x in feats)))
    with open(os.path.join(OUT, "rcv1-v2.topics.qrels"), "w") as fh:
        for _, d, _ in docs:
            k = rnd.choice([1, 2, 2, 3, 4])
            codes = set(rnd.sample(CODES, k))
            if rnd.random() < 0.6:   # CCAT (the positive class) on most documents: alone, last, or followed by E*/G*/M*
                codes.add("CCAT")
                if rnd.random() < 0.5:
                    codes = {c for c in codes if c <= "CCAT"}
            for code in sorted(codes):
                fh.write("%s %d 1\n" % (code, d))


if __name__ == "__main__":
    main()
