"""ORACLE (test infrastructure, NOT product code): a second, independent restatement of the reference's data loader,
`Dataset.rcv1` (utils/Dataset.scala:13-60), in plain Python -- to diff the product's native parser (csrc/rcv1.c)
against.  Written from the Scala source line by line; shares no code with the product.

    readData   (:19-34)   parts = line.split(' '); rowID = parts(0).toInt;
                          parts.drop(2).map(row => { elems = row.split(':'); elems(0).toInt -> elems(1).toDouble }).toMap
    readLabels (:36-45)   parts = line.split(' '); (parts(1).toInt, if (parts(0) == "CCAT") 1 else -1)
    labels     (:53)      readLabels(...).toMap            -- the LAST line of a document wins
    files      (:47-50)   the train file, then test_pt0..3 when `full`
    result     (:55-58)   rows in file order, labels(id) (throws on a missing id)

Java's `String.split(' ')` drops TRAILING empty strings and keeps the others; `Map` construction keeps the last value
of a repeated key; `.toInt` / `.toDouble` are Integer.parseInt / Double.parseDouble.
"""

from __future__ import annotations

import os
import re

import numpy as np

FILES = ["lyrl2004_vectors_train.dat"] + ["lyrl2004_vectors_test_pt%d.dat" % d for d in range(4)]
QRELS = "rcv1-v2.topics.qrels"
_INT = re.compile(r"^[+-]?[0-9]+$")


def java_split(line: str, sep: str):
    """String.split(sep) for a one-character, non-regex separator: trailing empty strings are removed, and an input
    without any separator comes back whole (so "" -> [""], but " " -> [])."""
    if sep not in line:
        return [line]
    parts = line.split(sep)
    while parts and parts[-1] == "":
        parts.pop()
    return parts


def to_int(tok: str) -> int:
    if not _INT.match(tok):
        raise ValueError("NumberFormatException: %r" % tok)
    return int(tok)


def to_double(tok: str) -> float:
    t = tok.strip()
    if t[-1:] in "dDfF":   # Double.parseDouble accepts a type suffix
        t = t[:-1]
    return float(t)        # (raises ValueError like NumberFormatException)


def read_data(path):
    out = []
    with open(path) as f:
        for line in f.read().split("\n")[:-1]:   # Source.getLines: the text before each line terminator
            parts = java_split(line, " ")
            row_id = to_int(parts[0])
            vec = {}
            for tok in parts[2:]:
                elems = java_split(tok, ":")
                vec[to_int(elems[0])] = to_double(elems[1])   # IndexError == ArrayIndexOutOfBoundsException
            out.append((row_id, vec))
    return out


def read_labels(path):
    labels = {}
    with open(path) as f:
        for line in f.read().split("\n")[:-1]:
            parts = java_split(line, " ")
            labels[to_int(parts[1])] = 1 if parts[0] == "CCAT" else -1
    return labels


def rcv1(folder: str, full: bool = True):
    """-> (row_ptr int64, col int32 in the order the Scala Map was filled, val float64, label int8, ids)."""
    labels = read_labels(os.path.join(folder, QRELS))
    row_ptr, col, val, lab, ids = [0], [], [], [], []
    for name in FILES[: 5 if full else 1]:
        for row_id, vec in read_data(os.path.join(folder, name)):
            for k, v in vec.items():
                col.append(k)
                val.append(v)
            row_ptr.append(len(col))
            lab.append(labels[row_id])   # KeyError == NoSuchElementException
            ids.append(row_id)
    return (np.asarray(row_ptr, np.int64), np.asarray(col, np.int32), np.asarray(val, np.float64),
            np.asarray(lab, np.int8), np.asarray(ids, np.int64))
