"""Substitution matrices in the layout of the reference's tables.

The eight matrices OSWALD offers (`-s`, reference host/src/arguments.c:76-88)
are the standard NCBI BLOSUM45/50/62/80/90 and PAM30/70/250 tables; they are
kept under oswald_amd/data/ in NCBI order and re-laid out here the way the
reference stores them (host/src/submat.c, host/src/submat.h:4-6): 24 rows x 32
columns of int8, rows/columns in the preprocessed alphabet order
A B C D E F G H I K L M N P Q R S T V W X Y Z (0..22); row 23 and columns
23..31 are zero, so the dummy residue 23 scores 0 against everything.
"""
from __future__ import annotations

import os

import numpy as np

OSWALD_ORDER = "ABCDEFGHIKLMNPQRSTVWXYZ"
NAMES = ("blosum45", "blosum50", "blosum62", "blosum80", "blosum90", "pam30", "pam70", "pam250")
ROWS, COLS = 24, 32

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
_cache: dict[str, np.ndarray] = {}


def parse_ncbi(path: str) -> dict[tuple[str, str], int]:
    cols, table = None, {}
    with open(path) as f:
        for line in f:
            if not line.strip() or line.startswith("#"):
                continue
            tok = line.split()
            if cols is None:
                cols = tok
                continue
            for c, v in zip(cols, tok[1:]):
                table[(tok[0], c)] = int(v)
    return table


def load(name: str) -> np.ndarray:
    """int8 [24, 32] table of matrix `name` (as on the reference's command line)."""
    name = name.lower()
    if name not in NAMES:
        raise ValueError(f"{name} is not a valid option for substitution matrix.")
    if name not in _cache:
        t = parse_ncbi(os.path.join(_DATA, name.upper() + ".txt"))
        m = np.zeros((ROWS, COLS), dtype=np.int8)
        for i, r in enumerate(OSWALD_ORDER):
            for j, c in enumerate(OSWALD_ORDER):
                m[i, j] = t[(r, c)]
        _cache[name] = m
    return _cache[name].copy()
