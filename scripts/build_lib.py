#!/usr/bin/env python3
"""Builds the library for gfx950:  python scripts/build_lib.py OUT.so [-DFLAG ...]   (see __graft_entry__.compile_library:
parallel translation units by default, one unit with any -D flag or KLNMF_SINGLE_TU=1)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry      # noqa: E402

if __name__ == '__main__':
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    entry.compile_library(sys.argv[1], sys.argv[2:])
