#!/usr/bin/env python3
"""Builds the library for gfx950:  python scripts/build_lib.py OUT.so [-DFLAG ...]   (__graft_entry__.compile_library: seven
translation units in parallel, the -D flags on every one; KLNMF_LIB=OUT.so selects the variant at run time)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry      # noqa: E402

if __name__ == '__main__':
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    entry.compile_library(sys.argv[1], sys.argv[2:])
