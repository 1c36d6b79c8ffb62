#!/usr/bin/env python3
"""Builds lab variants of the library: tools/build_variants.py name=-DFLAG,-DFLAG2 ..."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlvi_amd import _build
for spec in sys.argv[1:]:
    name, flags = spec.split("=", 1)
    lib = os.path.join(_build.HERE, f"librlvi_{name}.so")
    _build.build(extra_flags=[f for f in flags.split(",") if f], lib=lib)
    print(lib)
