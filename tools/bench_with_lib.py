#!/usr/bin/env python3
"""Run bench.py against an experiment build of the library: tools/bench_with_lib.py <lib.so> [bench.py args...]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import tsxform
tsxform._native.LIB_PATH = lib
import bench
bench.main()
