#!/bin/bash
timeout 900 python -m pytest tests/test_hip_gemm.py -m gpu -q -x -p no:cacheprovider -k "big_tile" 2>&1 | tail -6
